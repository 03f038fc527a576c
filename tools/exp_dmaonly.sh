cd $GRAFT_REPO_ROOT
echo "== refill + barriers only (IGEMM_EXP=3): no fragment reads, no MFMA"
MOFA_IGEMM_CFG=4 timeout 20 tools/igemm_trace_dmaonly.bin 115200 640 2560 0 | head -2
MOFA_IGEMM_CFG=4 timeout 20 tools/igemm_trace_dmaonly.bin 115200 640 2560 0 0 | head -2
MOFA_IGEMM_CFG=3 timeout 20 tools/igemm_trace_dmaonly.bin 28800 10240 1280 2 | head -2
MOFA_IGEMM_CFG=2 timeout 20 tools/igemm_trace_dmaonly.bin 115200 640 2560 0 | head -2
