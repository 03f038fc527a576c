cd $GRAFT_REPO_ROOT
for v in s4 s3 s2 nodma nomfma; do
  echo "== $v"
  MOFA_IGEMM_CFG=3 tools/igemm_trace_$v.bin 460800 2560 320 2 | head -2
  MOFA_IGEMM_CFG=3 tools/igemm_trace_$v.bin 28800 10240 1280 2 | head -2
  MOFA_IGEMM_CFG=4 tools/igemm_trace_$v.bin 115200 640 2560 | head -2
  MOFA_IGEMM_CFG=4 tools/igemm_trace_$v.bin 460800 320 1280 | head -2
done
