# On the GPU box (gpurun), after tools/build_ring3_tools.sh:  bash tools/exp_ring3.sh > gpurun_out/exp_ring3.log 2>&1
# 1. bit-exactness / determinism / time against the shipped kernels; 2. per-phase ticks of ring3 vs the shipped choice.
# Everything under `timeout`: an experimental kernel must not be able to hold the GPU.
cd $GRAFT_REPO_ROOT
timeout 90 tools/igemm_ring3_check.bin; echo "check rc=$?"
for shape in "115200 640 2560 0" "460800 320 1280 0" "28800 1280 5120 0" "115200 5120 640 2" "460800 2560 320 2"; do
  echo "== $shape"
  MOFA_IGEMM_CFG=5 timeout 30 tools/igemm_trace_ring3.bin $shape | head -3
  timeout 30 tools/igemm_trace_ring3.bin $shape | head -3
done
