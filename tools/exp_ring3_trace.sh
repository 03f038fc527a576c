cd $GRAFT_REPO_ROOT
for shape in "115200 640 2560 0" "28800 1280 5120 0" "115200 5120 640 2"; do
  echo "== $shape  (ring3, then shipped choice)"
  MOFA_IGEMM_CFG=5 timeout 20 tools/igemm_trace_ring3.bin $shape | head -3
  timeout 20 tools/igemm_trace_ring3.bin $shape | head -3
done
