cd $GRAFT_REPO_ROOT
for shape in "115200 640 2560 0" "460800 320 1280 0"; do
  echo "== $shape : normal, then ldx=0 (activation rows alias one row: refill reads hit L1/L2)"
  MOFA_IGEMM_CFG=4 timeout 20 tools/igemm_trace.bin $shape | head -2
  MOFA_IGEMM_CFG=4 timeout 20 tools/igemm_trace.bin $shape 0 | head -2
done
echo "== 28800 10240 1280 2 (256^2): normal, ldx=0"
MOFA_IGEMM_CFG=3 timeout 20 tools/igemm_trace.bin 28800 10240 1280 2 | head -2
MOFA_IGEMM_CFG=3 timeout 20 tools/igemm_trace.bin 28800 10240 1280 2 0 | head -2
