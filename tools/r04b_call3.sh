# r04b GPU call 3: spatial attention, XCD-aware work order (variant build) vs the 3-D grid: speed, bit identity, fabric reads
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for r in 1 2; do
timeout 300 python tools/attn_bench.py --iters 8 > $O/r04b_attn_base_$r.log 2>&1
timeout 300 python tools/attn_bench.py --iters 8 --lib tools/libmofa_hip_attnxcd.so > $O/r04b_attn_xcd_$r.log 2>&1
done
for f in $O/r04b_attn_base_1.log $O/r04b_attn_xcd_1.log $O/r04b_attn_base_2.log $O/r04b_attn_xcd_2.log; do echo "== $f"; grep "attn spatial" $f; done
cd /tmp
cat > /tmp/one_attn.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mofa_video_amd import lib, ops
lib.load()
fr, heads, S = 50, 5, 9216
Cc = heads * 64
qkv = torch.randn(fr * S, 3 * Cc, device="cuda").half()
for _ in range(3): ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], fr, heads, S)
torch.cuda.synchronize()
PY
for L in base xcd; do
  if [ $L = xcd ]; then export MOFA_HIP_LIB=$GRAFT_REPO_ROOT/tools/libmofa_hip_attnxcd.so; else unset MOFA_HIP_LIB; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmca_$L -- python /tmp/one_attn.py > /dev/null 2>&1
  python - <<PY
import csv, glob
fs = glob.glob("/tmp/pmca_$L/**/*counter_collection.csv", recursive=True)
rows = [r for f in fs for r in csv.DictReader(open(f))]
v = [float(r["Counter_Value"]) for r in rows if "attn_spatial" in r.get("Kernel_Name","") and r.get("Counter_Name")=="FETCH_SIZE"]
print("$L FETCH_SIZE per attn_spatial launch (raw counter, KB):", [round(x) for x in v[:4]])
PY
done > $GRAFT_REPO_ROOT/$O/r04b_attn_fetch.log 2>&1
unset MOFA_HIP_LIB
cat $GRAFT_REPO_ROOT/$O/r04b_attn_fetch.log
