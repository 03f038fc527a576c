# r04b GPU call 1: epilogue-emitted GroupNorm sums (tests, per-shape cost, clip A/B) + conv K-order experiment (speed, fabric reads)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gn_stats_gpu.py -x -q > $O/r04b_gn_stats_tests.log 2>&1; echo "stats tests rc=$?" 
tail -3 $O/r04b_gn_stats_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_config1_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/r04b_model_tests.log 2>&1; echo "model tests rc=$?"
tail -3 $O/r04b_model_tests.log
timeout 600 python tools/igemm_tiles_bench.py --tiles 320p,320s --only "conv,proj / attn" --rounds 3 > $O/r04b_stats_tiles.log 2>&1; tail -22 $O/r04b_stats_tiles.log
timeout 600 python tools/igemm_tiles_bench.py --tiles 320p --only "conv3x3" --rounds 3 > $O/r04b_korder_base.log 2>&1
timeout 600 python tools/igemm_tiles_bench.py --tiles 320p --only "conv3x3" --rounds 3 --lib tools/libmofa_hip_chunk.so > $O/r04b_korder_chunk.log 2>&1
paste <(cut -c1-75 $O/r04b_korder_base.log) <(cut -c62-75 $O/r04b_korder_chunk.log)
for v in 0 1 0 1; do timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --gn-stats $v > $O/r04b_bench_gnstats$v.log 2>&1; python - <<PY
import json
l=open("$O/r04b_bench_gnstats$v.log").read().strip().splitlines()[-1]
try:
    d=json.loads(l); print("gn_stats=$v", d["value"], d["config"]["clip_ms"], d["roofline"]["achieved"])
except Exception as e: print("gn_stats=$v FAILED", l[-300:])
PY
cp $O/r04b_bench_gnstats$v.log $O/r04b_bench_gnstats${v}_run$RANDOM.log; done
cd /tmp
for L in base chunk; do
  if [ $L = chunk ]; then export MOFA_HIP_LIB=$GRAFT_REPO_ROOT/tools/libmofa_hip_chunk.so; else unset MOFA_HIP_LIB; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$L -- python $GRAFT_REPO_ROOT/tools/one_shape.py conv 50 72 128 320 > /dev/null 2>&1
  python - <<PY
import csv, glob
fs = glob.glob("/tmp/pmc_$L/**/*counter_collection.csv", recursive=True)
rows = [r for f in fs for r in csv.DictReader(open(f))]
v = [float(r["Counter_Value"]) for r in rows if "igemm320" in r.get("Kernel_Name","") and r.get("Counter_Name")=="FETCH_SIZE"]
print("$L FETCH_SIZE per igemm320 launch (raw counter units, KB):", [round(x) for x in v[:6]])
PY
done > $GRAFT_REPO_ROOT/$O/r04b_korder_fetch.log 2>&1
unset MOFA_HIP_LIB
cat $GRAFT_REPO_ROOT/$O/r04b_korder_fetch.log
