# r04b GPU call 2: the whole GPU suite on the current state, then the clip A/B of the in-place concat (previous commit's host code,
# the same library, alternating on one box)
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r04b_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/r04b_gpu_tests.log
for v in prev cur prev cur; do
  if [ $v = prev ]; then (cd tools/ab_prev && MOFA_HIP_LIB=$GRAFT_REPO_ROOT/mofa_video_amd/libmofa_hip.so timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r04b_bench_concat_$v.log 2>&1)
  else timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r04b_bench_concat_$v.log 2>&1; fi
  python - <<PY
import json
l=open("$O/r04b_bench_concat_$v.log").read().strip().splitlines()[-1]
try:
    d=json.loads(l); print("$v", d["value"], d["config"]["clip_ms"], d["roofline"]["achieved"])
except Exception as e: print("$v FAILED", l[-400:])
PY
done | tee $O/r04b_concat_bench_ab.log
