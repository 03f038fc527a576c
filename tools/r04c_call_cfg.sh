# r04c: configs 3-5 bench lines (warm-up 1 + 3 timed clips) and the 1-GPU proxy of a rank of the 8- / 4-GPU layouts on the final state
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for w in 8 4; do timeout 300 python tools/shard_proxy.py --world $w --steps 4 2>&1 | grep "rank 0"; done > $O/r04c_shard_proxy.log 2>&1; cat $O/r04c_shard_proxy.log
for c in 3 4 5; do
  timeout 700 python bench.py --config $c --warmup 1 --steps 3 --no-cpu-baseline > $O/r04c_bench_config$c.log 2>&1
  tail -1 $O/r04c_bench_config$c.log > $O/r04c_bench_config$c.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r04c_bench_config$c.json").read()); print("config $c", d["value"], d["config"]["clip_ms"])
except Exception as e: print("config $c FAILED", open("$O/r04c_bench_config$c.log").read()[-600:])
PY
done
