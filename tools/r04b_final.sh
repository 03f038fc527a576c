# r04b final GPU call: kernel statistics + PMC passes + default bench of the committed state (tools/profile_round.sh), then the
# opt-in full-size CPU baseline step (bounded: it is a stated baseline, not a target)
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04b
cd $GRAFT_REPO_ROOT
MOFA_CPU_BASELINE_THREADS=64 timeout 560 python -c "
import json, bench
print(json.dumps(bench.cpu_baseline_full(timeout=540)))" > gpurun_out/r04b_cpu_baseline_full.json 2> gpurun_out/r04b_cpu_baseline_full.err; echo "cpu full rc=$?"; cut -c1-600 gpurun_out/r04b_cpu_baseline_full.json; tail -2 gpurun_out/r04b_cpu_baseline_full.err
