# usage (on the GPU box): bash tools/functional_shard.sh <tag>
# FUNCTIONAL check of the multi-process sharded bench path on ONE GPU: torch.distributed.run, N processes, gloo transport, all
# ranks on device 0, 1-2 denoise steps (MOFA_BENCH_DENOISE_STEPS).  The lines say "functional_only": true -- the transport is
# not RCCL and the timings mean nothing; what is checked is that every rank gets through the real process-group code path
# (sub-group creation, self-check, lockstep two-network schedule, exchanges, sharded decode) with finite output.
TAG=${1:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export MOFA_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # world config steps
  MOFA_BENCH_DENOISE_STEPS=$3 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 \
    --master-port $((29500 + $1 * 10 + $2)) bench.py --gpus $1 --config $2 --backend gloo --steps 1 --warmup 0 --no-cpu-baseline \
    > $O/${TAG}_functional_gloo$1_cfg$2.log 2>&1
  tail -1 $O/${TAG}_functional_gloo$1_cfg$2.log > $O/${TAG}_functional_shard_gloo$1_cfg$2.json
  python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_functional_shard_gloo$1_cfg$2.json").read())
    print("world $1 config $2:", d["config"]["output_finite"], d["config"]["comm_paths"], d["config"]["parallelism"][:120])
except Exception as e:
    print("world $1 config $2: FAILED", e); print(open("$O/${TAG}_functional_gloo$1_cfg$2.log").read()[-1500:])
PY
}
run 4 2 2
run 8 2 2
run 4 4 1
run 4 5 1
run 2 5 1
