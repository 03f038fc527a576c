# r04c: final state (XCD-aware attention order on top of r04b): GPU suite, kernel statistics + PMC passes + default bench, L1 attention QB check
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $O/r04c_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/r04c_gpu_tests.log
bash tools/profile_round.sh r04c
cd $GRAFT_REPO_ROOT
for qb in 0 1 2; do echo "== --qb $qb"; timeout 200 python tools/attn_bench.py --iters 8 --qb $qb 2>&1 | grep "attn spatial L"; done > $O/r04c_attn_qb.log 2>&1; cat $O/r04c_attn_qb.log
