# r04d: final state (norm kernels retuned on top of r04c): kernel statistics + PMC passes + default bench, then as much of the GPU suite as fits
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/profile_round.sh r04d > $O/r04d_profile_round.out 2>&1; tail -3 $O/r04d_profile_round.out | cut -c1-400
cd $GRAFT_REPO_ROOT
timeout 330 python -m pytest tests -x -q -m gpu --deselect tests/test_fullloop_gpu.py > $O/r04d_gpu_tests.log 2>&1; echo "gpu tests rc=$? (124 = cut off by the time limit)"; tail -3 $O/r04d_gpu_tests.log
