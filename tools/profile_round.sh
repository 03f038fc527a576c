cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/r01c_bench_under_rocprof.log 2>&1
python $R/tools/summarize_prof.py stats /tmp/st $O/r01c_kernel_stats_bench.md > /dev/null 2>&1
cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $O/r01c_kernel_stats_bench.csv
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -- python $R/tools/profile_step.py 1 > /dev/null 2>&1; python $R/tools/summarize_prof.py pmc /tmp/p_$c $O/r01c_pmc_$c.csv > /dev/null 2>&1; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma -- python $R/tools/profile_step.py 1 > /dev/null 2>&1; python $R/tools/summarize_prof.py pmc /tmp/p_mfma $O/r01c_pmc_mfma.csv > /dev/null 2>&1
cd $R && python bench.py > $O/r01c_bench_default.log 2>&1; tail -1 $O/r01c_bench_default.log > $O/r01c_bench_default.json
ls -la $O | tail -12; tail -1 $O/r01c_bench_under_rocprof.log | cut -c1-200; head -8 $O/r01c_kernel_stats_bench.md | cut -c1-160; cut -c1-600 $O/r01c_bench_default.json
