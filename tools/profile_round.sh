# usage (on the GPU box, from gpurun):  bash tools/profile_round.sh [tag]
# kernel statistics of one bench clip (single-stream order: exclusive kernel durations, the mode of the bench's roofline leg), the three PMC passes (separate runs, --kernel-trace only), the igemm fabric traffic /
# MFMA-busy summary derived from them, then the default bench line (which cites that summary).  Everything lands in
# gpurun_out/<tag>_*; copy what should be judged into profiles/.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-stream > $O/${TAG}_bench_under_rocprof.log 2>&1
python $R/tools/summarize_prof.py stats /tmp/st $O/${TAG}_kernel_stats_bench.md > /dev/null 2>&1
cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench.csv
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -- python $R/tools/profile_step.py 1 > /dev/null 2>&1; python $R/tools/summarize_prof.py pmc /tmp/p_$c $O/${TAG}_pmc_$c.csv > /dev/null 2>&1; done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma -- python $R/tools/profile_step.py 1 > /dev/null 2>&1; python $R/tools/summarize_prof.py pmc /tmp/p_mfma $O/${TAG}_pmc_mfma.csv > /dev/null 2>&1
python $R/tools/igemm_traffic.py $O $TAG && cp $O/${TAG}_igemm_traffic.json $R/profiles/
cd $R && python bench.py > $O/${TAG}_bench_default.log 2>&1; tail -1 $O/${TAG}_bench_default.log > $O/${TAG}_bench_default.json
ls -la $O | tail -12; tail -1 $O/${TAG}_bench_under_rocprof.log | cut -c1-200; head -8 $O/${TAG}_kernel_stats_bench.md | cut -c1-160; cut -c1-700 $O/${TAG}_bench_default.json
