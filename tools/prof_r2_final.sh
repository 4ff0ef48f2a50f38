set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python $R/bench.py --no-cpu-baseline --no-mvs > $O/bench_under_rocprof.log 2>&1
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
cd $R && python tools/trace_groups.py $(find $O/bench_trace -name "*kernel_trace.csv" | head -1) $O/kernel_groups_default.csv > /dev/null
cp $(find $O/bench_trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_default.csv
find $O -name "*kernel_trace.csv" -delete
du -sh $O
