set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in 1 0; do
  PVLM_MVS_SPEC=$S timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mvs_trace_spec$S -- python $R/tools/mvs_bench.py > $O/mvs_trace_spec$S.log 2>&1
  find $O/mvs_trace_spec$S -name "*kernel_stats.csv" -exec cat {} \; | grep -E "Name|propagate_diag" | cut -c1-60,300-
  find $O/mvs_trace_spec$S -name "*kernel_trace.csv" -exec cp {} $O/mvs_ktrace_spec$S.csv \;
done
ls -la $O/mvs_ktrace_spec*.csv
