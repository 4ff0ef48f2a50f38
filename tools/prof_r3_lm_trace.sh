set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PVLM_DRIVER_PREFIX="rocprofv3 --kernel-trace --stats --output-format csv -d $O/floor_trace --" timeout 900 python $R/tools/floor_like_odometry.py --scans 1593 --ranks "" --iters 2 > $O/r3_floor_traced.txt 2>&1
find $O/floor_trace -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-170 | head -24
find $O/floor_trace -name "*kernel_trace.csv" -size +20M -delete
