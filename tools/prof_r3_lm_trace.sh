set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PVLM_DRIVER_PREFIX="rocprofv3 --hip-trace --stats --output-format csv -d $O/lm_hip --" timeout 600 python $R/tools/room_like_odometry.py --scans 454 --iters 3 --lines 1 > $O/r3_room_like_lines454_hiptraced.txt 2>&1
find $O/lm_hip -name "*hip_api_stats.csv" -exec cat {} \; | cut -c1-150 | head -30
find $O/lm_hip -name "*hip_api_trace.csv" -exec cp {} $O/lm_hip_api_trace.csv \;
ls -la $O/lm_hip_api_trace.csv; find $O/lm_hip -name "*trace.csv" -size +20M -delete
