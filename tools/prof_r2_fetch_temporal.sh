#!/bin/bash
# calibration of the FETCH_SIZE counter on this kernel: the same bench command with a build of libpvlm.so whose k_eval_fused uses
# ordinary (temporal) column loads (-DPVLM_NT_LOADS=0 -> build/var/libpvlm_temporal.so).  Same bytes, same access pattern, only the hint differs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PVLM_LIB=$R/build/var/libpvlm_temporal.so timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch_t -- python $R/bench.py --no-cpu-baseline --no-mvs --no-projection --steps 10 > $O/bench_fetch_t.log 2>&1
grep '^{' $O/bench_fetch_t.log | tail -1 > $O/bench_fetch_t.json
cd $R && python tools/pmc_traffic.py $O/bench_fetch_t - $O/bench_fetch_t.json $O/r2_pmc_fetch_temporal_loads.json
find $O -name "*counter_collection.csv" -size +4M -delete
