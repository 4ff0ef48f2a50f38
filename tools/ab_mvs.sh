#!/bin/bash
# A/B of MVS sweep variants (build/var/libpvlm_<tag>.so, "base" = the in-tree library), same box, interleaved twice:
# sequential sweep (default launch form) and checkerboard colour pass at 1440 x 720, 4 neighbours.  VARIANTS="base onesite" tools/ab_mvs.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/ab_mvs.txt; : > $out
for rep in 1 2; do
  for v in ${VARIANTS:-base}; do
    lib=panovlm_amd/libpvlm.so; [ $v != base ] && lib=build/var/libpvlm_$v.so
    for mode in "" "--checkerboard"; do
      PVLM_LIB=$PWD/$lib python tools/mvs_seq_bench.py $mode ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %-12s %8.2f ms  %s' % ('$v', d['sweep'], d['ms_per_iteration'], d['sha256']))" >> $out
    done
  done
done
cat $out
