#!/bin/bash
for sc in 1.0 0.8 0.65 0.5 0.4 1.3; do
  echo -n "cell scale $sc: "; PVLM_CELL_SCALE=$sc python tools/assoc_workload.py --scans 128 --calls 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wall', [round(w*1e3,1) for w in d['wall_s']], 'accepted', d['accepted'])"
done
for sc in 1.0 0.65 0.5; do
  echo -n "raw targets, cell scale $sc: "; PVLM_CELL_SCALE=$sc python tools/assoc_workload.py --scans 32 --calls 3 --targets raw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wall', [round(w*1e3,1) for w in d['wall_s']], 'accepted', d['accepted'])"
done
