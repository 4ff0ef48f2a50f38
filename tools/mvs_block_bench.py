#!/usr/bin/env python3
"""The `mvs` block of bench.py on its own (resident views at 1440 x 720 and 5760 x 2880): K11, one K13 iteration, K12, the sequential sweep."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import panovlm_amd as pv
ctx = pv.Context(0)
if "--small" in sys.argv:          # 1440 x 720 only (counter passes: one pixel count per kernel)
    out = {"1440x720": bench.mvs_one_size(ctx, 720, 1440)}
else:
    out = bench.mvs_block(ctx, pv)
    out.pop("pmc", None)
print(json.dumps(out))
