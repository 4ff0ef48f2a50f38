#!/bin/bash
# A/B of streaming-access variants of the residual / map kernels.  Variant libraries (build/var/libpvlm_<tag>.so, built here, without a GPU, by
#   python -m panovlm_amd.build --variant temporal -DPVLM_NT_LOADS=0        (likewise -DPVLM_GLOBAL_LOADS / -DPVLM_NT_STORES / -DPVLM_NT_MAPS / ...)
# are selected through PVLM_LIB; "base" = the in-tree library.
# Prints per variant: value, roofline.frac, fused kernel ms, materialise GB/s, CamToImage / ImageToCam fraction of the HBM peak, wrench rows M/s.
mkdir -p gpurun_out/r2
for v in ${VARIANTS:-base nost base nost}; do
  lib=panovlm_amd/libpvlm.so; [ $v != base ] && lib=build/var/libpvlm_$v.so
  PVLM_LIB=$PWD/$lib timeout 170 python bench.py --no-mvs --no-projection --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | \
    python -c "
import sys, json
o = json.loads(sys.stdin.read()); p = o.get('panorama') or {}; m = o.get('materialise') or {}; pc = o.get('pcie') or {}
print('$v', round(o['value'], 1), round(o['roofline']['frac'], 4), round(o['roofline']['kernel_avg_ms'], 4), 'mat', round(m.get('GBps', 0), 1),
      'c2i', round((p.get('cam_to_image_f32') or {}).get('frac_of_hbm_peak', 0), 4), 'i2c', round((p.get('image_to_cam_f32') or {}).get('frac_of_hbm_peak', 0), 4),
      'pcie', json.dumps({k: round(v['M_evals_per_s_host_visible'], 1) for k, v in pc.items() if isinstance(v, dict) and 'M_evals_per_s_host_visible' in v}))"
done | tee gpurun_out/r2/ab_eval_loads.txt
