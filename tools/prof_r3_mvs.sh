# Round 3, MVS sequential sweep: four waves per pixel with speculative batches (k_mvs_propagate_diag_spec) against round 2's wave per pixel
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mvs_gpu.py tests/test_mvs_5p7k_gpu.py -x -q -m gpu 2>&1 | tail -4
for S in 1 0; do
  PVLM_MVS_SPEC=$S timeout 600 python tools/mvs_bench.py > $O/mvs_bench_spec$S.json 2> $O/mvs_bench_spec$S.err
  python - <<P
import json
d=json.load(open("$O/mvs_bench_spec$S.json"))
print("PVLM_MVS_SPEC=$S", "K13s", d.get("sweep_sequential"), "K13 ms/colour", d["sweep"]["kernel_ms_per_colour_pass"])
P
done
