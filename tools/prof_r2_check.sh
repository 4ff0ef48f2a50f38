#!/bin/bash
mkdir -p gpurun_out/r2
python bench.py --scans 64 --no-cpu-baseline 2>gpurun_out/r2/b64.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N1', d['value'], d['roofline']['frac'], list(d['mvs'].keys()), d['mvs']['5760x2880']['k13_patchmatch_iteration']['ms'], d['panorama']['image_to_cam_f32']['frac_of_hbm_peak'])"
tail -3 gpurun_out/r2/b64.err
PVLM_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --scans 64 > gpurun_out/r2/bench_2rank_shared.json 2> gpurun_out/r2/bench_2rank_shared.err
echo "rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|Gloo\|\*\*\*\|OMP_NUM" gpurun_out/r2/bench_2rank_shared.err | tail -5; python -c "
import json
d=json.loads(open('gpurun_out/r2/bench_2rank_shared.json').read().strip().splitlines()[-1]); print('N2 shared', d['value'], d['n_gpus'], d['step'], json.dumps(d['image_space_all_ranks']))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
