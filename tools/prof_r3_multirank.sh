# Round 3: bench.py --gpus 8 / 2 as the driver launches it, but with all ranks on the ONE GPU of the box (PVLM_BENCH_SHARED_GPU=1:
# gloo exchange — a functional check of the N > 1 control flow, watchdogs, per-rank reservation; not a measurement), then the
# default N = 1 line.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
PVLM_BENCH_SHARED_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 10 --warmup 2 > $O/r3_bench_shared_gpu_8ranks.log 2> $O/r3_bench_shared_gpu_8ranks.err
echo "rc=$?"; tail -1 $O/r3_bench_shared_gpu_8ranks.log | head -c 1500; echo; tail -5 $O/r3_bench_shared_gpu_8ranks.err
PVLM_BENCH_SHARED_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 2 --no-mvs > $O/r3_bench_shared_gpu_2ranks.log 2> $O/r3_bench_shared_gpu_2ranks.err
echo "rc=$?"; tail -1 $O/r3_bench_shared_gpu_2ranks.log | head -c 600; echo
timeout 900 python bench.py > $O/r3_bench_default.log 2>$O/r3_bench_default.err
echo "rc=$?"; tail -1 $O/r3_bench_default.log > $O/r3_bench_default.json; head -c 700 $O/r3_bench_default.json; echo
