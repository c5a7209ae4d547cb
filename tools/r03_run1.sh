set -x
export MISTARK_IPC_TIMEOUT_S=10
for N in 2 4; do
MISTARK_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2953$N tools/ipc_selftest.py > gpurun_out/ipc_selftest_$N.log 2>&1; echo rc=$?
tail -5 gpurun_out/ipc_selftest_$N.log
done
MISTARK_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 6 --warmup 2 --grid 12,12,12 --no-cpu-baseline > gpurun_out/bench_ipc2_small.log 2>&1; echo rc=$?
tail -3 gpurun_out/bench_ipc2_small.log
timeout 300 python bench.py --steps 6 --warmup 2 --grid 12,12,12 --no-cpu-baseline > gpurun_out/bench_1_small.log 2>&1; echo rc=$?
tail -1 gpurun_out/bench_1_small.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1; echo rc=$?
tail -5 gpurun_out/gpu_suite.log
