export MISTARK_IPC_TIMEOUT_S=20
for N in 2 4 8; do
MISTARK_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2953$N tools/ipc_selftest.py 2>/dev/null | grep '^{' > gpurun_out/r03_ipc_selftest_$N.json
cat gpurun_out/r03_ipc_selftest_$N.json
MISTARK_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03_shard${N}_bench.err | grep '^{' > gpurun_out/r03_shard${N}_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_shard${N}_bench.json"))
print($N, d["value"], d["ms_per_linear_solve"], d["cg_iterations_per_solve"], d["linear_solves"], d["sharded_cg_kernels_us"])
PY
done
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline | grep '^{' > gpurun_out/r03_shard1_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_shard1_bench.json"))
print(1, d["value"], d["ms_per_linear_solve"], d["cg_iterations_per_solve"], d["linear_solves"], d["host_timers_s"])
PY
