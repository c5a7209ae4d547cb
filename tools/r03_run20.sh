export MISTARK_IPC_TIMEOUT_S=15
timeout 1700 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu -x > gpurun_out/mp_suite.log 2>&1; echo rc=$?
tail -4 gpurun_out/mp_suite.log | cut -c1-600
for N in 4 8; do
MISTARK_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03_shard${N}_bench.err | grep '^{' > gpurun_out/r03_shard${N}_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_shard${N}_bench.json"))
print($N, d["value"], d["ms_per_linear_solve"], d["cg_iterations_per_solve"], d["linear_solves"], d["sharded_cg_kernels_us"])
PY
done
