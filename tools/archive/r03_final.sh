# the round's final validation and measurements on the GPU box (run through gpurun from the repo root)
export MISTARK_IPC_TIMEOUT_S=20
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/gpu_suite.log 2>&1; echo suite rc=$?
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/gpu_suite.log | tail -6 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh ${1:-r03_v2} 2>&1 | tail -3
for N in 2 4 8; do
MISTARK_BENCH_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2953$N tools/ipc_selftest.py 2>/dev/null | grep '^{' > gpurun_out/r03_ipc_selftest_$N.json
MISTARK_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N --steps 20 --warmup 4 --no-cpu-baseline 2>gpurun_out/r03_shard${N}_bench.err | grep '^{' > gpurun_out/r03_shard${N}_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_shard${N}_bench.json"))
print($N, round(d["value"],1), round(d["ms_per_linear_solve"],3), d["cg_iterations_per_solve"], d["linear_solves"], d["sharded_cg_kernels_us"])
PY
done
