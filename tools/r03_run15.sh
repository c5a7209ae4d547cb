timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiprocess.py > gpurun_out/gpu_suite.log 2>&1; echo rc=$?
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/gpu_suite.log | tail -30 | cut -c1-250
for o in "" "atomic_projection=1"; do
echo "== MISTARK_BENCH_OPTS=$o"; MISTARK_BENCH_OPTS=$o python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_linear_solve'], d['cg_iterations_per_solve'], d['linear_solves'], d['cg_iterations'], d['host_timers_s']['project'])"
done
