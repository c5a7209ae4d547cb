timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiprocess.py > gpurun_out/gpu_suite.log 2>&1; echo rc=$?
tail -30 gpurun_out/gpu_suite.log | cut -c1-300 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
for o in "" "row_order=1" "no_row_order=1"; do
echo "== MISTARK_OPTIONS=$o"; MISTARK_OPTIONS=$o python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_linear_solve'], d['cg_iterations_per_solve'], d['linear_solves'], d['roofline']['live']['device_clock']['launch_ms'], d['roofline']['live']['back_to_back']['launch_ms'])"
done
export MISTARK_SHIM_STATS=1 SHIM_THREADS=16
for scene in benchclamped benchblock; do
echo "== shim $scene 1M"; SHIM_GRID=44,44,43 timeout 900 oracle/_ref/shim_check $scene 6 2>&1 | grep -v "^shim_check" | tail -3
done
