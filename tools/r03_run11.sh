timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiprocess.py > gpurun_out/gpu_suite.log 2>&1; echo rc=$?
tail -40 gpurun_out/gpu_suite.log | cut -c1-400
python bench.py --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_linear_solve'], d['cg_iterations_per_solve'], d['linear_solves'], d['roofline']['live'])"
MISTARK_OPTIONS=no_row_order=1 python bench.py --no-cpu-baseline --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_linear_solve'], d['cg_iterations_per_solve'], d['linear_solves'], d['roofline']['live'])"
