timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_multiprocess.py > gpurun_out/gpu_suite.log 2>&1; echo rc=$?
tail -25 gpurun_out/gpu_suite.log | cut -c1-300
