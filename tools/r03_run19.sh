export MISTARK_IPC_TIMEOUT_S=15
timeout 1700 python -m pytest tests/test_gpu_multiprocess.py -q -m gpu > gpurun_out/mp_suite.log 2>&1; echo rc=$?
tail -4 gpurun_out/mp_suite.log | cut -c1-300
timeout 900 python tools/soak.py 2>&1 | tail -4
