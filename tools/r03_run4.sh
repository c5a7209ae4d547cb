export MISTARK_IPC_TIMEOUT_S=10
timeout 1700 python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu > gpurun_out/mp_suite.log 2>&1; echo rc=$?
tail -30 gpurun_out/mp_suite.log | cut -c1-1800
