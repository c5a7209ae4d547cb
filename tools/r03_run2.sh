export MISTARK_IPC_TIMEOUT_S=15
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_multiprocess.py -x -q -m gpu > gpurun_out/sharded_suite.log 2>&1; echo rc=$?
tail -30 gpurun_out/sharded_suite.log | cut -c1-400
