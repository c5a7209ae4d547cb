export MISTARK_IPC_TIMEOUT_S=5
timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "windows and tetbeam_full_4x1x1-2" > gpurun_out/sharded_dbg.log 2>&1; echo rc=$?
grep -n "EngineError\|passed\|failed" gpurun_out/sharded_dbg.log | head
