set -x
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scene.py tests/test_attach_by_distance.py tests/test_gpu_shim_user_potentials.py tests/test_gpu_contact.py -m gpu -q 2>&1 | tail -8
for pin in 0 1; do
SHIM_GRID=44,44,43 MISTARK_SHIM_STATS=1 SHIM_THREADS=1 MISTARK_SHIM_NO_PIN=$((1-pin)) timeout 300 oracle/_ref/shim_check_cd benchblock 16 > gpurun_out/r06_dropin_pin$pin.txt 2>&1
tail -6 gpurun_out/r06_dropin_pin$pin.txt
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['host_timers_s'])"
