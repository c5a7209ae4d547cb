export MISTARK_SHIM_STATS=1 SHIM_THREADS=16
timeout 600 python -m pytest tests/test_gpu_scene.py -x -q -m gpu -k shim 2>&1 | tail -3
for scene in benchclamped benchblock; do
for grid in 10,10,10 44,44,43; do
echo "== shim $scene $grid"; SHIM_GRID=$grid timeout 900 oracle/_ref/shim_check $scene 6 2>&1 | grep -v "^shim_check" | tail -3
done; done
echo "== mirror clamped 10"; python bench.py --scene clamped --grid 10,10,10 --no-cpu-baseline --steps 20 --warmup 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
echo "== mirror clamped 1M"; python bench.py --scene clamped --no-cpu-baseline --steps 20 --warmup 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
echo "== mirror contact 10"; python bench.py --grid 10,10,10 --no-cpu-baseline --steps 20 --warmup 4 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
