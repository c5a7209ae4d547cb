for o in "" "fuse_dir=1" "spmv_chunk_tiles=4" "spmv_chunk_tiles=16"; do
echo "== MISTARK_BENCH_OPTS=$o"; MISTARK_BENCH_OPTS=$o python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_linear_solve'], d['cg_iterations_per_solve'], d['linear_solves'], d['roofline']['live']['device_clock']['launch_ms'], d['roofline']['live']['back_to_back']['launch_ms'])"
done
