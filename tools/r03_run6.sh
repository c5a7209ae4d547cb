( time python bench.py ) > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo rc=$?
grep real gpurun_out/r03_bench_default.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms/solve", d["ms_per_linear_solve"])
r=d["roofline"]; print("frac", r["frac"], r["basis"]); print("live", json.dumps(r["live"])); print("hbm", json.dumps(r.get("hbm_resident")))
print("pinned", json.dumps(d.get("pinned_placement"))[:1500])
print("cpu", json.dumps(d["cpu_baseline"])[:1500])
PY
