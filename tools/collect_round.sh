bash tools/collect_profiles.sh r03b > gpurun_out/collect_r03b.log 2>&1
mkdir -p gpurun_out/r3sec2
for w in c0 notebook c1; do python bench.py --workload $w --steps 300 --warmup 30 > gpurun_out/r3sec2/bench_$w.json 2> gpurun_out/r3sec2/bench_$w.err || exit 1; done
python bench.py --workload c1dense --steps 10 --warmup 2 > gpurun_out/r3sec2/bench_c1dense.json 2> gpurun_out/r3sec2/bench_c1dense.err
python bench.py --workload c3 --steps 3 --warmup 1 > gpurun_out/r3sec2/bench_c3.json 2> gpurun_out/r3sec2/bench_c3.err
python bench.py --workload c3 --proj bf16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3sec2/bench_c3_bf16proj.json 2> gpurun_out/r3sec2/bench_c3_bf16.err
python bench.py --steps 10 --warmup 2 > gpurun_out/r3sec2/bench_c2.json 2> gpurun_out/r3sec2/bench_c2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3sec2/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],4), d["value"], (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
