#!/bin/bash
# Runs ON THE GPU BOX, part B: the bench lines of every workload at HEAD.
mkdir -p gpurun_out/final
for w in c0 notebook c1; do python bench.py --workload $w --steps 300 --warmup 30 > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err || exit 1; done
python bench.py --workload c1dense --steps 10 --warmup 2 > gpurun_out/final/bench_c1dense.json 2> gpurun_out/final/bench_c1dense.err
python bench.py --workload c3 --steps 5 --warmup 1 > gpurun_out/final/bench_c3.json 2> gpurun_out/final/bench_c3.err
python bench.py --workload c3 --proj bf16 --steps 5 --warmup 1 > gpurun_out/final/bench_c3_bf16proj.json 2> gpurun_out/final/bench_c3_bf16.err
python bench.py --steps 10 --warmup 2 > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
python bench.py --workload c4 --dtype f32 --steps 2 --warmup 1 > gpurun_out/final/bench_c4_f32.json 2> gpurun_out/final/bench_c4_f32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],4), round(d["value"],4), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"), (d.get("share_of_8gpu_job") or {}).get("ms_per_eval"), (d.get("roofline_region") or {}).get("frac"), (d.get("roofline_gram") or {}).get("frac_of_achievable"))
    except Exception as e: print(f, "ERR", e)
PY
