#!/bin/bash
# Runs ON THE GPU BOX: round-5 measurement items (VERDICT r4 "next round" 3b-3f).
OUT=gpurun_out/r5m
mkdir -p $OUT
python bench.py --steps 10 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err || exit 1
for w in c0 notebook c1; do python bench.py --workload $w --steps 300 --warmup 30 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || exit 1; done
LMM_BENCH_BACKEND=gloo LMM_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_2rank_selfspawn_gloo_shared_gpu.json 2> $OUT/bench_2rank.err || { tail -5 $OUT/bench_2rank.err; exit 1; }
timeout -k 10 300 python tools/gram_touch_probe.py > $OUT/gram_touch_probe.txt 2>&1 || { tail -5 $OUT/gram_touch_probe.txt; exit 1; }
timeout -k 10 400 python tools/c3_phases.py > $OUT/c3_phases.txt 2>&1 || { tail -5 $OUT/c3_phases.txt; exit 1; }
timeout -k 10 600 python bench.py --workload c4 --dtype f32 --steps 2 --warmup 1 > $OUT/bench_c4_f32.json 2> $OUT/bench_c4_f32.err || { tail -5 $OUT/bench_c4_f32.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5m/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],4), round(d["value"],4), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("sample"), (d.get("share_of_8gpu_job") or {}).get("ms_per_eval"), (d.get("roofline_gram") or {}).get("achievable_write_gbs"))
    except Exception as e: print(f, "ERR", e)
PY
cat $OUT/gram_touch_probe.txt $OUT/c3_phases.txt
