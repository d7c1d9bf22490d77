mkdir -p gpurun_out/r3fm
for i in 1 2; do
  (cd _ab_old && python bench.py --steps 5 --warmup 2 --no-cpu-baseline > ../gpurun_out/r3fm/old_$i.json 2>/dev/null)
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3fm/new_$i.json 2>/dev/null
  LMM_FUSE_BULK=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3fm/new_nofuse_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3fm/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["ms_per_step"],2), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["share_of_8gpu_job"]["ms_per_eval"])
PY
