for s in 1 0 1 0 1 0; do
  echo -n "c0 strict=$s: "
  LMM_STRICT_PROGRESS=$s python bench.py --workload c0 --no-cpu-baseline --no-roofline --steps 300 --warmup 5 2>/dev/null | tail -1 | python -c "import sys, re; print(re.search(r'\"ms_per_step\": ([\d.]+)', sys.stdin.read()).group(1))"
done
