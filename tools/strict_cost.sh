#!/bin/bash
# Cost of the strict-progress mode (tasks claimed at workgroup entry): bench lines of the small / mid shapes and a rank's share, strict vs index order.
set -e
out=gpurun_out/strict_cost.txt; : > $out
for w in c0 notebook c1; do
  for s in 1 0 1 0; do
    echo -n "workload $w LMM_STRICT_PROGRESS=$s ms_per_step " >> $out
    LMM_STRICT_PROGRESS=$s python bench.py --workload $w --no-cpu-baseline --no-roofline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys, re; print(re.search(r'\"ms_per_step\": ([\d.]+)', sys.stdin.read()).group(1))" >> $out
  done
done
for s in 1 0; do LMM_STRICT_PROGRESS=$s python tools/share_probe.py 2>/dev/null | grep share | sed "s/^/strict=$s /" >> $out; done
cat $out
