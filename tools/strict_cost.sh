#!/bin/bash
# Cost of the strict-progress mode (tasks by arrival ticket): the bench lines of the small / mid shapes and a rank's share, default vs strict.
set -e
out=gpurun_out/strict_cost.txt; : > $out
for w in c0 notebook c1; do
  for s in 0 1; do
    echo "== workload $w LMM_STRICT_PROGRESS=$s" >> $out
    LMM_STRICT_PROGRESS=$s python bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200 >> $out
  done
done
for s in 0 1; do LMM_STRICT_PROGRESS=$s python tools/share_probe.py 2>/dev/null | grep share | sed "s/^/strict=$s /" >> $out; done
cat $out
