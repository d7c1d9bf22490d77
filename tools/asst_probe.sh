#!/bin/bash
# Assistants of the region kernel's square also in launches that are not fully resident (LMM_REGION_ASST=2) against the plan's choice.
out=gpurun_out/asst_probe.txt; : > $out
for e in 1 2 1 2; do
  echo -n "c1 LMM_REGION_ASST=$e ms_per_step " >> $out
  LMM_REGION_ASST=$e python bench.py --workload c1 --no-cpu-baseline --no-roofline --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import sys, re; print(re.search(r'\"ms_per_step\": ([\d.]+)', sys.stdin.read()).group(1))" >> $out
done
for e in 1 2; do LMM_REGION_ASST=$e python tools/share_probe.py 2>/dev/null | grep share | sed "s/^/asst=$e /" >> $out; done
cat $out
