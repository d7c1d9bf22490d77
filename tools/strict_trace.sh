#!/bin/bash
# Region-launch timelines of an 8 x 4096 evaluation with and without the claim-based task assignment.
set -e
cat > /tmp/st_run.py <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, lmm_amd
from lmm_amd import workloads as O
lmm_amd.init(0)
P = O.synthetic_problem(8, 10, 4096, "matern52", True, s2=0.1, seed=0)
f = lmm_amd.ILMM(lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(8)]), lmm_amd.Orthogonal(P["U"], P["S"]))
fx = f(lmm_amd.MOInputIsotopicByOutputs(P["x"], 10), 0.1)
print(lmm_amd.logpdf(fx, P["y"]))
PY
for s in 1 0; do
  LMM_STRICT_PROGRESS=$s LMM_REGION_TRACE=1 python /tmp/st_run.py 2> gpurun_out/strict_trace_$s.txt
  for k in 0 1 2 3; do python tools/region_trace.py gpurun_out/strict_trace_$s.txt $k > gpurun_out/strict_trace_${s}_launch$k.txt; done
  echo "== strict=$s"; head -1 gpurun_out/strict_trace_${s}_launch*.txt | grep launch
done
