#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of two builds of the library (tools/ab/liblmm_hip_prev.so = the previous kernels, the in-tree
# liblmm_hip.so = the current ones) on the small / mid / share workloads.   bash tools/ab_lib.sh gpurun_out/<tag>
set -o pipefail
OUT=${1:-gpurun_out/ab}
mkdir -p $OUT
: > $OUT/ab.txt
PREV=$PWD/tools/ab/liblmm_hip_prev.so
for round in 1 2; do
for V in prev new; do
  if [ $V = prev ]; then export LMM_HIP_LIB=$PREV; else unset LMM_HIP_LIB; fi
  echo "== $V (round $round): mid sizes" >> $OUT/ab.txt
  timeout -k 10 300 python tools/mid_probe.py 200 3 552 20 1024 4 1024 32 2048 8 4096 8 >> $OUT/ab.txt 2>&1 || exit 1
  for W in c0 notebook; do
    timeout -k 10 300 python bench.py --workload $W --steps 2000 --warmup 200 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V $W', round(d['ms_per_step'], 4), 'ms/eval', d.get('logpdf'))" >> $OUT/ab.txt || exit 1
  done
done
done
for V in prev new; do
  if [ $V = prev ]; then export LMM_HIP_LIB=$PREV; else unset LMM_HIP_LIB; fi
  echo "== $V: share" >> $OUT/ab.txt
  timeout -k 10 400 python tools/share_profile.py 8 1 >> $OUT/ab.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $OUT/ab.txt
