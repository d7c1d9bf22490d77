#!/bin/bash
# Runs ON THE GPU BOX: quick same-box A/B (prev vs new library) of the region-kernel regime + a walker trace of the new build.
set -o pipefail
OUT=${1:-gpurun_out/ab}; mkdir -p $OUT; : > $OUT/ab.txt
PREV=$PWD/tools/ab/liblmm_hip_prev.so
for round in 1 2; do for V in prev new; do
  if [ $V = prev ]; then export LMM_HIP_LIB=$PREV; else unset LMM_HIP_LIB; fi
  echo "== $V (round $round)" >> $OUT/ab.txt
  timeout -k 10 300 python tools/mid_probe.py 200 3 552 20 1024 4 1024 32 2048 8 4096 8 >> $OUT/ab.txt 2>&1 || exit 1
done; done
unset LMM_HIP_LIB
LMM_REGION_TRACE=1 timeout -k 10 100 python tools/region_one.py 1024 4 2> $OUT/trace_1024.txt > /dev/null
grep "region-walker" $OUT/trace_1024.txt | tail -16 >> $OUT/ab.txt
grep -v amdgpu.ids $OUT/ab.txt
