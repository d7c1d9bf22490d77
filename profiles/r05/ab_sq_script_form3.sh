#!/bin/bash
set -o pipefail
OUT=${1:-gpurun_out/r5c}
mkdir -p $OUT
: > $OUT/ab.txt
for V in "LMM_SQ_FUSE=0" "LMM_SQ_FUSE=1" "LMM_SQ_FUSE=1 LMM_SQ_MIN_MS=1.2"; do
  echo "== $V" >> $OUT/ab.txt
  env $V timeout -k 10 300 python tools/classes_probe.py 16384 4 8192 8 4096 8 16384 16 >> $OUT/ab.txt 2>&1 || { tail -5 $OUT/ab.txt; exit 1; }
done
echo "== per-launch LMM_SQ_FUSE=1 m=4" >> $OUT/ab.txt
LMM_PROF_DUMP=1 timeout -k 10 200 python tools/classes_probe.py 16384 4 2>&1 | grep -E "cls=(1|5|6) " >> $OUT/ab.txt
grep -v "^\[prof\]" $OUT/ab.txt | grep -v amdgpu.ids
