#!/bin/bash
# Runs ON THE GPU BOX: A/B of the round-5 square-in-the-update-launch path (LMM_SQ_FUSE=1, default) against the round-4 path (=0):
# values must agree to rounding; class times of the share / N = 1 batch; mid sizes.
set -o pipefail
OUT=${1:-gpurun_out/r5b}
mkdir -p $OUT
: > $OUT/ab.txt
for V in 0 1; do
  echo "== LMM_SQ_FUSE=$V classes (n m)" >> $OUT/ab.txt
  LMM_SQ_FUSE=$V timeout -k 10 300 python tools/classes_probe.py 16384 4 4096 8 8192 8 3072 8 16384 16 >> $OUT/ab.txt 2>&1 || { tail -5 $OUT/ab.txt; exit 1; }
done
echo "== LMM_SQ_FUSE=1 per-launch, share" >> $OUT/ab.txt
LMM_PROF_DUMP=1 LMM_SQ_FUSE=1 timeout -k 10 200 python tools/classes_probe.py 16384 4 2>&1 | grep -E "cls=(1|5|6) " >> $OUT/ab.txt
for MS in 0.3 1.0; do
  echo "== LMM_SQ_FUSE=1 LMM_SQ_MIN_MS=$MS" >> $OUT/ab.txt
  LMM_SQ_MIN_MS=$MS timeout -k 10 300 python tools/classes_probe.py 16384 4 4096 8 8192 8 >> $OUT/ab.txt 2>&1 || exit 1
done
grep -v "^\[prof\]" $OUT/ab.txt | grep -v amdgpu.ids
