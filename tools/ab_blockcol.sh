#!/bin/bash
# Runs ON THE GPU BOX: A/B of the round-4 block-column factorisation (LMM_BLOCKCOL=1, default) against the round-3 recursion (=0):
# values at mid sizes (must agree to rounding), the share / N = 1 class times of C2, and mid-size wall times.
set -o pipefail
OUT=${1:-gpurun_out/r4b}
mkdir -p $OUT
for V in 0 1; do
  echo "== LMM_BLOCKCOL=$V mid" >> $OUT/ab.txt
  LMM_BLOCKCOL=$V timeout -k 10 300 python tools/mid_probe.py 1536 8 2048 8 3072 8 4096 8 8192 8 2048 16 5000 3 >> $OUT/ab.txt 2>&1 || exit 1
done
for V in 0 1 2; do
  echo "== LMM_BLOCKCOL=$V share" >> $OUT/ab.txt
  LMM_BLOCKCOL=$V timeout -k 10 400 python tools/share_profile.py 8 4 1 >> $OUT/ab.txt 2>&1 || exit 1
done
for SQ in region panel; do
  echo "== LMM_BLOCKCOL=1 LMM_BLOCKCOL_SQ=$SQ share" >> $OUT/ab.txt
  LMM_BLOCKCOL_SQ=$SQ timeout -k 10 400 python tools/share_profile.py 8 1 >> $OUT/ab.txt 2>&1 || exit 1
done
cat $OUT/ab.txt
