#!/bin/bash
# Timing ablations of gemm16p_kernel on the K = 8192 SYRK level (nb = 8): which instruction class costs what.  Runs ON THE GPU BOX.
# Build first (here or there): for v in FULL NOREAD NOLDSW NOGLOBAL NOBAR NOREAD_NOLDSW_NOGLOBAL_NOBAR; do ... done (see below)
set -o pipefail
for v in FULL NOREAD NOLDSW NOGLOBAL NOBAR ALLOFF; do
  echo "== $v"; timeout -k 5 120 tools/gemm16_ablate_$v 8 0 2 | grep -o "shipped): [0-9.]* ms [0-9.]* TF"
done
