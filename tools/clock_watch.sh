#!/bin/bash
# Samples the shader clock and power while an update-kernel binary runs (is the loaded kernel clock- or power-limited?).  Runs ON THE GPU BOX.
for v in FULL NOGLOBAL; do
  echo "== $v"
  ( for i in 1 2 3 4 5 6; do timeout -k 5 60 tools/gemm16_ablate_$v 8 0 2 > /dev/null; done ) &
  BG=$!
  sleep 2
  for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Socket" | tr '\n' ';' ; echo; sleep 1; done
  wait $BG
done
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Socket" | tr '\n' ';'; echo
