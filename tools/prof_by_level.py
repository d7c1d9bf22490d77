"""Aggregate the per-launch lines of LMM_PROF_DUMP=1 (stderr of an instrumented pass: tools/share_profile.py, bench.py) by kernel
class and K:   LMM_PROF_DUMP=1 python tools/share_profile.py 8 2> dump.txt; python tools/prof_by_level.py dump.txt"""
import collections
import re
import sys

CLS = {0: "gram", 1: "update(+leaf)", 2: "update_narrow", 3: "trsm/bulk", 4: "diag/leaf", 5: "region", 6: "update_short"}
acc = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"\[prof\] cls=(\d+) M=(\d+) N=(\d+) K=(\d+) ms=([\d.]+) tflops=([\d.]+)", line)
    if not m:
        continue
    c, M, N, K, ms, tf = int(m[1]), int(m[2]), int(m[3]), int(m[4]), float(m[5]), float(m[6])
    a = acc.setdefault((c, K), [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += tf * ms
for (c, K), (n, ms, w) in sorted(acc.items()):
    print(f"cls {c} {CLS.get(c, '?'):14s} K={K:6d}: {n:4d} launches {ms:9.3f} ms  avg {ms / n * 1e3:9.1f} us  {w / ms:7.2f} TFLOP/s")
