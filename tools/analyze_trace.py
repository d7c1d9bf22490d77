#!/usr/bin/env python3
"""Post-process a rocprofv3 --kernel-trace CSV: per-kernel stats, union-busy time, and overlap, optionally split
into the concurrent (timed) region and the serial-stream instrumented pass of bench.py."""
import csv, sys, collections
path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], int(r.get("Queue_Id", 0) or 0)))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"dispatches {len(rows)}  span {(t1 - t0) / 1e6:.1f} ms")
def stats(sel, label):
    by = collections.defaultdict(list)
    for s, e, n, q in sel: by[n].append(e - s)
    print(f"--- {label}: {len(sel)} dispatches, span {(max(e for _, e, _, _ in sel) - min(s for s, _, _, _ in sel)) / 1e6:.1f} ms")
    # union busy
    ev = sorted([(s, 1) for s, e, n, q in sel] + [(e, -1) for s, e, n, q in sel])
    busy = 0; depth = 0; last = None; wsum = 0
    for t, d in ev:
        if depth > 0: busy += t - last; wsum += (t - last) * depth
        depth += d; last = t
    print(f"    union busy {busy / 1e6:.1f} ms, mean concurrency while busy {wsum / max(busy, 1):.2f}")
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"    {n:60s} calls {len(v):6d} total {sum(v) / 1e6:9.2f} ms avg {sum(v) / len(v) / 1e3:9.1f} us")
stats(rows, "all")
qs = collections.Counter(q for *_, q in rows)
print("queues:", dict(qs))
