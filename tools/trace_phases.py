"""Split a rocprofv3 rocpd kernel trace (sqlite .db) into host-separated phases and print each phase's kernel totals.
    python tools/trace_phases.py gpurun_out/sec_prof/sec_results.db [min_ms]"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r'\(.*', '', n)[:44]
t0 = rows[0][1]; phases = []; cur = [rows[0]]
for r in rows[1:]:
    if r[1] - max(x[2] for x in cur[-8:]) > 400e3: phases.append(cur); cur = [r]
    else: cur.append(r)
phases.append(cur)
for ph in phases:
    dur = (max(x[2] for x in ph) - ph[0][1]) / 1e6
    if dur < min_ms: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e in ph:
        a = agg[short(n)]; a[0] += 1; a[1] += (e - s) / 1e6
    print(f"--- phase at {(ph[0][1]-t0)/1e6:9.1f} ms  dur {dur:8.2f} ms  kernels {len(ph)}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"     {k:46s} n={v[0]:5d} sum={v[1]:8.2f} ms  avg={v[1]/v[0]*1e3:8.1f} us")
