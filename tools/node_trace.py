"""Per-workgroup timeline of potrf_node_kernel launches:  LMM_NODE_TRACE=<K> python tools/trace_one_eval.py ... 2> trace.txt;
python tools/node_trace.py trace.txt [bin_us].  Per launch: durations by kind of work item (0 column-0 tile, 1 unsplit tile, 2 split-K
part, 3 bulk tile), resident workgroups over time, and the idle fraction of the 512 slots."""
import collections
import re
import sys

launches = []
for line in open(sys.argv[1]):
    if line.startswith("[node-trace] launch"):
        launches.append((line.strip(), []))
    elif line.startswith("[node-trace] wg="):
        d = dict(kv.split("=") for kv in line.split()[1:])
        launches[-1][1].append(d)
binw = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
KIND = {0: "col0", 1: "full", 2: "split", 3: "bulk"}
for head, items in launches:
    print(head)
    end = max(float(d["end_us"]) for d in items)
    by = collections.defaultdict(list)
    for d in items:
        by[int(d["kind"])].append(float(d["end_us"]) - float(d["start_us"]))
    for k, v in sorted(by.items()):
        v.sort()
        print(f"   {KIND[k]:6s} n={len(v):5d}  duration us: min {v[0]:8.1f}  median {v[len(v) // 2]:8.1f}  p90 {v[int(len(v) * 0.9)]:8.1f}  max {v[-1]:8.1f}   sum {sum(v) / 1e3:8.2f} ms")
    busy = sum(float(d["end_us"]) - float(d["start_us"]) for d in items)
    print(f"   end of launch {end:.1f} us; slot-time used {busy / 1e3:.2f} ms of {512 * end / 1e3:.2f} ms ({busy / (512 * end):.3f})")
    nb = int(end / binw) + 1
    occ = [0.0] * nb
    for d in items:
        s, e = float(d["start_us"]), float(d["end_us"])
        b0, b1 = int(s / binw), int(e / binw)
        for b in range(b0, b1 + 1):
            lo, hi = max(s, b * binw), min(e, (b + 1) * binw)
            if hi > lo:
                occ[b] += (hi - lo) / binw
    print("   resident workgroups per %.0f-us bin: " % binw + " ".join(f"{o:.0f}" for o in occ))
    cus = collections.Counter((d["xcc"], int(d["hw"]) >> 8 & 0xff) for d in items)
    per = sorted(cus.values())
    print(f"   distinct (xcc, se/sh/cu): {len(cus)}; items per CU min {per[0]} median {per[len(per) // 2]} max {per[-1]}")
