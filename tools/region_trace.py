"""Summarise LMM_REGION_TRACE=1 output (stderr) of one region launch: per role (walker / helper r / row streams) start and end times.
   LMM_REGION=1024 LMM_REGION_TRACE=1 python tools/shard_classes.py 4 2> trace.txt; python tools/region_trace.py trace.txt [launch#]"""
import re, sys
launches, cur = [], None
for line in open(sys.argv[1]):
    m = re.match(r"\[region-trace\] c0=(\d+) P=(\d+) R=(\d+) nb=(\d+) occ=(\d+)", line)
    if m:
        cur = {"hdr": tuple(map(int, m.groups())), "wg": []}; launches.append(cur); continue
    m = re.match(r"\[region-trace\] wg=(\d+) b=(\d+) idx=(\d+) start_us=([\d.]+) end_us=([\d.]+)", line)
    if m and cur is not None:
        cur["wg"].append((int(m[2]), int(m[3]), float(m[4]), float(m[5])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(launches) // 2
L = launches[k]
c0, P, R, nb, occ = L["hdr"]
print(f"launch {k} of {len(launches)}: c0={c0} P={P} R={R} nb={nb} occ={occ}; {len(L['wg'])} workgroups; end of launch {max(w[3] for w in L['wg']):.1f} us")
Q = 2 * P
NA = max(0, Q - 6) if len(L["wg"]) // nb >= 2 * Q - 6 + (R - P) and Q > 6 else 0      # assistant tasks (rows 6 ..) when the launch has them
for b in range(min(nb, 2)):
    ws = [w for w in L["wg"] if w[0] == b]
    print(f" matrix {b}:")
    for (_, idx, s, e) in ws:
        if idx == 0: print(f"   walker      start {s:8.1f} end {e:8.1f}")
        elif NA and 6 <= idx < Q + NA:             # (helper r, assistant r) pairs from row 6 on
            k = idx - 6
            print(f"   {'helper' if k % 2 == 0 else 'assist'} r={6 + k // 2:2d} start {s:8.1f} end {e:8.1f}")
        elif idx < Q: print(f"   helper r={idx:2d} start {s:8.1f} end {e:8.1f}")
    rows = [w for w in ws if w[1] >= Q + NA]
    if rows:
        print(f"   rows: {len(rows)}  start min/max {min(r[2] for r in rows):.1f}/{max(r[2] for r in rows):.1f}  end min/max {min(r[3] for r in rows):.1f}/{max(r[3] for r in rows):.1f}  "
              f"duration mean {sum(r[3] - r[2] for r in rows) / len(rows):.1f}")
