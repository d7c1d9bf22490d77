"""Summarise LMM_REGION_TRACE output (stderr of a run) per region launch: start / duration statistics of the square tasks and of the
row streams.  python tools/region_rows_stats.py trace.txt"""
import re, sys
import numpy as np
launch = None; launches = []
for line in open(sys.argv[1]):
    m = re.match(r"\[region-trace\] c0=(\d+) P=(\d+) R=(\d+) nb=(\d+) occ=(\d+)(?: square=(\d+) n128=(\d+))?", line)
    if m:
        launch = {"c0": int(m[1]), "P": int(m[2]), "R": int(m[3]), "nb": int(m[4]), "wg": [], "sq": int(m[6]) if m[6] else None,
                  "n128": int(m[7]) if m[7] else None}; launches.append(launch); continue
    m = re.match(r"\[region-trace\] wg=(\d+) b=(\d+) idx=(\d+) start_us=([\d.]+) end_us=([\d.]+)", line)
    if m and launch is not None: launch["wg"].append((int(m[3]), float(m[4]), float(m[5])))
for L in launches:
    P, R, nb = L["P"], L["R"], L["nb"]
    ntask = len(L["wg"]) // nb
    nsq = L["sq"] if L["sq"] is not None else ntask - (R - P)
    sq = np.array([(s, e) for i, s, e in L["wg"] if i < nsq]); rw = np.array([(s, e) for i, s, e in L["wg"] if i >= nsq])
    out = f"c0={L['c0']:6d} P={P} R={R:3d} nb={nb} tasks/matrix={ntask} (square {nsq}, n128 {L['n128']}): kernel end {max(e for _, _, e in L['wg']):8.1f} us; square end {sq[:, 1].max():7.1f}"
    if len(rw):
        d = rw[:, 1] - rw[:, 0]
        first = rw[rw[:, 0] < 50.0]
        out += f"; rows n={len(rw)} start p50 {np.median(rw[:, 0]):7.1f} max {rw[:, 0].max():7.1f}; duration min {d.min():6.1f} p50 {np.median(d):6.1f} max {d.max():6.1f}"
        late = rw[rw[:, 0] > sq[:, 1].max()]
        if len(late): out += f"; started after the square: n={len(late)} duration p50 {np.median(late[:, 1] - late[:, 0]):6.1f}"
    print(out)
