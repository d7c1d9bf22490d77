"""One evaluation's kernel timeline from a rocprofv3 --kernel-trace CSV of a repeated small workload: python tools/trace_one_eval.py trace.csv [first_kernel_substr]
Prints the kernels of the LAST complete evaluation with start offsets, durations and the gaps between them."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "tall_skinny"
starts = [i for i, r in enumerate(rows) if key in r[2] and (i == 0 or key not in rows[i - 1][2])]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]; prev_end = t0
for s, e, n in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {n}")
    prev_end = max(prev_end, e)
print(f"evaluation span (first kernel start -> next evaluation's first kernel start): {(rows[b][0] - t0) / 1e3:.1f} us")
