#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs -> profiles/pmc_traffic.json (bytes per launch per kernel).
FETCH_SIZE (KiB) is doubled (gfx950 counts 128-B requests at 64 B: MI355X_MICROARCH.md section HBM); WRITE_SIZE (KiB)
is exact.  Infinity-Cache hits are counted, so this is fabric (L2-miss) traffic, an upper bound on HBM bytes."""
import csv, json, sys, collections
# the dominant kernel under a stable key (bench.py reads it for roofline.traffic): the wide trailing update.  One update "launch" of
# the library (lmm_api.hip potrf_rec) is one gemm16p_kernel<DEPTH> launch plus, for a ragged last 64 rows, one gemm16h_kernel launch:
# bytes of both, divided by the gemm16p launches (the count bench.py's roofline.launches uses).
def aggregate_update(out):
    # round 3: the dominant kernel is potrf_node_kernel<2> (K >= 1024 updates + the next panel's leaf) plus gemm16h_kernel<true> on the
    # ragged last 64 rows of the same update: bytes of both per potrf_node_kernel<2> launch (= bench.py's roofline.launches)
    # (its two instantiations: <2, false>, and <2, true> when the next panel's bulk rows ride in the launch)
    node = [k for k in out if isinstance(out[k], dict) and (k.startswith("potrf_node_kernel<2>") or k.startswith("potrf_node_kernel<2,"))]
    if node:
        parts = node + [k for k in out if isinstance(out[k], dict) and k.startswith("gemm16h_kernel<true>")]
        n = sum(out[k]["launches"] for k in node)
        agg = {f: sum(out[k][f] * out[k]["launches"] for k in parts) / n
               for f in ("fetch_bytes_per_launch_x2", "write_bytes_per_launch", "bytes_per_launch")}
        out["update_kernel"] = dict(agg, launches=n, kernel=" + ".join(parts))
        return
    wide = [k for k in out if isinstance(out[k], dict) and k.startswith("gemm16p_kernel")]
    if not wide:
        wide = [k for k in out if isinstance(out[k], dict) and k.startswith("gemm44_kernel<128, false")]
    if not wide:
        return
    parts = wide + [k for k in out if isinstance(out[k], dict) and k.startswith("gemm16h_kernel")]
    # update launches = gemm16p launches + the launches made of 64-row tiles alone (gemm16h_kernel<false>: underfilled updates)
    n = sum(out[k]["launches"] for k in wide) + sum(out[k]["launches"] for k in out if isinstance(out[k], dict) and k.startswith("gemm16h_kernel<false>"))
    agg = {f: sum(out[k][f] * out[k]["launches"] for k in parts) / n
           for f in ("fetch_bytes_per_launch_x2", "write_bytes_per_launch", "bytes_per_launch")}
    out["update_kernel"] = dict(agg, launches=n, kernel=" + ".join(parts))
if len(sys.argv) == 3 and sys.argv[1] == "--reaggregate":      # recompute update_kernel of an existing file in place
    out = json.load(open(sys.argv[2])); out.pop("update_kernel", None); aggregate_update(out)
    json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out["update_kernel"], indent=1)); sys.exit(0)
def load(path, name):
    by = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            by[k][0] += 1; by[k][1] += float(r["Counter_Value"])
    return by
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"workload": sys.argv[3], "note": __doc__.strip().split("\n", 1)[1]}
for k in f:
    fb = 2 * f[k][1] * 1024 / f[k][0]; wb = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
    out[k] = {"launches": f[k][0], "fetch_bytes_per_launch_x2": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb}
aggregate_update(out)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "gemm" in k or "gram" in k or "potrf" in k or "leaf" in k or k == "update_kernel"}, indent=1))
