"""tools/stress_alternate.py for the large-matrix paths (panel recursion, fused bulk rows, split-K atomics, region base case): two
DIFFERENT problems of each shape in turn; values must agree with their first evaluation to 1e-11 relative (split-K atomics reorder sums)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O
lmm_amd.init(0)
bad = 0
for (n, m, rounds) in [(3072, 8, 12), (8192, 8, 8), (8192, 16, 6), (16384, 4, 6), (16384, 16, 3)]:
    probs = []
    for seed in (1, 2):
        P = O.synthetic_problem(m, 2 * m, n, "matern52" if seed == 1 else "matern32", True, s2=0.1 * seed, seed=seed)
        fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel(1.0 + 0.1 * seed, 1.0 + 0.2 * seed)) for _ in range(m)])
        fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1 * seed)
        probs.append((fx, torch.from_numpy(P["y"]).cuda()))
    ref = [lmm_amd.logpdf(fx, y) for fx, y in probs]
    worst = 0.0
    for it in range(rounds):
        for k, (fx, y) in enumerate(probs):
            v = lmm_amd.logpdf(fx, y)
            worst = max(worst, abs(v - ref[k]) / abs(ref[k]))
    print(f"n={n} m={m}: {rounds} rounds x 2 problems, worst relative deviation from the first evaluation {worst:.2e}", flush=True)
    bad += worst > 1e-11
sys.exit(1 if bad else 0)
