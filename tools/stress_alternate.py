"""Alternate DIFFERENT problems of the same shapes so that whatever a pooled buffer still holds from the previous call is WRONG for
the current one: a consumer that reads ahead of its producer then returns a different value (with one problem repeated, stale and fresh
data coincide and such a race stays invisible).   python tools/stress_alternate.py [reps] [all]   ("all": logpdf, gradient, posterior
marginals and posterior samples at small and mid shapes; run with LMM_DETERMINISTIC=1)"""
import sys
sys.path.insert(0, '.')
import numpy as np, lmm_amd
lmm_amd.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def problem(seed, n, d, p=4, m=3):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0, 6, n)) if d == 1 else rng.uniform(0, 4, size=(d, n))
    kinds = [lmm_amd.Matern52Kernel, lmm_amd.SEKernel, lmm_amd.Matern32Kernel]
    gps = [lmm_amd.GP(float(rng.normal()), kinds[l % 3](float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))) for l in range(m)]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    y = rng.standard_normal(n * p)
    fx = lmm_amd.ILMM(lmm_amd.independent_mogp(gps), lmm_amd.Orthogonal(U, S))(lmm_amd.MOInputIsotopicByOutputs(x, p), 0.3)
    return fx, y
def digest(fx, y, xs, which):
    if which == 0: return lmm_amd.logpdf(fx, y)
    if which == 1: return lmm_amd.logpdf_and_gradient(fx, y)["value"]
    post = lmm_amd.posterior(fx, y)
    mu, v = lmm_amd.mean_and_var(post(xs, 0.3))
    if which == 2: return float(np.sum(mu * np.arange(1, mu.size + 1))) + float(np.sum(v))
    s = lmm_amd.rand(np.random.default_rng(5), post(xs, 0.3), jitters=(1e-9, 1e-8, 1e-8))
    return float(np.sum(np.asarray(s) * np.arange(1, s.size + 1)))
def xs_like(fx, seed):
    x = fx.x.x; rng = np.random.default_rng(seed)
    xs = np.sort(rng.uniform(0, 6, 40)) if x.ndim == 1 else rng.uniform(0, 4, size=(x.shape[0], 40))
    return lmm_amd.MOInputIsotopicByOutputs(xs, fx.x.out_dim)
bad = 0
if len(sys.argv) > 2 and sys.argv[2] == "all":        # every verb, small and mid shapes (run with LMM_DETERMINISTIC=1: split-K atomics off)
    for shapes, mm, rr in [([(130, 2), (150, 1), (130, 1), (150, 2)], 3, reps), ([(552, 1), (600, 2), (640, 1)], 5, reps // 2),
                           ([(1100, 1), (1152, 2), (1030, 1)], 6, reps // 4), ([(2048, 1), (2000, 2)], 8, max(4, reps // 20)), ([(4096, 1), (4000, 1)], 4, max(3, reps // 40))]:
        probs = [problem(31 * i + n, n, d, p=mm + 1, m=mm) for i, (n, d) in enumerate(shapes)]
        xss = [xs_like(fx, 7 + i) for i, (fx, y) in enumerate(probs)]
        ref = [[digest(fx, y, xss[k], w) for w in range(4)] for k, (fx, y) in enumerate(probs)]
        nbad = 0
        for it in range(rr):
            for k, (fx, y) in enumerate(probs):
                w = it % 4
                v = digest(fx, y, xss[k], w)
                if v != ref[k][w]:
                    nbad += 1
                    if nbad <= 5: print("  MISMATCH", shapes[k], "iteration", it, "verb", w, v, "expected", ref[k][w], flush=True)
        print(f"all verbs, shapes {shapes}, {mm} latents: {rr} rounds, {nbad} mismatches", flush=True)
        bad += nbad
    sys.exit(1 if bad else 0)
for shapes in [[(130, 2), (150, 1), (130, 1), (150, 2)], [(552, 1), (530, 1), (600, 2)], [(1000, 1), (1024, 1), (960, 2)], [(200, 1), (250, 1), (256, 2)]]:
    probs = [problem(17 * i + n, n, d) for i, (n, d) in enumerate(shapes)]
    ref = [(lmm_amd.logpdf(fx, y), lmm_amd.logpdf_and_gradient(fx, y)["value"]) for fx, y in probs]
    nbad = 0
    for it in range(reps):
        for k, (fx, y) in enumerate(probs):
            v = lmm_amd.logpdf(fx, y) if it % 2 == 0 else lmm_amd.logpdf_and_gradient(fx, y)["value"]
            if v != ref[k][it % 2]:
                nbad += 1
                if nbad <= 5: print("  MISMATCH", shapes[k], "iteration", it, "path", "logpdf" if it % 2 == 0 else "gradient", v, "expected", ref[k][it % 2], flush=True)
    print(f"shapes {shapes}: {reps} rounds, {nbad} mismatches", flush=True)
    bad += nbad
sys.exit(1 if bad else 0)
