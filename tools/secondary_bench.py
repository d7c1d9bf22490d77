"""Secondary metrics of SURVEY.md section 8(d) on ONE GPU's share of BASELINE configs[3] / configs[4] (Float64):
posterior-create, posterior-predictive marginals, prior/posterior sampling.  Prints one JSON line per phase.

    python tools/secondary_bench.py [c3] [c4] [--reps R]
"""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd import workloads as O      # input generation only

lmm_amd.init(0)
lib = lmm_amd.load()
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
which = [a for a in sys.argv[1:] if a in ("c3", "c4", "c3small", "notebook")] or ["c3", "c4", "notebook"]


def timed(name, fn, flops, extra=None):
    fn(); torch.cuda.synchronize()
    out, ts = None, []
    for _ in range(reps):
        out = None              # drop the previous result first: its device buffers go back to the pool and are reused
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))   # median: host-side outliers (page faults of fresh result arrays) are not the device path
    line = {"phase": name, "ms": round(dt * 1e3, 2), "tflops": round(flops / dt / 1e12, 2)}
    if "--classes" in sys.argv:      # one more pass with per-launch events (perturbs small problems: not part of the timing)
        L.check(lib.lmm_profile_begin(0))
        fn(); torch.cuda.synchronize()
        ent = (L.ProfEntryT * len(L.PROF_CLASSES))()
        L.check(lib.lmm_profile_end(ent))
        line["event_ms_by_class"] = {c: round(float(ent[i].ms), 2) for i, c in enumerate(L.PROF_CLASSES) if ent[i].launches}
    line.update(extra or {})
    print(json.dumps(line), flush=True)
    return out


def model(P):
    K = {"se": lmm_amd.SEKernel, "matern32": lmm_amd.Matern32Kernel, "matern52": lmm_amd.Matern52Kernel}
    return lmm_amd.independent_mogp([lmm_amd.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in P["gps"]])


def run(tag, m, p, n, ns, share):
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
    fs, H = model(P), lmm_amd.Orthogonal(P["U"], P["S"])
    f = lmm_amd.ILMM(fs, H, shard=(0, share))
    xd = torch.from_numpy(P["x"]).cuda(); yd = torch.from_numpy(P["y"]).cuda()
    xin = lmm_amd.MOInputIsotopicByOutputs(xd, p)
    xs = torch.from_numpy(P["x"][:ns] + 0.5 * 20.0 / 575.0).cuda()
    xsin = lmm_amd.MOInputIsotopicByOutputs(xs, p)
    cfg = {"config": f"{tag}: one GPU's {share} of {m} latents, p={p}, n={n}, n*={ns}, f64"}
    post = timed(f"{tag} posterior(fx, y)", lambda: lmm_amd.posterior(f(xin, 0.1), yd), share * n ** 3 / 3.0, cfg)
    timed(f"{tag} marginals(post(x*))", lambda: lmm_amd.mean_and_var(post(xsin, 0.1)), share * float(n) ** 2 * ns, cfg)
    timed(f"{tag} mean(post(x*)) [means only]", lambda: lmm_amd.mean(post(xsin, 0.1)), 0.0, cfg)
    timed(f"{tag} marginals(prior(x))", lambda: lmm_amd.mean_and_var(f(xin, 0.1)), 0.0, cfg)
    rng = np.random.default_rng(0)      # normals are drawn on the host in the reference's order (part of the timed call)
    timed(f"{tag} rand(prior(x))", lambda: lmm_amd.rand(rng, f(xin, 0.1), jitters=(1e-9, 1e-8, 1e-8)),
          share * n ** 3 / 3.0, cfg)
    timed(f"{tag} rand(post(x*))", lambda: lmm_amd.rand(rng, post(xsin, 0.1), jitters=(1e-9, 1e-8, 1e-8)),
          share * (float(n) ** 2 * ns + float(n) * ns * ns + ns ** 3 / 3.0), cfg)
    del post
    lib.lmm_release_cached_memory()


if "c3small" in which:
    run("c3small", 64, 128, 2048, 2048, 8)
if "c3" in which:
    run("c3", 64, 128, 8192, 8192, 8)
if "c4" in which:
    run("c4", 128, 256, 32768, 4096, 16)


def notebook():
    """The reference notebook's posterior timings: p = 600, m = 20 Matern52 latents, 552 training
    points of a 576-point grid on [0, 20], the 24 held-out points as x*, sigma2 = 1e-6; published (unstated CPU, dense-H ILMM
    path): marginals 708.977 ms, rand 578.937 ms, logpdf 5.634 s (ILMM) / 172.541 ms (OILMM)."""
    m, p, s2 = 20, 600, 1e-6
    perm = np.random.default_rng(1).permutation(576)
    grid = np.linspace(0.0, 20.0, 576)
    x, xs = grid[np.sort(perm[:552])], grid[np.sort(perm[552:])]
    U, S, _ = np.linalg.svd(np.random.default_rng(2).uniform(size=(p, m)), full_matrices=False)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
    y = torch.from_numpy(np.random.default_rng(3).standard_normal(552 * p)).cuda()
    xin = lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(x).cuda(), p)
    xsin = lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(xs).cuda(), p)
    rng = np.random.default_rng(0)
    from lmm_amd import model as M
    for tag, H, pub in (("OILMM", lmm_amd.Orthogonal(np.ascontiguousarray(U), S), (172.541, None, None)),
                        ("ILMM dense-H", np.ascontiguousarray(U * np.sqrt(S)), (5634.0, 708.977, 578.937))):
        M.ILMM_ALLOW_DECOUPLED = False          # identical latent kernels: keep the reference's coupled (mn) x (mn) path
        f = lmm_amd.ILMM(fs, H)
        cfg = {"config": f"notebook {tag}: p=600, m=20, n=552, n*=24, sigma2=1e-6, f64"}
        timed(f"notebook {tag} logpdf", lambda: lmm_amd.logpdf(f(xin, s2), y), 0.0, dict(cfg, published_ms=pub[0]))
        post = timed(f"notebook {tag} posterior", lambda: lmm_amd.posterior(f(xin, s2), y), 0.0, cfg)
        timed(f"notebook {tag} marginals(post(x*))", lambda: lmm_amd.mean_and_var(post(xsin, s2)), 0.0, dict(cfg, published_ms=pub[1]))
        timed(f"notebook {tag} rand(post(x*))", lambda: lmm_amd.rand(rng, post(xsin, s2)), 0.0, dict(cfg, published_ms=pub[2]))
        del post
    M.ILMM_ALLOW_DECOUPLED = True


if "notebook" in which:
    notebook()
