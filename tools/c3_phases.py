"""Where a BASELINE configs[3] step (posterior + mean_and_var at n* = n = 8192, 64 latents, 128 outputs) spends its wall time, phase by phase
(host clock, device drained after each phase), next to the kernel-class sum of an instrumented pass over ALL 64 latents on one stream:
    python tools/c3_phases.py [m]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd.workloads import synthetic_problem
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p, n, s2 = 128, 8192, 0.1
lmm_amd.init(0)
lib = lmm_amd.load()
P = synthetic_problem(m, p, n, "matern52", True, s2=s2, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
xin = lmm_amd.MOInputIsotopicByOutputs(xd, p)
xs = lmm_amd.MOInputIsotopicByOutputs(xd + 0.5 * 20.0 / 575.0, p)
fx = lmm_amd.ILMM(fs, H)(xin, s2)
sync = torch.cuda.synchronize


def step(timed):
    t = [time.perf_counter()]
    post = lmm_amd.posterior(fx, yd); sync(); t.append(time.perf_counter())
    mv = lmm_amd.mean_and_var(post(xs, s2)); sync(); t.append(time.perf_counter())
    red = lmm_amd.sharded_mean_and_var(post(xs, s2)) if timed == 2 else None; sync(); t.append(time.perf_counter())
    del post; sync(); t.append(time.perf_counter())
    return [b - a for a, b in zip(t, t[1:])], mv, red


step(1)
rows = [step(1)[0] for _ in range(3)]
med = np.median(np.array(rows), axis=0) * 1e3
print(f"m={m}: posterior {med[0]:.1f} ms | mean_and_var {med[1]:.1f} ms | del post {med[3]:.2f} ms | sum {med[0] + med[1] + med[3]:.1f} ms")
t0 = time.perf_counter()
for _ in range(3):
    post = lmm_amd.posterior(fx, yd); lmm_amd.sharded_mean_and_var(post(xs, s2))
sync(); print(f"bench-style step (posterior + sharded_mean_and_var, handle dropped by rebinding): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
del post
L.check(lib.lmm_profile_begin(1))
lmm_amd.mean_and_var(lmm_amd.posterior(fx, yd)(xs, s2))
ent = (L.ProfEntryT * len(L.PROF_CLASSES))(); L.check(lib.lmm_profile_end(ent))
cls = {c: (int(ent[i].launches), round(float(ent[i].ms), 2)) for i, c in enumerate(L.PROF_CLASSES) if ent[i].launches}
print(f"instrumented pass over all {m} latents on ONE stream: classes (launches, ms) {cls}  sum {sum(v[1] for v in cls.values()):.1f} ms")
