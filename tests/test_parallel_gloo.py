"""world_size-2 gloo test of the multi-GPU layer (parallel.py) on CPU: latent sharding + ONE scalar all-reduce for
logpdf, and the p*n* all-reduce for posterior marginals.  The per-rank evaluator is injected (the oracle plays the
HIP path here -- tests may call the oracle; the product never does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lmm_amd
    from oracle import lmm_oracle as O
    P = O.synthetic_problem(5, 7, 40, "matern52", True, seed=1)
    K = {"matern52": lmm_amd.Matern52Kernel}
    fs = lmm_amd.independent_mogp([lmm_amd.GP(K[g["kind"]]()) for g in P["gps"]])
    f = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))
    x = lmm_amd.MOInputIsotopicByOutputs(P["x"], 7)

    def local_logpdf(fx, y, with_reg):                 # what lmm_oilmm_logpdf returns for a shard
        l0, l1 = fx.f.shard
        n = fx.x.n
        T, ST = O.project_orthogonal(P["U"], P["S"], fx.sigma2)
        Ty = T @ O.reshape_y(y, n)
        part = sum(O.gp_logpdf(P["gps"][l], P["x"], ST[l], Ty[l]) for l in range(l0, l1))
        return part + (O.regulariser_oilmm(P["U"], P["S"], fx.sigma2, O.reshape_y(y, n)) if with_reg else 0.0)

    total = lmm_amd.sharded_logpdf(f, x, 0.1, P["y"], local_fn=local_logpdf)

    post = O.oilmm_posterior(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    xs = P["x"][:9] + 0.02
    shard = lmm_amd.latent_shard(5, rank, world)

    def local_mv(fx, add_noise):                       # what lmm_oilmm_mean_and_var returns for a shard
        l0, l1 = fx.f.shard
        H = O.orthogonal_dense(P["U"], P["S"])[:, l0:l1]
        mv = [O.gp_mean_var(post[l], xs) for l in range(l0, l1)]
        M = H @ np.stack([a for a, _ in mv]); V = (H * H) @ np.stack([b + 1e-18 for _, b in mv])
        return M.reshape(-1), V.reshape(-1) + (fx.sigma2 if add_noise else 0.0)

    fxs = lmm_amd.ILMM(fs, f.H, shard=shard)(lmm_amd.MOInputIsotopicByOutputs(xs, 7), 0.1)
    mean, var = lmm_amd.sharded_mean_and_var(fxs, local_fn=local_mv)
    # rand: every rank draws the same normals; rank r mixes its latents; all-reduce; rank 0 adds the noise
    n = len(P["x"])
    def local_rand(fx, add_noise):
        l0, l1 = fx.f.shard
        g = np.random.default_rng(99); z = g.standard_normal(5 * n); eps = g.standard_normal(n * 7)
        X = np.stack([O.gp_rand(P["gps"][l], P["x"], 1e-6, z[l * n:(l + 1) * n]) for l in range(l0, l1)])
        part = (O.orthogonal_dense(P["U"], P["S"])[:, l0:l1] @ X).reshape(-1)
        return part + (np.sqrt(fx.sigma2) * eps if add_noise else 0.0)
    fxr = lmm_amd.ILMM(fs, f.H, shard=shard)(x, 0.1)
    smp = lmm_amd.sharded_rand(None, fxr, local_fn=local_rand)
    if rank == 0:
        q.put((total, mean.numpy(), var.numpy(), smp.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_logpdf_and_marginals_world2():
    sys.path.insert(0, ROOT)
    from oracle import lmm_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    total, mean, var, smp = q.get(timeout=150)
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    P = O.synthetic_problem(5, 7, 40, "matern52", True, seed=1)
    assert total == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-12)
    post = O.oilmm_posterior(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    mo, vo = O.oilmm_mean_var(post, P["U"], P["S"], P["x"][:9] + 0.02, 0.1)
    np.testing.assert_allclose(mean, mo, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(var, vo, rtol=1e-12)
    g = np.random.default_rng(99); z = g.standard_normal(5 * 40); eps = g.standard_normal(40 * 7)
    X = np.stack([O.gp_rand(P["gps"][l], P["x"], 1e-6, z[l * 40:(l + 1) * 40]) for l in range(5)])
    np.testing.assert_allclose(smp, (O.orthogonal_dense(P["U"], P["S"]) @ X).reshape(-1) + np.sqrt(0.1) * eps, rtol=1e-12, atol=1e-13)
