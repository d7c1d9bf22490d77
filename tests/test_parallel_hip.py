"""-m gpu: the multi-GPU layer with the REAL HIP evaluator on every rank (round-1 review: the gloo test only ever injected the
oracle).  Two ranks share device 0 (RCCL refuses two ranks on one device, so the process group is gloo); each rank evaluates
its latent shard through liblmm_hip.so and the partial results are summed by the all-reduce exactly as on two GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lmm_amd
    from lmm_amd.workloads import synthetic_problem
    lmm_amd.init(0)                                         # both ranks on the one GPU of the box
    P = synthetic_problem(5, 7, 700, "matern52", True, seed=1)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in P["gps"]])
    f = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))
    x = lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 7)
    y = torch.from_numpy(P["y"]).cuda()
    total = lmm_amd.sharded_logpdf(f, x, 0.1, y)            # HIP shard evaluator + ONE scalar all-reduce
    post = lmm_amd.sharded_posterior(f, x, 0.1, y)          # this rank's latents only
    xs = P["x"][:33] + 0.02
    mean, var = lmm_amd.sharded_mean_and_var(post(lmm_amd.MOInputIsotopicByOutputs(xs, 7), 0.1))
    shard = lmm_amd.latent_shard(5, rank, world)
    fxr = lmm_amd.ILMM(fs, f.H, shard=shard)(lmm_amd.MOInputIsotopicByOutputs(P["x"][:64], 7), 0.1)
    smp = lmm_amd.sharded_rand(np.random.default_rng(99), fxr, jitters=(1e-9, 1e-6, 1e-6))
    if rank == 0:
        q.put((total, mean.cpu().numpy(), var.cpu().numpy(), smp.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_hip_evaluator_one_device():
    sys.path.insert(0, ROOT)
    from oracle import lmm_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    total, mean, var, smp = q.get(timeout=500)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    P = O.synthetic_problem(5, 7, 700, "matern52", True, seed=1)
    assert total == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-9)
    post = O.oilmm_posterior(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    mo, vo = O.oilmm_mean_var(post, P["U"], P["S"], P["x"][:33] + 0.02, 0.1)
    np.testing.assert_allclose(mean, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, vo, rtol=1e-7)
    n = 64
    g = np.random.default_rng(99); z = g.standard_normal(5 * n); eps = g.standard_normal(n * 7)
    X = np.stack([O.gp_rand(P["gps"][l], P["x"][:n], 1e-6, z[l * n:(l + 1) * n]) for l in range(5)])
    np.testing.assert_allclose(smp, (O.orthogonal_dense(P["U"], P["S"]) @ X).reshape(-1) + np.sqrt(0.1) * eps, rtol=1e-7, atol=1e-9)
