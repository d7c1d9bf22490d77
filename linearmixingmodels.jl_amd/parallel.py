"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU
tests).  The reference has no parallelism; the latent processes of an OILMM are
independent, so rank r owns a contiguous block of latents, evaluates its partial sum through the C ABI with no
data-path collective, and ONE scalar all-reduce over xGMI finishes logpdf."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib as L
from . import model as M


def latent_shard(m: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of m latents: the first m % world ranks get one extra."""
    base, extra = divmod(m, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _abi_comm() -> Tuple[int, int]:
    """(rank, world) of the C ABI's own RCCL communicator (lmm_comm_init_rank); world 0 when there is none."""
    if L._lib is None or L._initialised_device is None:
        return 0, 0
    import ctypes as C
    r, w = C.c_int(), C.c_int()
    L.check(L._lib.lmm_comm_info(C.byref(r), C.byref(w)))
    return r.value, w.value


def _world() -> Tuple[int, int]:
    r, w = _abi_comm()
    if w > 0:
        return r, w
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


ABI_COLLECTIVE = "lmm_allreduce_sum_f64 (RCCL inside liblmm_hip.so)"


def _default_abi_init() -> int:
    L.comm_init_from_torch()
    return L.comm_world()


def _default_abi_probe(world: int) -> None:
    """First collective on the new communicator, on a host AND a device buffer: sum of (rank + 1) must be world (world + 1) / 2."""
    import torch
    rank = _abi_comm()[0]
    want = world * (world + 1) / 2.0
    h = np.array([rank + 1.0])
    L.allreduce_sum(h)
    d = torch.full((3,), rank + 1.0, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    L.allreduce_sum(d)
    if h[0] != want or not bool((d == want).all()):
        raise RuntimeError(f"ABI all-reduce probe returned {h[0]} / {d.tolist()}, expected {want}")


def _default_agree(ok: bool) -> bool:
    """True iff EVERY rank reports ok (MIN all-reduce through torch.distributed), so that all ranks take the same branch."""
    import torch
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return ok
    dev = torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    d.all_reduce(t, op=d.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def select_collective(backend: str, world: int, init_fn: Optional[Callable] = None, probe_fn: Optional[Callable] = None,
                      agree_fn: Optional[Callable] = None, destroy_fn: Optional[Callable] = None) -> Tuple[bool, Optional[str]]:
    """Which all-reduce finishes an N > 1 evaluation.  backend "nccl" (one process per GPU over RCCL): the C ABI's own
    communicator -- what a Julia / C caller binds -- created from torch's rendezvous store and PROVED by one probe collective;
    if creating or probing it fails on ANY rank, every rank drops it and reduces through torch.distributed instead, and the
    description carries the reason, so a multi-GPU run still reports a number.  Other backends (gloo rehearsals, CPU tests)
    reduce through torch.distributed.  Returns (use_abi_collective, description); the hooks exist for the CPU tests."""
    if world <= 1:
        return False, None
    if backend != "nccl":
        return False, "torch.distributed/" + backend
    init_fn = init_fn or _default_abi_init
    probe_fn = probe_fn or _default_abi_probe
    agree_fn = agree_fn or _default_agree
    destroy_fn = destroy_fn or L.comm_destroy
    err = None
    try:
        got = init_fn()
        if got != world:
            raise RuntimeError(f"ABI communicator has {got} ranks, expected {world}")
    except Exception as e:          # noqa: BLE001 -- any failure means "fall back", the reason is reported
        err = f"{type(e).__name__}: {e}"
    # Agree on the INIT outcome before anybody enters the probe collective: a rank whose init raised would otherwise skip the probe
    # while the others block inside an RCCL all-reduce that can never complete.
    if agree_fn(err is None):
        try:
            probe_fn(world)
        except Exception as e:      # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if agree_fn(err is None):
            return True, ABI_COLLECTIVE
    try:
        destroy_fn()
    except Exception:               # noqa: BLE001
        pass
    return False, f"torch.distributed/nccl (ABI RCCL failed: {err or 'on another rank'})"


def _all_reduce_sum(t):
    """Sum over ranks, in place.  The product path is lmm_allreduce_sum_f64 -- RCCL inside liblmm_hip.so, the same call a Julia
    or C caller makes -- whenever the ABI communicator exists; torch.distributed (gloo in the CPU tests) otherwise."""
    if _abi_comm()[1] > 0:
        if L._is_torch(t) and not t.is_cuda:
            a = t.numpy()                      # shares memory with t
            L.allreduce_sum(a)
        else:
            L.allreduce_sum(t)
        return t
    d = _dist()
    if d is not None and d.get_world_size() > 1:
        d.all_reduce(t, op=d.ReduceOp.SUM)
    return t


def _reduce_tensor(a):
    """`a` as a float64 tensor on the side the active collective reduces on: the ABI communicator takes host or device
    pointers (so the data stays where the HIP path left it); torch's nccl backend needs device tensors, gloo host tensors."""
    import torch
    t = torch.as_tensor(a, dtype=torch.float64)
    if _abi_comm()[1] > 0:
        return t if t.is_contiguous() else t.contiguous()
    d = _dist()
    if d is not None and d.get_backend() == "nccl":
        return t.to(torch.device("cuda", torch.cuda.current_device()))
    return t.cpu()


def sharded_logpdf(f: M.ILMM, x: M.MOInputIsotopicByOutputs, sigma2: float, y,
                   local_fn: Optional[Callable] = None, reduce: bool = True) -> float:
    """logpdf of an OILMM FiniteGP with the latents sharded over the ranks of the default process group.
    Rank 0 adds the regulariser.  `local_fn(fx_shard, y, with_regulariser)` defaults to the HIP path
    (model.logpdf); tests inject a checker to exercise the sharding + all-reduce on CPU/gloo."""
    import torch
    rank, world = _world()
    m = len(f.f.fs)
    shard = latent_shard(m, rank, world)
    fx = M.ILMM(f.f, f.H, shard=shard)(x, sigma2)
    fn = local_fn or M.logpdf
    part = fn(fx, y, rank == 0)
    if not reduce:
        return part
    t = _reduce_tensor(np.array([part]))
    return float(_all_reduce_sum(t)[0])


def sharded_mean_and_var(fx_shard: M.FiniteGP, local_fn: Optional[Callable] = None):
    """mean_and_var of a (posterior) OILMM whose latents are sharded: each rank mixes its latents through its
    columns of H, then ONE all-reduce of the p*n* partial means and variances (form (ii));
    sigma2 is added by rank 0 only."""
    import torch
    rank, _ = _world()
    fn = local_fn or M.mean_and_var
    mean, var = fn(fx_shard, rank == 0)
    t = _reduce_tensor(torch.stack([torch.as_tensor(mean, dtype=torch.float64), torch.as_tensor(var, dtype=torch.float64)]))
    _all_reduce_sum(t)
    return t[0], t[1]


def sharded_posterior(f: M.ILMM, x: M.MOInputIsotopicByOutputs, sigma2: float, y) -> M.ILMM:
    """posterior(fx, y) with the latents sharded: each rank conditions ITS block of latents (no collective; the posterior
    state stays sharded by latent and is never gathered -- SURVEY.md section 8e).  Returns this rank's shard model."""
    rank, world = _world()
    shard = latent_shard(len(f.f.fs), rank, world)
    return M.posterior(M.ILMM(f.f, f.H, shard=shard)(x, sigma2), y)


def sharded_rand(rng, fx_shard: M.FiniteGP, local_fn: Optional[Callable] = None, jitters=None):
    """rand(rng, fx) with sharded latents: every rank draws the SAME normals (same seed; the reference's draw order: m blocks of
    n latent normals, then n*p noise normals), mixes its own latents' samples through its columns of H, and ONE all-reduce of the
    n*p partial sums finishes the sample; rank 0 alone adds the noise term (rand row)."""
    import torch
    rank, _ = _world()
    fn = local_fn or (lambda fx, add_noise: M.rand(rng, fx, None, jitters, add_noise))
    part = fn(fx_shard, rank == 0)
    return _all_reduce_sum(_reduce_tensor(part))
