"""ctypes binding of liblmm_hip.so (include/lmm_hip.h).  No fallback: if the HIP library or a GPU is
missing every compute entry point raises -- the product path never routes through a CPU implementation."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LMM_HIP_LIB (the variable the Julia shim reads too): another build of the library, for same-box A/B runs of two kernel versions
LIB_PATH = os.environ.get("LMM_HIP_LIB") or os.path.join(_HERE, "liblmm_hip.so")

LMM_OK, LMM_ERR_DIM, LMM_ERR_NOT_ORTHOGONAL, LMM_ERR_NOT_PD, LMM_ERR_HIP, LMM_ERR_ARG, LMM_ERR_UNSUPPORTED, LMM_ERR_RCCL = range(8)
UNIQUE_ID_BYTES = 128
KERNEL_KINDS = {"se": 0, "matern32": 1, "matern52": 2}

# Every symbol include/lmm_hip.h declares (tests/test_abi.py checks the library exports each one).
SYMBOLS = [
    "lmm_init", "lmm_shutdown", "lmm_last_error_string", "lmm_last_error_detail", "lmm_device_synchronize", "lmm_release_cached_memory",
    "lmm_stream_wait_caller", "lmm_set_compute_dtype", "lmm_get_compute_dtype", "lmm_set_projection_dtype", "lmm_get_projection_dtype", "lmm_comm_get_unique_id", "lmm_comm_init_rank", "lmm_comm_info", "lmm_allreduce_sum_f64", "lmm_allreduce_max_f64",
    "lmm_comm_destroy",
    "lmm_set_strict_progress", "lmm_get_strict_progress", "lmm_dev_claim_scramble", "lmm_orthogonal_validate", "lmm_oilmm_logpdf", "lmm_oilmm_logpdf_grad", "lmm_oilmm_post_logpdf_grad", "lmm_oilmm_post_logpdf_grad_seq", "lmm_ilmm_logpdf_grad", "lmm_ilmm_post_logpdf_grad", "lmm_ilmm_post_logpdf_grad_seq", "lmm_ilmm_post_latent_logpdf_grad_seq", "lmm_oilmm_logpdf_multi", "lmm_reorder", "lmm_ilmm_logpdf", "lmm_ilmm_logpdf_ex", "lmm_ilmm_logpdf_multi", "lmm_mogp_logpdf", "lmm_mogp_logpdf_diag",
    "lmm_oilmm_posterior_create", "lmm_mogp_posterior_create", "lmm_post_condition", "lmm_ilmm_posterior_create", "lmm_post_destroy", "lmm_ilmm_post_latent_view", "lmm_ilmm_post_mean_and_var", "lmm_ilmm_post_mean_and_cov", "lmm_ilmm_post_condition", "lmm_ilmm_post_logpdf", "lmm_ilmm_post_rand",
    "lmm_latent_marginals", "lmm_oilmm_mean_and_var", "lmm_lmm_mean_and_cov", "lmm_mogp_cross_cov", "lmm_oilmm_post_logpdf", "lmm_lmm_rand", "lmm_lmm_rand_multi", "lmm_normals",
    "lmm_profile_begin", "lmm_profile_end",
    "lmm_dev_potrf", "lmm_dev_check_info", "lmm_dev_extent_check", "lmm_dev_region_plan", "lmm_dev_flag_epoch", "lmm_dev_gemm_nt_sub", "lmm_dev_gram", "lmm_dev_write_rate", "lmm_dev_mfma_f64_peak",
]


class GpT(C.Structure):
    _fields_ = [("kind", C.c_int), ("variance", C.c_double), ("lengthscale", C.c_double), ("mean", C.c_double)]


class JittersT(C.Structure):
    _fields_ = [("project_jitter", C.c_double), ("ilmm_rand_jitter", C.c_double), ("default_jitter", C.c_double)]


class GpGradT(C.Structure):
    _fields_ = [("variance", C.c_double), ("lengthscale", C.c_double), ("mean", C.c_double)]


class ProfEntryT(C.Structure):
    _fields_ = [("launches", C.c_longlong), ("ms", C.c_double), ("work", C.c_double), ("bytes", C.c_double)]


PROF_CLASSES = ["gram", "update", "update_narrow", "trsm", "diag", "region", "update_short", "solve", "solve_leaf", "strip"]


class PosDefException(ArithmeticError):
    """Julia's LinearAlgebra.PosDefException(info)."""

    def __init__(self, msg: str, latent: int, info: int):
        super().__init__(msg)
        self.latent, self.info = latent, info


class LMMError(RuntimeError):
    pass


_lib = None
_initialised_device: Optional[int] = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LMMError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  A process must hold ONE HIP
        # runtime, so when torch is present load it first: the dynamic linker then binds liblmm_hip.so's
        # libamdhip64.so.7 dependency to the copy torch already mapped (device pointers, streams and RCCL of
        # both sides then live in the same runtime).  Without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.lmm_last_error_string.restype = C.c_char_p
    return _lib


def init(device: Optional[int] = None) -> int:
    """lmm_init: one process per GPU.  device defaults to LOCAL_RANK (torch.distributed launch) or 0."""
    global _initialised_device
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    lib = load()
    check(lib.lmm_init(C.c_int(device)))
    _initialised_device = device
    return device


def ensure_init() -> None:
    if _initialised_device is None:
        init()


def check(rc: int) -> None:
    if rc == LMM_OK:
        return
    lib = load()
    msg = lib.lmm_last_error_string().decode()
    if rc == LMM_ERR_DIM:
        raise RuntimeError(msg)                      # Julia: ErrorException("out dim of x != out dim of f.")
    if rc == LMM_ERR_NOT_ORTHOGONAL:
        raise ValueError(msg)                        # Julia: ArgumentError
    if rc == LMM_ERR_NOT_PD:
        lat, info = C.c_int(), C.c_int()
        lib.lmm_last_error_detail(C.byref(lat), C.byref(info))
        raise PosDefException(msg, lat.value, info.value)
    if rc == LMM_ERR_ARG:
        raise ValueError(msg)
    if rc == LMM_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise LMMError(msg)


def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


class _OwnedPtr(C.c_void_p):
    """A void* that keeps the array it points into alive.  Call sites write `lib.f(Arr(a).ptr, ...)`: the temporary Arr dies as soon as
    `.ptr` has been read, and with it a converted copy it may own (a transposed (d, n) input, a non-contiguous or non-float64 array)
    -- the C call would then read freed memory.  (Found by tools/stress_alternate.py: gradients of d = 2 problems were occasionally
    evaluated on recycled memory; a plain c_void_p holds only the integer.)"""


def _ptr_of(addr, owner):
    p = _OwnedPtr(addr)
    p._owner = owner
    return p


# id(ndarray) -> (ndarray, pointer, size) of the last few SMALL read-only host arrays handed over (model matrices, inputs of the
# reference's regime): taking the address of a NumPy buffer through ctypes costs 2-4 us, several times per call, which shows at
# n = 200 (an evaluation is ~140 us).  The entry keeps the array alive, so an id cannot be reused by another object while it is
# cached; it is only trusted while the array still has the size it was cached with (an in-place ndarray.resize, the one way a live
# array's buffer moves, changes it).  Outputs, temporaries of conversions and anything above _PTRS_MAX_ELEMS are never inserted, so
# the cache cannot pin large buffers the caller has dropped.
_PTRS: dict = {}
_PTRS_MAX_ELEMS = 1 << 16


class Arr:
    """A Float64 array handed to the C ABI: a NumPy array (host pointer) or a CUDA/HIP torch tensor
    (device pointer).  Keeps the owner alive for the duration of the call."""

    def __init__(self, a, writable: bool = False):
        if _is_torch(a):
            import torch
            if a.dtype != torch.float64:
                raise TypeError("expected a float64 tensor")
            if not a.is_contiguous():
                if writable:
                    raise ValueError("output tensor must be contiguous")
                a = a.contiguous()
            self.owner = a
            self.ptr = _ptr_of(a.data_ptr(), a)
            self.size = a.numel()
            if a.is_cuda:
                order_after_torch()
        else:
            hit = None if writable else _PTRS.get(id(a))
            if hit is not None and hit[0] is a and a.size == hit[2]:      # the same, un-resized ndarray object as before
                self.owner, self.ptr, self.size = hit
                return
            ok = type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous
            if writable:
                if not ok:
                    raise ValueError("output array must be a C-contiguous float64 ndarray")
            elif not ok:
                a = np.ascontiguousarray(a, dtype=np.float64)
            self.owner = a
            self.ptr = _ptr_of(a.ctypes.data, a)
            self.size = a.size
            if ok and not writable and a.size <= _PTRS_MAX_ELEMS:
                if len(_PTRS) >= 64:
                    _PTRS.clear()
                _PTRS[id(a)] = (a, self.ptr, self.size)


def set_compute_dtype(dtype: str) -> None:
    """"f64" (default, parity mode) or "f32": Float32 matrices on v_mfma_f32 for the per-latent paths (include/lmm_hip.h)."""
    check(load().lmm_set_compute_dtype(C.c_int({"f64": 0, "f32": 1}[dtype])))


def get_compute_dtype() -> str:
    return "f32" if load().lmm_get_compute_dtype() == 1 else "f64"


def set_strict_progress(on: bool) -> None:
    """True (the default): the dataflow kernel's workgroups take their task index in turn from a counter, so its deadlock-freedom
    argument holds in any dispatch order; False: task = blockIdx.x, which relies on in-order dispatch (include/lmm_hip.h,
    conventions).  Same values either way."""
    check(load().lmm_set_strict_progress(C.c_int(1 if on else 0)))


def get_strict_progress() -> bool:
    return load().lmm_get_strict_progress() == 1


_PROJ = {"f64": 0, "native": 0, "bf16": 1, "bf16x2": 2}


def set_projection_dtype(dtype: str) -> None:
    """Dtype of the H unprojection of predictive marginals (reference src/oilmm.jl:69-72): "native" (Float64, default), "bf16"
    (v_mfma_f32_16x16x32_bf16, BASELINE configs[3]; tolerance 2^-7 * sum_l |H||M_lat|, include/lmm_hip.h) or "bf16x2"."""
    check(load().lmm_set_projection_dtype(C.c_int(_PROJ[dtype])))


def get_projection_dtype() -> str:
    return ["native", "bf16", "bf16x2"][load().lmm_get_projection_dtype()]


def wait_stream(stream) -> None:
    """Order the library's streams behind everything queued so far on `stream` (a torch.cuda.Stream, or any object with a
    `cuda_stream` handle; None = the legacy default stream): lmm_stream_wait_caller.  Call it before handing over a device tensor
    that was produced on a stream OTHER than torch's current one -- order_after_torch() below only looks at the current stream."""
    ensure_init()
    h = None if stream is None else getattr(stream, "cuda_stream", stream)
    check(load().lmm_stream_wait_caller(C.c_void_p(h)))


def order_after_torch() -> None:
    """The library runs on its own non-blocking HIP streams; torch produces (and recycles) device tensors asynchronously on ITS
    current stream.  Before a device pointer crosses the ABI, make the library's streams wait for everything torch has queued
    (lmm_stream_wait_caller: one event record + one stream wait, no host stall).  Skipped when torch's CURRENT stream is already
    idle; a tensor produced on another stream (torch.cuda.stream(s) blocks, side streams of a data loader) must be named by
    the caller: lmm_amd.wait_stream(s)."""
    import torch
    st = torch.cuda.current_stream()
    if st.query():
        return
    ensure_init()
    check(load().lmm_stream_wait_caller(C.c_void_p(st.cuda_stream)))


# ---- RCCL communicator of the C ABI (one process per GPU) ---------------------------------------------------------
def comm_world() -> int:
    """World size of the ABI's RCCL communicator (0: none)."""
    if _lib is None or _initialised_device is None:
        return 0
    r, w = C.c_int(), C.c_int()
    check(_lib.lmm_comm_info(C.byref(r), C.byref(w)))
    return w.value


def comm_get_unique_id() -> bytes:
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(load().lmm_comm_get_unique_id(buf))
    return buf.raw


def comm_init_rank(uid: bytes, rank: int, world: int) -> None:
    ensure_init()
    if len(uid) != UNIQUE_ID_BYTES:
        raise ValueError("unique id must be %d bytes" % UNIQUE_ID_BYTES)
    check(load().lmm_comm_init_rank(C.create_string_buffer(uid, UNIQUE_ID_BYTES), C.c_int(rank), C.c_int(world)))


def comm_init_from_torch() -> None:
    """Create the ABI's RCCL communicator for the ranks of torch.distributed's default group.  Rank 0 draws the unique id and
    ships it through the group's rendezvous STORE (a TCP key-value store: the out-of-band step a Julia caller does with
    MPI.bcast) -- no torch collective is involved, so torch's own NCCL communicator need not exist yet and the two RCCL
    initialisations never interleave.  Falls back to the object broadcast when the process group exposes no store."""
    import torch.distributed as dist
    global _comm_generation
    rank, world = dist.get_rank(), dist.get_world_size()
    _comm_generation += 1
    store = None
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if store is not None:
        key = f"lmm_rccl_unique_id/{_comm_generation}"
        if rank == 0:
            store.set(key, comm_get_unique_id())
        uid = bytes(store.get(key))          # blocks until rank 0 has set it
    else:
        box = [comm_get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    comm_init_rank(uid, rank, world)


_comm_generation = 0


def comm_destroy() -> None:
    if _lib is not None:
        check(_lib.lmm_comm_destroy())


def allreduce_sum(a, op: str = "sum"):
    """In-place sum (or max) all-reduce of a float64 NumPy array or CUDA tensor over the ABI communicator."""
    arr = Arr(a, True)
    fn = load().lmm_allreduce_sum_f64 if op == "sum" else load().lmm_allreduce_max_f64
    check(fn(arr.ptr, C.c_size_t(arr.size)))
    return a


def gps_array(gps: Sequence[dict]):
    arr = (GpT * max(len(gps), 1))()
    for l, g in enumerate(gps):
        a = arr[l]
        a.kind = KERNEL_KINDS[g["kind"]]
        a.variance = float(g.get("variance", 1.0))
        a.lengthscale = float(g.get("lengthscale", 1.0))
        a.mean = float(g.get("mean", 0.0))
    return arr


def jitters(j: Optional[Tuple[float, float, float]]):
    if j is None:
        return None
    return C.byref(JittersT(*map(float, j)))


def colmajor(a: np.ndarray) -> np.ndarray:
    """Column-major (Julia Array) image of a 2-D host matrix as a flat float64 vector (a view when `a` already is one)."""
    return np.ravel(np.asarray(a, dtype=np.float64), order="F")
