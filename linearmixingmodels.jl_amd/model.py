"""Host-side mirror of the LinearMixingModels.jl interface for the ILMM/OILMM inference hot path.

Julia is not available in the build image, so this module plays the role of the
Julia shim for tests and benchmarks: the same type names (`ILMM`, `IndependentMOGP`, `independent_mogp`,
`Orthogonal`, `get_latent_gp`; reference src/LinearMixingModels.jl:21-24) and the AbstractGPs verbs the
reference adds methods to (`logpdf`, `posterior`, `rand`, `marginals`, `mean_and_var`, `mean`, `var`),
each body being ONE call into liblmm_hip.so -- exactly what the `ccall` shim in
`linearmixingmodels.jl_amd/julia/LinearMixingModelsHIP.jl` does.  No arithmetic happens here.

Arrays may be NumPy (host) or float64 CUDA/HIP torch tensors (device pointers are passed straight
through the C ABI).
"""
from __future__ import annotations

import ctypes as C
import weakref
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


# ---- kernels / GPs (KernelFunctions.jl + AbstractGPs.jl names) ------------------------------------
class _Kernel:
    kind = ""

    def __init__(self, variance: float = 1.0, lengthscale: float = 1.0):
        self.variance, self.lengthscale = float(variance), float(lengthscale)

    def __eq__(self, o):
        return type(self) is type(o) and (self.variance, self.lengthscale) == (o.variance, o.lengthscale)

    def __repr__(self):
        return f"{type(self).__name__}(variance={self.variance}, lengthscale={self.lengthscale})"


class SEKernel(_Kernel):
    kind = "se"


class Matern32Kernel(_Kernel):
    kind = "matern32"


class Matern52Kernel(_Kernel):
    kind = "matern52"


class GP:
    """GP(kernel) or GP(mean_const, kernel)."""

    def __init__(self, *args):
        if len(args) == 1:
            self.mean, self.kernel = 0.0, args[0]
        else:
            self.mean, self.kernel = float(args[0]), args[1]

    def desc(self) -> dict:
        return {"kind": self.kernel.kind, "variance": self.kernel.variance, "lengthscale": self.kernel.lengthscale,
                "mean": self.mean}

    def __eq__(self, o):
        return isinstance(o, GP) and self.mean == o.mean and self.kernel == o.kernel


class IndependentMOGP:
    """reference src/independent_mogp.jl:10-12."""

    def __init__(self, fs: Sequence[GP], _post: Optional["_PostHandle"] = None):
        self.fs = list(fs)
        self._post = _post

    def __call__(self, x: "MOInputIsotopicByOutputs", sigma2=1e-18) -> "FiniteGP":
        """f(x, sigma2) or f(x, diag) with the diagonal of a general Diagonal noise (length n*p, ordered like x)."""
        # get_latent_gp(posterior(ilmm_dense(x, s2), y)): the latents of a dense-H posterior are COUPLED (one (mn) x (mn) state,
        # reference src/ilmm.jl:196-197); the verbs below serve them through the handle's latent view (H = I_m)
        if np.isscalar(sigma2) or getattr(sigma2, "ndim", 1) == 0:
            return FiniteGP(self, x, float(sigma2))
        return FiniteGP(self, x, sigma2)

    def __eq__(self, o):
        return isinstance(o, IndependentMOGP) and self.fs == o.fs and self._post is o._post


def independent_mogp(fs: Sequence[GP]) -> IndependentMOGP:
    """reference src/independent_mogp.jl:31."""
    return IndependentMOGP(fs)


class Orthogonal:
    """reference src/orthogonal_matrix.jl:11-34: H = U * sqrt(S); validates U'U ~ I."""

    def __init__(self, U, S, validate_fields: bool = True):
        # U is held column-major (what Julia holds and the C ABI takes): a row-major argument is copied once, here, and the ABI then
        # reads H.U's own buffer on every call (in-place edits of H.U are seen; edits of a row-major original are not)
        self.U = np.asfortranarray(np.asarray(U, dtype=np.float64))
        self.S = np.ascontiguousarray(np.asarray(S, dtype=np.float64).reshape(-1))      # the Diagonal's diag
        self._args = None                                         # (U object, S object, column-major image of U): see abi_args()
        if self.U.ndim != 2 or self.U.shape[1] != self.S.shape[0]:
            raise ValueError("U must be p x m and S of length m")
        if validate_fields:
            p, m = self.U.shape
            L.check(L.load().lmm_orthogonal_validate(L.Arr(L.colmajor(self.U)).ptr, C.c_int(p), C.c_int(m)))

    @property
    def shape(self) -> Tuple[int, int]:
        return self.U.shape

    def abi_args(self):
        """(U column-major, S) as the C ABI takes them.  A U that is already column-major (Fortran order, what Julia holds) goes over
        as it is, with no copy and therefore always current; a row-major U is re-imaged per call."""
        U, S = self.U, self.S
        a = self._args
        if a is not None and a[0] is U and a[1] is S:
            return a[2], a[3]
        Uc = L.colmajor(U)
        Ua, Sa = L.Arr(Uc), L.Arr(S)
        if Uc.base is U or Uc is U:                  # a view of the live buffer: safe to keep (in-place edits of U stay visible)
            self._args = (U, S, Ua, Sa)
        return Ua, Sa

    def collect(self) -> np.ndarray:
        """reference src/orthogonal_matrix.jl:27-30 (materialised H)."""
        return self.U * np.sqrt(self.S)[None, :]

    def __array__(self, dtype=None, copy=None):
        return self.collect()


class MOInputIsotopicByOutputs:
    """KernelFunctions.MOInputIsotopicByOutputs(x, out_dim): x is (n,) or ColVecs-style (d, n)."""

    def __init__(self, x, out_dim: int):
        self.x, self.out_dim = x, int(out_dim)

    @property
    def dim(self) -> int:
        return 1 if len(self.x.shape) == 1 else int(self.x.shape[0])

    @property
    def n(self) -> int:
        return int(self.x.shape[-1])

    def carr(self) -> L.Arr:
        x = self.x
        if len(x.shape) == 2:          # (d, n) -> d x n column-major == (n, d) C-order
            x = x.T.contiguous() if L._is_torch(x) else np.ascontiguousarray(np.asarray(x, dtype=np.float64).T)
        return L.Arr(x)

    def __len__(self):
        return self.n * self.out_dim


class MOInputIsotopicByFeatures:
    """KernelFunctions.MOInputIsotopicByFeatures(x, out_dim): index k -> (x[k // p], k % p) (all outputs of x_1, then x_2, ...).
    The reference supports it for IndependentMOGP only (src/independent_mogp.jl:128-229; ILMM `unpack` rejects it,
    src/ilmm.jl:45)."""

    def __init__(self, x, out_dim: int):
        self.x, self.out_dim = x, int(out_dim)

    @property
    def n(self) -> int:
        return int(self.x.shape[-1])

    def by_outputs(self) -> MOInputIsotopicByOutputs:
        return MOInputIsotopicByOutputs(self.x, self.out_dim)

    def __len__(self):
        return self.n * self.out_dim


def indices_which_reorder_features_to_outputs(x) -> np.ndarray:
    """reference src/independent_mogp.jl:141-145 (1-based, as in test/independent_mogp.jl:86-98): applied to a vector
    ordered by features it orders it by outputs."""
    return np.arange(1, len(x) + 1).reshape(x.n, x.out_dim).T.reshape(-1)


def indices_which_reorder_outputs_to_features(x) -> np.ndarray:
    """reference src/independent_mogp.jl:135-139."""
    return np.arange(1, len(x) + 1).reshape(x.out_dim, x.n).T.reshape(-1)


def _reorder(v, n: int, p: int, to_outputs: bool):
    """lmm_reorder: by-features <-> by-outputs (index permutation done by the library)."""
    L.ensure_init()
    a = L.Arr(v)
    out = _alloc_like(v, n * p)
    L.check(L.load().lmm_reorder(a.ptr, n, p, int(to_outputs), L.Arr(out, True).ptr))
    return out


class _PostHandle:
    """Owns an lmm_post_t* (device-resident posterior state); freed with the Python object, as the Julia
    shim does with a finalizer."""

    def __init__(self, ptr: C.c_void_p, l0: int, l1: int, dense: bool = False, train=None, latent: bool = False, parent=None, mix=None):
        self.ptr, self.l0, self.l1, self.dense = ptr, l0, l1, dense      # dense: coupled (mn) x (mn) state of a dense-H ILMM
        self.mix = mix                # dense only: the H (p x m) the conditioning batches were observed through (gradient of the latent view)
        # the conditioning batches [(x, sigma2, y), ...] the posterior was built from (references, no copies): the gradient of the
        # predictive logpdf is a total derivative through the posterior and needs them (one entry per posterior(...) call)
        self.train = train if (train is None or isinstance(train, list)) else [train]
        self.latent = latent          # dense only: this handle's H is I_m (it IS the latent PosteriorGP{IndependentMOGP})
        # A latent view does NOT reference its parent: the C side keeps the shared device state alive until the last of the two handles
        # is destroyed (either order), and a back-reference would make parent <-> view a cycle of objects with __del__, whose device
        # memory only the cyclic collector would free.
        self._parent = weakref.ref(parent) if parent is not None else None
        self._view = None

    def latent_view(self) -> "_PostHandle":
        """The latent PosteriorGP{IndependentMOGP} of a dense-H posterior (reference src/ilmm.jl:39 on :196-197) as a handle of
        its own: lmm_ilmm_post_latent_view (shares the factor, H = I_m)."""
        if self.latent:
            return self
        if self._view is None:
            h = C.c_void_p()
            L.check(L.load().lmm_ilmm_post_latent_view(self.ptr, C.byref(h)))
            self._view = _PostHandle(h, self.l0, self.l1, dense=True, latent=True, parent=self)
        return self._view

    def __del__(self):
        try:
            if self.ptr:
                L.load().lmm_post_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class ILMM:
    """reference src/ilmm.jl:16-19.  `ILMM(f, H)` with H a dense p x m matrix or an `Orthogonal` (=> OILMM,
    reference src/oilmm.jl:13).  `shard=(begin, end)` restricts this process to a block of latents
    (one process per GPU); partial results are combined by `parallel.py`."""

    def __init__(self, f: IndependentMOGP, H, shard: Optional[Tuple[int, int]] = None):
        self.f, self.H = f, H if isinstance(H, Orthogonal) else np.asarray(H, dtype=np.float64)
        m = len(f.fs)
        if self.H.shape[1] != m:
            raise ValueError(f"H has {self.H.shape[1]} columns but there are {m} latent processes")
        self.shard = (0, m) if shard is None else (int(shard[0]), int(shard[1]))

    @property
    def is_oilmm(self) -> bool:
        return isinstance(self.H, Orthogonal)

    def __call__(self, x: MOInputIsotopicByOutputs, sigma2: float = 1e-18) -> "FiniteGP":
        return FiniteGP(self, x, float(sigma2))


def OILMM(f: IndependentMOGP, H: Orthogonal, **kw) -> ILMM:
    if not isinstance(H, Orthogonal):
        raise TypeError("OILMM needs an Orthogonal mixing matrix")
    return ILMM(f, H, **kw)


def get_latent_gp(f: ILMM) -> IndependentMOGP:
    """reference src/ilmm.jl:39."""
    return f.f


class Normal:
    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma


class FiniteGP:
    """AbstractGPs.FiniteGP(f, x, Diagonal(Fill(sigma2, n*p))).  `sigma2` may also be a length n*p vector -- the diagonal of a
    general `Diagonal` noise, ordered like x -- which the reference accepts for IndependentMOGP logpdf only
    (src/independent_mogp.jl:149-159, 222-229; ILMM `noise_var` requires a Fill, src/ilmm.jl:41)."""

    def __init__(self, f, x: MOInputIsotopicByOutputs, sigma2):
        self.f, self.x, self.sigma2 = f, x, sigma2

    @property
    def heteroscedastic(self) -> bool:
        return not np.isscalar(self.sigma2) and getattr(self.sigma2, "ndim", 1) > 0

    def __len__(self):
        return len(self.x)


# ---- helpers ----------------------------------------------------------------------------------------
def noise_var(sigma2):
    """reference src/ilmm.jl:41."""
    return sigma2


def reshape_y(y, n: int):
    """reference src/ilmm.jl:43: reshape(y, N, :)'."""
    return np.asarray(y).reshape(-1, n)


def unpack(fx: FiniteGP):
    """reference src/ilmm.jl:45-54."""
    f = fx.f
    if fx.x.out_dim != f.H.shape[0]:
        raise RuntimeError("out dim of x != out dim of f.")
    return f.f, f.H, fx.sigma2, fx.x.x


def _merged_train(train, p: int):
    """The conditioning batches of a (sequentially conditioned) posterior as ONE set of points: posterior(posterior(f(x1, s1), y1)(x2, s2),
    y2) is the posterior given ([x1 x2], [y1; y2]) under per-batch noise (exact conditioning), which is what the *_post_logpdf_grad_seq
    entry points serve (one noise block per batch + one for the test points).  Returns (x_all, [s2 per batch], y_all, sizes); y is
    by-outputs, so the batches interleave per output."""
    if train is None:
        raise NotImplementedError("this posterior does not carry its training data (built outside posterior(fx, y))")
    if any(np.ndim(t[1]) > 0 for t in train):
        raise NotImplementedError("gradient of the predictive logpdf after conditioning with a per-point (Diagonal) noise is not built "
                                  "(scalar noise variances, one per conditioning batch, are)")
    if len(train) == 1:
        x0, s20, y0 = train[0]
        return x0, [float(s20)], y0, [x0.n]
    if len(train) > 7:
        raise NotImplementedError("gradient of the predictive logpdf after more than 7 conditioning batches is not built")
    xs = [np.asarray(t[0].x.cpu() if L._is_torch(t[0].x) else t[0].x, dtype=np.float64) for t in train]
    ys = [np.asarray(t[2].cpu() if L._is_torch(t[2]) else t[2], dtype=np.float64).reshape(p, -1) for t in train]
    x_all = np.concatenate(xs, axis=-1)
    y_all = np.concatenate(ys, axis=1).reshape(-1)
    return MOInputIsotopicByOutputs(x_all, p), [float(t[1]) for t in train], y_all, [t[0].n for t in train]


def _batch_args(s2b, sizes):
    """(batch_n, batch_sigma2, grad_batch_sigma2) ctypes arrays of a *_post_logpdf_grad_seq call."""
    k = len(sizes)
    return (C.c_int * k)(*sizes), (C.c_double * k)(*s2b), (C.c_double * k)()


def _train_noise_grad(gb, s2b):
    """d/d(training noise): ONE number when the batches share their variance (the derivative w.r.t. that shared value), else one per
    conditioning batch."""
    g = [float(v) for v in gb]
    return sum(g) if len(set(s2b)) == 1 else g


def _split_train_grad(gy, sizes, p: int):
    """d/dy of the merged batch back into one by-outputs vector per conditioning batch."""
    if len(sizes) == 1:
        return gy
    g = np.asarray(gy.cpu() if L._is_torch(gy) else gy).reshape(p, -1)
    out, o = [], 0
    for nb in sizes:
        out.append(np.ascontiguousarray(g[:, o:o + nb]).reshape(-1)); o += nb
    return out


def _gps_arg(mogp):
    """(descs, lmm_gp_t array) of an IndependentMOGP's latents.  The ctypes array is rebuilt only when a hyperparameter changed (the key
    is the tuple of current values: building it costs a third of filling the array, which at m = 20 is 25 us of a 380-us call)."""
    key = tuple((g.kernel.kind, g.kernel.variance, g.kernel.lengthscale, g.mean) for g in mogp.fs)
    hit = getattr(mogp, "_gps_cache", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    descs = [g.desc() for g in mogp.fs]
    arr = L.gps_array(descs)
    mogp._gps_cache = (key, descs, arr)
    return descs, arr


def _H_args(H):
    """(U or dense-H pointer, S pointer or None, p, m) for the C ABI."""
    if isinstance(H, Orthogonal):
        Ua, Sa = H.abi_args()
        return Ua, Sa, H.U.shape[0], H.U.shape[1]
    return L.Arr(L.colmajor(H)), None, H.shape[0], H.shape[1]


def _alloc_like(ref, count: int):
    """Output buffer on the same side (host / device) as `ref`."""
    if L._is_torch(ref) and ref.is_cuda:
        import torch
        return torch.empty(count, dtype=torch.float64, device=ref.device)
    return np.empty(count, dtype=np.float64)


# Dense-H ILMM logpdf: allow the identical-kernel decoupled shortcut (exact).  Set False to force
# the reference's single (mn) x (mn) factorisation.  ILMM_LAST_PATH records which ran.
ILMM_ALLOW_DECOUPLED = True
ILMM_LAST_PATH = None


# ---- the AbstractGPs verbs ----------------------------------------------------------------------------
def logpdf(fx: FiniteGP, y, with_regulariser: bool = True) -> float:
    """logpdf(fx, y).  ILMM/OILMM: reference src/oilmm.jl:79-93, src/ilmm.jl:150-163; IndependentMOGP:
    src/independent_mogp.jl:74-80.  Returns this process's shard of the sum (the whole value when the model
    is not sharded)."""
    L.ensure_init()
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    if isinstance(x, MOInputIsotopicByFeatures):          # reference src/independent_mogp.jl:222-229
        if not isinstance(f, IndependentMOGP):
            raise TypeError("ILMM needs MOInputIsotopicByOutputs (reference src/ilmm.jl:45)")
        if fx.heteroscedastic:                            # reorder_by_outputs(Sigma_y, x): src/independent_mogp.jl:149-151
            s2 = _reorder(s2, x.n, x.out_dim, True)
        return logpdf(FiniteGP(f, x.by_outputs(), s2), _reorder(y, x.n, x.out_dim, True))
    if fx.heteroscedastic:
        if not isinstance(f, IndependentMOGP) or f._post is not None:
            raise TypeError("per-point Diagonal noise is supported for the prior IndependentMOGP logpdf only "
                            "(reference src/ilmm.jl:41 requires Diagonal{<:Real,<:Fill})")
        if x.out_dim != len(f.fs):
            raise RuntimeError("out dim of x != out dim of f.")
        ya, na = L.Arr(y), L.Arr(s2)
        if ya.size != x.n * x.out_dim or na.size != x.n * x.out_dim:
            raise ValueError("length(y), length(diag(Sigma_y)) != n * out_dim")
        out = C.c_double()
        L.check(lib.lmm_mogp_logpdf_diag(x.carr().ptr, x.dim, x.n, ya.ptr, len(f.fs), na.ptr, L.gps_array([g.desc() for g in f.fs]),
                                         0, len(f.fs), C.byref(out)))
        return out.value
    if hasattr(y, "shape") and len(y.shape) == 2:         # logpdf(fx, Y::AbstractMatrix): one value per column
        return _logpdf_matrix(fx, y, with_regulariser)
    out = C.c_double()
    xa, ya = x.carr(), L.Arr(y)
    if isinstance(f, IndependentMOGP):
        if x.out_dim != len(f.fs):
            raise RuntimeError("out dim of x != out dim of f.")
        if ya.size != x.n * x.out_dim:
            raise ValueError("length(y) != n * out_dim")
        if f._post is not None and f._post.dense:      # latent PosteriorGP of a dense-H posterior: generic Gaussian logpdf
            L.check(lib.lmm_ilmm_post_logpdf(f._post.latent_view().ptr, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr,
                                             L.jitters((0.0, s2, 0.0)), C.byref(out)))
            return out.value
        if f._post is not None:
            return _post_logpdf(f._post, [g.desc() for g in f.fs], np.eye(len(f.fs)), np.ones(len(f.fs)), x, s2, ya, False)
        gps = L.gps_array([g.desc() for g in f.fs])
        L.check(lib.lmm_mogp_logpdf(xa.ptr, x.dim, x.n, ya.ptr, len(f.fs), C.c_double(s2), gps, 0, len(f.fs), C.byref(out)))
        return out.value
    unpack(fx)
    if ya.size != x.n * x.out_dim:
        raise ValueError("length(y) != n * out_dim")
    descs, gps = _gps_arg(f.f)
    Ua, Sa, p, m = _H_args(f.H)
    l0, l1 = f.shard
    if f.f._post is not None:
        if not f.is_oilmm:         # dense-H posterior: reference test/ilmm.jl:25
            L.check(lib.lmm_ilmm_post_logpdf(f.f._post.ptr, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr, None, C.byref(out)))
            return out.value
        return _post_logpdf(f.f._post, descs, f.H.U, f.H.S, x, s2, ya, with_regulariser)
    if f.is_oilmm:
        L.check(lib.lmm_oilmm_logpdf(xa.ptr, x.dim, x.n, ya.ptr, p, Ua.ptr, Sa.ptr, m, C.c_double(s2), gps, l0, l1,
                                     int(with_regulariser), C.byref(out)))
    else:
        if (l0, l1) != (0, m):
            raise NotImplementedError("dense-H ILMM does not shard (SURVEY.md 8e: replicas only)")
        path = C.c_int(0)
        L.check(lib.lmm_ilmm_logpdf_ex(xa.ptr, x.dim, x.n, ya.ptr, p, Ua.ptr, m, C.c_double(s2), gps, None,
                                       int(ILMM_ALLOW_DECOUPLED), C.byref(path), C.byref(out)))
        global ILMM_LAST_PATH
        ILMM_LAST_PATH = "decoupled" if path.value else "dense"
    return out.value


def logpdf_and_gradient(fx: FiniteGP, y, with_regulariser: bool = True) -> dict:
    """Value and gradient of logpdf(fx, y) -- what `Zygote.gradient(logpdf, fx, y)` differentiates in the reference's tests
    (test/oilmm.jl:31-32, test/ilmm.jl:31-32, test/independent_mogp.jl:65-66).

      prior OILMM / IndependentMOGP : {"value", "y", "sigma2", "S", "U", "gps": [{"variance","lengthscale","mean"}, ...]}
      prior dense-H ILMM            : {"value", "y", "sigma2", "H", "gps"}
      posterior dense-H ILMM        : as the posterior OILMM below with "H" in place of "S", "U" (test/ilmm.jl:32)
      posterior OILMM / MOGP        : fx = posterior(f(x, s2), y0)(xs, s2s); TOTAL derivatives of the predictive logpdf through the
                                      posterior: {"value", "y" (= d/d ys), "y_train", "sigma2" (= d/d s2s), "sigma2_train", "S", "U", "gps"}
                                      After sequential conditioning (equal noise variance per batch) "y_train" is a LIST with one
                                      by-outputs vector per conditioning batch, while "sigma2_train" stays ONE number: the derivative
                                      with respect to the variance the batches share (= the sum of the per-batch derivatives).
    Partial sums over the latent shard."""
    L.ensure_init()
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    mogp = isinstance(f, IndependentMOGP)
    post = f._post if mogp else (f.f._post if isinstance(f, ILMM) else None)
    if not mogp and not isinstance(f, ILMM):
        raise TypeError("logpdf_and_gradient needs an ILMM / OILMM / IndependentMOGP FiniteGP")
    if mogp and post is not None and post.dense:
        # logpdf(get_latent_gp(posterior(ilmm(x, s2), y))(xs, s2s), zs): the coupled latent PosteriorGP of a dense-H posterior (reference
        # src/ilmm.jl:39 on the ILMM of :196-197).  Joint density of the conditioning batches (observed through H) and the latent test
        # block (observed through [I; 0]) minus the marginal of the batches: lmm_ilmm_post_latent_logpdf_grad_seq.
        if post.latent or post.mix is None:
            raise NotImplementedError("gradient of the latent view after conditioning ON latent observations is not built")
        m = len(f.fs)
        if x.out_dim != m:
            raise RuntimeError("out dim of x != out dim of f.")
        Ha, _, p, _m = _H_args(post.mix)
        n = x.n
        x0, s2b, y0, sizes = _merged_train(post.train, p)
        bn, bs, gb = _batch_args(s2b, sizes)
        val, gs2 = C.c_double(), C.c_double()
        gy, gH = _alloc_like(y if L._is_torch(y) else x.x, n * m), np.empty(p * m)
        gy0 = _alloc_like(y0 if L._is_torch(y0) else x0.x, x0.n * p)
        gg = (L.GpGradT * m)()
        L.check(lib.lmm_ilmm_post_latent_logpdf_grad_seq(x0.carr().ptr, x0.dim, x0.n, bn, bs, len(sizes), L.Arr(y0).ptr, x.carr().ptr, n,
                                                         L.Arr(y).ptr, p, Ha.ptr, m, C.c_double(s2), L.gps_array([g.desc() for g in f.fs]),
                                                         None, C.byref(val), L.Arr(gy0, True).ptr, L.Arr(gy, True).ptr, gb, C.byref(gs2),
                                                         L.Arr(gH, True).ptr, gg))
        return {"value": val.value, "y": gy, "y_train": _split_train_grad(gy0, sizes, p), "sigma2": gs2.value,
                "sigma2_train": _train_noise_grad(gb, s2b), "H": gH.reshape(m, p).T.copy(),
                "gps": [{"variance": gg[l].variance, "lengthscale": gg[l].lengthscale, "mean": gg[l].mean} for l in range(m)]}
    if not mogp and not f.is_oilmm:
        unpack(fx)
        Ha, _, p, m = _H_args(f.H)
        n = x.n
        val, gs2 = C.c_double(), C.c_double()
        gy, gH = _alloc_like(y if L._is_torch(y) else x.x, n * p), np.empty(p * m)
        gg = (L.GpGradT * m)()
        if post is not None:          # reference test/ilmm.jl:32: gradient(logpdf, pi, y_test) on the dense-H posterior
            x0, s2b, y0, sizes = _merged_train(post.train, p)
            bn, bs, gb = _batch_args(s2b, sizes)
            gy0 = _alloc_like(y0 if L._is_torch(y0) else x0.x, x0.n * p)
            L.check(lib.lmm_ilmm_post_logpdf_grad_seq(x0.carr().ptr, x0.dim, x0.n, bn, bs, len(sizes), L.Arr(y0).ptr, x.carr().ptr, n,
                                                      L.Arr(y).ptr, p, Ha.ptr, m, C.c_double(s2), L.gps_array([g.desc() for g in f.f.fs]),
                                                      None, C.byref(val), L.Arr(gy0, True).ptr, L.Arr(gy, True).ptr, gb, C.byref(gs2),
                                                      L.Arr(gH, True).ptr, gg))
            return {"value": val.value, "y": gy, "y_train": _split_train_grad(gy0, sizes, p), "sigma2": gs2.value,
                    "sigma2_train": _train_noise_grad(gb, s2b),
                    "H": gH.reshape(m, p).T.copy(),
                    "gps": [{"variance": gg[l].variance, "lengthscale": gg[l].lengthscale, "mean": gg[l].mean} for l in range(m)]}
        L.check(lib.lmm_ilmm_logpdf_grad(x.carr().ptr, x.dim, n, L.Arr(y).ptr, p, Ha.ptr, m, C.c_double(s2),
                                         L.gps_array([g.desc() for g in f.f.fs]), None, C.byref(val), L.Arr(gy, True).ptr,
                                         C.byref(gs2), L.Arr(gH, True).ptr, gg))
        return {"value": val.value, "y": gy, "sigma2": gs2.value, "H": gH.reshape(m, p).T.copy(),
                "gps": [{"variance": gg[l].variance, "lengthscale": gg[l].lengthscale, "mean": gg[l].mean} for l in range(m)]}
    if mogp:                      # gradient(logpdf, fx, y) on an IndependentMOGP (reference test/independent_mogp.jl:65-66):
        m = p = len(f.fs)         # the OILMM with U = I, S = 1 (regulariser identically 0, so it is skipped)
        if x.out_dim != m:
            raise RuntimeError("out dim of x != out dim of f.")
        Ua, Sa, descs, shard, with_regulariser = L.Arr(L.colmajor(np.eye(m))), L.Arr(np.ones(m)), [g.desc() for g in f.fs], (0, m), False
    else:
        unpack(fx)
        Ua, Sa, p, m = _H_args(f.H)
        descs, shard = [g.desc() for g in f.f.fs], f.shard
    n = x.n
    val, gs2 = C.c_double(), C.c_double()
    gy, gS, gU = _alloc_like(y if L._is_torch(y) else x.x, n * p), np.empty(m), np.empty(p * m)
    gg = (L.GpGradT * m)()
    out = {}
    if post is None:
        L.check(lib.lmm_oilmm_logpdf_grad(x.carr().ptr, x.dim, n, L.Arr(y).ptr, p, Ua.ptr, Sa.ptr, m, C.c_double(s2),
                                          L.gps_array(descs), shard[0], shard[1], int(with_regulariser), C.byref(val),
                                          L.Arr(gy, True).ptr, C.byref(gs2), L.Arr(gS, True).ptr, L.Arr(gU, True).ptr, gg))
    else:
        x0, s2b, y0, sizes = _merged_train(post.train, p)
        bn, bs, gb = _batch_args(s2b, sizes)
        gy0 = _alloc_like(y0 if L._is_torch(y0) else x0.x, x0.n * p)
        L.check(lib.lmm_oilmm_post_logpdf_grad_seq(x0.carr().ptr, x0.dim, x0.n, bn, bs, len(sizes), L.Arr(y0).ptr, x.carr().ptr, n,
                                                   L.Arr(y).ptr, p, Ua.ptr, Sa.ptr, m, C.c_double(s2), L.gps_array(descs), shard[0],
                                                   shard[1], int(with_regulariser), C.byref(val), L.Arr(gy0, True).ptr,
                                                   L.Arr(gy, True).ptr, gb, C.byref(gs2), L.Arr(gS, True).ptr, L.Arr(gU, True).ptr, gg))
        out.update(y_train=_split_train_grad(gy0, sizes, p), sigma2_train=_train_noise_grad(gb, s2b))
    out.update({"value": val.value, "y": gy, "sigma2": gs2.value,
                "gps": [{"variance": gg[l].variance, "lengthscale": gg[l].lengthscale, "mean": gg[l].mean} for l in range(m)]})
    if not mogp:
        out["S"], out["U"] = gS, gU.reshape(m, p).T.copy()
    return out


def _logpdf_matrix(fx: FiniteGP, Y, with_regulariser: bool = True) -> np.ndarray:
    """logpdf(fx, Y) for Y of shape (n*p, ncol): ONE factorisation per latent, the columns ride as extra right-hand sides."""
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    ncol = int(Y.shape[1])
    Yc = Y.T.contiguous() if L._is_torch(Y) else np.ascontiguousarray(np.asarray(Y, dtype=np.float64).T)   # column-major image
    fpost = f._post if isinstance(f, IndependentMOGP) else f.f._post
    if fpost is not None:          # posterior models (TestUtils on po(x*, s2): reference test/oilmm.jl:36, test/ilmm.jl:36): the columns
        return np.array([logpdf(fx, Yc[c], with_regulariser) for c in range(ncol)])      # are evaluated one by one on the handle
    if isinstance(f, IndependentMOGP):
        m = len(f.fs)
        descs, Ua, Sa, p, shard, post = [g.desc() for g in f.fs], L.Arr(L.colmajor(np.eye(m))), L.Arr(np.ones(m)), m, (0, m), f._post
        with_regulariser = False                   # U = I, S = 1, p == m: the regulariser is identically 0 only up to its
        s2_eff = s2                                # log S = 0 and (p-m) = 0 terms; skip it exactly
    else:
        unpack(fx)
        if not f.is_oilmm:                         # dense H: one (mn) x (mn) factorisation, the columns ride it
            Ha, _, p, m = _H_args(f.H)
            if x.out_dim != p:
                raise RuntimeError("out dim of x != out dim of f.")
            out = np.empty(ncol)
            L.check(lib.lmm_ilmm_logpdf_multi(x.carr().ptr, x.dim, x.n, L.Arr(Yc).ptr, p, ncol, Ha.ptr, m, C.c_double(s2),
                                              L.gps_array([g.desc() for g in f.f.fs]), None, L.Arr(out, True).ptr))
            return out
        descs, (Ua, Sa, p, m), shard, post, s2_eff = [g.desc() for g in f.f.fs], _H_args(f.H), f.shard, f.f._post, s2
    if x.out_dim != p:
        raise RuntimeError("out dim of x != out dim of f.")
    out = np.empty(ncol)
    L.check(lib.lmm_oilmm_logpdf_multi(x.carr().ptr, x.dim, x.n, L.Arr(Yc).ptr, p, ncol, Ua.ptr, Sa.ptr, m, C.c_double(s2_eff),
                                       L.gps_array(descs), shard[0], shard[1], int(with_regulariser), L.Arr(out, True).ptr))
    return out


def _post_logpdf(post: _PostHandle, descs, U, S, x, s2, ya, with_reg) -> float:
    lib = L.load()
    out = C.c_double()
    Ua, Sa = L.Arr(L.colmajor(U)), L.Arr(S)
    xa = x.carr()
    L.check(lib.lmm_oilmm_post_logpdf(post.ptr, Ua.ptr, Sa.ptr, U.shape[0], U.shape[1], C.c_double(s2), xa.ptr, x.dim,
                                      x.n, ya.ptr, int(with_reg), C.byref(out)))
    return out.value


def _more_train(post: "_PostHandle", x, s2, y):
    """Conditioning batches of posterior(po(x, s2), y): those of `po` plus this one (None if `po` does not know its own)."""
    return None if post.train is None else post.train + [(x, s2, y)]


def posterior(fx: FiniteGP, y):
    """posterior(fx, y): reference src/oilmm.jl:116-134 (returns ILMM(independent_mogp(posteriors), H) -- again
    an OILMM with the same H) and src/independent_mogp.jl:119-126."""
    L.ensure_init()
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    if isinstance(x, MOInputIsotopicByFeatures):
        return posterior(FiniteGP(f, x.by_outputs(), s2), _reorder(y, x.n, x.out_dim, True))
    xa, ya = x.carr(), L.Arr(y)
    handle = C.c_void_p()
    if isinstance(f, IndependentMOGP):
        m = len(f.fs)
        if f._post is not None and f._post.dense:      # posterior(f_latent(x2, s2), y2) on the coupled latent PosteriorGP
            L.check(lib.lmm_ilmm_post_condition(f._post.latent_view().ptr, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr,
                                                L.jitters((0.0, s2, 0.0)), C.byref(handle)))
            return IndependentMOGP(f.fs, _PostHandle(handle, 0, m, dense=True, latent=True))
        if f._post is not None:        # sequential conditioning: posterior(po(x2, s2), y2)
            Ui, Si = L.Arr(L.colmajor(np.eye(m))), L.Arr(np.ones(m))
            L.check(lib.lmm_post_condition(f._post.ptr, Ui.ptr, Si.ptr, m, m, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr,
                                           C.byref(handle)))
            return IndependentMOGP(f.fs, _PostHandle(handle, 0, m, train=_more_train(f._post, x, s2, y)))
        gps = L.gps_array([g.desc() for g in f.fs])
        L.check(lib.lmm_mogp_posterior_create(xa.ptr, x.dim, x.n, ya.ptr, m, C.c_double(s2), gps, 0, m, C.byref(handle)))
        return IndependentMOGP(f.fs, _PostHandle(handle, 0, m, train=(x, s2, y)))
    unpack(fx)
    Ua, Sa, p, m = _H_args(f.H)
    l0, l1 = f.shard
    if f.f._post is not None:          # sequential conditioning of a posterior OILMM (same H: reference src/oilmm.jl:133)
        if not f.is_oilmm:         # dense-H posterior: both projected data sets condition the prior (reference src/ilmm.jl:184-198)
            L.check(lib.lmm_ilmm_post_condition(f.f._post.ptr, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr, None, C.byref(handle)))
            return ILMM(IndependentMOGP(f.f.fs, _PostHandle(handle, l0, l1, dense=True, train=_more_train(f.f._post, x, s2, y), mix=f.H)), f.H,
                        shard=f.shard)
        L.check(lib.lmm_post_condition(f.f._post.ptr, Ua.ptr, Sa.ptr, p, m, C.c_double(s2), xa.ptr, x.dim, x.n, ya.ptr,
                                       C.byref(handle)))
        return ILMM(IndependentMOGP(f.f.fs, _PostHandle(handle, l0, l1, train=_more_train(f.f._post, x, s2, y))), f.H, shard=f.shard)
    gps = L.gps_array([g.desc() for g in f.f.fs])
    if f.is_oilmm:
        L.check(lib.lmm_oilmm_posterior_create(xa.ptr, x.dim, x.n, ya.ptr, p, Ua.ptr, Sa.ptr, m, C.c_double(s2), gps, l0,
                                               l1, C.byref(handle)))
    else:
        L.check(lib.lmm_ilmm_posterior_create(xa.ptr, x.dim, x.n, ya.ptr, p, Ua.ptr, m, C.c_double(s2), gps, None,
                                              C.byref(handle)))
    return ILMM(IndependentMOGP(f.f.fs, _PostHandle(handle, l0, l1, dense=not f.is_oilmm, train=(x, s2, y), mix=None if f.is_oilmm else f.H)),
                f.H, shard=f.shard)


def mean_and_var(fx: FiniteGP, add_noise: bool = True):
    """mean_and_var(fx): reference src/oilmm.jl:57-76 (OILMM) and src/independent_mogp.jl:50,55.  For a sharded
    model the outputs are this shard's partial sums (add_noise only on one rank)."""
    L.ensure_init()
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    if isinstance(x, MOInputIsotopicByFeatures):          # reference src/independent_mogp.jl:169-215
        mo, vo = mean_and_var(FiniteGP(f, x.by_outputs(), s2), add_noise)
        return _reorder(mo, x.n, x.out_dim, False), _reorder(vo, x.n, x.out_dim, False)
    xa = x.carr()
    if isinstance(f, IndependentMOGP):
        m = len(f.fs)
        if x.out_dim != m:
            raise RuntimeError("out dim of x != out dim of f.")
        mean, var = _alloc_like(x.x, x.n * m), _alloc_like(x.x, x.n * m)
        ma, va = L.Arr(mean, True), L.Arr(var, True)
        if f._post is not None and f._post.dense:      # coupled latent PosteriorGP: diag of its joint covariance + sigma2
            L.check(lib.lmm_ilmm_post_mean_and_var(f._post.latent_view().ptr, C.c_double(s2), xa.ptr, x.dim, x.n,
                                                   L.jitters((0.0, s2, 0.0)), ma.ptr, va.ptr))
            return mean, var
        post = f._post.ptr if f._post is not None else None
        gps = L.gps_array([g.desc() for g in f.fs])
        L.check(lib.lmm_latent_marginals(post, gps, m, xa.ptr, x.dim, x.n, ma.ptr, va.ptr))
        return mean, var + s2                      # var(f, x) + Sigma_y diagonal
    unpack(fx)
    Ua, Sa, p, m = _H_args(f.H)
    if not f.is_oilmm and f.f._post is not None:
        # dense-H posterior: coupled latents (reference src/ilmm.jl:108-129 on the PosteriorGP of :196-197)
        mean, var = _alloc_like(x.x, x.n * p), _alloc_like(x.x, x.n * p)
        ma, va = L.Arr(mean, True), L.Arr(var, True)
        L.check(lib.lmm_ilmm_post_mean_and_var(f.f._post.ptr, C.c_double(s2), xa.ptr, x.dim, x.n, None, ma.ptr, va.ptr))
        return mean, var
    l0, l1 = f.shard
    mean, var = _alloc_like(x.x, x.n * p), _alloc_like(x.x, x.n * p)
    ma, va = L.Arr(mean, True), L.Arr(var, True)
    post = f.f._post.ptr if f.f._post is not None else None
    gps = L.gps_array([g.desc() for g in f.f.fs])
    # dense-H prior: latents are independent, so V = abs2.(H) * V_latent + s2 exactly as in the OILMM form (S == NULL)
    L.check(lib.lmm_oilmm_mean_and_var(post, gps, Ua.ptr, Sa.ptr if Sa is not None else None, p, m, l0, l1,
                                       C.c_double(s2), int(add_noise), xa.ptr, x.dim, x.n, None, ma.ptr, va.ptr))
    return mean, var


def mean_and_cov(fx: FiniteGP):
    """mean_and_cov(fx): reference src/ilmm.jl:132-139 (ILMM/OILMM) and AbstractGPs' generic form over
    src/independent_mogp.jl:60-63 (IndependentMOGP).  Returns (mean, C) with C (p n) x (p n), by-outputs order."""
    L.ensure_init()
    lib = L.load()
    f, x, s2 = fx.f, fx.x, fx.sigma2
    xa = x.carr()
    if isinstance(f, IndependentMOGP):
        m = len(f.fs)
        if x.out_dim != m:
            raise RuntimeError("out dim of x != out dim of f.")
        if f._post is not None and f._post.dense:      # coupled latent PosteriorGP: its joint covariance + sigma2 I
            n = x.n
            mean, cov = np.empty(n * m), np.empty((n * m) * (n * m))
            L.check(lib.lmm_ilmm_post_mean_and_cov(f._post.latent_view().ptr, C.c_double(s2), xa.ptr, x.dim, n,
                                                   L.jitters((0.0, s2, 0.0)), L.Arr(mean, True).ptr, L.Arr(cov, True).ptr))
            return mean, cov.reshape(n * m, n * m).T
        Ua, Sa, p, post, descs, shard = L.Arr(L.colmajor(np.eye(m))), None, m, f._post, [g.desc() for g in f.fs], (0, m)
        jit = L.jitters((1e-9, 0.0, 0.0))          # cov(f, x) + Sigma_y: no latent jitter for a bare MOGP
    else:
        unpack(fx)
        if not f.is_oilmm and f.f._post is not None:      # coupled latents: reference src/ilmm.jl:132-139 on the PosteriorGP
            n, p = x.n, f.H.shape[0]
            mean, cov = np.empty(n * p), np.empty((n * p) * (n * p))
            L.check(lib.lmm_ilmm_post_mean_and_cov(f.f._post.ptr, C.c_double(s2), xa.ptr, x.dim, n, None, L.Arr(mean, True).ptr,
                                                   L.Arr(cov, True).ptr))
            return mean, cov.reshape(n * p, n * p).T
        Ua, Sa, p, m = _H_args(f.H)
        post, descs, shard, jit = f.f._post, [g.desc() for g in f.f.fs], f.shard, None
    n = x.n
    mean, cov = np.empty(n * p), np.empty((n * p) * (n * p))
    L.check(lib.lmm_lmm_mean_and_cov(post.ptr if post is not None else None, L.gps_array(descs), Ua.ptr,
                                     Sa.ptr if Sa is not None else None, p, m, shard[0], shard[1], C.c_double(s2), 1, xa.ptr,
                                     x.dim, n, jit, L.Arr(mean, True).ptr, L.Arr(cov, True).ptr))
    return mean, cov.reshape(n * p, n * p).T       # column-major -> (row, col); symmetric


def cov(fx, x=None, y=None):
    """cov(fx): reference src/ilmm.jl:147.
    cov(f::IndependentMOGP, x, y): the two-input cross-covariance, reference src/independent_mogp.jl:66-71 (both inputs by outputs)
    and :184-215 (either one MOInputIsotopicByFeatures); cov(f::IndependentMOGP, x) = cov(f, x, x) (:60-63, :176-181).  Prior or
    (independent) posterior latents; (m n) x (m n2), one lmm_mogp_cross_cov call."""
    if isinstance(fx, IndependentMOGP):
        if x is None:
            raise TypeError("cov(f::IndependentMOGP, x[, y]) needs the inputs")
        return _mogp_cross_cov(fx, x, x if y is None else y)
    if x is not None or y is not None:
        raise TypeError("cov(f, x, y) is defined for an IndependentMOGP (reference src/independent_mogp.jl:66-71)")
    return mean_and_cov(fx)[1]


def _mogp_cross_cov(f: IndependentMOGP, x, y) -> np.ndarray:
    L.ensure_init()
    m = len(f.fs)
    if x.out_dim != m or y.out_dim != m:
        raise RuntimeError("out dim of x != out dim of f.")
    if f._post is not None and f._post.dense:
        raise NotImplementedError("cov(f, x, y) of the coupled latent PosteriorGP of a dense-H posterior is not built")
    xb = x.by_outputs() if isinstance(x, MOInputIsotopicByFeatures) else x
    yb = y.by_outputs() if isinstance(y, MOInputIsotopicByFeatures) else y
    if xb.dim != yb.dim:
        raise ValueError("x and y have different input dimensions")
    n, n2 = xb.n, yb.n
    out = np.empty((m * n) * (m * n2))
    post = f._post.ptr if f._post is not None else None
    L.check(L.load().lmm_mogp_cross_cov(post, L.gps_array([g.desc() for g in f.fs]), m, 0, m, xb.carr().ptr, xb.dim, n,
                                        int(isinstance(x, MOInputIsotopicByFeatures)), yb.carr().ptr, n2,
                                        int(isinstance(y, MOInputIsotopicByFeatures)), L.Arr(out, True).ptr))
    return out.reshape(m * n2, m * n).T          # column-major (m n) x (m n2) -> (row, col)


def mean(fx: FiniteGP):
    """reference src/ilmm.jl:142 (mean_and_var(fx)[1]).  For an OILMM (prior or posterior, by-outputs inputs) the means alone
    are computed -- mu + K(x*, x) alpha per latent, no triangular solve for variances that would be discarded."""
    f, x = fx.f, fx.x
    if isinstance(f, ILMM) and isinstance(x, MOInputIsotopicByOutputs) and (f.is_oilmm or f.f._post is None):
        L.ensure_init()
        unpack(fx)
        Ua, Sa, p, m = _H_args(f.H)
        l0, l1 = f.shard
        out = _alloc_like(x.x, x.n * p)
        post = f.f._post.ptr if f.f._post is not None else None
        L.check(L.load().lmm_oilmm_mean_and_var(post, L.gps_array([g.desc() for g in f.f.fs]), Ua.ptr,
                                               Sa.ptr if Sa is not None else None, p, m, l0, l1, C.c_double(fx.sigma2), 0,
                                               x.carr().ptr, x.dim, x.n, None, L.Arr(out, True).ptr, None))
        return out
    return mean_and_var(fx)[0]


def var(fx: FiniteGP):
    """reference src/ilmm.jl:145."""
    return mean_and_var(fx)[1]


def marginals(fx: FiniteGP) -> Normal:
    """AbstractGPs.marginals(fx) = Normal.(mean, sqrt.(var)) (vectorised)."""
    m, v = mean_and_var(fx)
    return Normal(m, v ** 0.5)


class DeviceNormals:
    """Standard normals generated on the GPU (lmm_normals: Philox4x32-10 + Box-Muller, Float64).  Pass it to `rand` in place of
    a numpy Generator when the reference's host random stream is not needed: the draw ORDER of the reference is kept (latent
    normals, then noise normals), the buffers never leave the device, and the sample comes back as a device tensor."""

    def __init__(self, seed: int):
        self.seed, self.stream = int(seed), 0

    def standard_normal(self, count: int):
        import torch
        L.ensure_init()
        out = torch.empty(int(count), dtype=torch.float64, device="cuda")
        L.check(L.load().lmm_normals(C.c_ulonglong(self.seed), C.c_ulonglong(self.stream), C.c_size_t(int(count)),
                                     C.c_void_p(out.data_ptr())))
        self.stream += 1
        return out


def _empty_for(rng, *shape):
    """Result / staging buffer on the side the normals live on."""
    if isinstance(rng, DeviceNormals):
        import torch
        return torch.empty(*shape, dtype=torch.float64, device="cuda")
    return np.empty(shape)


def rand(rng, fx: FiniteGP, N: Optional[int] = None, jitters=None, add_noise: bool = True):
    """rand(rng, fx[, N]): reference src/oilmm.jl:40-54, src/ilmm.jl:78-92, src/independent_mogp.jl:83-96.
    `rng` is a numpy Generator; standard normals are drawn on the host in the reference's order (m blocks of n
    latent normals, then n*p noise normals) and handed to the device, as the Julia shim does with randn(rng, ...)."""
    f, x, s2 = fx.f, fx.x, fx.sigma2
    if isinstance(x, MOInputIsotopicByFeatures):          # reference src/independent_mogp.jl:217-220
        s = rand(rng, FiniteGP(f, x.by_outputs(), s2), N, jitters, add_noise)
        if N is None:
            return _reorder(s, x.n, x.out_dim, False)
        return np.stack([_reorder(np.ascontiguousarray(s[:, q]), x.n, x.out_dim, False) for q in range(N)], axis=1)
    L.ensure_init()
    lib = L.load()
    xa = x.carr()
    n = x.n
    if isinstance(f, IndependentMOGP) and f._post is not None and f._post.dense:
        # rand(rng, f_latent(x, s2)) on the coupled latent PosteriorGP of a dense-H posterior: AbstractGPs' generic
        # mean + chol(cov + s2 I).U' z, one draw of m n normals per sample (N samples = N repeats, as the reference does)
        m = len(f.fs)
        view = f._post.latent_view()

        def one():
            z = rng.standard_normal(m * n)
            o = _empty_for(rng, n * m)
            L.check(lib.lmm_ilmm_post_rand(view.ptr, C.c_double(s2), 0, xa.ptr, x.dim, n, L.Arr(z).ptr, None,
                                           L.jitters((0.0, s2, 0.0)), L.Arr(o, True).ptr))
            return o
        if N is None:
            return one()
        cols = [one() for _ in range(N)]
        import torch
        return torch.stack(cols, dim=1) if L._is_torch(cols[0]) else np.stack(cols, axis=1)
    if N is not None:
        # reference src/ilmm.jl:90-92 / src/independent_mogp.jl:92-96 repeat the whole call N times; here ONE factorisation
        # serves all N samples (lmm_lmm_rand_multi).  Normals are still drawn sample by sample in the reference's order.
        if isinstance(f, IndependentMOGP):
            m = p = len(f.fs)
            Ua, Sa, descs, post, shard, jit, noise = L.Arr(L.colmajor(np.eye(m))), None, [g.desc() for g in f.fs], f._post, (0, m), \
                L.jitters((1e-9, s2, s2)), 0
        else:
            unpack(fx)
            Ua, Sa, p, m = _H_args(f.H)
            descs, post, shard, jit, noise = [g.desc() for g in f.f.fs], f.f._post, f.shard, L.jitters(jitters), int(add_noise)
        z = _empty_for(rng, N, m * n); eps = _empty_for(rng, N, n * p)
        for q in range(N):
            z[q] = rng.standard_normal(m * n)
            if not isinstance(f, IndependentMOGP):
                eps[q] = rng.standard_normal(n * p)
        out = _empty_for(rng, N, n * p)
        L.check(lib.lmm_lmm_rand_multi(post.ptr if post is not None else None, L.gps_array(descs), Ua.ptr,
                                       Sa.ptr if Sa is not None else None, p, m, shard[0], shard[1], C.c_double(s2), noise, xa.ptr,
                                       x.dim, n, N, L.Arr(z).ptr, L.Arr(eps).ptr if noise else None, jit, L.Arr(out, True).ptr))
        return out.T
    if isinstance(f, IndependentMOGP):
        # vcat(rand(rng, f_l(x, s2))): latent jitter = s2, H = I, no extra noise term
        m = len(f.fs)
        z = rng.standard_normal(m * n)
        out = _empty_for(rng, n * m)
        gps = L.gps_array([g.desc() for g in f.fs])
        post = f._post.ptr if f._post is not None else None
        Ua = L.Arr(L.colmajor(np.eye(m)))
        jit = L.jitters((1e-9, s2, s2))
        L.check(lib.lmm_lmm_rand(post, gps, Ua.ptr, None, m, m, 0, m, C.c_double(s2), 0, xa.ptr, x.dim, n, L.Arr(z).ptr,
                                 None, jit, L.Arr(out, True).ptr))
        return out
    unpack(fx)
    Ua, Sa, p, m = _H_args(f.H)
    z = rng.standard_normal(m * n)
    eps = rng.standard_normal(n * p)
    out = _empty_for(rng, n * p)
    if not f.is_oilmm and f.f._post is not None:      # dense-H posterior: coupled latents, reference src/ilmm.jl:78-87
        L.check(lib.lmm_ilmm_post_rand(f.f._post.ptr, C.c_double(s2), int(add_noise), xa.ptr, x.dim, n, L.Arr(z).ptr,
                                       L.Arr(eps).ptr, L.jitters(jitters), L.Arr(out, True).ptr))
        return out
    gps = L.gps_array([g.desc() for g in f.f.fs])
    post = f.f._post.ptr if f.f._post is not None else None
    l0, l1 = f.shard
    L.check(lib.lmm_lmm_rand(post, gps, Ua.ptr, Sa.ptr if Sa is not None else None, p, m, l0, l1, C.c_double(s2),
                             int(add_noise), xa.ptr, x.dim, n, L.Arr(z).ptr, L.Arr(eps).ptr, L.jitters(jitters),
                             L.Arr(out, True).ptr))
    return out
