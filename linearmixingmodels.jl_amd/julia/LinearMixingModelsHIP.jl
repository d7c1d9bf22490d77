# LinearMixingModelsHIP.jl -- the Julia-side binding a LinearMixingModels.jl maintainer would add to route the
# ILMM/OILMM inference hot path through liblmm_hip.so (include/lmm_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: Julia is absent from the build image and from the GPU box (SURVEY.md
# section 8c).  The file is argument marshalling only -- every method body is one `ccall`, so there is no
# arithmetic here to get wrong; parity is proven through the same C ABI from Python (tests/test_gpu_parity.py).
#
# It keeps the reference's own types (`ILMM`, `IndependentMOGP`, `Orthogonal`; reference
# src/LinearMixingModels.jl:21-24) and overrides the method bodies cited next to each definition.
module LinearMixingModelsHIP

using AbstractGPs, KernelFunctions, LinearAlgebra, Random, FillArrays
using LinearMixingModels
using LinearMixingModels: ILMM, OILMM, IndependentMOGP, Orthogonal, unpack, noise_var

const liblmm = get(ENV, "LMM_HIP_LIB", "liblmm_hip.so")

# ---- C structs (include/lmm_hip.h) -------------------------------------------------------------------------
struct LmmGp            # lmm_gp_t
    kind::Cint
    variance::Cdouble
    lengthscale::Cdouble
    mean::Cdouble
end

struct LmmJitters       # lmm_jitters_t
    project_jitter::Cdouble
    ilmm_rand_jitter::Cdouble
    default_jitter::Cdouble
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:lmm_last_error_string, liblmm), Cstring, ()))
    if rc == 1                                   # LMM_ERR_DIM            -> reference src/ilmm.jl:52
        throw(ErrorException("out dim of x != out dim of f."))
    elseif rc == 2                               # LMM_ERR_NOT_ORTHOGONAL -> reference src/orthogonal_matrix.jl:22
        throw(ArgumentError("`U` is not an orthogonal matrix"))
    elseif rc == 3                               # LMM_ERR_NOT_PD         -> LinearAlgebra.PosDefException(info)
        lat = Ref{Cint}(0); info = Ref{Cint}(0)
        ccall((:lmm_last_error_detail, liblmm), Cint, (Ref{Cint}, Ref{Cint}), lat, info)
        throw(PosDefException(info[]))
    else
        error(msg)
    end
end

__init__() = check(ccall((:lmm_init, liblmm), Cint, (Cint,), parse(Cint, get(ENV, "LOCAL_RANK", "0"))))

# ---- latent descriptors: kernel -> (kind, variance, lengthscale) ---------------------------------------------
_kind(::SEKernel) = Cint(0)
_kind(::Matern32Kernel) = Cint(1)
_kind(::Matern52Kernel) = Cint(2)
_desc(k::Kernel) = (_kind(k), 1.0, 1.0)
_desc(k::ScaledKernel) = ((kd, v, l) = _desc(k.kernel); (kd, v * only(k.σ²), l))
_desc(k::TransformedKernel{<:Kernel,<:ScaleTransform}) = ((kd, v, l) = _desc(k.kernel); (kd, v, l / only(k.transform.s)))
_mean(::AbstractGPs.ZeroMean) = 0.0
_mean(m::AbstractGPs.ConstMean) = Float64(m.c)
function _gps(fs::Vector{<:AbstractGP})
    return [begin (kd, v, l) = _desc(f.kernel); LmmGp(kd, v, l, _mean(f.mean)) end for f in fs]
end

# x as a d x n column-major matrix: Vector{Float64} => 1 x n; ColVecs => its X; RowVecs => transposed copy.
_xmat(x::AbstractVector{<:Real}) = reshape(collect(Float64, x), 1, :)
_xmat(x::ColVecs) = Matrix{Float64}(x.X)
_xmat(x::RowVecs) = Matrix{Float64}(x.X')

# ---- logpdf(fx::FiniteGP{<:OILMM}, y): replaces reference src/oilmm.jl:79-93 ---------------------------------
function AbstractGPs.logpdf(fx::FiniteGP{<:OILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)                       # keeps the reference's out-dim check (src/ilmm.jl:45-54)
    X = _xmat(x); d, n = size(X); p, m = size(H.U)
    gps = _gps(fs.fs); S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv U S gps check(ccall((:lmm_oilmm_logpdf, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp},
         Cint, Cint, Cint, Ref{Cdouble}),
        X, d, n, yv, p, U, S, m, σ², gps, 0, m, 1, out))
    return out[]
end

# ---- logpdf(fx::FiniteGP{<:ILMM}, y), dense H: replaces reference src/ilmm.jl:150-163 -----------------------------
function AbstractGPs.logpdf(fx::FiniteGP{<:ILMM{<:IndependentMOGP,<:Matrix}}, y::AbstractVector{<:Real})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H)
    gps = _gps(f.fs); Hm = Matrix{Float64}(H); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv Hm gps check(ccall((:lmm_ilmm_logpdf, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters},
         Ref{Cdouble}), X, d, n, yv, p, Hm, m, σ², gps, C_NULL, out))
    return out[]
end

# ---- logpdf(ft::IsotropicByOutputsFiniteIndependentMOGP, y): replaces reference src/independent_mogp.jl:74-80 -----
function AbstractGPs.logpdf(ft::LinearMixingModels.IsotropicByOutputsFiniteIndependentMOGP, y::AbstractVector{<:Real})
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs)
    gps = _gps(ft.f.fs); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv gps check(ccall((:lmm_mogp_logpdf, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Ref{Cdouble}),
        X, d, n, yv, m, Float64(ft.Σy[1]), gps, 0, m, out))
    return out[]
end

# ---- general Diagonal noise on a by-outputs IndependentMOGP (what reference src/independent_mogp.jl:222-229 reaches after
# reorder_by_outputs, :149-159): per-point noise variances ride the Gram diagonal ------------------------------------
function AbstractGPs.logpdf(
    ft::FiniteGP{<:IndependentMOGP,<:MOInputIsotopicByOutputs,<:Diagonal{<:Real,<:Vector}}, y::AbstractVector{<:Real}
)
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs)
    gps = _gps(ft.f.fs); yv = Vector{Float64}(y); nv = Vector{Float64}(ft.Σy.diag)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv nv gps check(ccall((:lmm_mogp_logpdf_diag, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{LmmGp}, Cint, Cint, Ref{Cdouble}),
        X, d, n, yv, m, nv, gps, 0, m, out))
    return out[]
end

# ---- posterior: device-resident state behind an opaque handle ---------------------------------------------------
# The reference returns ILMM(independent_mogp(posteriors), H) (src/oilmm.jl:133).  The shim returns the same ILMM
# whose latent container is a `HIPPosteriorMOGP` (an AbstractGP holding the handle), so `post(x*, σ²)` builds a
# FiniteGP on which the methods below dispatch.
mutable struct HIPPosteriorMOGP{Tfs<:Vector{<:AbstractGP}} <: AbstractGP
    fs::Tfs                  # the prior latents (kept for get_latent_gp / printing)
    handle::Ptr{Cvoid}       # lmm_post_t*
    function HIPPosteriorMOGP(fs::Tfs, h::Ptr{Cvoid}) where {Tfs}
        obj = new{Tfs}(fs, h)
        finalizer(o -> ccall((:lmm_post_destroy, liblmm), Cint, (Ptr{Cvoid},), o.handle), obj)
        return obj
    end
end

# replaces reference src/oilmm.jl:116-134
function AbstractGPs.posterior(fx::FiniteGP{<:OILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H.U)
    gps = _gps(fs.fs); S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U); yv = Vector{Float64}(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve X yv U S gps check(ccall((:lmm_oilmm_posterior_create, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp},
         Cint, Cint, Ref{Ptr{Cvoid}}), X, d, n, yv, p, U, S, m, σ², gps, 0, m, h))
    return ILMM(HIPPosteriorMOGP(fs.fs, h[]), H)
end

const HIPPosteriorOILMM = ILMM{<:HIPPosteriorMOGP,<:Orthogonal}

# mean_and_var / marginals of the posterior OILMM at x*: replaces reference src/oilmm.jl:57-76
function AbstractGPs.mean_and_var(fx::FiniteGP{<:HIPPosteriorOILMM})
    H = fx.f.H; σ² = noise_var(fx.Σy); post = fx.f.f
    X = _xmat(fx.x.x); d, ns = size(X); p, m = size(H.U)
    fx.x.out_dim == p || throw(error("out dim of x != out dim of f."))
    S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U)
    M = Vector{Float64}(undef, ns * p); V = similar(M)
    GC.@preserve X U S M V check(ccall((:lmm_oilmm_mean_and_var, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble},
         Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        post.handle, C_NULL, U, S, p, m, 0, m, σ², 1, X, d, ns, C_NULL, M, V))
    return M, V
end
# mean alone: var_out = C_NULL selects mu + K(x*, x) alpha per latent (no triangular solve for variances that would be discarded)
function AbstractGPs.mean(fx::FiniteGP{<:HIPPosteriorOILMM})
    H = fx.f.H; post = fx.f.f
    X = _xmat(fx.x.x); d, ns = size(X); p, m = size(H.U)
    S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U)
    M = Vector{Float64}(undef, ns * p)
    GC.@preserve X U S M check(ccall((:lmm_oilmm_mean_and_var, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble},
         Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        post.handle, C_NULL, U, S, p, m, 0, m, noise_var(fx.Σy), 0, X, d, ns, C_NULL, M, C_NULL))
    return M
end
AbstractGPs.var(fx::FiniteGP{<:HIPPosteriorOILMM}) = mean_and_var(fx)[2]

# logpdf(po(x*, σ²), y*) (reference test/oilmm.jl:25)
function AbstractGPs.logpdf(fx::FiniteGP{<:HIPPosteriorOILMM}, y::AbstractVector{<:Real})
    H = fx.f.H; σ² = noise_var(fx.Σy); post = fx.f.f
    X = _xmat(fx.x.x); d, ns = size(X); p, m = size(H.U)
    S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X U S yv check(ccall((:lmm_oilmm_post_logpdf, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint,
         Ref{Cdouble}), post.handle, U, S, p, m, σ², X, d, ns, yv, 1, out))
    return out[]
end

# ---- rand: the normals are drawn HERE, in the reference's order (src/oilmm.jl:47,53: m blocks of n latent draws,
# then n*p noise draws), so the same `rng` gives the same sample as the reference -----------------------------------
function _rand(rng::AbstractRNG, handle, gps, U, S, p, m, σ², X)
    d, ns = size(X)
    z = randn(rng, ns * m); ε = randn(rng, ns * p)
    out = Vector{Float64}(undef, ns * p)
    GC.@preserve X U S gps z ε out check(ccall((:lmm_lmm_rand, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint,
         Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmJitters}, Ptr{Cdouble}),
        handle, gps, U, S, p, m, 0, m, σ², 1, X, d, ns, z, ε, C_NULL, out))
    return out
end

# replaces reference src/oilmm.jl:40-54
function AbstractGPs.rand(rng::AbstractRNG, fx::FiniteGP{<:OILMM})
    fs, H, σ², x = unpack(fx)
    p, m = size(H.U)
    return _rand(rng, C_NULL, _gps(fs.fs), Matrix{Float64}(H.U), Vector{Float64}(H.S.diag), p, m, σ², _xmat(x))
end
function AbstractGPs.rand(rng::AbstractRNG, fx::FiniteGP{<:HIPPosteriorOILMM})
    H = fx.f.H; p, m = size(H.U)
    return _rand(rng, fx.f.f.handle, Ptr{LmmGp}(C_NULL), Matrix{Float64}(H.U), Vector{Float64}(H.S.diag), p, m,
                 noise_var(fx.Σy), _xmat(fx.x.x))
end
# replaces reference src/ilmm.jl:78-87 (dense H: S == NULL selects the 1e-12 latent jitter of src/ilmm.jl:84)
function AbstractGPs.rand(rng::AbstractRNG, fx::FiniteGP{<:ILMM{<:IndependentMOGP,<:Matrix}})
    f, H, σ², x = unpack(fx)
    p, m = size(H)
    return _rand(rng, C_NULL, _gps(f.fs), Matrix{Float64}(H), Ptr{Cdouble}(C_NULL), p, m, σ², _xmat(x))
end

# ---- dense-H ILMM posterior (reference src/ilmm.jl:184-198 and the methods of :108-163 on its PosteriorGP latent): ONE
# coupled (mn) x (mn) factorisation behind the handle ---------------------------------------------------------------
const HIPPosteriorILMM = ILMM{<:HIPPosteriorMOGP,<:Matrix}

function AbstractGPs.posterior(fx::FiniteGP{<:ILMM{<:IndependentMOGP,<:Matrix}}, y::AbstractVector{<:Real})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H)
    gps = _gps(f.fs); Hm = Matrix{Float64}(H); yv = Vector{Float64}(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve X yv Hm gps check(ccall((:lmm_ilmm_posterior_create, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters},
         Ref{Ptr{Cvoid}}), X, d, n, yv, p, Hm, m, σ², gps, C_NULL, h))
    return ILMM(HIPPosteriorMOGP(f.fs, h[]), H)
end

# posterior(pi(x₂, σ²), y₂): AbstractGPs.TestUtils on `pi` (reference test/ilmm.jl:34-37)
function AbstractGPs.posterior(fx::FiniteGP{<:HIPPosteriorILMM}, y::AbstractVector{<:Real})
    X = _xmat(fx.x.x); d, n2 = size(X); yv = Vector{Float64}(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve X yv check(ccall((:lmm_ilmm_post_condition, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{LmmJitters}, Ref{Ptr{Cvoid}}),
        fx.f.f.handle, noise_var(fx.Σy), X, d, n2, yv, C_NULL, h))
    return ILMM(HIPPosteriorMOGP(fx.f.f.fs, h[]), fx.f.H)
end

function AbstractGPs.mean_and_var(fx::FiniteGP{<:HIPPosteriorILMM})
    X = _xmat(fx.x.x); d, ns = size(X); p = size(fx.f.H, 1)
    M = Vector{Float64}(undef, ns * p); V = similar(M)
    GC.@preserve X M V check(ccall((:lmm_ilmm_post_mean_and_var, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        fx.f.f.handle, noise_var(fx.Σy), X, d, ns, C_NULL, M, V))
    return M, V
end

# replaces reference src/ilmm.jl:132-147 on the posterior
function AbstractGPs.mean_and_cov(fx::FiniteGP{<:HIPPosteriorILMM})
    X = _xmat(fx.x.x); d, ns = size(X); p = size(fx.f.H, 1)
    M = Vector{Float64}(undef, ns * p); Cm = Matrix{Float64}(undef, ns * p, ns * p)
    GC.@preserve X M Cm check(ccall((:lmm_ilmm_post_mean_and_cov, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        fx.f.f.handle, noise_var(fx.Σy), X, d, ns, C_NULL, M, Cm))
    return M, Cm
end
AbstractGPs.cov(fx::FiniteGP{<:HIPPosteriorILMM}) = mean_and_cov(fx)[2]

function AbstractGPs.logpdf(fx::FiniteGP{<:HIPPosteriorILMM}, y::AbstractVector{<:Real})
    X = _xmat(fx.x.x); d, ns = size(X); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv check(ccall((:lmm_ilmm_post_logpdf, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{LmmJitters}, Ref{Cdouble}),
        fx.f.f.handle, noise_var(fx.Σy), X, d, ns, yv, C_NULL, out))
    return out[]
end

function AbstractGPs.rand(rng::AbstractRNG, fx::FiniteGP{<:HIPPosteriorILMM})
    X = _xmat(fx.x.x); d, ns = size(X); p, m = size(fx.f.H)
    z = randn(rng, ns * m); ε = randn(rng, ns * p); out = Vector{Float64}(undef, ns * p)
    GC.@preserve X z ε out check(ccall((:lmm_ilmm_post_rand, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmJitters}, Ptr{Cdouble}),
        fx.f.f.handle, noise_var(fx.Σy), 1, X, d, ns, z, ε, C_NULL, out))
    return out
end

# ---- gradients: ChainRulesCore.rrule around the ccall (reference tests: `gradient(logpdf, oilmmx, y) isa Tuple`,
# test/oilmm.jl:31-32).  lmm_oilmm_logpdf_grad returns d/dy, d/dsigma2, d/dS, d/dU and per-latent (variance, lengthscale,
# mean) cotangents in one pass; they are mapped back onto the reference's structs as Tangents. -------------------------
using ChainRulesCore

struct LmmGpGrad      # lmm_gp_grad_t
    variance::Cdouble
    lengthscale::Cdouble
    mean::Cdouble
end

function ChainRulesCore.rrule(::typeof(AbstractGPs.logpdf), fx::FiniteGP{<:OILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H.U)
    gps = _gps(fs.fs); S = Vector{Float64}(H.S.diag); U = Matrix{Float64}(H.U); yv = Vector{Float64}(y)
    val = Ref{Cdouble}(0.0); gσ = Ref{Cdouble}(0.0)
    gy = Vector{Float64}(undef, n * p); gS = Vector{Float64}(undef, m); gU = Matrix{Float64}(undef, p, m)
    gg = Vector{LmmGpGrad}(undef, m)
    GC.@preserve X yv U S gps gy gS gU gg check(ccall((:lmm_oilmm_logpdf_grad, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint,
         Ref{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
        X, d, n, yv, p, U, S, m, σ², gps, 0, m, 1, val, gy, gσ, gS, gU, gg))
    function logpdf_pullback(Δ)
        dH = Tangent{typeof(H)}(; U=Δ .* gU, S=Tangent{typeof(H.S)}(; diag=Δ .* gS))
        # kernel-parameter cotangents (gg[l].variance / .lengthscale / .mean) attach to fs.fs[l].kernel / .mean according
        # to how the kernel was built (ScaledKernel.σ², ScaleTransform.s = 1/ℓ ⇒ ∂/∂s = -ℓ² ∂/∂ℓ); left to the maintainer's
        # preferred parameterisation.  The noise cotangent is Δ*gσ on each entry's share of Fill(σ², n*p).
        dfx = Tangent{typeof(fx)}(; f=Tangent{typeof(fx.f)}(; H=dH), Σy=Tangent{typeof(fx.Σy)}(; diag=Tangent{typeof(fx.Σy.diag)}(; value=Δ * gσ[])))
        return NoTangent(), dfx, Δ .* gy
    end
    return val[], logpdf_pullback
end

end # module
