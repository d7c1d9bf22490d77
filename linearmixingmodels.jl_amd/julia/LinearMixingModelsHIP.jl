# LinearMixingModelsHIP.jl -- the Julia-side binding a LinearMixingModels.jl maintainer would add to route the
# ILMM/OILMM inference hot path through liblmm_hip.so (include/lmm_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY: Julia is absent from the build image and from the GPU box (SURVEY.md section 8c).
# The file is argument marshalling only -- every method body is one `ccall`, so there is no arithmetic here to get
# wrong; parity is proven through the same C ABI from Python (tests/test_gpu_parity*.py).
#
# Design: NO method of LinearMixingModels is overwritten (overwriting another module's methods breaks precompilation on
# Julia >= 1.10).  The reference's types `ILMM` and `Orthogonal` are kept; the opt-in is the latent container: `hip(f)`
# swaps the `IndependentMOGP` inside an ILMM for a `HIPMOGP` (same `fs` field, plus a device handle once conditioned), and
# every method below dispatches on `FiniteGP{<:ILMM{<:HIPMOGP, ...}}` -- strictly more specific than the reference's
# `FiniteGP{<:ILMM}` / `FiniteGP{<:OILMM}` signatures, so Julia picks it without ambiguity:
#
#     f  = hip(ILMM(independent_mogp(fs), Orthogonal(U, S)))     # or ILMM(hip(independent_mogp(fs)), H)
#     fx = f(MOInputIsotopicByOutputs(x, p), σ²)
#     logpdf(fx, y); po = posterior(fx, y); marginals(po(xs, σ²)); rand(rng, fx); Zygote.gradient(logpdf, fx, y)
module LinearMixingModelsHIP

using AbstractGPs, KernelFunctions, LinearAlgebra, Random, FillArrays, ChainRulesCore, Distributions
using LinearMixingModels
using LinearMixingModels: ILMM, IndependentMOGP, Orthogonal, unpack, noise_var

export hip, HIPMOGP

const liblmm = get(ENV, "LMM_HIP_LIB", "liblmm_hip.so")

# ---- C structs (include/lmm_hip.h) -------------------------------------------------------------------------
struct LmmGp            # lmm_gp_t
    kind::Cint
    variance::Cdouble
    lengthscale::Cdouble
    mean::Cdouble
end
struct LmmGpGrad        # lmm_gp_grad_t
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
    else                                         # LMM_ERR_HIP / _ARG / _UNSUPPORTED / _RCCL
        error(msg)
    end
end

__init__() = check(ccall((:lmm_init, liblmm), Cint, (Cint,), parse(Cint, get(ENV, "LOCAL_RANK", "0"))))

# ---- the opt-in latent container ----------------------------------------------------------------------------------
# Prior: handle == C_NULL.  Posterior: lmm_post_t* (device-resident factors, alpha, x) + the data it was built from
# (train: needed for TOTAL derivatives of the predictive logpdf; one entry per conditioning batch, in conditioning order).
mutable struct HIPMOGP{Tfs<:Vector{<:AbstractGP}} <: AbstractGPs.AbstractGP
    fs::Tfs
    handle::Ptr{Cvoid}
    train::Any               # nothing | Vector of (X::Matrix{Float64}, σ²::Float64, y::Vector{Float64}), one per batch
    # filled by the posterior logpdf rrule: the total-derivative cotangents of the predictive logpdf w.r.t. the training data and
    # training noise, (y_train = ..., sigma2_train = ...), which the pullback itself cannot route anywhere (see the rrule)
    last_train_cotangents::Base.RefValue{Any}
    mix::Any                 # latent view of a dense-H posterior only: the H (p x m) its conditioning batches were observed through
    function HIPMOGP(fs::Tfs, h::Ptr{Cvoid}=C_NULL, train=nothing, mix=nothing) where {Tfs<:Vector{<:AbstractGP}}
        obj = new{Tfs}(fs, h, train, Ref{Any}(nothing), mix)
        h == C_NULL || finalizer(o -> ccall((:lmm_post_destroy, liblmm), Cint, (Ptr{Cvoid},), o.handle), obj)
        return obj
    end
end
hip(f::IndependentMOGP) = HIPMOGP(f.fs)
hip(f::ILMM) = ILMM(hip(f.f), f.H)
LinearMixingModels.get_latent_gp(f::ILMM{<:HIPMOGP}) = f.f

const HIPOILMM = ILMM{<:HIPMOGP,<:Orthogonal}
const HIPDenseILMM = ILMM{<:HIPMOGP,<:Matrix}
const ByOutputsFill{F} = FiniteGP{<:F,<:MOInputIsotopicByOutputs,<:Diagonal{<:Real,<:Fill}}
isposterior(f::HIPMOGP) = f.handle != C_NULL

# ---- latent descriptors: kernel -> (kind, variance, lengthscale) ---------------------------------------------
_kind(::SEKernel) = Cint(0)
_kind(::Matern32Kernel) = Cint(1)
_kind(::Matern52Kernel) = Cint(2)
_desc(k::Kernel) = (_kind(k), 1.0, 1.0)
_desc(k::ScaledKernel) = ((kd, v, l) = _desc(k.kernel); (kd, v * only(k.σ²), l))
_desc(k::TransformedKernel{<:Kernel,<:ScaleTransform}) = ((kd, v, l) = _desc(k.kernel); (kd, v, l / only(k.transform.s)))
_mean(::AbstractGPs.ZeroMean) = 0.0
_mean(m::AbstractGPs.ConstMean) = Float64(m.c)
_gps(fs::Vector{<:AbstractGP}) = [begin (kd, v, l) = _desc(f.kernel); LmmGp(kd, v, l, _mean(f.mean)) end for f in fs]

# x as a d x n column-major matrix: Vector{Float64} => 1 x n; ColVecs => its X; RowVecs => transposed copy.
_xmat(x::AbstractVector{<:Real}) = reshape(collect(Float64, x), 1, :)
_xmat(x::ColVecs) = Matrix{Float64}(x.X)
_xmat(x::RowVecs) = Matrix{Float64}(x.X')

# (U, S-or-NULL, p, m) of the mixing matrix: Orthogonal passes U and diag(S) -- never collect(H), whose getindex
# materialises U sqrt(S) per element (reference src/orthogonal_matrix.jl:27-30)
_hargs(H::Orthogonal) = (Matrix{Float64}(H.U), Vector{Float64}(H.S.diag), size(H.U)...)
_hargs(H::AbstractMatrix) = (Matrix{Float64}(H), nothing, size(H)...)
_ptr(::Nothing) = Ptr{Cdouble}(C_NULL)
_ptr(a::Array{Float64}) = pointer(a)

# ---- logpdf -------------------------------------------------------------------------------------------------------
# reference src/oilmm.jl:79-93 (prior) and test/oilmm.jl:25 (posterior: the posterior is again an OILMM, src/oilmm.jl:133)
function AbstractGPs.logpdf(fx::ByOutputsFill{HIPOILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)                       # keeps the reference's out-dim check (src/ilmm.jl:45-54)
    X = _xmat(x); d, n = size(X); U, S, p, m = _hargs(H); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    if isposterior(fs)
        GC.@preserve X U S yv check(ccall((:lmm_oilmm_post_logpdf, liblmm), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ref{Cdouble}),
            fs.handle, U, S, p, m, σ², X, d, n, yv, 1, out))
    else
        gps = _gps(fs.fs)
        GC.@preserve X yv U S gps check(ccall((:lmm_oilmm_logpdf, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ref{Cdouble}),
            X, d, n, yv, p, U, S, m, σ², gps, 0, m, 1, out))
    end
    return out[]
end

# logpdf(fx, Y::AbstractMatrix): one value per column from ONE factorisation per latent (AbstractGPs.TestUtils calls it)
function AbstractGPs.logpdf(fx::ByOutputsFill{HIPOILMM}, Y::AbstractMatrix{<:Real})
    fs, H, σ², x = unpack(fx)
    isposterior(fs) && return [logpdf(fx, Y[:, c]) for c in axes(Y, 2)]
    X = _xmat(x); d, n = size(X); U, S, p, m = _hargs(H); Ym = Matrix{Float64}(Y); gps = _gps(fs.fs)
    out = Vector{Float64}(undef, size(Ym, 2))
    GC.@preserve X Ym U S gps out check(ccall((:lmm_oilmm_logpdf_multi, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ptr{Cdouble}),
        X, d, n, Ym, p, size(Ym, 2), U, S, m, σ², gps, 0, m, 1, out))
    return out
end

# the same for the dense-H ILMM (lmm_ilmm_logpdf_multi: the columns ride the one (mn) x (mn) factorisation) and the IndependentMOGP
# (the OILMM with U = I, S = 1, no regulariser); on posterior models the columns are evaluated one by one on the handle
function AbstractGPs.logpdf(fx::ByOutputsFill{HIPDenseILMM}, Y::AbstractMatrix{<:Real})
    f, H, σ², x = unpack(fx)
    isposterior(f) && return [logpdf(fx, Y[:, c]) for c in axes(Y, 2)]
    X = _xmat(x); d, n = size(X); p, m = size(H); Ym = Matrix{Float64}(Y); gps = _gps(f.fs); Hm = Matrix{Float64}(H)
    out = Vector{Float64}(undef, size(Ym, 2))
    GC.@preserve X Ym Hm gps out check(ccall((:lmm_ilmm_logpdf_multi, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ptr{Cdouble}),
        X, d, n, Ym, p, size(Ym, 2), Hm, m, σ², gps, C_NULL, out))
    return out
end
function AbstractGPs.logpdf(ft::ByOutputsFill{HIPMOGP}, Y::AbstractMatrix{<:Real})
    isposterior(ft.f) && return [logpdf(ft, Y[:, c]) for c in axes(Y, 2)]
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs); σ² = noise_var(ft.Σy)
    ft.x.out_dim == m || throw(ErrorException("out dim of x != out dim of f."))
    U = Matrix{Float64}(I, m, m); S = ones(m); Ym = Matrix{Float64}(Y); gps = _gps(ft.f.fs)
    out = Vector{Float64}(undef, size(Ym, 2))
    GC.@preserve X Ym U S gps out check(ccall((:lmm_oilmm_logpdf_multi, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ptr{Cdouble}),
        X, d, n, Ym, m, size(Ym, 2), U, S, m, σ², gps, 0, m, 0, out))
    return out
end

# reference src/ilmm.jl:150-163 (prior) and test/ilmm.jl:25 (posterior), dense H
function AbstractGPs.logpdf(fx::ByOutputsFill{HIPDenseILMM}, y::AbstractVector{<:Real})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H); yv = Vector{Float64}(y)
    out = Ref{Cdouble}(0.0)
    if isposterior(f)
        GC.@preserve X yv check(ccall((:lmm_ilmm_post_logpdf, liblmm), Cint,
            (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{LmmJitters}, Ref{Cdouble}),
            f.handle, σ², X, d, n, yv, C_NULL, out))
    else
        gps = _gps(f.fs); Hm = Matrix{Float64}(H)
        GC.@preserve X yv Hm gps check(ccall((:lmm_ilmm_logpdf, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ref{Cdouble}),
            X, d, n, yv, p, Hm, m, σ², gps, C_NULL, out))
    end
    return out[]
end

# reference src/independent_mogp.jl:74-80 (by-outputs, scalar noise); posterior MOGP == posterior OILMM with U = I, S = 1
function AbstractGPs.logpdf(ft::ByOutputsFill{HIPMOGP}, y::AbstractVector{<:Real})
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs); yv = Vector{Float64}(y); σ² = noise_var(ft.Σy)
    ft.x.out_dim == m || throw(ErrorException("out dim of x != out dim of f."))
    out = Ref{Cdouble}(0.0)
    if isposterior(ft.f)
        U = Matrix{Float64}(I, m, m); S = ones(m)
        GC.@preserve X U S yv check(ccall((:lmm_oilmm_post_logpdf, liblmm), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ref{Cdouble}),
            ft.f.handle, U, S, m, m, σ², X, d, n, yv, 0, out))
    else
        gps = _gps(ft.f.fs)
        GC.@preserve X yv gps check(ccall((:lmm_mogp_logpdf, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Ref{Cdouble}),
            X, d, n, yv, m, σ², gps, 0, m, out))
    end
    return out[]
end

# general Diagonal noise on a by-outputs IndependentMOGP (what reference src/independent_mogp.jl:222-229 reaches after
# reorder_by_outputs, :149-159): per-point noise variances ride the Gram diagonal
function AbstractGPs.logpdf(ft::FiniteGP{<:HIPMOGP,<:MOInputIsotopicByOutputs,<:Diagonal{<:Real,<:Vector}}, y::AbstractVector{<:Real})
    isposterior(ft.f) && error("logpdf with a general Diagonal noise is served for the prior IndependentMOGP only")
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs)
    ft.x.out_dim == m || throw(ErrorException("out dim of x != out dim of f."))
    gps = _gps(ft.f.fs); yv = Vector{Float64}(y); nv = Vector{Float64}(ft.Σy.diag)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv nv gps check(ccall((:lmm_mogp_logpdf_diag, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{LmmGp}, Cint, Cint, Ref{Cdouble}),
        X, d, n, yv, m, nv, gps, 0, m, out))
    return out[]
end

# ---- posterior ----------------------------------------------------------------------------------------------------
# the conditioning batches a posterior was built from (nothing: unknown, e.g. the latent view of a dense-H posterior)
_push_train(train, X, σ², yv) = train === nothing ? nothing : vcat(train, [(X, Float64(σ²), yv)])
# ... as ONE set of points for the *_post_logpdf_grad_seq entries: X (d x n), sizes, variances, y by outputs over the n points
function _merged_train(train, p::Integer)
    length(train) <= 7 || error("gradient of the predictive logpdf after more than 7 conditioning batches is not built")
    X0 = reduce(hcat, [t[1] for t in train])
    y0 = vec(reduce(vcat, [reshape(t[3], :, p) for t in train]))            # (n_b x p) blocks stacked per output
    return X0, Cint[size(t[1], 2) for t in train], Cdouble[t[2] for t in train], y0
end
# d/dy of the merged points back to one by-outputs vector per batch; d/dσ² per batch (a scalar for a single batch)
function _split_train(gy0, gb, bn, p::Integer)
    length(bn) == 1 && return (y_train=gy0, sigma2_train=gb[1])
    G = reshape(gy0, :, p); o = cumsum(vcat(0, bn))
    return (y_train=[vec(G[o[b]+1:o[b+1], :]) for b in eachindex(bn)], sigma2_train=copy(gb))
end
# reference src/oilmm.jl:116-134; on a posterior: sequential conditioning (TestUtils on `po`, test/oilmm.jl:34-37)
function AbstractGPs.posterior(fx::ByOutputsFill{HIPOILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); U, S, p, m = _hargs(H); yv = Vector{Float64}(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if isposterior(fs)
        GC.@preserve X yv U S check(ccall((:lmm_post_condition, liblmm), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
            fs.handle, U, S, p, m, σ², X, d, n, yv, h))
        return ILMM(HIPMOGP(fs.fs, h[], _push_train(fs.train, X, σ², yv)), H)
    end
    gps = _gps(fs.fs)
    GC.@preserve X yv U S gps check(ccall((:lmm_oilmm_posterior_create, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Ref{Ptr{Cvoid}}),
        X, d, n, yv, p, U, S, m, σ², gps, 0, m, h))
    return ILMM(HIPMOGP(fs.fs, h[], [(X, Float64(σ²), yv)]), H)    # again an ILMM with the same H (src/oilmm.jl:133)
end

# reference src/independent_mogp.jl:119-126; on a posterior: sequential conditioning (test/independent_mogp.jl:68-76)
function AbstractGPs.posterior(ft::ByOutputsFill{HIPMOGP}, y::AbstractVector{<:Real})
    X = _xmat(ft.x.x); d, n = size(X); m = length(ft.f.fs); yv = Vector{Float64}(y); σ² = noise_var(ft.Σy)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if isposterior(ft.f)
        U = Matrix{Float64}(I, m, m); S = ones(m)
        GC.@preserve X yv U S check(ccall((:lmm_post_condition, liblmm), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
            ft.f.handle, U, S, m, m, σ², X, d, n, yv, h))
        return HIPMOGP(ft.f.fs, h[], _push_train(ft.f.train, X, σ², yv))
    end
    gps = _gps(ft.f.fs)
    GC.@preserve X yv gps check(ccall((:lmm_mogp_posterior_create, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Ref{Ptr{Cvoid}}),
        X, d, n, yv, m, σ², gps, 0, m, h))
    return HIPMOGP(ft.f.fs, h[], [(X, Float64(σ²), yv)])
end

# reference src/ilmm.jl:184-198 (one coupled (mn) x (mn) factorisation); on a posterior: TestUtils on `pi` (test/ilmm.jl:34-37)
function AbstractGPs.posterior(fx::ByOutputsFill{HIPDenseILMM}, y::AbstractVector{<:Real})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H); yv = Vector{Float64}(y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if isposterior(f)
        GC.@preserve X yv check(ccall((:lmm_ilmm_post_condition, liblmm), Cint,
            (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{LmmJitters}, Ref{Ptr{Cvoid}}),
            f.handle, σ², X, d, n, yv, C_NULL, h))
        # (a latent view conditioned ON latent observations keeps no training record: its batches were observed through different H's)
        return ILMM(HIPMOGP(f.fs, h[], f.mix === nothing ? _push_train(f.train, X, σ², yv) : nothing), H)
    end
    gps = _gps(f.fs); Hm = Matrix{Float64}(H)
    GC.@preserve X yv Hm gps check(ccall((:lmm_ilmm_posterior_create, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ref{Ptr{Cvoid}}),
        X, d, n, yv, p, Hm, m, σ², gps, C_NULL, h))
    return ILMM(HIPMOGP(f.fs, h[], [(X, Float64(σ²), yv)]), H)
end

# ---- mean_and_var / marginals / mean / var / mean_and_cov / cov --------------------------------------------------------
# Independent latents (OILMM prior or posterior, dense-H prior): reference src/oilmm.jl:57-76, src/ilmm.jl:108-145.
# want_var = false: means only -- mu + K(x*, x) alpha per latent, no triangular solve (the reference computes and discards
# the variances, src/ilmm.jl:142).
function _mean_var(fx, want_var::Bool)
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, ns = size(X); U, S, p, m = _hargs(H)
    gps = isposterior(f) ? LmmGp[] : _gps(f.fs)
    M = Vector{Float64}(undef, ns * p); V = want_var ? similar(M) : Float64[]
    GC.@preserve X U S gps M V check(ccall((:lmm_oilmm_mean_and_var, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint,
         Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        f.handle, isposterior(f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), U, _ptr(S), p, m, 0, m, σ², 1, X, d, ns, C_NULL,
        M, want_var ? pointer(V) : Ptr{Cdouble}(C_NULL)))
    return M, V
end
AbstractGPs.mean_and_var(fx::ByOutputsFill{HIPOILMM}) = _mean_var(fx, true)
AbstractGPs.mean(fx::ByOutputsFill{HIPOILMM}) = _mean_var(fx, false)[1]
AbstractGPs.var(fx::ByOutputsFill{HIPOILMM}) = _mean_var(fx, true)[2]

function AbstractGPs.mean_and_var(fx::ByOutputsFill{HIPDenseILMM})
    f = fx.f.f
    isposterior(f) || return _mean_var(fx, true)             # prior latents are independent: same mixing as the OILMM form
    unpack(fx)
    X = _xmat(fx.x.x); d, ns = size(X); p = size(fx.f.H, 1)
    M = Vector{Float64}(undef, ns * p); V = similar(M)
    GC.@preserve X M V check(ccall((:lmm_ilmm_post_mean_and_var, liblmm), Cint,
        (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        f.handle, noise_var(fx.Σy), X, d, ns, C_NULL, M, V))
    return M, V
end
AbstractGPs.mean(fx::ByOutputsFill{HIPDenseILMM}) = isposterior(fx.f.f) ? mean_and_var(fx)[1] : _mean_var(fx, false)[1]
AbstractGPs.var(fx::ByOutputsFill{HIPDenseILMM}) = mean_and_var(fx)[2]

# reference src/independent_mogp.jl:50,55: vcat of the latent marginals (+ Σy on the variances)
function AbstractGPs.mean_and_var(ft::ByOutputsFill{HIPMOGP})
    X = _xmat(ft.x.x); d, ns = size(X); m = length(ft.f.fs)
    gps = isposterior(ft.f) ? LmmGp[] : _gps(ft.f.fs)
    M = Vector{Float64}(undef, ns * m); V = similar(M)
    GC.@preserve X gps M V check(ccall((:lmm_latent_marginals, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
        ft.f.handle, isposterior(ft.f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), m, X, d, ns, M, V))
    return M, V .+ noise_var(ft.Σy)
end
AbstractGPs.mean(ft::ByOutputsFill{HIPMOGP}) = mean_and_var(ft)[1]
AbstractGPs.var(ft::ByOutputsFill{HIPMOGP}) = mean_and_var(ft)[2]

# reference src/ilmm.jl:132-147: full (p n*) x (p n*) covariance, small n* only (as in the reference)
function AbstractGPs.mean_and_cov(fx::Union{ByOutputsFill{HIPOILMM},ByOutputsFill{HIPDenseILMM}})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, ns = size(X); U, S, p, m = _hargs(H)
    M = Vector{Float64}(undef, ns * p); Cm = Matrix{Float64}(undef, ns * p, ns * p)
    if isposterior(f) && S === nothing                         # coupled latents of the dense-H posterior
        GC.@preserve X M Cm check(ccall((:lmm_ilmm_post_mean_and_cov, liblmm), Cint,
            (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint, Cint, Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
            f.handle, σ², X, d, ns, C_NULL, M, Cm))
        return M, Cm
    end
    gps = isposterior(f) ? LmmGp[] : _gps(f.fs)
    GC.@preserve X U S gps M Cm check(ccall((:lmm_lmm_mean_and_cov, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint,
         Ptr{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        f.handle, isposterior(f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), U, _ptr(S), p, m, 0, m, σ², 1, X, d, ns, C_NULL, M, Cm))
    return M, Cm
end
AbstractGPs.cov(fx::Union{ByOutputsFill{HIPOILMM},ByOutputsFill{HIPDenseILMM}}) = mean_and_cov(fx)[2]

# ---- rand: the normals are drawn HERE, in the reference's order (src/oilmm.jl:47,53: m blocks of n latent draws, then
# n*p noise draws; N samples = N repeats, src/ilmm.jl:90-92), so the same `rng` gives the same samples as the reference;
# ONE factorisation per latent serves all N samples (lmm_lmm_rand_multi) --------------------------------------------------
function _rand(rng::AbstractRNG, fx, N::Int)
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, ns = size(X); U, S, p, m = _hargs(H)
    z = Matrix{Float64}(undef, ns * m, N); ε = Matrix{Float64}(undef, ns * p, N)
    for q in 1:N
        z[:, q] = randn(rng, ns * m); ε[:, q] = randn(rng, ns * p)
    end
    out = Matrix{Float64}(undef, ns * p, N)
    if isposterior(f) && S === nothing                         # dense-H posterior: coupled latents, reference src/ilmm.jl:78-87
        for q in 1:N
            zq = z[:, q]; εq = ε[:, q]; oq = Vector{Float64}(undef, ns * p)
            GC.@preserve X zq εq oq check(ccall((:lmm_ilmm_post_rand, liblmm), Cint,
                (Ptr{Cvoid}, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmJitters}, Ptr{Cdouble}),
                f.handle, σ², 1, X, d, ns, zq, εq, C_NULL, oq))
            out[:, q] = oq
        end
        return out
    end
    gps = isposterior(f) ? LmmGp[] : _gps(f.fs)
    GC.@preserve X U S gps z ε out check(ccall((:lmm_lmm_rand_multi, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint, Cint,
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmJitters}, Ptr{Cdouble}),
        f.handle, isposterior(f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), U, _ptr(S), p, m, 0, m, σ², 1, X, d, ns, N, z, ε, C_NULL, out))
    return out
end
const HIPLMMFinite = Union{ByOutputsFill{HIPOILMM},ByOutputsFill{HIPDenseILMM}}
AbstractGPs.rand(rng::AbstractRNG, fx::HIPLMMFinite) = vec(_rand(rng, fx, 1))          # src/oilmm.jl:40-54, src/ilmm.jl:78-87
AbstractGPs.rand(rng::AbstractRNG, fx::HIPLMMFinite, N::Int) = _rand(rng, fx, N)       # src/ilmm.jl:90-92

# reference src/independent_mogp.jl:83-96: vcat(rand(rng, f_l(x, σ²))): latent jitter = σ², H = I, no extra noise term
function _rand_mogp(rng::AbstractRNG, ft, N::Int)
    X = _xmat(ft.x.x); d, ns = size(X); m = length(ft.f.fs); σ² = noise_var(ft.Σy)
    U = Matrix{Float64}(I, m, m)
    z = Matrix{Float64}(undef, ns * m, N)
    for q in 1:N
        z[:, q] = randn(rng, ns * m)
    end
    out = Matrix{Float64}(undef, ns * m, N)
    jit = Ref(LmmJitters(1e-9, σ², σ²))
    gps = isposterior(ft.f) ? LmmGp[] : _gps(ft.f.fs)
    GC.@preserve X U gps z out check(ccall((:lmm_lmm_rand_multi, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint, Cint,
         Ptr{Cdouble}, Ptr{Cdouble}, Ref{LmmJitters}, Ptr{Cdouble}),
        ft.f.handle, isposterior(ft.f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), U, C_NULL, m, m, 0, m, σ², 0, X, d, ns, N, z, C_NULL, jit, out))
    return out
end
AbstractGPs.rand(rng::AbstractRNG, ft::ByOutputsFill{HIPMOGP}) = vec(_rand_mogp(rng, ft, 1))
AbstractGPs.rand(rng::AbstractRNG, ft::ByOutputsFill{HIPMOGP}, N::Int) = _rand_mogp(rng, ft, N)

# reference src/ilmm.jl:95-106 and src/independent_mogp.jl:102-113: `rand!(rng, fx, y)` (what AbstractGPs.TestUtils reaches through
# Distributions) fills y with one sample (vector, or one column) or N samples (N columns) -- the draws come from the methods above, so the
# normals are consumed in the reference's order
function Distributions._rand!(rng::AbstractRNG, fx::Union{HIPLMMFinite,ByOutputsFill{HIPMOGP}}, y::AbstractVecOrMat{<:Real})
    N = size(y, 2)
    if N == 1
        y .= AbstractGPs.rand(rng, fx)
    else
        y .= AbstractGPs.rand(rng, fx, N)
    end
end

# ---- MOInputIsotopicByFeatures (reference src/independent_mogp.jl:128-229) --------------------------------------------------------
# The reference serves by-features inputs of an IndependentMOGP by re-ordering to by-outputs, calling the by-outputs method and
# permuting the result back (indices_which_reorder_*, :135-147).  Same here: the by-outputs methods above do the work, the
# permutation of the length-(n p) vectors is the library's (lmm_reorder: to_outputs = 1 is `v[indices_which_reorder_features_to_outputs]`,
# 0 the inverse), and the (n p) x (n p) covariance is permuted on the host with the reference's own index vectors.
const ByFeatures{F} = FiniteGP{<:F,<:MOInputIsotopicByFeatures,<:Diagonal{<:Real}}
const ByFeaturesFill{F} = FiniteGP{<:F,<:MOInputIsotopicByFeatures,<:Diagonal{<:Real,<:Fill}}
function _reorder(v::AbstractVector{<:Real}, n::Integer, p::Integer, to_outputs::Bool)
    vin = Vector{Float64}(v); out = Vector{Float64}(undef, n * p)
    GC.@preserve vin out check(ccall((:lmm_reorder, liblmm), Cint, (Ptr{Cdouble}, Cint, Cint, Cint, Ptr{Cdouble}),
        vin, n, p, to_outputs ? 1 : 0, out))
    return out
end
_nx(x::MOInputIsotopicByFeatures) = length(x.x)
_by_outputs(x::MOInputIsotopicByFeatures) = MOInputIsotopicByOutputs(x.x, x.out_dim)                      # src/independent_mogp.jl:149
_by_outputs(Σy::Diagonal{<:Real,<:Fill}, x::MOInputIsotopicByFeatures) = Σy                                # :155
_by_outputs(Σy::Diagonal{<:Real}, x::MOInputIsotopicByFeatures) = Diagonal(_reorder(Σy.diag, _nx(x), x.out_dim, true))   # :151-153
_by_outputs(ft::ByFeatures{HIPMOGP}) = FiniteGP(ft.f, _by_outputs(ft.x), _by_outputs(ft.Σy, ft.x))        # :157-159
_to_features(v::AbstractVector{<:Real}, x::MOInputIsotopicByFeatures) = _reorder(v, _nx(x), x.out_dim, false)

# src/independent_mogp.jl:222-229 (any Diagonal noise: a Fill stays a Fill and reaches the ByOutputsFill method; a general diagonal is
# permuted with the data into a Diagonal{Float64,Vector{Float64}} and reaches the per-point-noise method above, lmm_mogp_logpdf_diag)
AbstractGPs.logpdf(ft::ByFeatures{HIPMOGP}, y::AbstractVector{<:Real}) =
    logpdf(_by_outputs(ft), _reorder(y, _nx(ft.x), ft.x.out_dim, true))
# src/independent_mogp.jl:217-220: the by-outputs sample, permuted (the normals are drawn latent by latent, as in the reference)
AbstractGPs.rand(rng::AbstractRNG, ft::ByFeaturesFill{HIPMOGP}) = _to_features(rand(rng, _by_outputs(ft)), ft.x)
function AbstractGPs.rand(rng::AbstractRNG, ft::ByFeaturesFill{HIPMOGP}, N::Int)
    return reduce(hcat, [rand(rng, ft) for _ in 1:N])
end
# src/independent_mogp.jl:165-175 (mean, var) on the finite GP: by-outputs marginals, permuted
function AbstractGPs.mean_and_var(ft::ByFeaturesFill{HIPMOGP})
    M, V = mean_and_var(_by_outputs(ft))
    return _to_features(M, ft.x), _to_features(V, ft.x)
end
AbstractGPs.mean(ft::ByFeaturesFill{HIPMOGP}) = mean_and_var(ft)[1]
AbstractGPs.var(ft::ByFeaturesFill{HIPMOGP}) = mean_and_var(ft)[2]
# src/independent_mogp.jl:177-182: C_by_outputs[idx, idx] (block-diagonal latent covariances + Σy, lmm_lmm_mean_and_cov with U = I)
function AbstractGPs.mean_and_cov(ft::ByOutputsFill{HIPMOGP})
    X = _xmat(ft.x.x); d, ns = size(X); m = length(ft.f.fs); σ² = noise_var(ft.Σy)
    U = Matrix{Float64}(I, m, m)
    gps = isposterior(ft.f) ? LmmGp[] : _gps(ft.f.fs)
    M = Vector{Float64}(undef, ns * m); Cm = Matrix{Float64}(undef, ns * m, ns * m)
    jit = Ref(LmmJitters(1e-9, 0.0, 0.0))                   # cov(f, x) + Σy of a bare MOGP: no latent jitter (src/independent_mogp.jl:60-63)
    GC.@preserve X U gps M Cm check(ccall((:lmm_lmm_mean_and_cov, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cdouble}, Cint, Cint,
         Ref{LmmJitters}, Ptr{Cdouble}, Ptr{Cdouble}),
        ft.f.handle, isposterior(ft.f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), U, Ptr{Cdouble}(C_NULL), m, m, 0, m, σ², 1, X, d, ns, jit, M, Cm))
    return M, Cm
end
AbstractGPs.cov(ft::ByOutputsFill{HIPMOGP}) = mean_and_cov(ft)[2]
function AbstractGPs.mean_and_cov(ft::ByFeaturesFill{HIPMOGP})
    M, Cm = mean_and_cov(_by_outputs(ft))
    idx = LinearMixingModels.indices_which_reorder_outputs_to_features(_by_outputs(ft.x))
    return _to_features(M, ft.x), Cm[idx, idx]
end
AbstractGPs.cov(ft::ByFeaturesFill{HIPMOGP}) = mean_and_cov(ft)[2]
# conditioning on by-features data: reorder, then the by-outputs posterior (the reference reaches the same through reorder_by_outputs)
AbstractGPs.posterior(ft::ByFeaturesFill{HIPMOGP}, y::AbstractVector{<:Real}) =
    posterior(_by_outputs(ft), _reorder(y, _nx(ft.x), ft.x.out_dim, true))
function Distributions._rand!(rng::AbstractRNG, ft::ByFeaturesFill{HIPMOGP}, y::AbstractVecOrMat{<:Real})
    N = size(y, 2)
    if N == 1
        y .= AbstractGPs.rand(rng, ft)
    else
        y .= AbstractGPs.rand(rng, ft, N)
    end
end

# ---- cov(f, x, y), mean(f, x), var(f, x) on the GP itself (AbstractGPs' internal interface) --------------------------------------
# reference src/independent_mogp.jl:66-71 (x, y by outputs: Matrix(BlockDiagonal(cov(f_l, x.x, y.x)))) and :184-215 (either input by
# features: the same blocks at rows / columns permuted with indices_which_reorder_outputs_to_features) -- the library writes every
# block at its final place (lmm_mogp_cross_cov); cov(f, x) = cov(f, x, x) (:60-63, :177-182).  Reference tests:
# test/independent_mogp.jl:136-141.
const MOIsotopic = Union{MOInputIsotopicByOutputs,MOInputIsotopicByFeatures}
_byfeat(::MOInputIsotopicByOutputs) = Cint(0)
_byfeat(::MOInputIsotopicByFeatures) = Cint(1)
function AbstractGPs.cov(f::HIPMOGP, x::MOIsotopic, y::MOIsotopic)
    m = length(f.fs)
    (x.out_dim == m && y.out_dim == m) || throw(ErrorException("out dim of x != out dim of f."))
    X = _xmat(x.x); Y = _xmat(y.x); d, n = size(X); n2 = size(Y, 2)
    gps = isposterior(f) ? LmmGp[] : _gps(f.fs)
    Cm = Matrix{Float64}(undef, m * n, m * n2)
    GC.@preserve X Y gps Cm check(ccall((:lmm_mogp_cross_cov, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Cint, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}),
        f.handle, isposterior(f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), m, 0, m, X, d, n, _byfeat(x), Y, n2, _byfeat(y), Cm))
    return Cm
end
AbstractGPs.cov(f::HIPMOGP, x::MOIsotopic) = cov(f, x, x)
# reference src/independent_mogp.jl:50,55 (by outputs: vcat of the latent marginals) and :169-175 (by features: permuted)
function _mean_var(f::HIPMOGP, x::MOIsotopic)
    m = length(f.fs)
    x.out_dim == m || throw(ErrorException("out dim of x != out dim of f."))
    X = _xmat(x.x); d, ns = size(X)
    gps = isposterior(f) ? LmmGp[] : _gps(f.fs)
    M = Vector{Float64}(undef, ns * m); V = similar(M)
    GC.@preserve X gps M V check(ccall((:lmm_latent_marginals, liblmm), Cint,
        (Ptr{Cvoid}, Ptr{LmmGp}, Cint, Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
        f.handle, isposterior(f) ? Ptr{LmmGp}(C_NULL) : pointer(gps), m, X, d, ns, M, V))
    x isa MOInputIsotopicByFeatures && return _to_features(M, x), _to_features(V, x)
    return M, V
end
AbstractGPs.mean(f::HIPMOGP, x::MOIsotopic) = _mean_var(f, x)[1]
AbstractGPs.var(f::HIPMOGP, x::MOIsotopic) = _mean_var(f, x)[2]

# ---- gradients: ChainRulesCore.rrule around the ccall --------------------------------------------------------------------
# The reference's tests take Zygote.gradient(logpdf, fx, y) on prior and posterior models (test/oilmm.jl:31-32,
# test/ilmm.jl:31-32, test/independent_mogp.jl:65-66).  A ccall is opaque to Zygote, so the pullbacks come from the library
# (lmm_oilmm_logpdf_grad, lmm_ilmm_logpdf_grad, lmm_oilmm_post_logpdf_grad_seq, lmm_ilmm_post_logpdf_grad_seq) and are mapped onto the reference's structs.

# kernel cotangent: the library differentiates w.r.t. the EFFECTIVE (variance, lengthscale); the chain rule through the
# kernel's construction: ScaledKernel: v = v_inner σ² -> d/dσ² = gv v_inner; ScaleTransform: ℓ = ℓ_inner / s -> d/ds = -gl ℓ_inner / s².
_ktangent(k::Kernel, gv, gl) = NoTangent()                                    # SEKernel() etc. carry no parameters
function _ktangent(k::ScaledKernel, gv, gl)
    (_, vin, _) = _desc(k.kernel)
    return Tangent{typeof(k)}(; kernel=_ktangent(k.kernel, gv * only(k.σ²), gl), σ²=[gv * vin])
end
function _ktangent(k::TransformedKernel{<:Kernel,<:ScaleTransform}, gv, gl)
    (_, _, lin) = _desc(k.kernel); s = only(k.transform.s)
    return Tangent{typeof(k)}(; kernel=_ktangent(k.kernel, gv, gl / s), transform=Tangent{typeof(k.transform)}(; s=[-gl * lin / s^2]))
end
_mtangent(::AbstractGPs.ZeroMean, g) = NoTangent()
_mtangent(m::AbstractGPs.ConstMean, g) = Tangent{typeof(m)}(; c=g)
_fstangent(fs::Vector{<:AbstractGP}, gg::Vector{LmmGpGrad}, Δ) =
    [Tangent{typeof(f)}(; mean=_mtangent(f.mean, Δ * g.mean), kernel=_ktangent(f.kernel, Δ * g.variance, Δ * g.lengthscale)) for (f, g) in zip(fs, gg)]
_noise_tangent(fx, g) = Tangent{typeof(fx.Σy)}(; diag=Tangent{typeof(fx.Σy.diag)}(; value=g))     # Fill(σ², n p): one parameter
_htangent(H::Orthogonal, gU, gS) = Tangent{typeof(H)}(; U=gU, S=Tangent{typeof(H.S)}(; diag=gS))

function ChainRulesCore.rrule(::typeof(AbstractGPs.logpdf), fx::ByOutputsFill{HIPOILMM}, y::AbstractVector{<:Real})
    fs, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); U, S, p, m = _hargs(H); gps = _gps(fs.fs); yv = Vector{Float64}(y)
    val = Ref{Cdouble}(0.0); gσ = Ref{Cdouble}(0.0)
    gy = Vector{Float64}(undef, n * p); gS = Vector{Float64}(undef, m); gU = Matrix{Float64}(undef, p, m)
    gg = Vector{LmmGpGrad}(undef, m)
    if isposterior(fs)
        fs.train === nothing && error("this posterior does not carry its training data")
        X0, bn, bs, y0 = _merged_train(fs.train, p); n0 = size(X0, 2); gy0 = Vector{Float64}(undef, n0 * p); gb = similar(bs)
        GC.@preserve X0 bn bs y0 X yv U S gps gy0 gy gb gS gU gg check(ccall((:lmm_oilmm_post_logpdf_grad_seq, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
             Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble},
             Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X0, d, n0, bn, bs, length(bn), y0, X, n, yv, p, U, S, m, σ², gps, 0, m, 1, val, gy0, gy, gb, gσ, gS, gU, gg))
        # The library returns TOTAL derivatives through the posterior, including those w.r.t. the training data (gy0) and the
        # training noise (gb, one per conditioning batch).  The posterior model object has no differentiable slot for (x, σ², y) -- they entered through
        # `posterior`, whose own rrule would be the place to receive them -- so THIS pullback propagates the cotangents of the
        # latent GPs, H, the predictive noise and y* only; gy0 / gb are NOT propagated by it.  Callers who differentiate
        # θ -> logpdf(posterior(f_θ(x, σ²), y)(x*, σ²*), y*) end to end use `predictive_logpdf_and_gradient` below, which returns them.
        fs.last_train_cotangents[] = _split_train(gy0, gb, bn, p)      # one entry per conditioning batch when there are several
    else
        GC.@preserve X yv U S gps gy gS gU gg check(ccall((:lmm_oilmm_logpdf_grad, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint,
             Ref{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X, d, n, yv, p, U, S, m, σ², gps, 0, m, 1, val, gy, gσ, gS, gU, gg))
    end
    function logpdf_pullback(Δ)
        dlat = Tangent{typeof(fs)}(; fs=_fstangent(fs.fs, gg, Δ))
        dfx = Tangent{typeof(fx)}(; f=Tangent{typeof(fx.f)}(; f=dlat, H=_htangent(H, Δ .* gU, Δ .* gS)), Σy=_noise_tangent(fx, Δ * gσ[]))
        return NoTangent(), dfx, Δ .* gy
    end
    return val[], logpdf_pullback
end

# Value, pullback-at-1 and the training cotangents of the predictive logpdf in one call: what an end-to-end differentiation of
# θ -> logpdf(posterior(f_θ(x, σ²), y)(x*, σ²*), y*) needs beyond the rrule above (whose pullback cannot return d/dy_train, d/dσ²_train).
function predictive_logpdf_and_gradient(fx::ByOutputsFill{HIPOILMM}, y::AbstractVector{<:Real})
    val, back = ChainRulesCore.rrule(AbstractGPs.logpdf, fx, y)
    _, dfx, dy = back(1.0)
    tr = unpack(fx)[1].last_train_cotangents[]
    return (value=val, fx=dfx, y=dy, y_train=(tr === nothing ? nothing : tr.y_train), sigma2_train=(tr === nothing ? nothing : tr.sigma2_train))
end

# IndependentMOGP (reference test/independent_mogp.jl:65-66): the OILMM with U = I, S = 1 and no regulariser
function ChainRulesCore.rrule(::typeof(AbstractGPs.logpdf), ft::ByOutputsFill{HIPMOGP}, y::AbstractVector{<:Real})
    f = ft.f; X = _xmat(ft.x.x); d, n = size(X); m = length(f.fs); σ² = noise_var(ft.Σy)
    U = Matrix{Float64}(I, m, m); S = ones(m); gps = _gps(f.fs); yv = Vector{Float64}(y)
    val = Ref{Cdouble}(0.0); gσ = Ref{Cdouble}(0.0)
    gy = Vector{Float64}(undef, n * m); gg = Vector{LmmGpGrad}(undef, m)
    if isposterior(f)
        f.train === nothing && error("this posterior does not carry its training data")
        X0, bn, bs, y0 = _merged_train(f.train, m); n0 = size(X0, 2)
        GC.@preserve X0 bn bs y0 X yv U S gps gy gg check(ccall((:lmm_oilmm_post_logpdf_grad_seq, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
             Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble},
             Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X0, d, n0, bn, bs, length(bn), y0, X, n, yv, m, U, S, m, σ², gps, 0, m, 0, val, C_NULL, gy, C_NULL, gσ, C_NULL, C_NULL, gg))
    else
        GC.@preserve X yv U S gps gy gg check(ccall((:lmm_oilmm_logpdf_grad, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint,
             Ref{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X, d, n, yv, m, U, S, m, σ², gps, 0, m, 0, val, gy, gσ, C_NULL, C_NULL, gg))
    end
    function logpdf_pullback(Δ)
        dft = Tangent{typeof(ft)}(; f=Tangent{typeof(f)}(; fs=_fstangent(f.fs, gg, Δ)), Σy=_noise_tangent(ft, Δ * gσ[]))
        return NoTangent(), dft, Δ .* gy
    end
    return val[], logpdf_pullback
end

# dense-H ILMM, prior and posterior (reference test/ilmm.jl:31-32): the reference's (mn) x (mn) operation + its explicit inverse;
# the posterior's predictive logpdf as the joint density of (y, y*) under two-block noise minus the density of y
function ChainRulesCore.rrule(::typeof(AbstractGPs.logpdf), fx::ByOutputsFill{HIPDenseILMM}, y::AbstractVector{<:Real})
    f, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); p, m = size(H); gps = _gps(f.fs); Hm = Matrix{Float64}(H); yv = Vector{Float64}(y)
    val = Ref{Cdouble}(0.0); gσ = Ref{Cdouble}(0.0)
    gy = Vector{Float64}(undef, n * p); gH = Matrix{Float64}(undef, p, m); gg = Vector{LmmGpGrad}(undef, m)
    if isposterior(f) && f.mix !== nothing
        # the latent view of a dense-H posterior (latent_view; reference src/ilmm.jl:39 on :196-197): here H = I_m, p = m, y = latent
        # observations; the conditioning batches were observed through f.mix.  d/d(mix) and the training cotangents have no slot in
        # this model's tangent (its H is the constant I): they are left in last_train_cotangents.
        f.train === nothing && error("gradient of the latent view after conditioning ON latent observations is not built")
        Hp = f.mix::Matrix{Float64}; pp = size(Hp, 1)
        X0, bn, bs, y0 = _merged_train(f.train, pp); n0 = size(X0, 2); gy0 = Vector{Float64}(undef, n0 * pp); gb = similar(bs)
        gHp = Matrix{Float64}(undef, pp, m)
        GC.@preserve X0 bn bs y0 X yv Hp gps gy0 gy gb gHp gg check(ccall((:lmm_ilmm_post_latent_logpdf_grad_seq, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint,
             Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X0, d, n0, bn, bs, length(bn), y0, X, n, yv, pp, Hp, m, σ², gps, C_NULL, val, gy0, gy, gb, gσ, gHp, gg))
        f.last_train_cotangents[] = merge(_split_train(gy0, gb, bn, pp), (H_train=gHp,))
        fill!(gH, 0.0)
    elseif isposterior(f)
        f.train === nothing && error("this posterior does not carry its training data")
        X0, bn, bs, y0 = _merged_train(f.train, p); n0 = size(X0, 2); gy0 = Vector{Float64}(undef, n0 * p); gb = similar(bs)
        GC.@preserve X0 bn bs y0 X yv Hm gps gy0 gy gb gH gg check(ccall((:lmm_ilmm_post_logpdf_grad_seq, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cint}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint,
             Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X0, d, n0, bn, bs, length(bn), y0, X, n, yv, p, Hm, m, σ², gps, C_NULL, val, gy0, gy, gb, gσ, gH, gg))
        f.last_train_cotangents[] = _split_train(gy0, gb, bn, p)
    else
        GC.@preserve X yv Hm gps gy gH gg check(ccall((:lmm_ilmm_logpdf_grad, liblmm), Cint,
            (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Ptr{LmmJitters}, Ref{Cdouble}, Ptr{Cdouble},
             Ref{Cdouble}, Ptr{Cdouble}, Ptr{LmmGpGrad}),
            X, d, n, yv, p, Hm, m, σ², gps, C_NULL, val, gy, gσ, gH, gg))
    end
    function logpdf_pullback(Δ)
        dfx = Tangent{typeof(fx)}(; f=Tangent{typeof(fx.f)}(; f=Tangent{typeof(f)}(; fs=_fstangent(f.fs, gg, Δ)), H=Δ .* gH),
                                  Σy=_noise_tangent(fx, Δ * gσ[]))
        return NoTangent(), dfx, Δ .* gy
    end
    return val[], logpdf_pullback
end

# ---- modes -------------------------------------------------------------------------------------------------------------
# compute dtype of the per-latent matrices: :f64 (parity mode) | :f32 (BASELINE configs[4])
set_compute_dtype(d::Symbol) = check(ccall((:lmm_set_compute_dtype, liblmm), Cint, (Cint,), d === :f32 ? 1 : 0))
# the region kernel's task hand-out: true (default) = in turn to started workgroups (no reliance on dispatch order), false = task = blockIdx.x
set_strict_progress(on::Bool) = check(ccall((:lmm_set_strict_progress, liblmm), Cint, (Cint,), on ? 1 : 0))
# dtype of the H unprojection of predictive marginals (reference src/oilmm.jl:69-72): :native | :bf16 (BASELINE configs[3]:
# v_mfma_f32_16x16x32_bf16, tolerance 2^-7 Σ_l |H||M_lat|) | :bf16x2 (hi + lo split, <= 2^-15 Σ_l |H||M_lat|, include/lmm_hip.h)
set_projection_dtype(d::Symbol) =
    check(ccall((:lmm_set_projection_dtype, liblmm), Cint, (Cint,), d === :bf16 ? 1 : (d === :bf16x2 ? 2 : 0)))

# get_latent_gp(posterior(fx::FiniteGP{<:ILMM}, y)) for a dense H (reference src/ilmm.jl:39 on the ILMM of :196-197): the coupled
# latent PosteriorGP{IndependentMOGP} as a handle that shares the posterior's device state with H = I_m; the lmm_ilmm_post_*
# entry points then answer for the m latent outputs (jitters {0, σ², 0}: see include/lmm_hip.h).
function latent_view(f::ILMM{<:HIPMOGP,<:Matrix})
    isposterior(f.f) || return f.f
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lmm_ilmm_post_latent_view, liblmm), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), f.f.handle, h))
    # (train and mix ride along for the gradient of the view's logpdf: lmm_ilmm_post_latent_logpdf_grad_seq)
    return ILMM(HIPMOGP(f.f.fs, h[], f.f.train, Matrix{Float64}(f.H)), Matrix{Float64}(I, length(f.f.fs), length(f.f.fs)))
end

# ---- multi-GPU: one Julia process per GPU; the collective lives in the library (RCCL over xGMI) ----------------------------
# rank 0: id = unique_id(); ship the 128 bytes to the other ranks (MPI.bcast!(id, 0, comm)); all: comm_init_rank(id, rank, world).
unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:lmm_comm_get_unique_id, liblmm), Cint, (Ptr{UInt8},), id)); id)
comm_init_rank(id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:lmm_comm_init_rank, liblmm), Cint, (Ptr{UInt8}, Cint, Cint), id, rank, world))
allreduce_sum!(buf::Vector{Float64}) = (check(ccall((:lmm_allreduce_sum_f64, liblmm), Cint, (Ptr{Cdouble}, Csize_t), buf, length(buf))); buf)
comm_destroy() = check(ccall((:lmm_comm_destroy, liblmm), Cint, ()))

# contiguous block partition of m latents over `world` ranks (the first m % world ranks get one extra)
function latent_shard(m::Integer, rank::Integer, world::Integer)
    base, extra = divrem(m, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (rank < extra ? 1 : 0)
end

# logpdf of an OILMM with the latents sharded over the ranks: each rank evaluates its block (no data-path collective), rank 0
# adds the regulariser, ONE 8-byte all-reduce finishes it (SURVEY.md section 8e)
function sharded_logpdf(fx::ByOutputsFill{HIPOILMM}, y::AbstractVector{<:Real}, rank::Integer, world::Integer)
    fs, H, σ², x = unpack(fx)
    X = _xmat(x); d, n = size(X); U, S, p, m = _hargs(H); gps = _gps(fs.fs); yv = Vector{Float64}(y)
    l0, l1 = latent_shard(m, rank, world)
    out = Ref{Cdouble}(0.0)
    GC.@preserve X yv U S gps check(ccall((:lmm_oilmm_logpdf, liblmm), Cint,
        (Ptr{Cdouble}, Cint, Cint, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{LmmGp}, Cint, Cint, Cint, Ref{Cdouble}),
        X, d, n, yv, p, U, S, m, σ², gps, l0, l1, rank == 0 ? 1 : 0, out))
    return allreduce_sum!([out[]])[1]
end

end # module
