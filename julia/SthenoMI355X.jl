# SthenoMI355X.jl -- the Julia-side binding a Stheno.jl maintainer would add so that
# `logpdf / rand / posterior / elbo` on Stheno FiniteGPs run in libsthenomi.so (HIP, gfx950).
#
# NOT EXECUTED in the build container (no Julia toolchain; see INTEGRATION.md).  It is kept
# mechanically simple: flatten the GP tree into kernel terms (SURVEY.md Appendix B -- the same
# algorithm as stheno.jl_amd/flatten.py, which IS tested against the reference's recursion),
# fill the C structs of include/sthenomi.h, `ccall`.  User code and the @gppp machinery are
# untouched: the methods below are more specific than AbstractGPs' `FiniteGP{<:AbstractGP}`
# methods, exactly like the existing override in Stheno's src/gp/util.jl:12-14.
module SthenoMI355X

using Stheno, AbstractGPs, KernelFunctions, LinearAlgebra, Random
using Stheno: GPPP, SthenoAbstractGP, AtomicGP, DerivedGP, BlockData, GPPPInput, extract_components
import AbstractGPs: logpdf, rand, posterior, elbo, FiniteGP, VFE

const LIB = get(ENV, "STHENOMI_LIB", "libsthenomi.so")
const SthenoFGP = FiniteGP{<:Union{GPPP,SthenoAbstractGP}}

# ---- C structs (include/sthenomi.h) --------------------------------------------------------
struct CInput; dim::Int64; n::Int64; ld::Int64; x::Ptr{Float64}; end
struct CTerm
    kind::Int32; row_input::Int32; col_input::Int32; reserved::Int32
    coef::Float64; param::Float64; row_scale::Ptr{Float64}; col_scale::Ptr{Float64}
end
struct CSpec
    n_row_blocks::Int32; n_col_blocks::Int32; row_len::Ptr{Int64}; col_len::Ptr{Int64}
    n_inputs::Int32; inputs::Ptr{CInput}; term_ptr::Ptr{Int32}; terms::Ptr{CTerm}
    symmetric::Int32; reserved::Int32
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)
# One context per process.  ENV["STHENOMI_DEVICES"] = "0,1,2,3,4,5,6,7" makes it a multi-GPU context
# (sgp_ctx_create_multi): `logpdf`, `posterior` (+ its predictions), `rand` and `elbo` are then sharded over the
# listed GPUs inside the library (RCCL / peer copies over xGMI); covariances, gradients and the sparse posterior run
# on the first device -- the Julia side does not change.
function ctx()
    if CTX[] == C_NULL
        devs = get(ENV, "STHENOMI_DEVICES", "")
        rc = if isempty(devs)
            ccall((:sgp_ctx_create, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}), 0, CTX)
        else
            d = Cint[parse(Cint, t) for t in split(devs, ",")]
            ccall((:sgp_ctx_create_multi, LIB), Cint, (Ptr{Cint}, Cint, Ptr{Ptr{Cvoid}}), d, length(d), CTX)
        end
        rc == 0 || error(unsafe_string(ccall((:sgp_last_error, LIB), Cstring, ())))
    end
    return CTX[]
end
function check(rc)
    rc == 0 && return
    rc > 0 && throw(PosDefException(rc))      # same exception `cholesky` throws in the reference
    error(unsafe_string(ccall((:sgp_last_error, LIB), Cstring, ())))
end

# ---- flattening (SURVEY.md Appendix B) ------------------------------------------------------
struct Path; key::Tuple; atom::AtomicGP; c::Float64; r::Union{Nothing,Vector{Float64}}; X::Matrix{Float64}; end
mat(x::ColVecs) = x.X
mat(x::AbstractVector{<:Real}) = reshape(collect(Float64, x), 1, :)

paths(f::AtomicGP, x, c, r, key) = f.gp isa GP ? [Path((key..., objectid(f)), f, c, r, mat(x))] :
    paths(f.gp, x, c, r, (key..., objectid(f)))
paths(f::GPPP, x, c, r, key) = paths(extract_components(f, x)..., c, r, key)
paths(f::DerivedGP, x, c, r, key) = paths(f.args, x, c, r, key)
paths((_, fa, fb)::Tuple{typeof(+),AbstractGP,AbstractGP}, x, c, r, key) =
    vcat(paths(fa, x, c, r, key), paths(fb, x, c, r, key))
paths((_, b, f)::Tuple{typeof(+),Any,AbstractGP}, x, c, r, key) = paths(f, x, c, r, key)
paths((_, s, f)::Tuple{typeof(*),Real,AbstractGP}, x, c, r, key) = paths(f, x, c * s, r, key)
function paths((_, s, f)::Tuple{typeof(*),Any,AbstractGP}, x, c, r, key)
    sx = Float64.(s.(x))
    return paths(f, x, c, r === nothing ? sx : r .* sx, key)
end
paths((_, f, g)::Tuple{typeof(∘),AbstractGP,Any}, x, c, r, key) = paths(f, g.(x), c, r, key)

# KernelFunctions kernel -> [(kind, coef, param, chain)]: `chain` is the ordered list of input-transform steps the
# host applies to the points before upload -- (:scale, s) for ScaleTransform(s) / with_lengthscale, (:periodic, f) for
# PeriodicTransform(f) (examples/extended_mauna_loa/script.jl:129: x -> [sin(2 pi f x); cos(2 pi f x)]) -- outermost
# transform first, exactly as stheno.jl_amd/kernels.py (`_push` / `apply_chain`, which IS tested against the reference
# semantics): (k o PeriodicTransform(f)) o ScaleTransform(a) and nested periodic transforms are ordinary chains.
const Chain = Vector{Tuple{Symbol,Float64}}
leaf(::SEKernel) = [(0, 1.0, 0.0, Chain())]
leaf(::Matern12Kernel) = [(1, 1.0, 0.0, Chain())]       # == ExponentialKernel
leaf(::Matern32Kernel) = [(2, 1.0, 0.0, Chain())]
leaf(::Matern52Kernel) = [(3, 1.0, 0.0, Chain())]
leaf(::WhiteKernel) = [(4, 1.0, 0.0, Chain())]
leaf(k::ConstantKernel) = [(5, 1.0, only(k.c), Chain())]
leaf(k::ScaledKernel) = [(a, c * only(k.σ²), p, ch) for (a, c, p, ch) in leaf(k.kernel)]
leaf(k::KernelSum) = reduce(vcat, leaf.(k.kernels))
# k o t evaluates k(t(x), t(y)): t is applied to the raw points first, the inner kernel's own chain after it
leaf(k::TransformedKernel{<:Any,<:ScaleTransform}) =
    [(a, c, p, vcat([(:scale, Float64(only(k.transform.s)))], ch)) for (a, c, p, ch) in leaf(k.kernel)]
leaf(k::TransformedKernel{<:Any,<:PeriodicTransform}) =
    [(a, c, p, vcat([(:periodic, Float64(only(k.transform.f)))], ch)) for (a, c, p, ch) in leaf(k.kernel)]
function apply_input(X, chain::Chain)
    for (op, v) in chain
        if op === :scale
            X = v == 1.0 ? X : v .* X
        else                                 # :periodic -- 1-D points only, as PeriodicTransform itself
            θ = (2π * v) .* X
            X = vcat(sin.(θ), cos.(θ))
        end
    end
    return X
end

blocks_of(f::GPPP, x) = blocks_of(extract_components(f, x)...)
blocks_of(f::DerivedGP, x::BlockData) = f.args[1] === cross ?
    reduce(vcat, [blocks_of(g, b) for (g, b) in zip(f.args[2], x.X)]) : [(f, x)]
blocks_of(f, x) = [(f, x)]

# Owns every buffer a CSpec points into; use inside GC.@preserve.
mutable struct Spec
    c::CSpec; keep::Vector{Any}; N::Int; M::Int
end
function build_spec(f, x, f2 = f, x2 = nothing)
    sym = x2 === nothing
    rows = blocks_of(f, x); cols = sym ? rows : blocks_of(f2, x2)
    rp = [paths(n, v, 1.0, nothing, ()) for (n, v) in rows]
    cp = sym ? rp : [paths(n, v, 1.0, nothing, ()) for (n, v) in cols]
    inputs = Matrix{Float64}[]; cin = CInput[]; terms = CTerm[]; tptr = Int32[0]
    function input!(X, s)
        Xt = apply_input(X, s)
        push!(inputs, Xt)
        push!(cin, CInput(size(Xt, 1), size(Xt, 2), size(Xt, 1), pointer(inputs[end])))
        return Int32(length(inputs) - 1)
    end
    for pi in rp, pj in cp
        for p in pi, q in pj
            p.key == q.key || continue
            for (kind, kc, param, s) in leaf(p.atom.gp.kernel)
                push!(terms, CTerm(kind, input!(p.X, s), input!(q.X, s), 0, p.c * q.c * kc, param,
                    p.r === nothing ? C_NULL : pointer(p.r), q.r === nothing ? C_NULL : pointer(q.r)))
            end
        end
        push!(tptr, Int32(length(terms)))
    end
    rl = Int64[length(v) for (_, v) in rows]; cl = Int64[length(v) for (_, v) in cols]
    c = CSpec(length(rl), length(cl), pointer(rl), pointer(cl), length(cin), pointer(cin),
              pointer(tptr), pointer(terms), sym ? 1 : 0, 0)
    return Spec(c, Any[rp, cp, inputs, cin, terms, tptr, rl, cl], sum(rl), sum(cl))
end

# ---- block order (round 5) ----------------------------------------------------------------------
# WHICH tiles of the factor are structurally zero depends on the order of the blocks of the BlockData (f3 = f1 + f2 observed
# as (f3, f1, f2) fills the factor in and the factorisation silently takes the dense time).  The library suggests a
# fill-reducing order (sgp_cov_spec_suggest_order: greedy minimum fill on the block graph; host-only, no GPU work);
# logpdf -- whose value does not depend on the order of the observations beyond rounding -- applies it to the blocks, y,
# the mean and the noise.  `rand` and `posterior` keep the caller's order: the factor's layout is part of their results.
function suggest_order(sp::Spec)
    nb = Int(sp.c.n_row_blocks); perm = zeros(Int32, nb); ch = Ref{Int32}(0)
    GC.@preserve sp perm check(ccall((:sgp_cov_spec_suggest_order, LIB), Cint, (Ref{CSpec}, Ptr{Int32}, Ref{Int32}),
        sp.c, perm, ch))
    return Int.(perm) .+ 1, ch[] != 0
end
# the observation indices of the blocks of x in the order `perm` (one range per block, block lengths from the spec)
function block_permutation(sp::Spec, perm)
    rl = sp.keep[7]; offs = cumsum(vcat(0, rl))
    return reduce(vcat, [collect((offs[b] + 1):offs[b + 1]) for b in perm])
end
permute_noise(Σ::AbstractGPs.ScalMat, idx) = Σ
permute_noise(Σ::Diagonal, idx) = Diagonal(diag(Σ)[idx])
permute_noise(Σ::AbstractMatrix, idx) = Σ[idx, idx]

noise_args(Σ::AbstractGPs.ScalMat) = (0, [Σ.value])          # f(x, s2)   (and the 1e-18 default)
noise_args(Σ::Diagonal) = (1, collect(Float64, diag(Σ)))     # f(x, v)
noise_args(Σ::AbstractMatrix) = (2, Matrix{Float64}(Σ))      # f(x, S)

# ---- operator surface ------------------------------------------------------------------------
function logpdf(fx::SthenoFGP, Y::AbstractMatrix{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x))
    Σ = fx.Σy; Yd = Matrix{Float64}(Y)
    if fx.x isa BlockData && !(Σ isa AbstractMatrix && !(Σ isa Diagonal) && !(Σ isa AbstractGPs.ScalMat))
        perm, changes = suggest_order(sp)            # a dense Σy has no structural zeros to keep: left alone
        # (`perm` indexes the spec's FLATTENED row blocks: a nested programme observed through BlockData can have more of them
        # than fx.x.X has entries -- then the caller's order is kept; advisor, round 5)
        if changes && length(perm) == length(fx.x.X) # a fill-reducing block order: same value, the skipped tile products back
            idx = block_permutation(sp, perm)
            sp = build_spec(fx.f, BlockData(fx.x.X[perm])); m = m[idx]; Yd = Yd[idx, :]; Σ = permute_noise(Σ, idx)
        end
    end
    kind, nz = noise_args(Σ); out = zeros(size(Yd, 2))
    GC.@preserve sp m nz Yd out check(ccall((:sgp_logpdf, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
        ctx(), sp.c, m, kind, nz, Yd, size(Yd, 1), size(Yd, 2), out))
    return out
end
logpdf(fx::SthenoFGP, y::AbstractVector{<:Real}) = only(logpdf(fx, reshape(y, :, 1)))

# logpdf of several INDEPENDENT models in one call (sgp_logpdf_batch, round 6): restarts of an optimiser, cross-validation
# folds, a population of hyper-parameter candidates -- [logpdf(fx, y) for (fx, y) in zip(fxs, ys)], every value bit-equal to
# the member's own call.  Members of one padded size (equal N or N within one 128-column tile; scalar / diagonal Σy) are
# factored as ONE task pool of the dataflow kernel:
# at N <= 8192 one factorisation is bound by its diagonal chain and the B chains hide each other (N = 4096: 0.13 -> 0.49 of
# the fp64 MFMA peak at B = 8).  A member that is not positive definite gives NaN instead of throwing (its LAPACK info in
# the second result), so that one bad candidate does not lose the others.
function logpdf_batch(fxs::AbstractVector{<:SthenoFGP}, ys::AbstractVector{<:AbstractVector{<:Real}})
    length(fxs) == length(ys) || throw(DimensionMismatch("logpdf_batch: one y per model"))
    B = length(fxs)
    kinds = [noise_args(fx.Σy)[1] for fx in fxs]
    if B == 0 || !allequal(kinds) || first(kinds) == 2        # mixed or dense noise kinds: member by member
        vals = Float64[]; infos = Cint[]
        for (fx, y) in zip(fxs, ys)
            try
                push!(vals, logpdf(fx, y)); push!(infos, 0)
            catch e
                e isa PosDefException || rethrow()
                push!(vals, NaN); push!(infos, e.info)
            end
        end
        return vals, infos
    end
    sps = [build_spec(fx.f, fx.x) for fx in fxs]
    ms = [collect(Float64, mean(fx.f, fx.x)) for fx in fxs]
    nzs = [noise_args(fx.Σy)[2] for fx in fxs]
    yv = [collect(Float64, y) for y in ys]
    specs = [Ptr{CSpec}(pointer_from_objref(sp)) for sp in sps]     # (`c` is the first field of the mutable Spec: its address)
    out = zeros(B); infos = zeros(Cint, B)
    GC.@preserve sps ms nzs yv specs out infos check(ccall((:sgp_logpdf_batch, LIB), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Ptr{CSpec}}, Ptr{Ptr{Float64}}, Cint, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Cint}),
        ctx(), B, specs, pointer.(ms), first(kinds), pointer.(nzs), pointer.(yv), out, infos))
    return out, infos
end

# Float32 models (test/gp/util.jl:76-88: `logpdf(fx, y) isa Float32`): fp32 assembly + fp32 Cholesky on the device
# (sgp_logpdf_f32).  The spec is passed in Float64 (an exact conversion); the library rounds it to fp32 once.
# The element type of the POINTS, looking through GPPPInput / BlockData (whose own eltype is a Tuple / a Union): what decides
# between the fp32 and the fp64 device path for GPPP models too.
point_eltype(x::AbstractVector{<:Real}) = eltype(x)
point_eltype(x::ColVecs) = eltype(x.X)
point_eltype(x::GPPPInput) = point_eltype(x.x)
point_eltype(x::BlockData) = promote_type(map(point_eltype, x.X)...)
point_eltype(x) = Float64
# limits of the fp32 kernels (include/sthenomi.h): input dimension <= 16, terms per block pair x dimension <= 64
function f32_supported(sp::Spec)
    tptr = sp.keep[6]; terms = sp.keep[5]; inputs = sp.keep[3]
    for p in 1:(length(tptr) - 1)
        t0, t1 = tptr[p] + 1, tptr[p + 1]
        t1 < t0 && continue
        d = maximum(size(inputs[terms[t].row_input + 1], 1) for t in t0:t1)
        dmax = nextpow(2, max(d, 1))
        (dmax > 16 || (t1 - t0 + 1) * dmax > 64) && return false
    end
    return true
end
function logpdf(fx::SthenoFGP, y::AbstractVector{Float32})
    point_eltype(fx.x) === Float32 || return only(logpdf(fx, reshape(y, :, 1)))
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    # dense Sigma_y and models beyond the fp32 kernels' limits: fp64 arithmetic, Float32 result (type stability)
    (kind == 2 || !f32_supported(sp)) && return Float32(only(logpdf(fx, reshape(y, :, 1))))
    yd = collect(Float64, y); out = zeros(1)
    GC.@preserve sp m nz yd out check(ccall((:sgp_logpdf_f32, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx(), sp.c, m, kind, collect(Float64, nz), yd, out))
    return Float32(out[1])
end

function rand(rng::AbstractRNG, fx::SthenoFGP, S::Int)
    point_eltype(fx.x) === Float32 && return rand_f32(rng, fx, S)
    Z = randn(rng, Float64, length(fx), S)      # the caller's integer RNG stream, column-major fill
    return rand_with(fx, Z)
end
# `rand(rng, fx) isa Vector{Float32}` for Float32 models (test/gp/util.jl:76-88): the draw stays Float32 on the caller's
# RNG stream (randn(rng, Float32, ...), as AbstractGPs does), the fp32 factor and product run on the device (sgp_rand_f32)
function rand_f32(rng::AbstractRNG, fx::SthenoFGP, S::Int)
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    Z32 = randn(rng, Float32, length(fx), S)
    (kind == 2 || !f32_supported(sp)) && return Float32.(rand_with(fx, Float64.(Z32)))
    Z = Float64.(Z32); out = zeros(Float32, size(Z))
    GC.@preserve sp m nz Z out check(ccall((:sgp_rand_f32, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float32}, Int64),
        ctx(), sp.c, m, kind, collect(Float64, nz), Z, size(Z, 1), S, out, size(out, 1)))
    return out
end
function rand_with(fx::SthenoFGP, Z::Matrix{Float64})          # m .+ L Z for a given draw (fp64 device path)
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    out = similar(Z)
    GC.@preserve sp m nz Z out check(ccall((:sgp_rand, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        ctx(), sp.c, m, kind, nz, Z, size(Z, 1), size(Z, 2), out, size(out, 1)))
    return out
end
rand(rng::AbstractRNG, fx::SthenoFGP) = vec(rand(rng, fx, 1))

mutable struct MI355XPosterior{Tf,Tx} <: AbstractGPs.AbstractGP
    prior::Tf; x::Tx; α::Vector{Float64}; δ::Vector{Float64}; handle::Ptr{Cvoid}
end
function posterior(fx::SthenoFGP, y::AbstractVector{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    yd = collect(Float64, y); α = zeros(length(yd)); h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve sp m nz yd α check(ccall((:sgp_posterior_create, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Cvoid}}),
        ctx(), sp.c, m, kind, nz, yd, α, h))
    post = MI355XPosterior(fx.f, fx.x, α, yd .- m, h[])
    finalizer(p -> ccall((:sgp_posterior_destroy, LIB), Cint, (Ptr{Cvoid},), p.handle), post)
    return post
end
function predict(p::MI355XPosterior, xs; want_cov = false)
    cr = build_spec(p.prior, xs, p.prior, p.x); ss = build_spec(p.prior, xs)
    ms = collect(Float64, mean(p.prior, xs)); n = length(ms)
    μ = zeros(n); v = zeros(n); C = want_cov ? zeros(n, n) : zeros(0, 0)
    GC.@preserve cr ss ms μ v C check(ccall((:sgp_posterior_predict, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ref{CSpec}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
        p.handle, cr.c, ss.c, ms, μ, v, want_cov ? pointer(C) : C_NULL, max(n, 1)))
    return μ, v, C
end
AbstractGPs.mean(p::MI355XPosterior, xs::AbstractVector) = predict(p, xs)[1]
AbstractGPs.var(p::MI355XPosterior, xs::AbstractVector) = predict(p, xs)[2]
AbstractGPs.cov(p::MI355XPosterior, xs::AbstractVector) = predict(p, xs; want_cov = true)[3]
AbstractGPs.mean_and_var(p::MI355XPosterior, xs::AbstractVector) = predict(p, xs)[1:2]

function elbo(v::VFE, fx::SthenoFGP, y::AbstractVector{<:Real})
    fz = v.fz; @assert fz.f === fx.f
    zz = build_spec(fz.f, fz.x); xz = build_spec(fx.f, fx.x, fz.f, fz.x)
    varx = collect(Float64, var(fx.f, fx.x)); m = collect(Float64, mean(fx.f, fx.x))
    kx, nx = noise_args(fx.Σy); kz, nz = noise_args(fz.Σy); yd = collect(Float64, y); out = zeros(1)
    GC.@preserve zz xz varx m nx nz yd out check(ccall((:sgp_elbo, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ref{CSpec}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}), ctx(), zz.c, xz.c, varx, m, kx, nx, kz, nz, yd, out))
    return out[1]
end

# ---- reverse mode (what an rrule for logpdf / elbo on Stheno FiniteGPs would call) -----------------
# Returns the value and the raw per-term gradients of include/sthenomi.h (d/d coef, d/d input scale
# of every flattened term, in spec order); mapping them back onto the kernel parameters of the
# programme is the pullback's job.
function logpdf_and_gradient(fx::SthenoFGP, y::AbstractVector{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    @assert kind != 2 "dense observation noise has no device gradient"
    yd = collect(Float64, y); n = length(yd); nt = max(1, length(sp.keep[5]))
    lp = zeros(1); gy = zeros(n); gm = zeros(n); gn = zeros(kind == 1 ? n : 1); gc = zeros(nt); gs = zeros(nt)
    GC.@preserve sp m nz yd check(ccall((:sgp_logpdf_grad, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx(), sp.c, m, kind, nz, yd, lp, gy, gm, gn, gc, gs))
    return (logpdf = lp[1], y = gy, mean = gm, noise = gn, coef = gc, inscale = gs)
end

function elbo_and_gradient(v::VFE, fx::SthenoFGP, y::AbstractVector{<:Real})
    fz = v.fz; @assert fz.f === fx.f
    zz = build_spec(fz.f, fz.x); xz = build_spec(fx.f, fx.x, fz.f, fz.x); xx = build_spec(fx.f, fx.x)
    varx = collect(Float64, var(fx.f, fx.x)); m = collect(Float64, mean(fx.f, fx.x))
    kx, nx = noise_args(fx.Σy); kz, nz = noise_args(fz.Σy); yd = collect(Float64, y)
    n = length(yd); mz = length(fz)
    out = zeros(1); gy = zeros(n); gm = zeros(n); gv = zeros(n); gn = zeros(kx == 1 ? n : 1); gzn = zeros(kz == 1 ? mz : 1)
    ntz = max(1, length(zz.keep[5])); ntx = max(1, length(xz.keep[5])); ntd = max(1, length(xx.keep[5]))
    gcz = zeros(ntz); gsz = zeros(ntz); gcx = zeros(ntx); gsx = zeros(ntx); gcd = zeros(ntd); gsd = zeros(ntd)
    GC.@preserve zz xz varx m nx nz yd check(ccall((:sgp_elbo_grad, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ref{CSpec}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx(), zz.c, xz.c, varx, m, kx, nx, kz, nz, yd, out, gy, gm, gn, gv, gzn, gcz, gsz, gcx, gsx))
    GC.@preserve xx gv check(ccall((:sgp_kernelmatrix_diag_grad, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx(), xx.c, gv, gcd, gsd))
    return (elbo = out[1], y = gy, mean = gm, noise = gn, z_noise = gzn,
            zz = (coef = gcz, inscale = gsz), xz = (coef = gcx, inscale = gsx), xx = (coef = gcd, inscale = gsd))
end

# ---- statistics of FiniteGPs: cov(fx), cov(fx, gx), var, mean_and_*, marginals -------------------------
# replaces src/gp/util.jl:12-14 and the AbstractGPs FiniteGP defaults that call cov(f::GPPP, x)
# (gaussian_process_probabilistic_programme.jl:51-64) -> sgp_kernelmatrix / sgp_kernelmatrix_diag
function kernelmatrix_of(sp::Spec)
    K = zeros(sp.N, sp.M)
    (sp.N == 0 || sp.M == 0) && return K
    GC.@preserve sp K check(ccall((:sgp_kernelmatrix, LIB), Cint, (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Int64),
        ctx(), sp.c, K, sp.N))
    return K
end
function kernelmatrix_diag_of(sp::Spec)
    v = zeros(sp.N)
    sp.N == 0 && return v
    GC.@preserve sp v check(ccall((:sgp_kernelmatrix_diag, LIB), Cint, (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}),
        ctx(), sp.c, v))
    return v
end
AbstractGPs.cov(fx::SthenoFGP, gx::SthenoFGP) = kernelmatrix_of(build_spec(fx.f, fx.x, gx.f, gx.x))  # no noise
AbstractGPs.cov(fx::SthenoFGP) = kernelmatrix_of(build_spec(fx.f, fx.x)) + fx.Σy
AbstractGPs.var(fx::SthenoFGP) = kernelmatrix_diag_of(build_spec(fx.f, fx.x)) .+ diag(fx.Σy)
AbstractGPs.mean_and_cov(fx::SthenoFGP) = (mean(fx.f, fx.x), cov(fx))
AbstractGPs.mean_and_var(fx::SthenoFGP) = (mean(fx.f, fx.x), var(fx))
AbstractGPs.marginals(fx::SthenoFGP) = ((m, v) = mean_and_var(fx); AbstractGPs.Normal.(m, sqrt.(v)))

# ---- posterior(VFE(fz), fx, y) (src/gp/sparse_finite_gp.jl:60-62) ------------------------------------------
# Stheno's SparseFiniteGP methods (sparse_finite_gp.jl:52-62) call elbo(VFE(f.finducing), f.fobs, y) and
# posterior(VFE(f.finducing), f.fobs, y): both land on the overloads of this module, nothing to add.
mutable struct MI355XSparsePosterior{Tf,Tz} <: AbstractGPs.AbstractGP
    prior::Tf; z::Tz; handle::Ptr{Cvoid}
end
function posterior(v::VFE, fx::SthenoFGP, y::AbstractVector{<:Real})
    fz = v.fz; @assert fz.f === fx.f
    zz = build_spec(fz.f, fz.x); xz = build_spec(fx.f, fx.x, fz.f, fz.x)
    m = collect(Float64, mean(fx.f, fx.x)); kx, nx = noise_args(fx.Σy); kz, nz = noise_args(fz.Σy)
    yd = collect(Float64, y); h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve zz xz m nx nz yd check(ccall((:sgp_sparse_posterior_create, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64},
         Ptr{Ptr{Cvoid}}), ctx(), zz.c, xz.c, m, kx, nx, kz, nz, yd, h))
    post = MI355XSparsePosterior(fx.f, fz.x, h[])
    finalizer(p -> ccall((:sgp_sparse_posterior_destroy, LIB), Cint, (Ptr{Cvoid},), p.handle), post)
    return post
end
function predict(p::MI355XSparsePosterior, xs; want_cov = false)
    cr = build_spec(p.prior, xs, p.prior, p.z); ss = build_spec(p.prior, xs)
    ms = collect(Float64, mean(p.prior, xs)); n = length(ms)
    μ = zeros(n); v = zeros(n); C = want_cov ? zeros(n, n) : zeros(0, 0)
    GC.@preserve cr ss ms μ v C check(ccall((:sgp_sparse_posterior_predict, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ref{CSpec}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
        p.handle, cr.c, ss.c, ms, μ, v, want_cov ? pointer(C) : C_NULL, max(n, 1)))
    return μ, v, C
end
AbstractGPs.mean(p::MI355XSparsePosterior, xs::AbstractVector) = predict(p, xs)[1]
AbstractGPs.var(p::MI355XSparsePosterior, xs::AbstractVector) = predict(p, xs)[2]
AbstractGPs.cov(p::MI355XSparsePosterior, xs::AbstractVector) = predict(p, xs; want_cov = true)[3]
AbstractGPs.mean_and_var(p::MI355XSparsePosterior, xs::AbstractVector) = predict(p, xs)[1:2]

# ---- ChainRulesCore.rrule for logpdf(fx, y): what `Zygote.gradient(θ -> logpdf(build_gp(θ)(x, σ²), y), θ)`
# needs (reference contract: test/gaussian_process_probabilistic_programme.jl:99-104 "does not error";
# training loops: examples/getting_started/script.jl:154-213).  The device returns d/dy, d/dmean, d/dΣy,
# per-term d/dcoef and the gradient w.r.t. every uploaded input array (sgp_logpdf_grad_x); the pullback
# walks the GP tree once more, in the same order as `paths`, and turns them into a structural tangent:
#   (*, σ::Real, f)      ∂σ  = Σ_{terms through this node} d_coef_t · coef_t · (#times the node is on the term's two paths) / σ
#   (∘, f, Stretch(l))   ∂l  = ⟨∂X_out, X_in⟩ (scalar l) ;  ∂X_in = l · ∂X_out
#   leaf ScaledKernel σ² ∂σ² = Σ d_coef_t · coef_t / σ² ;  leaf ScaleTransform s: ∂s = ⟨∂X_out, X_in⟩
#   AtomicGP inputs      ∂x  = what is left of ∂X at the leaf
# Coverage: real `*` scales, scalar `Stretch`, the input points, Σy and y.  Everything else that carries a
# differentiable parameter -- hyper-parameters INSIDE `GP(kernel)` (ScaledKernel σ², ScaleTransform s,
# PeriodicTransform f, ConstantKernel c, kernel sums of those), function-valued scales σ(x) (the device returns
# d logpdf / d σ.(x): logpdf_and_gradient_xs below, but this rule has no RuleConfig to call back into the AD system),
# matrix / vector Stretch, Shift, Select, Periodic and custom warps -- is NOT differentiated by this rule, and the rule
# REFUSES such a model (`uncovered` below, ArgumentError from the rrule) instead of handing the optimiser a silent zero:
# before this rule existed Zygote failed loudly on those models, and a training loop must keep failing loudly.  Write
# such parameters at the Stheno level (σ * stretch(GP(SEKernel()), 1 / l)), which is covered, or take the per-term
# gradients of `logpdf_and_gradient_x` / `_xs` and chain them by hand.
using ChainRulesCore
const ParameterFreeKernel = Union{SEKernel,Matern12Kernel,Matern32Kernel,Matern52Kernel,WhiteKernel}
uncovered(k::ParameterFreeKernel) = String[]
uncovered(k::Kernel) = ["hyper-parameters of the leaf kernel $(nameof(typeof(k))) inside GP(kernel)"]
uncovered(f::AtomicGP) = f.gp isa GP ? uncovered(f.gp.kernel) : uncovered(f.gp)
uncovered(f::GPPP) = reduce(vcat, map(uncovered, collect(values(f.fs))); init = String[])
uncovered(f::DerivedGP) = uncovered(f.args)
uncovered((_, fa, fb)::Tuple{typeof(+),AbstractGP,AbstractGP}) = vcat(uncovered(fa), uncovered(fb))
uncovered((_, b, f)::Tuple{typeof(+),Any,AbstractGP}) = uncovered(f)            # a known shift only moves the mean
uncovered((_, s, f)::Tuple{typeof(*),Real,AbstractGP}) = uncovered(f)
uncovered((_, s, f)::Tuple{typeof(*),Any,AbstractGP}) = vcat(["function-valued scale σ(x) * f"], uncovered(f))
uncovered((_, f, g)::Tuple{typeof(∘),AbstractGP,Any}) =
    g isa Stheno.Stretch{<:Real} ? uncovered(f) : vcat(["input transformation $(nameof(typeof(g)))"], uncovered(f))
uncovered(args::Tuple) = ["GP node $(args[1]) without a gradient rule"]
function logpdf_and_gradient_x(fx::SthenoFGP, y::AbstractVector{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    @assert kind != 2 "dense observation noise has no device gradient"
    yd = collect(Float64, y); n = length(yd); nt = max(1, length(sp.keep[5]))
    lp = zeros(1); gy = zeros(n); gm = zeros(n); gn = zeros(kind == 1 ? n : 1); gc = zeros(nt); gs = zeros(nt)
    gx = [zeros(size(X)) for X in sp.keep[3]]; px = [pointer(g) for g in gx]
    GC.@preserve sp m nz yd gx px check(ccall((:sgp_logpdf_grad_x, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Float64}}),
        ctx(), sp.c, m, kind, nz, yd, lp, gy, gm, gn, gc, gs, px))
    return (logpdf = lp[1], y = gy, mean = gm, noise = gn, coef = gc, inscale = gs, inputs = gx, spec = sp)
end

# Same call with the gradient w.r.t. the row-scale vectors of function-scaled processes σ(x) * f
# (src/affine_transformations/product.jl:25-48): rowscale[t] = d logpdf / d (r_t)_i for every term whose row
# carries a scale vector r_t = Π σ_k.(x) (sgp_logpdf_grad_xs); terms sharing one vector add up, and
# ∂σ_k.(x) = (Σ_t rowscale[t]) .* Π_{l≠k} σ_l.(x).  A pullback that wants ∂θ of σ(x; θ) hands that vector to
# the AD system's own pullback of `x -> σ.(x)` (`rrule_via_ad(config, x -> σ.(x), x)` under a RuleConfig).
function logpdf_and_gradient_xs(fx::SthenoFGP, y::AbstractVector{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    @assert kind != 2 "dense observation noise has no device gradient"
    yd = collect(Float64, y); n = length(yd); terms = sp.keep[5]; tptr = sp.keep[6]; nt = max(1, length(terms))
    lp = zeros(1); gy = zeros(n); gm = zeros(n); gn = zeros(kind == 1 ? n : 1); gc = zeros(nt); gs = zeros(nt)
    gx = [zeros(size(X)) for X in sp.keep[3]]; px = [pointer(g) for g in gx]
    rl = [length(v) for (_, v) in blocks_of(fx.f, fx.x)]; nrb = length(rl)          # row_len per block
    gr = Vector{Vector{Float64}}(undef, length(terms)); pr = fill(Ptr{Float64}(C_NULL), nt)
    for I in 1:nrb, J in 1:nrb, t in (tptr[(I - 1) * nrb + J] + 1):tptr[(I - 1) * nrb + J + 1]
        gr[t] = terms[t].row_scale == C_NULL ? Float64[] : zeros(rl[I])
        isempty(gr[t]) || (pr[t] = pointer(gr[t]))
    end
    GC.@preserve sp m nz yd gx px gr pr check(ccall((:sgp_logpdf_grad_xs, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}),
        ctx(), sp.c, m, kind, nz, yd, lp, gy, gm, gn, gc, gs, px, pr))
    return (logpdf = lp[1], y = gy, mean = gm, noise = gn, coef = gc, inscale = gs, inputs = gx, rowscale = gr, spec = sp)
end

# parameter bookkeeping of one path: the real-scale nodes it passed (with their σ) and the Stretch warps
# (with the inputs they were applied to), collected by `trace`, a twin of `paths`
struct Trace; scales::Vector{Any}; warps::Vector{Any}; atom::AtomicGP; end
trace(f::AtomicGP, x, sc, wp) = f.gp isa GP ? [Trace(copy(sc), copy(wp), f)] : trace(f.gp, x, sc, wp)
trace(f::GPPP, x, sc, wp) = trace(extract_components(f, x)..., sc, wp)
trace(f::DerivedGP, x, sc, wp) = trace(f.args, f, x, sc, wp)
trace((_, fa, fb)::Tuple{typeof(+),AbstractGP,AbstractGP}, node, x, sc, wp) = vcat(trace(fa, x, sc, wp), trace(fb, x, sc, wp))
trace((_, b, f)::Tuple{typeof(+),Any,AbstractGP}, node, x, sc, wp) = trace(f, x, sc, wp)
trace((_, s, f)::Tuple{typeof(*),Real,AbstractGP}, node, x, sc, wp) = trace(f, x, vcat(sc, Any[(node, Float64(s))]), wp)
trace((_, s, f)::Tuple{typeof(*),Any,AbstractGP}, node, x, sc, wp) = trace(f, x, sc, wp)
trace((_, f, g)::Tuple{typeof(∘),AbstractGP,Any}, node, x, sc, wp) = trace(f, g.(x), sc, vcat(wp, Any[(node, g, x)]))

function ChainRulesCore.rrule(::typeof(logpdf), fx::SthenoFGP, y::AbstractVector{<:Real})
    missing_rules = unique(uncovered(fx.f))
    isempty(missing_rules) || throw(ArgumentError("SthenoMI355X: the logpdf rrule does not differentiate " *
        join(missing_rules, "; ") * " -- it would return a silent zero gradient for them.  Move the parameter to a " *
        "Stheno-level scale / stretch, or use logpdf_and_gradient_x / logpdf_and_gradient_xs and chain by hand."))
    g = logpdf_and_gradient_x(fx, y)
    function logpdf_pullback(Δ̄)
        Δ = unthunk(Δ̄)
        sp = g.spec; rp = sp.keep[1]; terms = sp.keep[5]; tptr = sp.keep[6]
        rows = blocks_of(fx.f, fx.x)
        tr = [trace(n, v, Any[], Any[]) for (n, v) in rows]          # same order as rp
        dσ = IdDict{Any,Float64}(); dl = IdDict{Any,Any}(); dX = [zeros(size(mat(v))) for (_, v) in rows]
        # coefficients: term t of block pair (I, J) pairs path p of I with path q of J, in `build_spec` order
        t = 0
        for I in eachindex(rp), J in eachindex(rp), (ip, p) in enumerate(rp[I]), (iq, q) in enumerate(rp[J])
            p.key == q.key || continue
            for _ in leaf(p.atom.gp.kernel)
                t += 1
                for (node, σ) in vcat(tr[I][ip].scales, tr[J][iq].scales)
                    dσ[node] = get(dσ, node, 0.0) + g.coef[t] * terms[t].coef / σ
                end
            end
        end
        # inputs: spec input k was registered by term order as well (row input, then column input)
        k = 0
        for I in eachindex(rp), J in eachindex(rp), (ip, p) in enumerate(rp[I]), (iq, q) in enumerate(rp[J])
            p.key == q.key || continue
            for _ in leaf(p.atom.gp.kernel)
                for (B, ib) in ((I, ip), (J, iq))
                    k += 1
                    G = copy(g.inputs[k])          # (covered models have parameter-free leaf kernels: empty chains)
                    for (node, w, xin) in reverse(tr[B][ib].warps)
                        if w isa Stheno.Stretch{<:Real}
                            dl[node] = get(dl, node, 0.0) + sum(G .* mat(xin))
                            G = w.l .* G
                        end                                            # (other warps never get here: `uncovered`)
                    end
                    size(G) == size(dX[B]) && (dX[B] .+= G)
                end
            end
        end
        tangent(f::AtomicGP) = NoTangent()      # parameter-free leaf kernels only (checked by `uncovered` above)
        tangent(f::GPPP) = Tangent{typeof(f)}(fs = map(tangent, f.fs))
        function tangent(f::DerivedGP)
            a = f.args
            if a[1] === (*) && a[2] isa Real
                return Tangent{typeof(f)}(args = (NoTangent(), Δ * get(dσ, f, 0.0), tangent(a[3])))
            elseif a[1] === (∘) && a[3] isa Stheno.Stretch{<:Real}
                return Tangent{typeof(f)}(args = (NoTangent(), tangent(a[2]), Tangent{typeof(a[3])}(l = Δ * get(dl, f, 0.0))))
            elseif a[1] === (+) && a[2] isa AbstractGP
                return Tangent{typeof(f)}(args = (NoTangent(), tangent(a[2]), tangent(a[3])))
            elseif length(a) == 3 && a[3] isa AbstractGP
                return Tangent{typeof(f)}(args = (NoTangent(), NoTangent(), tangent(a[3])))
            end
            return NoTangent()
        end
        ∂Σ = g.noise isa Vector && length(g.noise) == 1 ? Tangent{typeof(fx.Σy)}(value = Δ * g.noise[1]) :
             Tangent{typeof(fx.Σy)}(diag = Δ .* g.noise)
        ∂fx = Tangent{typeof(fx)}(f = tangent(fx.f), x = Δ .* dX, Σy = ∂Σ)
        return NoTangent(), ∂fx, Δ .* g.y
    end
    return g.logpdf, logpdf_pullback
end

end # module
