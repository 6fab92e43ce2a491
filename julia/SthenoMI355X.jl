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
function ctx()
    if CTX[] == C_NULL
        rc = ccall((:sgp_ctx_create, LIB), Cint, (Cint, Ptr{Ptr{Cvoid}}), 0, CTX)
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

# KernelFunctions kernel -> [(kind, coef, param, input_scale)]
leaf(::SEKernel) = [(0, 1.0, 0.0, 1.0)]
leaf(::Matern12Kernel) = [(1, 1.0, 0.0, 1.0)]       # == ExponentialKernel
leaf(::Matern32Kernel) = [(2, 1.0, 0.0, 1.0)]
leaf(::Matern52Kernel) = [(3, 1.0, 0.0, 1.0)]
leaf(::WhiteKernel) = [(4, 1.0, 0.0, 1.0)]
leaf(k::ConstantKernel) = [(5, 1.0, only(k.c), 1.0)]
leaf(k::ScaledKernel) = [(a, c * only(k.σ²), p, s) for (a, c, p, s) in leaf(k.kernel)]
leaf(k::KernelSum) = reduce(vcat, leaf.(k.kernels))
leaf(k::TransformedKernel{<:Any,<:ScaleTransform}) =
    [(a, c, p, s * only(k.transform.s)) for (a, c, p, s) in leaf(k.kernel)]

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
        push!(inputs, s == 1.0 ? X : s .* X)
        push!(cin, CInput(size(X, 1), size(X, 2), size(X, 1), pointer(inputs[end])))
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

noise_args(Σ::AbstractGPs.ScalMat) = (0, [Σ.value])          # f(x, s2)   (and the 1e-18 default)
noise_args(Σ::Diagonal) = (1, collect(Float64, diag(Σ)))     # f(x, v)
noise_args(Σ::AbstractMatrix) = (2, Matrix{Float64}(Σ))      # f(x, S)

# ---- operator surface ------------------------------------------------------------------------
function logpdf(fx::SthenoFGP, Y::AbstractMatrix{<:Real})
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x))
    kind, nz = noise_args(fx.Σy); Yd = Matrix{Float64}(Y); out = zeros(size(Yd, 2))
    GC.@preserve sp m nz Yd out check(ccall((:sgp_logpdf, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}),
        ctx(), sp.c, m, kind, nz, Yd, size(Yd, 1), size(Yd, 2), out))
    return out
end
logpdf(fx::SthenoFGP, y::AbstractVector{<:Real}) = only(logpdf(fx, reshape(y, :, 1)))

function rand(rng::AbstractRNG, fx::SthenoFGP, S::Int)
    sp = build_spec(fx.f, fx.x); m = collect(Float64, mean(fx.f, fx.x)); kind, nz = noise_args(fx.Σy)
    Z = randn(rng, Float64, length(fx), S)      # the caller's integer RNG stream, column-major fill
    out = similar(Z)
    GC.@preserve sp m nz Z out check(ccall((:sgp_rand, LIB), Cint,
        (Ptr{Cvoid}, Ref{CSpec}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Int64),
        ctx(), sp.c, m, kind, nz, Z, size(Z, 1), S, out, size(out, 1)))
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

end # module
