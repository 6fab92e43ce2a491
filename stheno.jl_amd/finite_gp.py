"""FiniteGP operator surface: f(x, s2), logpdf, rand, posterior, marginals, mean/cov/var,
VFE / elbo, SparseFiniteGP -- the AbstractGPs.jl API Stheno inherits and
src/gp/sparse_finite_gp.jl:30-62, routed through the C-ABI (include/sthenomi.h).

Every number comes out of libsthenomi.so (HIP, gfx950).  The host only flattens the model
(flatten.py), evaluates prior means (O(N)) and draws Z for `rand` from the caller's RNG.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .flatten import build_spec, chain_input_gradients, zero_spec
from .gp import SthenoAbstractGP, mean_vector
from .gppp import GPPP
from .inputs import BlockData, eltype as _eltype


def _ctx():
    return _lib.default_context()


def _f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["F" if order == "F" else "C", "A"])


class FiniteGP:
    """f(x, Sigma_y): Sigma_y real -> s2*I, vector -> Diagonal, matrix -> dense; default 1e-18."""

    def __init__(self, f, x, noise=1e-18):
        self.f, self.x, self.noise = f, x, noise

    def __len__(self):
        return len(self.x)


# ---- a prior-like object is anything below; dispatch by type ---------------------------------
def _is_prior(f):
    return isinstance(f, (GPPP, SthenoAbstractGP))


def _prior_spec(f, x, x2=None):
    spec, _, _ = build_spec(f, x, None, x2)
    return spec


def _kernelmatrix(spec):
    K = np.zeros((spec.N, spec.M), order="F")
    if spec.N and spec.M:
        _lib.check(_ctx().lib.sgp_kernelmatrix(_ctx().handle, spec.ref(), _lib.dptr(K), spec.N), "sgp_kernelmatrix")
    return K


def _kernelmatrix_f32(spec):
    K = np.zeros((spec.N, spec.M), dtype=np.float32, order="F")
    if spec.N and spec.M:
        rc = _ctx().lib.sgp_kernelmatrix_f32(_ctx().handle, spec.ref(), K.ctypes.data_as(C.POINTER(C.c_float)), spec.N)
        _lib.check(rc, "sgp_kernelmatrix_f32")
    return K


def _is_f32(fx, y=None):
    """Float32 model (the reference is type-stable in Float32, test/gp/util.jl:76-88): inputs given in
    Float32 (and, where there are observations, y in Float32) select the fp32 device path."""
    if _eltype(fx.x) != np.float32:
        return False
    return y is None or getattr(y, "dtype", None) == np.float32


def _kernelmatrix_diag(spec):
    out = np.zeros(spec.N)
    if spec.N:
        _lib.check(_ctx().lib.sgp_kernelmatrix_diag(_ctx().handle, spec.ref(), _lib.dptr(out)), "sgp_kernelmatrix_diag")
    return out


def _noise_dense(noise, n):
    a = np.asarray(noise, dtype=np.float64)
    if a.ndim == 0:
        return float(a) * np.eye(n)
    if a.ndim == 1:
        return np.diag(a)
    return a


def _noise_diag(noise, n):
    a = np.asarray(noise, dtype=np.float64)
    if a.ndim == 0:
        return np.full(n, float(a))
    if a.ndim == 1:
        return a
    return np.diag(a).copy()


# ---- statistics of the underlying process ------------------------------------------------------
def prior_mean(f, x):
    if _is_prior(f):
        return mean_vector(f, x)
    return f.mean(x)


def _model_type(out, *inputs):
    """ONE output-type rule for the whole operator surface: results come back in the element type of the inputs they were
    asked at -- Float32 when every input is Float32 (test/gp/util.jl:76-88 checks `rand` / `logpdf` of Float32 models for
    exactly that), Float64 otherwise -- whichever arithmetic produced them (the fp32 device paths cover logpdf, cov, rand
    and posterior moments; everything else is computed in fp64 and rounded here).  None / tuples pass through."""
    if out is None or not all(_eltype(x) == np.float32 for x in inputs):
        return out
    if isinstance(out, tuple):
        return tuple(_model_type(o, *inputs) for o in out)
    return np.float32(out) if np.ndim(out) == 0 else np.asarray(out).astype(np.float32, copy=False)


def prior_cov(f, x, x2=None):
    if _is_prior(f):
        if _eltype(x) == np.float32 and (x2 is None or _eltype(x2) == np.float32):
            spec = _prior_spec(f, x, x2)
            if spec.f32_supported():
                return _kernelmatrix_f32(spec)                   # fp32 assembly on the device
            # beyond the fp32 kernels' limits (input dimension > 16, very many terms per block pair): the fp64
            # kernels take any dimension and term count; the result is rounded to the model's type
            return _kernelmatrix(spec).astype(np.float32)
        return _kernelmatrix(_prior_spec(f, x, x2))
    return f.cov(x, x2)          # (posteriors round to the model's type themselves)


def prior_var(f, x):
    if _is_prior(f):
        return _model_type(_kernelmatrix_diag(_prior_spec(f, x)), x)
    return f.var(x)


def mean(fx):
    if isinstance(fx, SparseFiniteGP):     # sparse_finite_gp.jl:37
        return mean(fx.fobs)
    m = prior_mean(fx.f, fx.x)
    return m.astype(np.float32) if _eltype(fx.x) == np.float32 else m


def cov(fx, gx=None):
    if isinstance(fx, SparseFiniteGP):     # sparse_finite_gp.jl:39-43: explicit error, use cov(f.fobs)
        raise RuntimeError(_COV_ERR)
    if gx is None:
        K = prior_cov(fx.f, fx.x)
        return K + _noise_dense(fx.noise, len(fx)).astype(K.dtype)
    # src/gp/util.jl:12-14: cov(fx, gx) = cov(fx.f, gx.f, fx.x, gx.x) -- no noise
    if _is_prior(fx.f) and _is_prior(gx.f):
        spec, _, _ = build_spec(fx.f, fx.x, gx.f, gx.x)
        return _model_type(_kernelmatrix(spec), fx.x, gx.x)
    raise TypeError("cov(fx, gx) needs two FiniteGPs of one Stheno model")


def var(fx):
    if isinstance(fx, SparseFiniteGP):
        return var(fx.fobs)
    return _model_type(prior_var(fx.f, fx.x) + _noise_diag(fx.noise, len(fx)), fx.x)


def mean_and_cov(fx):
    return mean(fx), cov(fx)


def mean_and_var(fx):
    if isinstance(fx, SparseFiniteGP):
        return mean_and_var(fx.fobs)
    if isinstance(fx.f, (PosteriorGP, ApproxPosteriorGP)):
        m, v = fx.f.mean_and_var(fx.x)
        v = v + _noise_diag(fx.noise, len(fx))
        if _eltype(fx.x) == np.float32:
            return m.astype(np.float32), v.astype(np.float32)
        return m, v
    m, v = mean(fx), var(fx)
    return m, (v.astype(np.float32) if _eltype(fx.x) == np.float32 else v)


class Normal:
    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma

    def __eq__(self, other):
        return self.mu == other.mu and self.sigma == other.sigma


def marginals(fx):
    """Vector of Normal(mean_i, sqrt(var_i)) (test/gp/util.jl:15-20)."""
    if isinstance(fx, SparseFiniteGP):
        return marginals(fx.fobs)
    m, v = mean_and_var(fx)
    return [Normal(a, b) for a, b in zip(m, np.sqrt(v))]


# ---- logpdf / rand ---------------------------------------------------------------------------
def _spec_mean_noise(fx):
    """(spec, mean, noise_kind, noise_buf) for the C-ABI; explicit-covariance processes
    (posteriors) enter as a zero-term spec + dense noise = their covariance."""
    n = len(fx)
    if _is_prior(fx.f):
        spec = _prior_spec(fx.f, fx.x)
        kind, buf = _lib._noise_args(fx.noise, n)
        return spec, mean_vector(fx.f, fx.x), kind, buf
    Cm = fx.f.cov(fx.x) + _noise_dense(fx.noise, n)
    return zero_spec(n), fx.f.mean(fx.x), _lib.NOISE_DENSE, np.asfortranarray(Cm)


def logpdf(fx, y):
    """logpdf(fx, y::Vector) -> float;  logpdf(fx, Y::Matrix) -> one value per column."""
    if isinstance(fx, SparseFiniteGP):
        Y = np.asarray(y, dtype=np.float64)
        if Y.ndim == 2:
            return np.array([elbo(VFE(fx.finducing), fx.fobs, Y[:, j]) for j in range(Y.shape[1])])
        return elbo(VFE(fx.finducing), fx.fobs, Y)
    f32 = _is_prior(fx.f) and _is_f32(fx, y)
    if f32 and np.ndim(y) == 1 and np.ndim(fx.noise) <= 1 and _prior_spec(fx.f, fx.x).f32_supported():
        return logpdf_f32(fx, y)
    # Float32 models the fp32 kernels do not cover (matrix Y, dense Sigma_y, input dimension > 16, more than
    # 64 / dimension terms per block pair) are computed in fp64 and returned in the model's type, so that
    # `logpdf(fx, y) isa Float32` (test/gp/util.jl:76-88) holds whichever path ran
    Y = np.asarray(y, dtype=np.float64)
    vec = Y.ndim == 1
    Y = _f64(Y.reshape(len(fx), -1))
    if Y.shape[0] != len(fx):
        raise ValueError("length(y) != length(fx)")
    spec, m, kind, nbuf = _spec_mean_noise(fx)
    m = _f64(m)
    out = np.zeros(Y.shape[1])
    rc = _ctx().lib.sgp_logpdf(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf), _lib.dptr(Y),
                               Y.shape[0], Y.shape[1], _lib.dptr(out))
    _lib.check(rc, "sgp_logpdf")
    if f32:
        return np.float32(out[0]) if vec else out.astype(np.float32)
    return float(out[0]) if vec else out


def logpdf_batch(fxs, ys, return_infos=False):
    """[logpdf(fx, y) for fx, y in zip(fxs, ys)] in ONE library call (sgp_logpdf_batch): the members are independent models
    -- restarts of an optimiser, cross-validation folds, a population of hyper-parameter candidates.  Members of one padded
    size (equal N, or folds that differ by a point or two inside one 128-column tile) with scalar / diagonal noise are
    factored as one task pool of the dataflow kernel (their diagonal chains hide each other);
    every value is bit-equal to the member's own `logpdf`.  A member that is not positive definite gives NaN (and, with
    return_infos=True, its LAPACK info in the second result) instead of raising, so that one bad candidate does not lose the
    others."""
    fxs, ys = list(fxs), list(ys)
    if len(fxs) != len(ys):
        raise ValueError("logpdf_batch: one y per model")
    if not fxs:
        return (np.zeros(0), np.zeros(0, dtype=np.int32)) if return_infos else np.zeros(0)
    keep = []          # (spec, mean, noise buffer, y) stay alive until the call returns
    kinds = set()
    for fx, y in zip(fxs, ys):
        if isinstance(fx, SparseFiniteGP):
            raise NotImplementedError("logpdf_batch takes FiniteGPs (use elbo for a SparseFiniteGP)")
        yv = _f64(np.asarray(y, dtype=np.float64).ravel())
        if yv.shape[0] != len(fx):
            raise ValueError("length(y) != length(fx)")
        spec, m, kind, nbuf = _spec_mean_noise(fx)
        kinds.add(kind)
        keep.append((spec, _f64(m), nbuf, yv))
    if len(kinds) != 1 or _lib.NOISE_DENSE in kinds:
        # mixed or dense noise kinds: member by member (same values; sgp_logpdf_batch takes one noise kind)
        vals, infos = [], []
        for fx, y in zip(fxs, ys):
            try:
                vals.append(float(logpdf(fx, y)))
                infos.append(0)
            except _lib.PosDefException as e:
                vals.append(float("nan"))
                infos.append(e.info)
        return (np.array(vals), np.array(infos, dtype=np.int32)) if return_infos else np.array(vals)
    nb = len(keep)
    specs = (C.POINTER(_lib.sgp_cov_spec) * nb)(*[C.pointer(k[0].c) for k in keep])
    means = (C.POINTER(C.c_double) * nb)(*[_lib.dptr(k[1]) for k in keep])
    noises = (C.POINTER(C.c_double) * nb)(*[_lib.dptr(k[2]) for k in keep])
    yp = (C.POINTER(C.c_double) * nb)(*[_lib.dptr(k[3]) for k in keep])
    out = np.zeros(nb)
    infos = np.zeros(nb, dtype=np.int32)
    rc = _ctx().lib.sgp_logpdf_batch(_ctx().handle, nb, specs, means, kinds.pop(), noises, yp, _lib.dptr(out),
                                     infos.ctypes.data_as(C.POINTER(C.c_int)))
    _lib.check(rc, "sgp_logpdf_batch")
    return (out, infos) if return_infos else out


def logpdf_f32(fx, y):
    """logpdf(fx, y) on the fp32 device path (sgp_logpdf_f32: fp32 assembly, fp32 blocked Cholesky on
    v_mfma_f32_32x32x2_f32, fp32 forward substitution) -> np.float32.  Selected automatically by `logpdf`
    when the inputs and y are Float32."""
    n = len(fx)
    yv = _f64(np.asarray(y, dtype=np.float64).ravel())
    if yv.shape[0] != n:
        raise ValueError("length(y) != length(fx)")
    spec = _prior_spec(fx.f, fx.x)
    m = _f64(mean_vector(fx.f, fx.x))
    kind, nbuf = _lib._noise_args(fx.noise, n)
    if kind == _lib.NOISE_DENSE:
        # (`logpdf` itself takes a Float32 model with dense Sigma_y through the fp64 factorisation and returns Float32)
        raise ValueError("logpdf_f32 is the explicit fp32 kernel path (scalar / diagonal Sigma_y); call logpdf for a dense Sigma_y")
    out = np.zeros(1)
    rc = _ctx().lib.sgp_logpdf_f32(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf), _lib.dptr(yv), _lib.dptr(out))
    _lib.check(rc, "sgp_logpdf_f32")
    return np.float32(out[0])


def _scale_records(spec, grs, more=()):
    """Per-term row-scale gradients -> one record per function scale `sigma * f` and input collection it was
    mapped over: {node, x, values (= sigma.(x)), d_values (= d logpdf / d values)}.  A path's scale vector is the
    product of its factors' values; a vector carried by several terms gets the sum of their gradients.
    `more`: further (spec, per-term gradients, "row" | "col") triples whose vectors join the same records (the ELBO
    reads one scale vector through K(z,z), K(x,z) and diag K(x,x))."""
    by_vec = {}
    for sp, grads, side in ((spec, grs, "row"),) + tuple(more):
        vecs = sp.term_row_scale if side == "row" else sp.term_col_scale
        for t, g in enumerate(grads or []):
            if g is None:
                continue
            r = vecs[t]
            if id(r) in by_vec:
                by_vec[id(r)][1] += g
            else:
                by_vec[id(r)] = [r, g.copy()]
    out, index = [], {}
    for r, g in by_vec.values():
        fac = getattr(r, "factors", [])
        for k, (node, x, vals) in enumerate(fac):
            others = np.ones(len(vals))
            for l, (_, _, v) in enumerate(fac):
                if l != k:
                    others = others * v
            key = (id(node), id(x))
            if key in index:
                out[index[key]]["d_values"] += g * others
            else:
                index[key] = len(out)
                out.append(dict(node=node, x=x, values=np.asarray(vals, dtype=np.float64).copy(), d_values=g * others))
    return out


def logpdf_and_gradient(fx, y, inputs=False, scales=False):
    """logpdf(fx, y) and its reverse-mode gradient (what Zygote derives on the reference path).
    scales=True adds `scales`: one record per function-valued scale (`sigma * f`, product.jl:25-48) and input
    collection it is mapped over, {node, x, values, d_values} with d_values[i] = d logpdf / d sigma(x_i); the
    gradient of a parameter theta of sigma is sum_i d_values[i] * d sigma(x_i) / d theta.
    inputs=True adds `inputs`: one (dim, n) array per spec input (g["_spec"].inputs[k]) with
    d logpdf / d (the points the terms read -- after Stretch / Select / Periodic; for
    stretch(f, a) the gradient w.r.t. the user's x is a * that array).

    Returns a dict: logpdf; y, mean (d/dy, d/d mean, N each); noise (scalar for f(x, s2), vector for
    f(x, v)); terms: one record per flattened covariance term of the lower block pairs
    {I, J, kind, coef, row_input, col_input, d_coef, d_inscale} where d_coef / d_inscale already
    include the mirror-image pair (J, I).  d_inscale is the derivative w.r.t. a common scale g of the
    term's inputs (stretch(f, g)); a lengthscale l = 1/g gives d/dl = -g^2 d_inscale."""
    if not _is_prior(fx.f):
        raise NotImplementedError("gradients are implemented for prior Stheno processes")
    n = len(fx)
    yv = _f64(np.asarray(y, dtype=np.float64).ravel())
    spec = _prior_spec(fx.f, fx.x)
    m = _f64(mean_vector(fx.f, fx.x))
    kind, nbuf = _lib._noise_args(fx.noise, n)
    lp = np.zeros(1)
    gy, gm = np.zeros(n), np.zeros(n)
    # dense Sigma_y (round 4): the gradient w.r.t. the matrix is the cotangent G = (alpha alpha' - C^-1) / 2 itself, N x N
    gn = np.zeros((n, n), order="F") if kind == _lib.NOISE_DENSE else np.zeros(n if kind == _lib.NOISE_DIAG else 1)
    nt = max(1, spec.n_terms)
    gc, gs = np.zeros(nt), np.zeros(nt)
    gx = None
    grs = None
    if scales:
        if inputs:
            gx = [np.zeros(np.asarray(a).shape, order="F") for a in spec.inputs]
            ptrs = (C.POINTER(C.c_double) * len(gx))(*[_lib.dptr(a) for a in gx])
        else:
            ptrs = None
        rl = np.asarray(spec.row_len)
        ncb = len(spec.col_len)
        grs = [None] * nt
        for p in range(len(rl) * ncb):
            for t in range(spec._term_ptr[p], spec._term_ptr[p + 1]):
                if spec.term_row_scale[t] is not None:
                    grs[t] = np.zeros(int(rl[p // ncb]))
        sptrs = (C.POINTER(C.c_double) * nt)(*[_lib.dptr(a) if a is not None else None for a in grs])
        rc = _ctx().lib.sgp_logpdf_grad_xs(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf),
                                           _lib.dptr(yv), _lib.dptr(lp), _lib.dptr(gy), _lib.dptr(gm), _lib.dptr(gn),
                                           _lib.dptr(gc), _lib.dptr(gs), ptrs, sptrs)
        _lib.check(rc, "sgp_logpdf_grad_xs")
    elif inputs:
        gx = [np.zeros(np.asarray(a).shape, order="F") for a in spec.inputs]
        ptrs = (C.POINTER(C.c_double) * len(gx))(*[_lib.dptr(a) for a in gx])
        rc = _ctx().lib.sgp_logpdf_grad_x(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf),
                                          _lib.dptr(yv), _lib.dptr(lp), _lib.dptr(gy), _lib.dptr(gm), _lib.dptr(gn),
                                          _lib.dptr(gc), _lib.dptr(gs), ptrs)
        _lib.check(rc, "sgp_logpdf_grad_x")
    else:
        rc = _ctx().lib.sgp_logpdf_grad(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf),
                                        _lib.dptr(yv), _lib.dptr(lp), _lib.dptr(gy), _lib.dptr(gm), _lib.dptr(gn),
                                        _lib.dptr(gc), _lib.dptr(gs))
        _lib.check(rc, "sgp_logpdf_grad")
    terms = _term_records(spec, gc, gs, True)
    # x: the same gradient mapped back through the model's input transformations onto the blocks of
    # fx.x (one (D, n) array per block; blocks sharing one input object get their joint gradient in
    # the first of them)
    xb = chain_input_gradients(spec, gx)[0] if inputs else None
    return dict(logpdf=float(lp[0]), y=gy, mean=gm, noise=(gn if kind != _lib.NOISE_SCALAR else float(gn[0])),
                terms=terms, inputs=gx, x=xb, scales=(_scale_records(spec, grs) if scales else None),
                _raw=(gc, gs), _rowscale=grs, _spec=spec)


def _draw(rng, n, s):
    """Z = randn(rng, n, s) in Julia's column-major fill order, from the caller's RNG."""
    if hasattr(rng, "standard_normal"):
        z = rng.standard_normal(n * s)
    else:
        z = rng.randn(n * s)
    return np.asfortranarray(np.asarray(z, dtype=np.float64).reshape((n, s), order="F"))


def rand(rng, fx, S=None, Z=None):
    """rand(rng, fx) -> vector; rand(rng, fx, S) -> N x S.  m .+ L Z with Z from `rng`
    (pass Z explicitly to reuse a caller-side draw)."""
    if isinstance(fx, SparseFiniteGP):
        return rand(rng, fx.fobs, S, Z)     # sparse_finite_gp.jl:47-50: samples from the dense fobs
    n = len(fx)
    s = 1 if S is None else int(S)
    if Z is None:
        Z = _draw(rng, n, s)
    Z = _f64(np.asarray(Z, dtype=np.float64).reshape(n, s))
    spec, m, kind, nbuf = _spec_mean_noise(fx)
    m = _f64(m)
    if _is_prior(fx.f) and _eltype(fx.x) == np.float32 and kind != _lib.NOISE_DENSE and spec.f32_supported():
        # Float32 model: the fp32 factor and an fp32 MFMA product L Z (sgp_rand_f32); `rand(rng, fx) isa Vector{Float32}`
        out32 = np.zeros((n, s), dtype=np.float32, order="F")
        rc = _ctx().lib.sgp_rand_f32(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf), _lib.dptr(Z), n, s,
                                     out32.ctypes.data_as(C.POINTER(C.c_float)), n)
        _lib.check(rc, "sgp_rand_f32")
        return out32[:, 0].copy() if S is None else out32
    out = np.zeros((n, s), order="F")
    rc = _ctx().lib.sgp_rand(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf), _lib.dptr(Z), n, s,
                             _lib.dptr(out), n)
    _lib.check(rc, "sgp_rand")
    if _eltype(fx.x) == np.float32:      # Float32 model: computed in fp64 on the device, returned in the model's type
        out = out.astype(np.float32)
    return out[:, 0].copy() if S is None else out


# ---- exact posterior ---------------------------------------------------------------------------
class PosteriorGP:
    """posterior(fx, y): keeps L and L^-1 (y - m) in HBM (sgp_post); data = (alpha, x, delta)."""

    def __init__(self, prior, x, handle, alpha, delta, noise=None, y=None, mean_x=None):
        self.prior, self.x, self._h, self._alpha, self.delta = prior, x, handle, alpha, delta
        self.noise, self.y = noise, y      # kept for sequential conditioning (posterior of a posterior)
        self._mean_x = mean_x

    def _ensure(self):
        """The fp64 factor in HBM.  A Float32 model (handle None at construction) answers mean / var from the fp32
        one-shot path (sgp_posterior_mean_var_f32) and only builds the fp64 factor when cov / alpha are asked for."""
        if self._h is None:
            n = len(self.y)
            spec = _prior_spec(self.prior, self.x)
            kind, nbuf = _lib._noise_args(self.noise, n)
            alpha = np.zeros(n)
            h = C.c_void_p()
            rc = _ctx().lib.sgp_posterior_create(_ctx().handle, spec.ref(), _lib.dptr(self._mean_x), kind, _lib.dptr(nbuf),
                                                 _lib.dptr(self.y), _lib.dptr(alpha), C.byref(h))
            _lib.check(rc, "sgp_posterior_create")
            self._h, self._alpha = h, alpha
        return self._h

    @property
    def alpha(self):
        self._ensure()
        return self._alpha

    def __del__(self):  # pragma: no cover
        try:
            if self._h:
                _lib.load().sgp_posterior_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __call__(self, xs, noise=1e-18):
        return FiniteGP(self, xs, noise)

    def _train_inputs(self):
        return (self.x,)

    def _predict(self, xs, want_mean, want_var, want_cov):
        cross, _, _ = build_spec(self.prior, xs, self.prior, self.x)
        pss = _prior_spec(self.prior, xs) if (want_var or want_cov) else None
        ns = cross.N
        ms = _f64(mean_vector(self.prior, xs))
        self._ensure()
        mo = np.zeros(ns) if want_mean else None
        vo = np.zeros(ns) if want_var else None
        co = np.zeros((ns, ns), order="F") if want_cov else None
        rc = _ctx().lib.sgp_posterior_predict(self._h, cross.ref(), pss.ref() if pss is not None else None,
                                              _lib.dptr(ms), _lib.dptr(mo), _lib.dptr(vo), _lib.dptr(co), max(ns, 1))
        _lib.check(rc, "sgp_posterior_predict")
        return mo, vo, co

    def mean(self, xs):
        return _model_type(self._predict(xs, True, False, False)[0], xs, *self._train_inputs())

    def var(self, xs):
        return _model_type(self._predict(xs, False, True, False)[1], xs, *self._train_inputs())

    def cov(self, xs, zs=None):
        if zs is None:
            return _model_type(self._predict(xs, False, False, True)[2], xs, *self._train_inputs())
        # cov(post, x*, z*) = K(x*, z*) - V_x*' V_z*: the off-diagonal block of the joint covariance
        joint = self._predict(BlockData([xs, zs]) if not isinstance(xs, BlockData) else _concat(xs, zs),
                              False, False, True)[2]
        nx = len(xs)
        return _model_type(joint[:nx, nx:], xs, zs, *self._train_inputs())

    def mean_and_var(self, xs):
        m, v, _ = self._predict(xs, True, True, False)
        return _model_type((m, v), xs, *self._train_inputs())

    def mean_and_cov(self, xs):
        m, _, c = self._predict(xs, True, False, True)
        return _model_type((m, c), xs, *self._train_inputs())


def _concat(xs, zs):
    zb = zs.X if isinstance(zs, BlockData) else [zs]
    return BlockData(list(xs.X) + list(zb))


def posterior(fx, y, y_vfe=None):
    """posterior(fx, y)  |  posterior(VFE(fz), fx, y)  |  posterior(SparseFiniteGP, y)"""
    if isinstance(fx, VFE):                  # posterior(VFE(fz), fx, y)  (sparse_finite_gp.jl:60-62)
        return posterior_vfe(fx, y, y_vfe)
    if isinstance(fx, SparseFiniteGP):
        return posterior_vfe(VFE(fx.finducing), fx.fobs, y)
    if isinstance(fx.f, PosteriorGP):
        # Sequential conditioning (AbstractGPs: posterior(f_post(x2, s2), y2) is again a PosteriorGP): the
        # posterior given (x1, y1) and then (x2, y2) IS the prior conditioned on the stacked data, which is
        # one factorisation of the joint covariance on the device instead of a Schur-complement update.
        p1 = fx.f
        if p1.noise is None:
            raise NotImplementedError("this posterior does not carry its observation model")
        n1, n2 = len(p1.x), len(fx)
        a1, a2 = np.asarray(p1.noise, dtype=np.float64), np.asarray(fx.noise, dtype=np.float64)
        if a1.ndim > 1 or a2.ndim > 1:
            # a dense Sigma_y on either side: the stacked observation noise is block diagonal (dense noise kind)
            noise = np.zeros((n1 + n2, n1 + n2))
            noise[:n1, :n1] = _noise_dense(a1, n1)
            noise[n1:, n1:] = _noise_dense(a2, n2)
        elif a1.ndim == 0 and a2.ndim == 0 and float(a1) == float(a2):
            noise = float(a1)
        else:
            noise = np.concatenate([_noise_diag(a1, n1), _noise_diag(a2, n2)])
        x1 = p1.x if isinstance(p1.x, BlockData) else BlockData([p1.x])
        xx = _concat(x1, fx.x)
        yy = np.concatenate([p1.y, np.asarray(y, dtype=np.float64).ravel()])
        return posterior(FiniteGP(p1.prior, xx, noise), yy)
    if not _is_prior(fx.f):
        # a process that is not a prior Stheno process but answers mean / var / cov -- the approximate (VFE) posterior, which the
        # reference returns as an ordinary AbstractGP (sparse_finite_gp.jl:60-62): conditioned through explicit covariances
        return ExplicitPosteriorGP(fx.f, fx.x, fx.noise, y)
    y = _f64(np.asarray(y, dtype=np.float64).ravel())
    n = len(fx)
    if y.shape[0] != n:
        raise ValueError("length(y) != length(fx)")
    m = _f64(mean_vector(fx.f, fx.x))
    post = PosteriorGP(fx.f, fx.x, None, None, y - m, fx.noise, y.copy(), mean_x=m)
    # Factor ONCE, here, and keep the handle -- for Float32 models as well (advisor, round 3: a Float32 posterior used to
    # re-assemble and re-factorise the whole covariance in fp32 on EVERY mean / var call, reported a covariance that is not
    # positive definite only at the first prediction, and switched to fp64-rounded values once `cov` / `alpha` had built
    # the fp64 factor, so a result depended on the call history).  Every prediction now runs against this one fp64
    # factor; Float32 models get their results rounded by the one output-type rule (_model_type).  The one-shot fp32
    # path (one fp32 factorisation with x* riding along as bordered rows) is the explicit posterior_mean_and_var_f32.
    post._ensure()                 # PosDefException here, as `cholesky` throws in the reference's posterior
    return post


def posterior_mean_and_var_f32(fx, y, xs):
    """mean_and_var(posterior(fx, y)(xs)) of a Float32 model in ONE fp32 factorisation (sgp_posterior_mean_var_f32: K(x*, x)
    rides through the fp32 Cholesky as bordered rows) -- the explicit fast path for "condition once, predict once"; nothing
    is kept.  fx: FiniteGP of a prior process with scalar / diagonal noise within the fp32 kernels' limits."""
    if not _is_prior(fx.f):
        raise NotImplementedError("the fp32 one-shot posterior takes a prior process")
    y = _f64(np.asarray(y, dtype=np.float64).ravel())
    spec = _prior_spec(fx.f, fx.x)
    cross, _, _ = build_spec(fx.f, xs, fx.f, fx.x)
    pss = _prior_spec(fx.f, xs)
    kind, nbuf = _lib._noise_args(fx.noise, len(y))
    if kind == _lib.NOISE_DENSE or not spec.f32_supported() or not cross.f32_supported():
        raise NotImplementedError("beyond the fp32 kernels' limits (dense noise, input dimension > 16, too many terms)")
    ns = cross.N
    m, ms = _f64(mean_vector(fx.f, fx.x)), _f64(mean_vector(fx.f, xs))
    mo32, vo32 = np.zeros(ns, dtype=np.float32), np.zeros(ns, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rc = _ctx().lib.sgp_posterior_mean_var_f32(_ctx().handle, spec.ref(), _lib.dptr(m), kind, _lib.dptr(nbuf), _lib.dptr(y),
                                               cross.ref(), pss.ref(), _lib.dptr(ms), fp(mo32), fp(vo32))
    _lib.check(rc, "sgp_posterior_mean_var_f32")
    return mo32, vo32


# ---- VFE / sparse --------------------------------------------------------------------------------
class VFE:
    def __init__(self, fz):
        self.fz = fz


class SparseFiniteGP:
    """SparseFiniteGP(fobs, finducing): logpdf == elbo, posterior == VFE posterior."""

    def __init__(self, fobs, finducing):
        self.fobs, self.finducing = fobs, finducing

    def __len__(self):
        return len(self.fobs)


_COV_ERR = ("The covariance matrix of a sparse GP can often be dense and can cause the computer to run out of "
            "memory. If you are sure you have enough memory, you can use `cov(f.fobs)`.")


def sparse_cov(f):
    raise RuntimeError(_COV_ERR)      # sparse_finite_gp.jl:39-43


def _vfe_args(vfe, fx):
    fz = vfe.fz
    if fz.f is not fx.f:
        raise AssertionError("VFE requires fz.f === fx.f")
    if not _is_prior(fx.f):
        raise NotImplementedError("VFE needs a prior Stheno process")
    n, m = len(fx), len(fz)
    zz = _prior_spec(fz.f, fz.x)
    xz, _, _ = build_spec(fx.f, fx.x, fz.f, fz.x)
    a = np.asarray(fx.noise, dtype=np.float64)
    if a.ndim > 1:
        raise ValueError("elbo needs isotropic or diagonal observation noise")
    nk, nbuf = _lib._noise_args(fx.noise, n)
    zk, zbuf = _lib._noise_args(fz.noise, m)
    mean_x = _f64(mean_vector(fx.f, fx.x))
    return zz, xz, mean_x, nk, nbuf, zk, zbuf


def elbo(vfe, fx, y=None):
    if isinstance(vfe, SparseFiniteGP):      # elbo(f::SparseFiniteGP, y)
        return elbo(VFE(vfe.finducing), vfe.fobs, fx)
    zz, xz, mean_x, nk, nbuf, zk, zbuf = _vfe_args(vfe, fx)
    y = _f64(np.asarray(y, dtype=np.float64).ravel())
    var_x = _f64(prior_var(fx.f, fx.x))
    out = np.zeros(1)
    rc = _ctx().lib.sgp_elbo(_ctx().handle, zz.ref(), xz.ref(), _lib.dptr(var_x), _lib.dptr(mean_x), nk,
                             _lib.dptr(nbuf), zk, _lib.dptr(zbuf), _lib.dptr(y), _lib.dptr(out))
    _lib.check(rc, "sgp_elbo")
    return float(out[0])


def _term_records(spec, gc, gs, symmetric):
    """One record per flattened term.  For a symmetric spec the mirror-image pair (J, I) is folded
    into the lower pair (I >= J).  The mirror of a term with row / column scale vectors (rs, cs) is
    the term of (J, I) with the inputs AND the scales swapped -- several terms of one block pair may
    differ only in their scales (f3 = f1 + sin * f1 over two blocks), so the scale identities are
    part of the match."""
    nrb, ncb = len(spec.row_len), len(spec.col_len)
    tp, out, index = spec._term_ptr, [], {}
    sid = spec.term_scale_ids
    for I in range(nrb):
        for J in range(ncb):
            for t in range(tp[I * ncb + J], tp[I * ncb + J + 1]):
                T = spec._terms[t]
                if symmetric and I < J:
                    continue
                index[(I, J, T.kind, T.row_input, T.col_input, T.param, sid[t][0], sid[t][1])] = len(out)
                out.append(dict(I=I, J=J, kind=T.kind, coef=T.coef, row_input=T.row_input, col_input=T.col_input,
                                row_scaled=sid[t][0] is not None, col_scaled=sid[t][1] is not None,
                                t=t, mirror_t=None, d_coef=float(gc[t]), d_inscale=float(gs[t])))
    if symmetric:
        for I in range(nrb):
            for J in range(I + 1, ncb):
                for t in range(tp[I * ncb + J], tp[I * ncb + J + 1]):
                    T = spec._terms[t]
                    k = index.get((J, I, T.kind, T.col_input, T.row_input, T.param, sid[t][1], sid[t][0]))
                    if k is None:
                        raise AssertionError("symmetric spec without a mirror term: flattener invariant broken")
                    out[k]["mirror_t"] = t
                    out[k]["d_coef"] += float(gc[t])
                    out[k]["d_inscale"] += float(gs[t])
    return out


def _scale_arrays(spec, side):
    """one zero array per term that carries a row ("row") / column ("col") scale vector, None elsewhere + the ctypes
    pointer table"""
    nt = max(1, spec.n_terms)
    vecs = spec.term_row_scale if side == "row" else spec.term_col_scale
    arrs = [None] * nt
    for t in range(spec.n_terms):
        if vecs[t] is not None:
            arrs[t] = np.zeros(len(vecs[t]))
    return arrs, (C.POINTER(C.c_double) * nt)(*[_lib.dptr(a) if a is not None else None for a in arrs])


def elbo_and_gradient(vfe, fx, y=None, inputs=False, scales=False):
    """elbo(VFE(fz), fx, y) and its reverse-mode gradient (what Zygote derives through
    AbstractGPs.elbo on the reference path; sgp_elbo_grad).

    Returns a dict: elbo; y, mean (N each); noise (scalar or N); z_noise (scalar, M, or M x M for a dense Sigma_z:
    d/d Sigma_z);
    zz_terms / xz_terms / xx_terms: per flattened covariance term of K(z,z), K(x,z) and diag K(x,x)
    {I, J, kind, coef, row_input, col_input, d_coef, d_inscale} (see logpdf_and_gradient).
    inputs=True adds zz_inputs / xz_inputs: d elbo / d (input points) per entry of
    g["_specs"]["zz"].inputs and g["_specs"]["xz"].inputs (the inducing points appear in both).
    scales=True adds `scales`: one record per function-valued scale (`sigma * f`, product.jl:25-48) and input collection
    it is mapped over -- {node, x, values, d_values} with d_values[i] = d elbo / d sigma(x_i), summed over the three
    places the bound reads sigma: K(z,z), K(x,z) (row side at x, column side at z) and diag K(x,x)."""
    if isinstance(vfe, SparseFiniteGP):
        return elbo_and_gradient(VFE(vfe.finducing), vfe.fobs, fx, inputs=inputs, scales=scales)
    zz, xz, mean_x, nk, nbuf, zk, zbuf = _vfe_args(vfe, fx)
    n, m = len(fx), len(vfe.fz)
    yv = _f64(np.asarray(y, dtype=np.float64).ravel())
    xx = _prior_spec(fx.f, fx.x)
    var_x = _f64(_kernelmatrix_diag(xx))
    out = np.zeros(1)
    gy, gm, gv = np.zeros(n), np.zeros(n), np.zeros(n)
    gn = np.zeros(n if nk == _lib.NOISE_DIAG else 1)
    gzn = np.zeros((m, m), order="F") if zk == _lib.NOISE_DENSE else np.zeros(m if zk == _lib.NOISE_DIAG else 1)
    gcz, gsz = np.zeros(max(1, zz.n_terms)), np.zeros(max(1, zz.n_terms))
    gcx, gsx = np.zeros(max(1, xz.n_terms)), np.zeros(max(1, xz.n_terms))
    lib = _ctx().lib
    gxz = gxx = None
    args = (_ctx().handle, zz.ref(), xz.ref(), _lib.dptr(var_x), _lib.dptr(mean_x), nk, _lib.dptr(nbuf), zk,
            _lib.dptr(zbuf), _lib.dptr(yv), _lib.dptr(out), _lib.dptr(gy), _lib.dptr(gm), _lib.dptr(gn), _lib.dptr(gv),
            _lib.dptr(gzn), _lib.dptr(gcz), _lib.dptr(gsz), _lib.dptr(gcx), _lib.dptr(gsx))
    pz = px = None
    if inputs:
        gxz = [np.zeros(a.shape, order="F") for a in zz.inputs]
        gxx = [np.zeros(a.shape, order="F") for a in xz.inputs]
        pz = (C.POINTER(C.c_double) * max(1, len(gxz)))(*[_lib.dptr(a) for a in gxz])
        px = (C.POINTER(C.c_double) * max(1, len(gxx)))(*[_lib.dptr(a) for a in gxx])
    srz = srx = scx = None
    if scales:
        srz, prz = _scale_arrays(zz, "row")
        srx, prx = _scale_arrays(xz, "row")
        scx, pcx = _scale_arrays(xz, "col")
        _lib.check(lib.sgp_elbo_grad_xs(*args, pz, px, prz, prx, pcx), "sgp_elbo_grad_xs")
    elif inputs:
        _lib.check(lib.sgp_elbo_grad_x(*args, pz, px), "sgp_elbo_grad_x")
    else:
        _lib.check(lib.sgp_elbo_grad(*args), "sgp_elbo_grad")
    gcd, gsd = np.zeros(max(1, xx.n_terms)), np.zeros(max(1, xx.n_terms))
    gdx = pd = None
    if inputs:      # var(f, x) depends on x wherever a diagonal term reads two different views of x
        gdx = [np.zeros(a.shape, order="F") for a in xx.inputs]
        pd = (C.POINTER(C.c_double) * max(1, len(gdx)))(*[_lib.dptr(a) for a in gdx])
    sdr = sdc = None
    if scales:      # ... and on sigma(x): var_i = sum_t coef rs_i cs_i k_t
        sdr, pdr = _scale_arrays(xx, "row")
        sdc, pdc = _scale_arrays(xx, "col")
        rc = lib.sgp_kernelmatrix_diag_grad_xs(_ctx().handle, xx.ref(), _lib.dptr(gv), _lib.dptr(gcd), _lib.dptr(gsd), pd,
                                               pdr, pdc)
    elif inputs:
        rc = lib.sgp_kernelmatrix_diag_grad_x(_ctx().handle, xx.ref(), _lib.dptr(gv), _lib.dptr(gcd), _lib.dptr(gsd), pd)
    else:
        rc = lib.sgp_kernelmatrix_diag_grad(_ctx().handle, xx.ref(), _lib.dptr(gv), _lib.dptr(gcd), _lib.dptr(gsd))
    _lib.check(rc, "sgp_kernelmatrix_diag_grad")
    nb = len(xx.row_len)
    xx_terms = [r for r in _term_records(xx, gcd, gsd, False) if r["I"] == r["J"]] if nb else []
    xb = zb = None
    if inputs:   # chain rule back onto the blocks of fx.x and fz.x
        zr, _ = chain_input_gradients(zz, gxz)
        xr, zc = chain_input_gradients(xz, gxx)
        xd, _ = chain_input_gradients(xx, gdx)
        xb, zb = [a + b for a, b in zip(xr, xd)], [a + b for a, b in zip(zr, zc)]
    return dict(elbo=float(out[0]), y=gy, mean=gm, noise=(gn if nk == _lib.NOISE_DIAG else float(gn[0])), x=xb, z=zb,
                z_noise=(gzn if zk != _lib.NOISE_SCALAR else float(gzn[0])), var=gv,
                zz_terms=_term_records(zz, gcz, gsz, True), xz_terms=_term_records(xz, gcx, gsx, False),
                xx_terms=xx_terms, zz_inputs=gxz, xz_inputs=gxx,
                scales=(_scale_records(zz, srz, more=((xz, srx, "row"), (xz, scx, "col"), (xx, sdr, "row"), (xx, sdc, "col")))
                        if scales else None),
                _raw=dict(zz=(gcz, gsz), xz=(gcx, gsx), xx=(gcd, gsd)),
                _specs=dict(zz=zz, xz=xz, xx=xx))


class ApproxPosteriorGP:
    def __init__(self, prior, z, handle):
        self.prior, self.z, self._h = prior, z, handle

    def __del__(self):  # pragma: no cover
        try:
            if self._h:
                _lib.load().sgp_sparse_posterior_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __call__(self, xs, noise=1e-18):
        return FiniteGP(self, xs, noise)

    def _train_inputs(self):
        return (self.z,)

    def _predict(self, xs, want_mean, want_var, want_cov):
        cross, _, _ = build_spec(self.prior, xs, self.prior, self.z)
        pss = _prior_spec(self.prior, xs) if (want_var or want_cov) else None
        ns = cross.N
        ms = _f64(mean_vector(self.prior, xs))
        mo = np.zeros(ns) if want_mean else None
        vo = np.zeros(ns) if want_var else None
        co = np.zeros((ns, ns), order="F") if want_cov else None
        rc = _ctx().lib.sgp_sparse_posterior_predict(self._h, cross.ref(), pss.ref() if pss is not None else None,
                                                     _lib.dptr(ms), _lib.dptr(mo), _lib.dptr(vo), _lib.dptr(co),
                                                     max(ns, 1))
        _lib.check(rc, "sgp_sparse_posterior_predict")
        return mo, vo, co

    def mean(self, xs):
        return _model_type(self._predict(xs, True, False, False)[0], xs, *self._train_inputs())

    def var(self, xs):
        return _model_type(self._predict(xs, False, True, False)[1], xs, *self._train_inputs())

    def cov(self, xs, zs=None):
        if zs is None:
            return _model_type(self._predict(xs, False, False, True)[2], xs, *self._train_inputs())
        # cov(post, x*, z*): the off-diagonal block of the joint covariance over [x*; z*] (as PosteriorGP.cov)
        joint = self._predict(BlockData([xs, zs]) if not isinstance(xs, BlockData) else _concat(xs, zs),
                              False, False, True)[2]
        nx = len(xs)
        return _model_type(joint[:nx, nx:], xs, zs, *self._train_inputs())

    def mean_and_var(self, xs):
        m, v, _ = self._predict(xs, True, True, False)
        return _model_type((m, v), xs, *self._train_inputs())

    def mean_and_cov(self, xs):
        m, _, c = self._predict(xs, True, False, True)
        return _model_type((m, c), xs, *self._train_inputs())


class ExplicitPosteriorGP:
    """posterior(g(x2, S2), y2) for a process g whose covariance is not a sum of kernel terms (round 5): g answers
    mean / var / cov itself (the approximate posterior does, on the device); the factor of cov(g, x2) + S2 is built by
    sgp_posterior_create on the zero-term spec with dense noise, predictions go through sgp_posterior_predict_explicit with
    cov(g, x*, x2), var / cov(g, x*) and mean(g, x*) evaluated by g.  AbstractGPs [EXT]: posterior(fx::FiniteGP, y) for any
    AbstractGP; reference: test/gp/sparse_finite_gp.jl (the VFE posterior used as a GP)."""

    def __init__(self, base, x, noise, y):
        n = len(x)
        y = _f64(np.asarray(y, dtype=np.float64).ravel())
        if y.shape[0] != n:
            raise ValueError("length(y) != length(fx)")
        self.base, self.x, self.noise, self.y = base, x, noise, y
        m, Cm = base.mean_and_cov(x)
        self._mean_x = _f64(np.asarray(m, dtype=np.float64))
        Cm = np.asfortranarray(np.asarray(Cm, dtype=np.float64) + _noise_dense(noise, n))
        self._spec = zero_spec(n)
        alpha = np.zeros(n)
        h = C.c_void_p()
        rc = _ctx().lib.sgp_posterior_create(_ctx().handle, self._spec.ref(), _lib.dptr(self._mean_x), _lib.NOISE_DENSE,
                                             _lib.dptr(Cm), _lib.dptr(y), _lib.dptr(alpha), C.byref(h))
        _lib.check(rc, "sgp_posterior_create")
        self._h, self.alpha, self.delta = h, alpha, y - self._mean_x

    def __del__(self):  # pragma: no cover
        try:
            if self._h:
                _lib.load().sgp_posterior_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __call__(self, xs, noise=1e-18):
        return FiniteGP(self, xs, noise)

    def _train_inputs(self):
        return (self.x,)

    def _predict(self, xs, want_mean, want_var, want_cov):
        ns = len(xs)
        cross = np.asfortranarray(np.asarray(self.base.cov(xs, self.x), dtype=np.float64))       # ns x N
        ms = _f64(np.asarray(self.base.mean(xs), dtype=np.float64))
        pv = _f64(np.asarray(self.base.var(xs), dtype=np.float64)) if want_var else None
        pc = np.asfortranarray(np.asarray(self.base.cov(xs), dtype=np.float64)) if want_cov else None
        mo = np.zeros(ns) if want_mean else None
        vo = np.zeros(ns) if want_var else None
        co = np.zeros((ns, ns), order="F") if want_cov else None
        rc = _ctx().lib.sgp_posterior_predict_explicit(self._h, _lib.dptr(cross), max(ns, 1), ns, _lib.dptr(pv), _lib.dptr(pc),
                                                       max(ns, 1), _lib.dptr(ms), _lib.dptr(mo), _lib.dptr(vo), _lib.dptr(co),
                                                       max(ns, 1))
        _lib.check(rc, "sgp_posterior_predict_explicit")
        return mo, vo, co

    def mean(self, xs):
        return self._predict(xs, True, False, False)[0]

    def var(self, xs):
        return self._predict(xs, False, True, False)[1]

    def cov(self, xs, zs=None):
        if zs is None:
            return self._predict(xs, False, False, True)[2]
        joint = self._predict(BlockData([xs, zs]) if not isinstance(xs, BlockData) else _concat(xs, zs), False, False, True)[2]
        nx = len(xs)
        return joint[:nx, nx:]

    def mean_and_var(self, xs):
        m, v, _ = self._predict(xs, True, True, False)
        return m, v

    def mean_and_cov(self, xs):
        m, _, c = self._predict(xs, True, False, True)
        return m, c


def posterior_vfe(vfe, fx, y):
    zz, xz, mean_x, nk, nbuf, zk, zbuf = _vfe_args(vfe, fx)
    y = _f64(np.asarray(y, dtype=np.float64).ravel())
    h = C.c_void_p()
    rc = _ctx().lib.sgp_sparse_posterior_create(_ctx().handle, zz.ref(), xz.ref(), _lib.dptr(mean_x), nk,
                                                _lib.dptr(nbuf), zk, _lib.dptr(zbuf), _lib.dptr(y), C.byref(h))
    _lib.check(rc, "sgp_sparse_posterior_create")
    return ApproxPosteriorGP(fx.f, vfe.fz.x, h)
