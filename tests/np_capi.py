"""NumPy double of the C-ABI of include/sthenomi.h, for the CPU suite.

TEST DOUBLE ONLY.  The product has no CPU path (tests/test_capi_symbols.py::test_no_cpu_fallback) and never imports
this file.  What it is for: the host mirror (stheno.jl_amd/finite_gp.py, flatten.py, kernels.py, gp.py) is Python
logic -- flattening a Stheno programme into a covariance spec, marshalling arguments, mapping the library's per-term
outputs back onto the model (gradient records, mirror terms, scale records, input-gradient chain rule), dispatching
SparseFiniteGP / sequential conditioning / Float32 tagging.  On the GPU box that logic is tested end to end against
the oracle through libsthenomi.so; with `install()` the same test bodies run on a machine without a GPU, the
library replaced by the dense NumPy / SciPy statement below of what each entry point is documented to compute
(every function cites its declaration).  It says nothing about the HIP kernels.

Handles: a created posterior is kept in a registry keyed by the identity of the caller's c_void_p, whose value stays
NULL -- so that a `__del__` running after the double is gone never hands a fake pointer to the real library.
"""
import ctypes as C

import numpy as np
import scipy.linalg as sla

from stheno_jl_amd import lib as L

LOG2PI = 1.8378770664093453


def _struct(ref):
    return ref._obj if hasattr(ref, "_obj") else ref.contents


def _vec(ptr, n):
    """numpy view of n doubles behind a ctypes pointer (None for NULL)"""
    if ptr is None or not ptr or n == 0:
        return None if (ptr is None or not ptr) else np.zeros(0)
    return np.ctypeslib.as_array(ptr, shape=(int(n),))


def _mat(ptr, nr, nc, ld):
    """column-major nr x nc view with leading dimension ld"""
    if ptr is None or not ptr:
        return None
    flat = np.ctypeslib.as_array(ptr, shape=(int(ld) * int(nc),))
    return flat.reshape((int(nc), int(ld)))[:, : int(nr)].T


def _kern(kind, d2, param):
    d = np.sqrt(d2)
    if kind == L.SE:
        return np.exp(-0.5 * d2)
    if kind == L.MATERN12:
        return np.exp(-d)
    if kind == L.MATERN32:
        return (1.0 + np.sqrt(3.0) * d) * np.exp(-np.sqrt(3.0) * d)
    if kind == L.MATERN52:
        return (1.0 + np.sqrt(5.0) * d + 5.0 * d2 / 3.0) * np.exp(-np.sqrt(5.0) * d)
    if kind == L.WHITE:
        return (d2 == 0.0).astype(np.float64)
    if kind == L.CONST:
        return np.full_like(d2, param)
    raise ValueError(kind)


def _dkern_dd2(kind, d2):
    """kappa'(d2) = d k / d (d^2); 0 where the kernel is not differentiable (Matern-1/2 at d = 0) or constant"""
    d = np.sqrt(d2)
    if kind == L.SE:
        return -0.5 * np.exp(-0.5 * d2)
    if kind == L.MATERN12:
        with np.errstate(divide="ignore", invalid="ignore"):
            out = np.where(d > 0, -np.exp(-d) / (2.0 * d), 0.0)
        return out
    if kind == L.MATERN32:
        return -1.5 * np.exp(-np.sqrt(3.0) * d)
    if kind == L.MATERN52:
        return -(5.0 / 6.0) * (1.0 + np.sqrt(5.0) * d) * np.exp(-np.sqrt(5.0) * d)
    return np.zeros_like(d2)


class _Spec:
    """what a sgp_cov_spec says (include/sthenomi.h:57-131), read back from the raw ctypes struct"""

    def __init__(self, ref):
        c = _struct(ref)
        self.nrb, self.ncb, self.symmetric = c.n_row_blocks, c.n_col_blocks, bool(c.symmetric)
        self.row_len = [int(c.row_len[i]) for i in range(self.nrb)]
        self.col_len = [int(c.col_len[j]) for j in range(self.ncb)]
        self.N, self.M = sum(self.row_len), sum(self.col_len)
        self.roff = np.concatenate([[0], np.cumsum(self.row_len)]).astype(int)
        self.coff = np.concatenate([[0], np.cumsum(self.col_len)]).astype(int)
        self.inputs = []
        for k in range(c.n_inputs):
            inp = c.inputs[k]
            dim, n, ld = int(inp.dim), int(inp.n), int(inp.ld)
            if n == 0:
                self.inputs.append(np.zeros((dim, 0)))
                continue
            flat = np.ctypeslib.as_array(inp.x, shape=(ld * n,))
            self.inputs.append(flat.reshape((n, ld))[:, :dim].T.copy())
        self.terms = []
        for I in range(self.nrb):
            for J in range(self.ncb):
                p = I * self.ncb + J
                for t in range(c.term_ptr[p], c.term_ptr[p + 1]):
                    T = c.terms[t]
                    rs = _vec(T.row_scale, self.row_len[I])
                    cs = _vec(T.col_scale, self.col_len[J])
                    self.terms.append((I, J, int(T.kind), int(T.row_input), int(T.col_input), float(T.coef), float(T.param),
                                       None if rs is None else rs.copy(), None if cs is None else cs.copy()))

    def d2(self, t):
        (_, _, _, ri, ci, _, _, _, _) = self.terms[t]
        X, Y = self.inputs[ri], self.inputs[ci]
        if X.shape[0] != Y.shape[0]:
            raise ValueError("spec: row / col input dimension mismatch")
        return ((X[:, :, None] - Y[:, None, :]) ** 2).sum(0)

    def block(self, t):
        """(rows slice, cols slice) of term t and its unscaled kernel block k and scaled weight pieces"""
        (I, J, kind, _, _, coef, param, rs, cs) = self.terms[t]
        return slice(self.roff[I], self.roff[I + 1]), slice(self.coff[J], self.coff[J + 1])

    def dense(self):
        K = np.zeros((self.N, self.M))
        for t, (I, J, kind, ri, ci, coef, param, rs, cs) in enumerate(self.terms):
            r, c = self.block(t)
            blk = coef * _kern(kind, self.d2(t), param)
            if rs is not None:
                blk = rs[:, None] * blk
            if cs is not None:
                blk = blk * cs[None, :]
            K[r, c] += blk
        if self.symmetric:          # the device assembles the lower triangle and mirrors it: exactly symmetric
            K = np.tril(K) + np.tril(K, -1).T
        return K


def _noise_matrix(kind, noise, n):
    if kind == L.NOISE_SCALAR:
        return float(noise[0]) * np.eye(n)
    if kind == L.NOISE_DIAG:
        return np.diag(np.ctypeslib.as_array(noise, shape=(n,)).copy())
    return np.ctypeslib.as_array(noise, shape=(n * n,)).reshape((n, n)).T.copy()     # dense, column-major


def _chol(Cm):
    """lower Cholesky; (L, 0) or (None, info) with LAPACK potrf's info (first failing leading minor)"""
    Lm, info = sla.lapack.dpotrf(np.asfortranarray(Cm), lower=1, clean=1)
    return (np.asarray(Lm), 0) if info == 0 else (None, int(info))


class FakeLib:
    """one method per C-ABI entry point the host mirror calls; same argument order as include/sthenomi.h"""

    def __init__(self):
        self.posts, self.sposts = {}, {}
        self.err = b""

    # -- errors --------------------------------------------------------------------------------------
    def sgp_last_error(self):
        return self.err

    def _fail(self, msg, rc=-1):
        self.err = msg.encode()
        return rc

    # -- covariance (sthenomi.h:133-137) ---------------------------------------------------------------
    def sgp_kernelmatrix(self, ctx, spec, K, ldk):
        s = _Spec(spec)
        _mat(K, s.N, s.M, ldk)[:, :] = s.dense()
        return 0

    def sgp_kernelmatrix_diag(self, ctx, spec, out):
        s = _Spec(spec)
        _vec(out, s.N)[:] = np.diag(s.dense())
        return 0

    # -- fp32 instantiation (sthenomi.h:145-152): the double computes in fp64 and rounds the result, which is inside
    #    every fp32 tolerance; what it serves is the host mirror's Float32 tagging / type stability ------------------
    def _f32_limits(self, s):
        """csrc/f32.hip: assemble_f32 -- input dimension <= 16 and (terms per block pair) x (dimension rounded up to a
        power of two) <= 64, else rc = -1 (the host mirror must not send such a model down the fp32 path)"""
        per_pair = {}
        for (I, J, _, ri, _, _, _, _, _) in s.terms:
            d = s.inputs[ri].shape[0]
            dmax = 1
            while dmax < d:
                dmax *= 2
            n, dm = per_pair.get((I, J), (0, 1))
            per_pair[(I, J)] = (n + 1, max(dm, dmax))
        for (n, dm) in per_pair.values():
            if dm > 16 or n * dm > 64:
                return self._fail("fp32 path: input dimension <= 16 and (terms per block pair) x dimension <= 64")
        return 0

    def sgp_kernelmatrix_f32(self, ctx, spec, K, ldk):
        s = _Spec(spec)
        rc = self._f32_limits(s)
        if rc:
            return rc
        flat = np.ctypeslib.as_array(K, shape=(int(ldk) * s.M,))
        flat.reshape((s.M, int(ldk)))[:, : s.N].T[:, :] = s.dense().astype(np.float32)
        return 0

    def sgp_logpdf_f32(self, ctx, spec, mean, kind, noise, y, out):
        s = _Spec(spec)
        rc = self._f32_limits(s)
        if rc:
            return rc
        return self.sgp_logpdf(ctx, spec, mean, kind, noise, y, s.N, 1, out)

    def sgp_rand_f32(self, ctx, spec, mean, kind, noise, Z, ldz, S, out, ldo):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        rc = rc or self._f32_limits(s)
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        flat = np.ctypeslib.as_array(out, shape=(int(ldo) * int(S),))
        flat.reshape((int(S), int(ldo)))[:, : s.N].T[:, :] = (m[:, None] + Lm @ _mat(Z, s.N, S, ldz)).astype(np.float32)
        return 0

    def sgp_posterior_mean_var_f32(self, ctx, spec, mean, kind, noise, y, cross, prior_ss, mean_s, mean_out, var_out):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        rc = rc or self._f32_limits(s) or self._f32_limits(_Spec(cross))
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        Kx = _Spec(cross).dense()
        ns = Kx.shape[0]
        ms = np.zeros(ns) if not mean_s else _vec(mean_s, ns)
        z = sla.solve_triangular(Lm, _vec(y, s.N) - m, lower=True, check_finite=False)
        V = sla.solve_triangular(Lm, Kx.T, lower=True, check_finite=False)
        if mean_out:
            np.ctypeslib.as_array(mean_out, shape=(ns,))[:] = (ms + V.T @ z).astype(np.float32)
        if var_out:
            np.ctypeslib.as_array(var_out, shape=(ns,))[:] = (np.diag(_Spec(prior_ss).dense()) - (V * V).sum(0)).astype(np.float32)
        return 0

    # -- the observation model C = K + Sigma_y ---------------------------------------------------------
    def _observed(self, spec, mean, kind, noise):
        s = _Spec(spec)
        if not s.symmetric:
            return None, None, None, self._fail("spec must be symmetric")
        m = np.zeros(s.N) if not mean else _vec(mean, s.N).copy()
        Cm = s.dense() + _noise_matrix(kind, noise, s.N)
        return s, m, Cm, 0

    # -- logpdf (sthenomi.h:139-143) -------------------------------------------------------------------
    def sgp_logpdf(self, ctx, spec, mean, kind, noise, Y, ldy, ncols, out):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        Ym = _mat(Y, s.N, ncols, ldy)
        Z = sla.solve_triangular(Lm, Ym - m[:, None], lower=True, check_finite=False)
        _vec(out, ncols)[:] = -0.5 * (s.N * LOG2PI + 2.0 * np.log(np.diag(Lm)).sum() + (Z * Z).sum(0))
        return 0

    # -- rand (sthenomi.h:194-198) -----------------------------------------------------------------------
    def sgp_logpdf_batch(self, ctx, nspec, specs, means, kind, noises, ys, out, infos):
        """include/sthenomi.h: member b = (specs[b], means[b], noises[b], ys[b]); out[b] its logpdf, NaN + infos[b] for a member
        that is not positive definite (rc stays 0 when infos is given)."""
        o = _vec(out, nspec)
        inf = np.ctypeslib.as_array(infos, shape=(nspec,)) if infos else None
        first_bad = 0
        for b in range(nspec):
            one = np.zeros(1)
            n = _Spec(specs[b]).N
            rc = self.sgp_logpdf(ctx, specs[b], means[b], kind, noises[b], ys[b], n, 1, one.ctypes.data_as(C.POINTER(C.c_double)))
            if rc < 0:
                return rc
            o[b] = one[0] if rc == 0 else np.nan
            if inf is not None:
                inf[b] = rc
            if rc > 0 and not first_bad:
                first_bad = rc
        return 0 if inf is not None else first_bad

    def sgp_rand(self, ctx, spec, mean, kind, noise, Z, ldz, S, out, ldo):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        _mat(out, s.N, S, ldo)[:, :] = m[:, None] + Lm @ _mat(Z, s.N, S, ldz)
        return 0

    # -- exact posterior (sthenomi.h:200-214) --------------------------------------------------------------
    def sgp_posterior_create(self, ctx, spec, mean, kind, noise, y, alpha_out, out):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        alpha = sla.cho_solve((Lm, True), _vec(y, s.N) - m, check_finite=False)
        if alpha_out:
            _vec(alpha_out, s.N)[:] = alpha
        self.posts[id(_struct(out))] = (Lm, alpha)
        return 0

    def sgp_posterior_predict(self, post, cross, prior_ss, mean_s, mean_out, var_out, cov_out, ldcov):
        Lm, alpha = self.posts[id(post)]
        Kx = _Spec(cross).dense()                      # Ns x N
        ns = Kx.shape[0]
        ms = np.zeros(ns) if not mean_s else _vec(mean_s, ns)
        if mean_out:
            _vec(mean_out, ns)[:] = ms + Kx @ alpha
        if var_out or cov_out:
            V = sla.solve_triangular(Lm, Kx.T, lower=True, check_finite=False)
            Kss = _Spec(prior_ss).dense()
            if var_out:
                _vec(var_out, ns)[:] = np.diag(Kss) - (V * V).sum(0)
            if cov_out:
                _mat(cov_out, ns, ns, ldcov)[:, :] = Kss - V.T @ V
        return 0

    def sgp_posterior_predict_explicit(self, post, cross, ldc, ns, prior_var, prior_cov, ldp, mean_s, mean_out, var_out, cov_out, ldcov):
        Lm, alpha = self.posts[id(post)]
        n = Lm.shape[0]
        Kx = _mat(cross, ns, n, ldc)                   # ns x N, given
        ms = np.zeros(ns) if not mean_s else _vec(mean_s, ns)
        if mean_out:
            _vec(mean_out, ns)[:] = ms + Kx @ alpha
        if var_out or cov_out:
            V = sla.solve_triangular(Lm, Kx.T, lower=True, check_finite=False)
            if var_out:
                _vec(var_out, ns)[:] = _vec(prior_var, ns) - (V * V).sum(0)
            if cov_out:
                _mat(cov_out, ns, ns, ldcov)[:, :] = _mat(prior_cov, ns, ns, ldp) - V.T @ V
        return 0

    def sgp_posterior_destroy(self, post):
        return 0

    # -- VFE (App. A.6; sthenomi.h: sgp_elbo, sgp_sparse_posterior_*) ---------------------------------------
    def _vfe_parts(self, zz, xz, mean_x, nk, noise_x, zk, z_noise, y):
        sz, sx = _Spec(zz), _Spec(xz)
        if nk == L.NOISE_DENSE:
            return None, self._fail("vfe: Sigma_y must be isotropic or diagonal")
        M, N = sz.N, sx.N
        Kzz = sz.dense() + _noise_matrix(zk, z_noise, M)
        Lz, info = _chol(Kzz)
        if info:
            return None, self._fail("matrix is not positive definite", info)
        sy = np.full(N, float(noise_x[0])) if nk == L.NOISE_SCALAR else _vec(noise_x, N).copy()
        Kxz = sx.dense()                                           # N x M
        A = sla.solve_triangular(Lz, Kxz.T, lower=True, check_finite=False) / np.sqrt(sy)[None, :]
        Le, info = _chol(A @ A.T + np.eye(M))
        if info:
            return None, self._fail("matrix is not positive definite", info)
        m = np.zeros(N) if not mean_x else _vec(mean_x, N)
        delta = (_vec(y, N) - m) / np.sqrt(sy)
        return (Lz, A, Le, delta, sy), 0

    def sgp_elbo(self, ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, out):
        parts, rc = self._vfe_parts(zz, xz, mean_x, nk, noise_x, zk, z_noise, y)
        if rc:
            return rc
        Lz, A, Le, delta, sy = parts
        n = len(delta)
        b = sla.solve_triangular(Le, A @ delta, lower=True, check_finite=False)
        tmp = np.log(sy).sum() + 2.0 * np.log(np.diag(Le)).sum() + delta @ delta - b @ b
        out[0] = -0.5 * (n * LOG2PI + tmp) - 0.5 * ((_vec(var_x, n) / sy).sum() - (A * A).sum())
        return 0

    def sgp_sparse_posterior_create(self, ctx, zz, xz, mean_x, nk, noise_x, zk, z_noise, y, out):
        parts, rc = self._vfe_parts(zz, xz, mean_x, nk, noise_x, zk, z_noise, y)
        if rc:
            return rc
        Lz, A, Le, delta, _ = parts
        m_eps = sla.cho_solve((Le, True), A @ delta, check_finite=False)
        alpha = sla.solve_triangular(Lz, m_eps, lower=True, trans="T", check_finite=False)
        self.sposts[id(_struct(out))] = (Lz, Le, alpha)
        return 0

    def sgp_sparse_posterior_predict(self, post, cross, prior_ss, mean_s, mean_out, var_out, cov_out, ldcov):
        Lz, Le, alpha = self.sposts[id(post)]
        Ksz = _Spec(cross).dense()                     # Ns x M
        ns = Ksz.shape[0]
        ms = np.zeros(ns) if not mean_s else _vec(mean_s, ns)
        if mean_out:
            _vec(mean_out, ns)[:] = ms + Ksz @ alpha
        if var_out or cov_out:
            B = sla.solve_triangular(Lz, Ksz.T, lower=True, check_finite=False)
            Cb = sla.solve_triangular(Le, B, lower=True, check_finite=False)
            Kss = _Spec(prior_ss).dense()
            if var_out:
                _vec(var_out, ns)[:] = np.diag(Kss) - (B * B).sum(0) + (Cb * Cb).sum(0)
            if cov_out:
                _mat(cov_out, ns, ns, ldcov)[:, :] = Kss - B.T @ B + Cb.T @ Cb
        return 0

    def sgp_sparse_posterior_destroy(self, post):
        return 0

    # -- logpdf and its reverse-mode gradient (sthenomi.h:154-192) -------------------------------------------
    def _term_grads(self, s, G, grad_coef, grad_inscale, grad_inputs, grad_rowscale, grad_colscale=None):
        gx = None
        if grad_inputs:
            gx = [np.zeros(X.shape) for X in s.inputs]
        for t, (I, J, kind, ri, ci, coef, param, rs, cs) in enumerate(s.terms):
            r, c = s.block(t)
            d2 = s.d2(t)
            k, dk = _kern(kind, d2, param), _dkern_dd2(kind, d2)
            w = G[r, c].copy()
            if rs is not None:
                w = w * rs[:, None]
            if cs is not None:
                w = w * cs[None, :]
            if grad_coef is not None:
                grad_coef[t] = (w * k).sum()
            if grad_inscale is not None:
                # both inputs scaled by g: d2 -> g^2 d2, d/dg at 1 = kappa'(d2) 2 d2
                grad_inscale[t] = coef * (w * dk * 2.0 * d2).sum()
            if gx is not None:
                # d K_ij / d x_i = coef rs_i cs_j kappa'(d2_ij) 2 (x_i - x'_j).  Symmetric spec (header's convention):
                # sum_j 2 G_ij ... on the row side only, the mirror term of pair (J, I) covers the column side.  Cross
                # spec (K(x, z) of the ELBO): the matrix appears once, row points and column points get their own sums.
                X, Y = s.inputs[ri], s.inputs[ci]
                diff = X[:, :, None] - Y[:, None, :]
                core = coef * 2.0 * (w * dk)[None, :, :] * diff
                if s.symmetric:
                    gx[ri] += 2.0 * core.sum(2)
                else:
                    gx[ri] += core.sum(2)
                    gx[ci] -= core.sum(1)
            if grad_rowscale and grad_rowscale[t] and rs is not None:
                wc = G[r, c] * coef * k
                if cs is not None:
                    wc = wc * cs[None, :]
                # symmetric spec: "row side x 2" (the mirror term covers the column role); cross spec: once
                _vec(grad_rowscale[t], s.row_len[I])[:] = (2.0 if s.symmetric else 1.0) * wc.sum(1)
            if grad_colscale and grad_colscale[t] and cs is not None:
                wr = G[r, c] * coef * k
                if rs is not None:
                    wr = wr * rs[:, None]
                _vec(grad_colscale[t], s.col_len[J])[:] = wr.sum(0)
        if gx is not None:
            for k_, g in enumerate(gx):
                if grad_inputs[k_]:
                    dim, n = g.shape
                    _mat(grad_inputs[k_], dim, n, dim)[:, :] = g
        return 0

    def sgp_logpdf_grad_xs(self, ctx, spec, mean, kind, noise, y, lp_out, gy, gm, gn, gc, gs, grad_inputs, grad_rowscale):
        s, m, Cm, rc = self._observed(spec, mean, kind, noise)
        if rc:
            return rc
        Lm, info = _chol(Cm)
        if info:
            return self._fail("matrix is not positive definite", info)
        delta = _vec(y, s.N) - m
        alpha = sla.cho_solve((Lm, True), delta, check_finite=False)
        Cinv = sla.cho_solve((Lm, True), np.eye(s.N), check_finite=False)
        z = sla.solve_triangular(Lm, delta, lower=True, check_finite=False)
        G = 0.5 * (np.outer(alpha, alpha) - Cinv)
        if lp_out:
            lp_out[0] = -0.5 * (s.N * LOG2PI + 2.0 * np.log(np.diag(Lm)).sum() + z @ z)
        if gy:
            _vec(gy, s.N)[:] = -alpha
        if gm:
            _vec(gm, s.N)[:] = alpha
        if gn:
            if kind == L.NOISE_SCALAR:
                gn[0] = np.trace(G)
            elif kind == L.NOISE_DENSE:
                _mat(gn, s.N, s.N, s.N)[:, :] = G
            else:
                _vec(gn, s.N)[:] = np.diag(G)
        nt = len(s.terms)
        return self._term_grads(s, G, _vec(gc, nt) if gc else None, _vec(gs, nt) if gs else None, grad_inputs,
                                grad_rowscale)

    def sgp_logpdf_grad_x(self, ctx, spec, mean, kind, noise, y, lp_out, gy, gm, gn, gc, gs, grad_inputs):
        return self.sgp_logpdf_grad_xs(ctx, spec, mean, kind, noise, y, lp_out, gy, gm, gn, gc, gs, grad_inputs, None)

    def sgp_logpdf_grad(self, ctx, spec, mean, kind, noise, y, lp_out, gy, gm, gn, gc, gs):
        return self.sgp_logpdf_grad_xs(ctx, spec, mean, kind, noise, y, lp_out, gy, gm, gn, gc, gs, None, None)


    # -- gradient of var = sgp_kernelmatrix_diag(spec) (sthenomi.h: sgp_kernelmatrix_diag_grad[_x]) -----------
    def sgp_kernelmatrix_diag_grad_x(self, ctx, spec, w, gc, gs, grad_inputs):
        return self.sgp_kernelmatrix_diag_grad_xs(ctx, spec, w, gc, gs, grad_inputs, None, None)

    def sgp_kernelmatrix_diag_grad_xs(self, ctx, spec, w, gc, gs, grad_inputs, grad_rowscale, grad_colscale):
        s = _Spec(spec)
        nt = len(s.terms)
        wv = _vec(w, s.N)
        gcv, gsv = (_vec(gc, nt) if gc else None), (_vec(gs, nt) if gs else None)
        gx = [np.zeros(X.shape) for X in s.inputs] if grad_inputs else None
        for t, (I, J, kind, ri, ci, coef, param, rs, cs) in enumerate(s.terms):
            if gcv is not None:
                gcv[t] = 0.0
            if gsv is not None:
                gsv[t] = 0.0
            if I != J:
                continue
            X, Y = s.inputs[ri], s.inputs[ci]
            d2 = ((X - Y) ** 2).sum(0)                       # the diagonal of the block only
            k, dk = _kern(kind, d2, param), _dkern_dd2(kind, d2)
            ww = wv[s.roff[I]:s.roff[I + 1]].copy()
            if rs is not None:
                ww = ww * rs
            if cs is not None:
                ww = ww * cs
            if gcv is not None:
                gcv[t] = (ww * k).sum()
            if gsv is not None:
                gsv[t] = coef * (ww * dk * 2.0 * d2).sum()
            if gx is not None:
                core = coef * 2.0 * (ww * dk)[None, :] * (X - Y)
                gx[ri] += core
                gx[ci] -= core
            base = wv[s.roff[I]:s.roff[I + 1]] * coef * k
            if grad_rowscale and grad_rowscale[t] and rs is not None:
                _vec(grad_rowscale[t], s.row_len[I])[:] = base * (cs if cs is not None else 1.0)
            if grad_colscale and grad_colscale[t] and cs is not None:
                _vec(grad_colscale[t], s.row_len[I])[:] = base * (rs if rs is not None else 1.0)
        if gx is not None:
            for k_, g in enumerate(gx):
                if grad_inputs[k_]:
                    _mat(grad_inputs[k_], g.shape[0], g.shape[1], g.shape[0])[:, :] = g
        return 0

    def sgp_kernelmatrix_diag_grad(self, ctx, spec, w, gc, gs):
        return self.sgp_kernelmatrix_diag_grad_x(ctx, spec, w, gc, gs, None)

    # -- elbo and its reverse-mode gradient (sthenomi.h:216-250; cotangents: oracle/abstractgps.py derivation) -----
    def sgp_elbo_grad_x(self, ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out, gy, gm, gn, gv, gzn,
                        gc_zz, gs_zz, gc_xz, gs_xz, gin_zz, gin_xz):
        return self.sgp_elbo_grad_xs(ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out, gy, gm, gn, gv, gzn,
                                     gc_zz, gs_zz, gc_xz, gs_xz, gin_zz, gin_xz, None, None, None)

    def sgp_elbo_grad_xs(self, ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out, gy, gm, gn, gv, gzn,
                         gc_zz, gs_zz, gc_xz, gs_xz, gin_zz, gin_xz, grs_zz, grs_xz, gcs_xz):
        parts, rc = self._vfe_parts(zz, xz, mean_x, nk, noise_x, zk, z_noise, y)
        if rc:
            return rc
        Lz, A, Le, delta, sy = parts
        M, N = A.shape
        rc = self.sgp_elbo(ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out)
        if rc:
            return rc
        I = np.eye(M)
        B = A @ A.T + I
        u = sla.cho_solve((Le, True), A @ delta, check_finite=False)
        Binv = sla.cho_solve((Le, True), I, check_finite=False)
        Z = I - Binv - np.outer(u, u)
        S = B + Binv - 2.0 * I + np.outer(u, u)
        J = sla.solve_triangular(Lz, I, lower=True, check_finite=False).T
        rsig = 1.0 / np.sqrt(sy)
        dA_T = A.T @ Z + np.outer(delta, u)
        dKxz = rsig[:, None] * (dA_T @ J.T)
        dKzz = -0.5 * J @ S @ J.T
        ddelta = -delta + A.T @ u
        dy = ddelta * rsig
        v = _vec(var_x, N)
        dsy = -0.5 / sy + 0.5 * v / sy ** 2 - 0.5 * (ddelta * delta + (A.T * dA_T).sum(1)) / sy
        if gy:
            _vec(gy, N)[:] = dy
        if gm:
            _vec(gm, N)[:] = -dy
        if gn:
            if nk == L.NOISE_SCALAR:
                gn[0] = dsy.sum()
            else:
                _vec(gn, N)[:] = dsy
        if gv:
            _vec(gv, N)[:] = -0.5 / sy
        if gzn:
            if zk == L.NOISE_SCALAR:
                gzn[0] = np.trace(dKzz)
            elif zk == L.NOISE_DENSE:       # M x M, ld M: the cotangent itself
                _vec(gzn, M * M)[:] = np.asfortranarray(dKzz).ravel(order="F")
            else:
                _vec(gzn, M)[:] = np.diag(dKzz)
        sz, sx = _Spec(zz), _Spec(xz)
        self._term_grads(sz, dKzz, _vec(gc_zz, len(sz.terms)) if gc_zz else None,
                         _vec(gs_zz, len(sz.terms)) if gs_zz else None, gin_zz, grs_zz)
        self._term_grads(sx, dKxz, _vec(gc_xz, len(sx.terms)) if gc_xz else None,
                         _vec(gs_xz, len(sx.terms)) if gs_xz else None, gin_xz, grs_xz, gcs_xz)
        return 0

    def sgp_elbo_grad(self, ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out, gy, gm, gn, gv, gzn, gc_zz,
                      gs_zz, gc_xz, gs_xz):
        return self.sgp_elbo_grad_x(ctx, zz, xz, var_x, mean_x, nk, noise_x, zk, z_noise, y, elbo_out, gy, gm, gn, gv, gzn,
                                    gc_zz, gs_zz, gc_xz, gs_xz, None, None)


class FakeContext:
    def __init__(self):
        self.lib, self.handle, self.device = FakeLib(), None, 0

    def factor_work(self):
        """(executed, dense) tile products of the last factorisation: the double factors densely (nothing is skipped)."""
        return 0.0, 0.0


def install(monkeypatch):
    """Route the host mirror's calls through the NumPy double for the duration of one test."""
    ctx = FakeContext()
    monkeypatch.setattr(L, "_default_ctx", ctx)
    monkeypatch.setattr(L, "load", lambda: ctx.lib)
    return ctx
