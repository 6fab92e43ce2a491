"""NumPy test double of stheno_jl_amd.dist.HipOps (TEST INFRASTRUCTURE ONLY).

Drives the multi-rank orchestration of stheno.jl_amd/dist.py on CPU tensors over gloo so that
panel ownership, broadcast order, look-ahead indexing and the final reductions are covered by
the CPU suite (the HIP kernels themselves are covered by the -m gpu tests).  Mirrors the
semantics of the C-ABI building blocks in include/sthenomi.h; the product never imports it."""
import numpy as np
import torch

import np_terms


class NumpyOps:
    def __init__(self):
        self.calls = []

    def empty(self, n):
        return torch.full((n,), float("nan"), dtype=torch.float64)

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def izeros(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).clone()

    def to_host(self, t):
        return t.numpy()

    def make_dspec(self, spec):
        return {"K": np_terms.dense_from_spec(spec)}

    def free_dspec(self, h):
        pass

    def synchronize(self):
        pass

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def panel_context(self):
        import contextlib
        return contextlib.nullcontext()

    def fork_updates(self):
        pass

    def join_updates(self):
        pass

    def pool_context(self, idx):
        import contextlib
        return contextlib.nullcontext()

    def record(self, name):
        self.calls.append(("record", name, None))

    def wait(self, name):
        self.calls.append(("wait", name, None))

    @staticmethod
    def _mat(A, off, ld, ncols):
        """view (ld, ncols) of a packed panel: row index = global row - first stored row"""
        return A.numpy()[off:off + ld * ncols].reshape(ncols, ld).T

    def assemble_cols(self, ds, N, c0, nc, A, off, ld, m_tot, mean, sigma2, Y, ncols):
        self.calls.append(("assemble", c0, nc))
        assert ld == m_tot - c0
        n_pad = max(128, (N + 127) // 128 * 128)       # m_tot may hold more bordered rows (test points) below
        M = self._mat(A, off, ld, nc)
        K = ds["K"]
        for lc in range(nc):
            gc = c0 + lc
            col = np.zeros(m_tot)
            if gc < N:
                col[:N] = K[:, gc]
                col[gc] += sigma2
                for s in range(ncols):
                    col[n_pad + s] = Y.numpy().reshape(-1)[gc + s * N] - (mean.numpy()[gc] if mean is not None else 0.0)
            else:
                col[gc] = 1.0
            M[:, lc] = col[c0:]       # rows above c0 are not stored at all

    def panel_factor(self, A, off, ld, m, J0, w, logdet, info):
        self.calls.append(("factor", J0, w))
        assert m <= ld          # (m < ld: a sub-panel of a packed panel -- only its first m rows are the panel's)
        P = self._strided(A, off, ld, m, w)
        D = np.tril(P[:w, :w]) + np.tril(P[:w, :w], -1).T
        try:
            L = np.linalg.cholesky(D)
        except np.linalg.LinAlgError:
            if int(info[0]) == 0:
                # first failing leading minor, LAPACK convention
                for k in range(1, w + 1):
                    try:
                        np.linalg.cholesky(D[:k, :k])
                    except np.linalg.LinAlgError:
                        info[0] = J0 + k
                        break
            P[:] = np.nan
            return
        P[:w, :w] = L
        P[w:, :] = np.linalg.solve(L, P[w:, :].T).T
        logdet[0] += 2.0 * np.log(np.diag(L)).sum()

    @staticmethod
    def _strided(A, off, ld, nrows, ncols):
        """writable (nrows, ncols) view: element (r, k) at A[off + r + k * ld]"""
        a = A.numpy()
        return np.lib.stride_tricks.as_strided(a[off:], shape=(nrows, ncols), strides=(a.itemsize, a.itemsize * ld))

    def panel_update_batch(self, srcs, dsts, m_tot):
        """sgp_dev_panel_update_batch: destination (tensor, off, ld, c0, w, s_first, s_count) -= sum over its sources
        (tensor, off, ld, row0, w) of P[rows >= c0] P[rows c0 .. c0 + w]'  (lower trapezoid by 128-tile)"""
        self.calls.append(("update_batch", len(srcs), len(dsts)))
        for (A, off, ld, c0, w, s0, sn) in dsts:
            M = self._strided(A, off, ld, m_tot - c0, w)
            low = (np.arange(m_tot - c0)[:, None] // 128) >= (np.arange(w)[None, :] // 128)
            for (Pt, p_off, ldp, row0, sw) in srcs[s0:s0 + sn]:
                self.calls.append(("update", row0, c0))      # (source panel's first column, destination's first column)
                Pm = self._strided(Pt, p_off, ldp, m_tot - row0, sw)
                rows = Pm[c0 - row0:, :]
                upd = rows @ Pm[c0 - row0:c0 - row0 + w, :].T
                M[low] -= upd[low]

    def panel_update(self, Pt, p_off, ldp, J0, w, A, off, ld, c0, nc, m_tot):
        self.calls.append(("update", J0, c0))
        assert ldp == m_tot - J0 and ld == m_tot - c0
        P = self._mat(Pt, p_off, ldp, w)                 # rows J0..m_tot
        M = self._mat(A, off, ld, nc)                    # rows c0..m_tot
        rows = P[c0 - J0:, :]
        M[:, :] -= rows @ P[c0 - J0:c0 - J0 + nc, :].T

    # -- posterior on the sharded factor (sgp_dev_assemble_cross_rows / rows_dot / rows_gram)
    def assemble_cross_rows(self, dx, c0, nc, A, off, ld, row0):
        self.calls.append(("cross", c0, nc))
        Kx = dx["K"]                                     # n* x N
        ns, Nx = Kx.shape
        ns_pad = (ns + 127) // 128 * 128
        M = self._mat(A, off, ld, nc)                    # rows c0..m_tot
        blk = np.zeros((ns_pad, nc))
        v = max(0, min(nc, Nx - c0))
        blk[:ns, :v] = Kx[:, c0:c0 + v]
        M[row0 - c0: row0 - c0 + ns_pad, :] = blk

    def rows_dot(self, A, r_off, ld, nrows, nc, z_off, pv, ns_pad):
        a = A.numpy()
        for r in range(nrows):
            row = a[r_off + r: r_off + r + ld * nc: ld]
            z = a[z_off: z_off + ld * nc: ld]
            pv[r] += float((row ** 2).sum())
            pv[ns_pad + r] += float((row * z).sum())

    def rows_gram(self, A, r_off, ld, nrows_pad, nc, G):
        a = A.numpy()
        R = np.stack([a[r_off + r: r_off + r + ld * nc: ld] for r in range(nrows_pad)])
        G.numpy()[:] += (R @ R.T).reshape(-1)

    def rowsumsq(self, A, off, ld, nc, nrows, out):
        for s in range(nrows):
            out[s] += float((A.numpy()[off + s: off + s + ld * nc: ld] ** 2).sum())

    # -- sharded ELBO: same "part" contract as sgp_dev_elbo_partial / sgp_dev_elbo_finish (sums over the
    # rank's data slice that add up across ranks), stated with dense NumPy algebra (App. A.6)
    def prior_var(self, f, x):
        import stheno_jl_amd as P
        if len(x) == 0:
            return np.zeros(0)
        return np.diag(np_terms.dense_from_spec(P.build_spec(f, x)[0])).copy()

    def prior_cov(self, f, x):
        import stheno_jl_amd as P
        K = np_terms.dense_from_spec(P.build_spec(f, x)[0])
        return np.tril(K) + np.tril(K, -1).T

    def elbo_part(self, M):
        return torch.zeros(M * M + M + 4, dtype=torch.float64)

    def elbo_partial(self, zz, xz, var_x, mean_x, nk, nbuf, zk, zbuf, y, part):
        self.calls.append(("elbo_partial", xz.N, xz.M))
        M = zz.N
        Kzz = np_terms.dense_from_spec(zz)
        Sz = (float(zbuf[0]) * np.eye(M)) if zk == 0 else (np.diag(zbuf) if zk == 1 else np.asarray(zbuf).reshape(M, M))
        Lz = np.linalg.cholesky(Kzz + Sz)
        n = xz.N
        p = part.numpy()
        p[:] = 0.0
        if n == 0:
            return
        s2 = np.full(n, float(nbuf[0])) if nk == 0 else np.asarray(nbuf)
        Kxz = np_terms.dense_from_spec(xz)
        A = np.linalg.solve(Lz, (Kxz / np.sqrt(s2)[:, None]).T)          # M x n
        delta = (np.asarray(y) - (mean_x if mean_x is not None else 0.0)) / np.sqrt(s2)
        p[:M * M] = (A @ A.T).ravel()
        p[M * M:M * M + M] = A @ delta
        p[M * M + M:] = [np.log(s2).sum(), delta @ delta, (var_x / s2).sum(), (A * A).sum()]

    def elbo_finish(self, M, N_total, part):
        p = part.numpy()
        B = p[:M * M].reshape(M, M) + np.eye(M)
        Ad = p[M * M:M * M + M]
        h0, h1, h2, h3 = p[M * M + M:]
        Le = np.linalg.cholesky(B)
        w = np.linalg.solve(Le, Ad)
        tmp = h0 + 2.0 * np.log(np.diag(Le)).sum() + h1 - w @ w
        return float(-0.5 * (N_total * np.log(2.0 * np.pi) + tmp) - 0.5 * (h2 - h3))
