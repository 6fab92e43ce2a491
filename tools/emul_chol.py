"""NumPy emulation of the device Cholesky's numerics (explicit-inverse sub-panel / panel solves) to
study accuracy on ill-conditioned covariances without a GPU.  Dev tool."""
import sys
import numpy as np, scipy.linalg as sla
import mpmath as mp

def micro(Ab):
    """16x16 Cholesky + inverse by carrying identity rows (as potrf_diag wave 0)."""
    n = Ab.shape[0]
    R = np.vstack([np.tril(Ab) + np.tril(Ab, -1).T, np.eye(n)])
    for j in range(n):
        d = np.sqrt(R[j, j]); rinv = 1.0 / d
        R[:, j] *= rinv
        R[j, j] = d
        for c in range(j + 1, n):
            R[:, c] -= R[:, j] * R[c, j]
    L = np.tril(R[:n]); InvT = R[n:]   # rows of X = I L^-T  -> InvT = inv(L)^T
    return L, np.triu(InvT).T

def potrf_diag(A, refine=False):
    T = np.tril(A).copy(); n = T.shape[0]; nb = 16
    invd = []
    for cb in range(n // nb):
        c0 = cb * nb
        if cb:
            T[c0:, c0:c0+nb] -= T[c0:, :c0] @ T[c0:c0+nb, :c0].T
        L, Inv = micro(T[c0:c0+nb, c0:c0+nb])
        T[c0:c0+nb, c0:c0+nb] = L
        invd.append(Inv)
        B = T[c0+nb:, c0:c0+nb]
        X = B @ Inv.T
        if refine:
            X = X + (B - X @ L.T) @ Inv.T
        T[c0+nb:, c0:c0+nb] = X
    return T, invd

def trtri(L, invd):
    n = L.shape[0]; nb = 16; W = np.zeros_like(L)
    for cb in range(n // nb):
        W[cb*nb:(cb+1)*nb, cb*nb:(cb+1)*nb] = invd[cb]
        for rb in range(cb + 1, n // nb):
            S = L[rb*nb:(rb+1)*nb, cb*nb:rb*nb] @ W[cb*nb:rb*nb, cb*nb:(cb+1)*nb]
            W[rb*nb:(rb+1)*nb, cb*nb:(cb+1)*nb] = -invd[rb] @ S
    return W

def solve16(B, L, invd):
    """blocked substitution over 16-column blocks with refined diagonal solves (panel_solve_kernel)."""
    nb = 16; nB = -B.copy(); X = np.zeros_like(B)
    for c in range(L.shape[0] // nb):
        sl = slice(c*nb, (c+1)*nb)
        T = -nB[:, sl]
        X1 = T @ invd[c].T
        R = T - X1 @ L[sl, sl].T
        X[:, sl] = X1 + R @ invd[c].T
        for c2 in range(c + 1, L.shape[0] // nb):
            s2 = slice(c2*nb, (c2+1)*nb)
            nB[:, s2] += X[:, sl] @ L[s2, sl].T
    return X

def chol(A, refine16=False, refine128=False, sub16=False):
    A = np.tril(A).copy(); n = A.shape[0]; nb = 128
    npad = -(-n // nb) * nb
    P = np.eye(npad); P[:n, :n] = A; A = np.tril(P)
    for j in range(0, npad, nb):
        L, invd = potrf_diag(A[j:j+nb, j:j+nb], refine16)
        A[j:j+nb, j:j+nb] = L
        if j + nb < npad:
            W = trtri(L, invd)
            B = A[j+nb:, j:j+nb]
            X = solve16(B, L, invd) if sub16 else B @ W.T
            if refine128 and not sub16:
                X = X + (B - X @ L.T) @ W.T
            A[j+nb:, j:j+nb] = X
            A[j+nb:, j+nb:] -= np.tril(X @ X.T)
    return A[:n, :n]

def lp(L, y):
    z = sla.solve_triangular(L, y, lower=True)
    return -0.5 * (len(y) * np.log(2*np.pi) + 2*np.log(np.diag(L)).sum() + z @ z)

rng = np.random.default_rng(1)
for N, s2 in [(200, 1e-6), (200, 1e-9), (200, 1e-12), (400, 1e-8), (130, 1e-10)]:
    x = np.sort(rng.uniform(-3, 3, N)); y = rng.standard_normal(N)
    K = np.exp(-0.5 * (x[:, None] - x[None, :])**2) + s2 * np.eye(N)
    mp.mp.dps = 60
    Km = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            Km[i, j] = mp.exp(-(mp.mpf(x[i]) - mp.mpf(x[j]))**2 / 2)
        Km[i, i] += mp.mpf(s2)
    Lm = mp.cholesky(Km); zm = mp.lu_solve(Lm, mp.matrix(list(y)))
    t = float(-(N*mp.log(2*mp.pi) + 2*sum(mp.log(Lm[i, i]) for i in range(N)) + sum(v*v for v in zm))/2)
    res = {"lapack": lp(sla.cholesky(K, lower=True), y)}
    for nm, kw in [("dev", {}), ("r128", dict(refine128=True)), ("r16+r128", dict(refine16=True, refine128=True)), ("r16+sub16", dict(refine16=True, sub16=True))]:
        with np.errstate(all="ignore"):
            L = chol(K, **kw)
            res[nm] = lp(L, y) if np.isfinite(L).all() else np.nan
    print(N, s2, {k: f"{abs(v-t)/abs(t):.1e}" for k, v in res.items()}, flush=True)
