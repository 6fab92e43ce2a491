"""First GPU bring-up: exercise every kernel through the C-ABI against NumPy/SciPy formulas.
Run on the GPU box:  python tools/gpu_trip1.py  (writes gpurun_out/trip1.json)."""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)

out = {}
ctx = L.Context(0)
lib = ctx.lib


def kern(kind, d2):
    d = np.sqrt(d2)
    if kind == L.SE:
        return np.exp(-0.5 * d2)
    if kind == L.MATERN12:
        return np.exp(-d)
    if kind == L.MATERN32:
        return (1 + np.sqrt(3) * d) * np.exp(-np.sqrt(3) * d)
    if kind == L.MATERN52:
        return (1 + np.sqrt(5) * d + 5 * d2 / 3) * np.exp(-np.sqrt(5) * d)
    raise ValueError


def sqdist(X, Y):
    return ((X[:, :, None] - Y[:, None, :]) ** 2).sum(0)


def single_spec(X, kind, coef=1.0, X2=None):
    if X2 is None:
        return L.Spec([X.shape[1]], [X.shape[1]], [X], {(0, 0): [(kind, 0, 0, coef, 0.0, None, None)]}, True)
    return L.Spec([X.shape[1]], [X2.shape[1]], [X, X2], {(0, 0): [(kind, 0, 1, coef, 0.0, None, None)]}, False)


# ---- micro benchmarks -----------------------------------------------------------------
tf = C.c_double()
err = C.c_double()
L.check(lib.sgp_bench_mfma_f64(ctx.handle, 20000, C.byref(tf), C.byref(err)), "mfma bench")
out["mfma_f64_tflops"] = tf.value
out["mfma_layout_maxerr"] = err.value
print("mfma f64 peak TF/s", tf.value, "layout err", err.value, flush=True)
w = C.c_double()
cp = C.c_double()
L.check(lib.sgp_bench_hbm(ctx.handle, 4 << 30, 10, C.byref(w), C.byref(cp)), "hbm bench")
out["hbm_write_gbs"] = w.value
out["hbm_copy_gbs"] = cp.value
print("hbm write GB/s", w.value, "copy GB/s", cp.value, flush=True)
for (m, n, k, lo) in [(1024, 1024, 128, 0), (8192, 8192, 512, 1), (16384, 16384, 512, 1), (16384, 16384, 128, 1),
                      (32768, 32768, 512, 1)]:
    L.check(lib.sgp_bench_gemm(ctx.handle, m, n, k, lo, 5, C.byref(tf), C.byref(err)), "gemm bench")
    out[f"gemm_{m}_{n}_{k}_{lo}"] = {"tflops": tf.value, "maxerr": err.value}
    print("gemm", m, n, k, lo, "TF/s", tf.value, "maxerr", err.value, flush=True)

# ---- kernelmatrix -----------------------------------------------------------------------
rng = np.random.default_rng(123456)
res = {}
for kind in (L.SE, L.MATERN12, L.MATERN32, L.MATERN52):
    for (D, n, m) in [(1, 5, 7), (2, 300, 129), (8, 513, 400), (3, 128, 128), (17, 200, 260)]:
        X = np.asfortranarray(rng.standard_normal((D, n)))
        Y = np.asfortranarray(rng.standard_normal((D, m)))
        sp = single_spec(X, kind, 1.7, Y)
        K = np.zeros((n, m), order="F")
        L.check(lib.sgp_kernelmatrix(ctx.handle, sp.ref(), L.dptr(K), n), "kernelmatrix")
        ref = 1.7 * kern(kind, sqdist(X, Y))
        e = np.abs(K - ref).max()
        res[f"k{kind}_D{D}_{n}x{m}"] = e
        sp = single_spec(X, kind)
        K = np.zeros((n, n), order="F")
        L.check(lib.sgp_kernelmatrix(ctx.handle, sp.ref(), L.dptr(K), n), "kernelmatrix")
        ref = kern(kind, sqdist(X, X))
        e2 = np.abs(K - ref).max()
        res[f"k{kind}_D{D}_{n}sym"] = e2
        assert np.array_equal(K, K.T)
        if kind == L.SE:
            assert np.all(np.diag(K) == 1.0)
        dg = np.zeros(n)
        L.check(lib.sgp_kernelmatrix_diag(ctx.handle, sp.ref(), L.dptr(dg)), "diag")
        assert np.array_equal(dg, np.diag(K)), (dg, np.diag(K))
out["kernelmatrix_maxerr"] = max(res.values())
print("kernelmatrix max err", out["kernelmatrix_maxerr"], flush=True)


# ---- logpdf / rand / posterior ------------------------------------------------------------
def ref_logpdf(Kn, Y, m):
    Lc = np.linalg.cholesky(Kn)
    Z = sla.solve_triangular(Lc, Y - m[:, None], lower=True)
    N = Kn.shape[0]
    return -0.5 * (N * np.log(2 * np.pi) + 2 * np.log(np.diag(Lc)).sum() + (Z ** 2).sum(0))


lp_res = {}
for (kind, D, N, S, s2) in [(L.SE, 2, 100, 1, 0.1), (L.SE, 2, 128, 3, 0.1), (L.MATERN52, 3, 300, 2, 0.05),
                            (L.SE, 8, 640, 1, 0.1), (L.MATERN32, 4, 1000, 4, 0.2), (L.SE, 2, 2048, 1, 0.1),
                            (L.MATERN52, 8, 4096, 2, 0.1)]:
    X = np.asfortranarray(rng.standard_normal((D, N)) / np.sqrt(D))
    Y = np.asfortranarray(rng.standard_normal((N, S)))
    mean = rng.standard_normal(N)
    sp = single_spec(X, kind)
    o = np.zeros(S)
    nz = np.array([s2])
    t0 = time.time()
    L.check(lib.sgp_logpdf(ctx.handle, sp.ref(), L.dptr(mean), L.NOISE_SCALAR, L.dptr(nz), L.dptr(Y), N, S, L.dptr(o)), "logpdf")
    dt = time.time() - t0
    Kn = kern(kind, sqdist(X, X)) + s2 * np.eye(N)
    ref = ref_logpdf(Kn, Y, mean)
    rel = np.abs((o - ref) / ref).max()
    lp_res[f"k{kind}_N{N}_S{S}"] = rel
    print("logpdf", kind, D, N, S, "rel", rel, "time", dt, flush=True)
    # diagonal noise
    nd = 0.05 + rng.random(N)
    L.check(lib.sgp_logpdf(ctx.handle, sp.ref(), L.dptr(mean), L.NOISE_DIAG, L.dptr(nd), L.dptr(Y), N, S, L.dptr(o)), "logpdf")
    ref = ref_logpdf(kern(kind, sqdist(X, X)) + np.diag(nd), Y, mean)
    lp_res[f"k{kind}_N{N}_S{S}_diag"] = np.abs((o - ref) / ref).max()
    if N <= 1000:
        A = rng.standard_normal((N, N))
        Sd = np.asfortranarray(A @ A.T / N + 0.1 * np.eye(N))
        L.check(lib.sgp_logpdf(ctx.handle, sp.ref(), L.dptr(mean), L.NOISE_DENSE, L.dptr(Sd), L.dptr(Y), N, S, L.dptr(o)), "logpdf")
        ref = ref_logpdf(kern(kind, sqdist(X, X)) + Sd, Y, mean)
        lp_res[f"k{kind}_N{N}_S{S}_dense"] = np.abs((o - ref) / ref).max()
    # rand
    Sr = 3
    Z = np.asfortranarray(rng.standard_normal((N, Sr)))
    R = np.zeros((N, Sr), order="F")
    L.check(lib.sgp_rand(ctx.handle, sp.ref(), L.dptr(mean), L.NOISE_SCALAR, L.dptr(nz), L.dptr(Z), N, Sr, L.dptr(R), N), "rand")
    refR = mean[:, None] + np.linalg.cholesky(Kn) @ Z
    lp_res[f"rand_k{kind}_N{N}"] = np.abs(R - refR).max() / np.abs(refR).max()
    # posterior
    y = Y[:, 0].copy()
    alpha = np.zeros(N)
    ph = C.c_void_p()
    L.check(lib.sgp_posterior_create(ctx.handle, sp.ref(), L.dptr(mean), L.NOISE_SCALAR, L.dptr(nz), L.dptr(y), L.dptr(alpha), C.byref(ph)), "post")
    aref = np.linalg.solve(Kn, y - mean)
    lp_res[f"alpha_k{kind}_N{N}"] = np.abs(alpha - aref).max() / np.abs(aref).max()
    Ns = 77
    Xs = np.asfortranarray(rng.standard_normal((D, Ns)) / np.sqrt(D))
    cross = single_spec(Xs, kind, 1.0, X)
    pss = single_spec(Xs, kind)
    ms = rng.standard_normal(Ns)
    mo = np.zeros(Ns)
    vo = np.zeros(Ns)
    co = np.zeros((Ns, Ns), order="F")
    L.check(lib.sgp_posterior_predict(ph, cross.ref(), pss.ref(), L.dptr(ms), L.dptr(mo), L.dptr(vo), L.dptr(co), Ns), "predict")
    Ksx = kern(kind, sqdist(Xs, X))
    Kss = kern(kind, sqdist(Xs, Xs))
    mref = ms + Ksx @ aref
    V = sla.solve_triangular(np.linalg.cholesky(Kn), Ksx.T, lower=True)
    cref = Kss - V.T @ V
    lp_res[f"pmean_k{kind}_N{N}"] = np.abs(mo - mref).max() / max(1, np.abs(mref).max())
    lp_res[f"pvar_k{kind}_N{N}"] = np.abs(vo - np.diag(cref)).max()
    lp_res[f"pcov_k{kind}_N{N}"] = np.abs(co - cref).max()
    lib.sgp_posterior_destroy(ph)
out["logpdf_etc"] = lp_res
print(json.dumps(lp_res, indent=1), flush=True)

# non-PD must surface as info > 0
X = np.asfortranarray(np.zeros((1, 10)))
sp = single_spec(X, L.SE)
o = np.zeros(1)
nz = np.array([-0.5])
rc = lib.sgp_logpdf(ctx.handle, sp.ref(), None, L.NOISE_SCALAR, L.dptr(nz), L.dptr(np.zeros(10)), 10, 1, L.dptr(o))
out["nonpd_rc"] = rc
print("non-PD rc", rc, L.last_error(), flush=True)

# ---- elbo -------------------------------------------------------------------------------
el = {}
for (kind, D, N, M, s2) in [(L.SE, 2, 300, 40, 0.1), (L.MATERN32, 1, 1000, 129, 0.5), (L.SE, 4, 5000, 256, 0.1)]:
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Zi = np.asfortranarray(X[:, :M].copy())
    y = rng.standard_normal(N)
    mean = 0.1 * rng.standard_normal(N)
    zz = single_spec(Zi, kind)
    xz = single_spec(X, kind, 1.0, Zi)
    varx = np.ones(N)
    nz = np.array([s2])
    jz = np.array([1e-6])
    o = np.zeros(1)
    L.check(lib.sgp_elbo(ctx.handle, zz.ref(), xz.ref(), L.dptr(varx), L.dptr(mean), L.NOISE_SCALAR, L.dptr(nz), L.NOISE_SCALAR, L.dptr(jz), L.dptr(y), L.dptr(o)), "elbo")
    Kzz = kern(kind, sqdist(Zi, Zi)) + 1e-6 * np.eye(M)
    Kxz = kern(kind, sqdist(X, Zi))
    Lz = np.linalg.cholesky(Kzz)
    A = sla.solve_triangular(Lz, Kxz.T, lower=True) / np.sqrt(s2)
    Le = np.linalg.cholesky(A @ A.T + np.eye(M))
    delta = (y - mean) / np.sqrt(s2)
    tmp = N * np.log(s2) + 2 * np.log(np.diag(Le)).sum() + delta @ delta - (sla.solve_triangular(Le, A @ delta, lower=True) ** 2).sum()
    ref = -0.5 * (N * np.log(2 * np.pi) + tmp) - 0.5 * (varx.sum() / s2 - (A ** 2).sum())
    el[f"elbo_k{kind}_N{N}_M{M}"] = abs((o[0] - ref) / ref)
    print("elbo", kind, N, M, o[0], ref, flush=True)
out["elbo"] = el

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "trip1.json"), "w"), indent=1)
print("TRIP1 DONE")
