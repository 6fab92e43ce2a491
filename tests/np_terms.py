"""NumPy evaluator of a flattened covariance spec (stheno_jl_amd.lib.Spec).

TEST DOUBLE ONLY: lets the CPU suite check the *host-side flattener* (tree -> kernel terms)
against the oracle's literal recursion without a GPU.  It restates what
stheno.jl_amd/csrc/kernelmatrix.hip computes per term; the product never imports it."""
import numpy as np

from stheno_jl_amd import lib as L


def _kern(kind, d2, param):
    d = np.sqrt(d2)
    if kind == L.SE:
        return np.exp(-0.5 * d2)
    if kind == L.MATERN12:
        return np.exp(-d)
    if kind == L.MATERN32:
        l = np.sqrt(3.0) * d
        return (1.0 + l) * np.exp(-l)
    if kind == L.MATERN52:
        l = np.sqrt(5.0) * d
        return (1.0 + l + l * l / 3.0) * np.exp(-l)
    if kind == L.WHITE:
        return (d2 == 0.0).astype(np.float64)
    if kind == L.CONST:
        return np.full_like(d2, param)
    raise ValueError(kind)


def spec_terms(spec):
    """[(I, J, kind, row_input, col_input, coef, param, rs, cs)] read back from the ctypes spec."""
    out = []
    nrb, ncb = len(spec.row_len), len(spec.col_len)
    tp = spec._term_ptr
    for I in range(nrb):
        for J in range(ncb):
            p = I * ncb + J
            for t in range(tp[p], tp[p + 1]):
                T = spec._terms[t]
                nr, nc = int(spec.row_len[I]), int(spec.col_len[J])
                rs = np.ctypeslib.as_array(T.row_scale, shape=(nr,)).copy() if T.row_scale else None
                cs = np.ctypeslib.as_array(T.col_scale, shape=(nc,)).copy() if T.col_scale else None
                out.append((I, J, T.kind, T.row_input, T.col_input, T.coef, T.param, rs, cs))
    return out


def dense_from_spec(spec):
    roff = np.concatenate([[0], np.cumsum(spec.row_len)])
    coff = np.concatenate([[0], np.cumsum(spec.col_len)])
    K = np.zeros((spec.N, spec.M))
    for (I, J, kind, ri, ci, coef, param, rs, cs) in spec_terms(spec):
        X, Y = spec.inputs[ri], spec.inputs[ci]
        d2 = np.zeros((X.shape[1], Y.shape[1]))
        for d in range(X.shape[0]):
            df = X[d][:, None] - Y[d][None, :]
            d2 += df * df
        blk = coef * _kern(kind, d2, param)
        if rs is not None:
            blk = rs[:, None] * blk
        if cs is not None:
            blk = blk * cs[None, :]
        K[roff[I]:roff[I + 1], coff[J]:coff[J + 1]] += blk
    return K
