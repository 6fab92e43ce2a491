"""Model flattener: a Stheno GP tree evaluated at (BlockData) inputs -> kernel terms per block
pair, i.e. the `sgp_cov_spec` the HIP library consumes (include/sthenomi.h).

The reference evaluates cov(f_i, f_j, x_I, x_J) by recursive dispatch on creation order
(/root/reference/src/gp/derived_gp.jl:31-44) through dense N x N' temporaries.  Covariance is
bilinear and distinct AtomicGPs are independent (src/gp/atomic_gp.jl:36-38), so the same
matrix is the *path expansion* of SURVEY.md Appendix B:

  paths(f, x) = [(atom, c, r, X)]   : f(x) = sum_p c_p * diag(r_p) * atom_p(X_p) (+ mean)
  K[I, J] = sum_{p in paths(I), q in paths(J), atom_p is atom_q}
                c_p c_q diag(r_p) k_atom(X_p, X_q) diag(r_q)

with  +            -> union of paths                 (addition.jl:20-47; known-function add :67-86)
      * real       -> c <- c * sigma                  (product.jl:54-70; -f = (-1) * f, :73)
      * function   -> r <- r .* sigma.(x)             (product.jl:25-48; x as seen at that node)
      o g          -> X <- g.(X)                      (compose.jl:16-28, fast warps :36-127)
      cross / GPPP -> one process per block           (cross.jl:54-93, gppp.jl:25-43)
and the leaf kernel expanded into SimpleKernel terms (ScaledKernel -> coefficient, KernelSum ->
several terms, ScaleTransform -> input scale).  Block pairs without a common atom get no
terms, which the library writes as exact zeros (test/gp/atomic_gp.jl:33).
"""
from __future__ import annotations

import numpy as np

from . import gp as _gp
from . import lib as _lib
from .gppp import GPPP, extract_components
from .inputs import BlockData, ColVecs, GPPPInput, as_matrix, blocks


class _Path:
    __slots__ = ("key", "atom", "c", "r", "X")

    def __init__(self, key, atom, c, r, X):
        self.key, self.atom, self.c, self.r, self.X = key, atom, c, r, X


class _MatCache:
    """as_matrix with identity caching so equal inputs are uploaded once."""

    def __init__(self):
        self.m = {}
        self.keep = []

    def __call__(self, x):
        k = id(x)
        if k not in self.m:
            self.m[k] = np.asfortranarray(as_matrix(x))
            self.keep.append(x)
        return self.m[k]


def block_list(f, x):
    """[(process node, inputs)] per block of x, in order (cross.jl / gppp.jl semantics)."""
    if isinstance(f, GPPP):
        node, v = extract_components(f, x)
        return block_list(node, v)
    if isinstance(f, _gp.DerivedGP) and f.args[0] == "cross":
        if not isinstance(x, BlockData) or len(blocks(x)) != len(f.args[1]):
            raise ValueError("a cross(...) process must be indexed with a BlockData of matching length")
        out = []
        for g, b in zip(f.args[1], blocks(x)):
            out.extend(block_list(g, b))
        return out
    return [(f, x)]


def _paths(f, x, c, r, key, mat):
    if isinstance(f, GPPP):  # a GPPP wrapped in atomic(...) (nested programmes, test gppp.jl:107-120)
        node, v = extract_components(f, x)
        if isinstance(node, _gp.DerivedGP) and node.args[0] == "cross":
            raise NotImplementedError("a nested GPPP must be indexed one process at a time")
        return _paths(node, v, c, r, key, mat)
    if isinstance(f, _gp.AtomicGP):
        k2 = key + (id(f),)
        if isinstance(f.gp, _gp.GP):
            return [_Path(k2, f, c, r, mat(x))]
        return _paths(f.gp, x, c, r, k2, mat)
    op = f.args[0]
    if op == "+":
        return _paths(f.args[1], x, c, r, key, mat) + _paths(f.args[2], x, c, r, key, mat)
    if op == "+known":
        return _paths(f.args[2], x, c, r, key, mat)
    if op == "*":
        s = f.args[1]
        if _gp._is_real(s):
            return _paths(f.args[2], x, c * float(s), r, key, mat)
        sx = _gp._map_points(s, x)
        return _paths(f.args[2], x, c, sx if r is None else r * sx, key, mat)
    if op == "o":
        return _paths(f.args[1], _gp.warp(f.args[2], x), c, r, key, mat)
    if op == "cross":
        raise ValueError("cross(...) can only appear at block level")
    raise ValueError(op)


def _merge_paths(ps):
    out, index = [], {}
    for p in ps:
        k = (p.key, id(p.X), id(p.r) if p.r is not None else None)
        if k in index:
            index[k].c += p.c
        else:
            q = _Path(p.key, p.atom, p.c, p.r, p.X)
            index[k] = q
            out.append(q)
    return out


class _InputTable:
    def __init__(self):
        self.arrays, self.index = [], {}

    def get(self, X, scale):
        k = (id(X), float(scale))
        if k not in self.index:
            self.index[k] = len(self.arrays)
            self.arrays.append(X if scale == 1.0 else np.asfortranarray(scale * X))
        return self.index[k]


def build_spec(f, x, f2=None, x2=None):
    """lib.Spec for cov(f, x) (symmetric) or cov(f, f2, x, x2) (cross; no noise).

    f / f2: GPPP or Stheno GP nodes of one model (same GPC).  Returns (spec, row_blocks,
    col_blocks) where the block lists are [(node, inputs)]."""
    symmetric = x2 is None
    mat = _MatCache()
    rows = block_list(f, x)
    cols = rows if symmetric else block_list(f if f2 is None else f2, x2)
    gpcs = {id(n.gpc) for n, _ in rows} | {id(n.gpc) for n, _ in cols}
    if len(gpcs) > 1:
        raise AssertionError("f.gpc === f'.gpc violated: processes come from different GPCs")
    rpaths = [_merge_paths(_paths(n, v, 1.0, None, (), mat)) for n, v in rows]
    cpaths = rpaths if symmetric else [_merge_paths(_paths(n, v, 1.0, None, (), mat)) for n, v in cols]
    table = _InputTable()
    pairs = {}
    for I, pi in enumerate(rpaths):
        for J, pj in enumerate(cpaths):
            merged, order = {}, []
            for p in pi:
                for q in pj:
                    if p.key != q.key:
                        continue
                    if p.X.shape[0] != q.X.shape[0]:
                        raise ValueError("input dimension mismatch between two views of one process")
                    for (kind, kc, param, s) in p.atom.gp.kernel.leaf_terms():
                        ri, ci = table.get(p.X, s), table.get(q.X, s)
                        k = (kind, param, ri, ci, id(p.r) if p.r is not None else None,
                             id(q.r) if q.r is not None else None)
                        if k in merged:
                            merged[k][3] += p.c * q.c * kc
                        else:
                            merged[k] = [kind, ri, ci, p.c * q.c * kc, param, p.r, q.r]
                            order.append(k)
            if order:
                pairs[(I, J)] = [tuple(merged[k]) for k in order]
    spec = _lib.Spec([len(v) for _, v in rows], [len(v) for _, v in cols], table.arrays, pairs, symmetric)
    spec._mat_keep = mat  # keep the source arrays alive (ids are identity keys)
    return spec, rows, cols


def zero_spec(n):
    """A spec with no terms: K == 0 exactly (used to feed an explicit covariance as dense noise)."""
    return _lib.Spec([n], [n], [], {}, True)
