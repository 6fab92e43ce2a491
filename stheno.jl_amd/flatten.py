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
from . import kernels as _kernels
from . import lib as _lib
from .gppp import GPPP, extract_components
from .inputs import BlockData, ColVecs, GPPPInput, as_matrix, blocks, is_pair_vector, regroup_pairs


class _Path:
    __slots__ = ("key", "atom", "c", "r", "X", "chain")

    def __init__(self, key, atom, c, r, X, chain=()):
        # chain: the input warps applied on the way down, outermost first, as (warp, inputs before it)
        self.key, self.atom, self.c, self.r, self.X, self.chain = key, atom, c, r, X, chain


class _ScaleVector(np.ndarray):
    """The row-scale vector r of a path (a product of sigma.(x) factors) that remembers its factors:
    `factors` = [(node, x, values)] with node the `sigma * f` process, x the inputs sigma was mapped over and
    values = sigma.(x) -- what the chain rule of the scale gradients needs (finite_gp.logpdf_and_gradient)."""

    def __new__(cls, values, factors):
        obj = np.asarray(values, dtype=np.float64).view(cls)
        obj.factors = list(factors)
        return obj

    def __array_finalize__(self, obj):
        self.factors = getattr(obj, "factors", [])


class _MatCache:
    """as_matrix with identity caching so equal inputs are uploaded once -- and the memo of the walk down the tree:
    a shared sub-tree is reached along several paths (f5 = f3 + f4 with f4 = f1 + f3: the reference's recursion
    re-evaluates it once per path, test/affine_transformations/addition.jl:7-9); the warped inputs of a compose node
    and the scale vector of a `sigma * f` node are computed ONCE per (node, inputs[, incoming scale]) and the same
    object is handed to every path, so that paths which read the same points with the same scales merge into one
    kernel term (identity keys) instead of one term per path."""

    def __init__(self):
        self.m = {}
        self.keep = []
        self.warped = {}
        self.scaled = {}

    def warp(self, f, g, x):
        k = (id(f), id(x))
        if k not in self.warped:
            self.warped[k] = _gp.warp(g, x)
            self.keep.append((f, x))
        return self.warped[k]

    def scale(self, f, s, x, r):
        k = (id(f), id(x), id(r) if r is not None else None)
        if k not in self.scaled:
            sx = np.asarray(_gp._map_points(s, x), dtype=np.float64)
            fac = ([] if r is None else list(r.factors)) + [(f, x, sx)]
            self.scaled[k] = _ScaleVector(sx if r is None else np.asarray(r) * sx, fac)
            self.keep.append((f, x, r))
        return self.scaled[k]

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
    if is_pair_vector(x):
        # (key, value) pairs on their way to a NESTED programme below f (gppp.jl:32-43 regroups them by key when they reach
        # it, changing the element order): regrouped here, so that every block reads ONE inner process
        x = regroup_pairs(x)
    if isinstance(x, BlockData):
        # BlockData is an ordinary AbstractVector for any GP (input_collection_types.jl:61-95): the
        # same process evaluated on each block
        out = []
        for b in blocks(x):
            out.extend(block_list(f, b))
        return out
    return [(f, x)]


def _paths(f, x, c, r, key, mat, chain=()):
    if isinstance(f, GPPP):  # a GPPP wrapped in atomic(...) (nested programmes, test gppp.jl:107-120)
        node, v = extract_components(f, x)
        if isinstance(node, _gp.DerivedGP) and node.args[0] == "cross":
            # (block_list splits BlockData and regroups pair vectors before any path is walked)
            raise ValueError("internal: a nested GPPP reached with inputs of several of its processes")
        return _paths(node, v, c, r, key, mat, chain)
    if isinstance(f, _gp.AtomicGP):
        k2 = key + (id(f),)
        if isinstance(f.gp, _gp.GP):
            return [_Path(k2, f, c, r, mat(x), chain)]
        return _paths(f.gp, x, c, r, k2, mat, chain)
    op = f.args[0]
    if op == "+":
        return _paths(f.args[1], x, c, r, key, mat, chain) + _paths(f.args[2], x, c, r, key, mat, chain)
    if op == "+known":
        return _paths(f.args[2], x, c, r, key, mat, chain)
    if op == "*":
        s = f.args[1]
        if _gp._is_real(s):
            return _paths(f.args[2], x, c * float(s), r, key, mat, chain)
        return _paths(f.args[2], x, c, mat.scale(f, s, x, r), key, mat, chain)
    if op == "o":
        return _paths(f.args[1], mat.warp(f, f.args[2], x), c, r, key, mat, chain + ((f.args[2], x),))
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
            q = _Path(p.key, p.atom, p.c, p.r, p.X, p.chain)
            index[k] = q
            out.append(q)
    # paths that cancel exactly (f - f, 2 f - f - f, ...) leave no term: the block is an exact zero, as in the
    # reference's recursion, which adds and subtracts identical matrices
    return [q for q in out if q.c != 0.0]


class _InputTable:
    def __init__(self):
        self.arrays, self.index, self.origin = [], {}, []

    def get(self, X, chain, origin=None):
        """chain: the kernel-level input transformation (kernels.apply_chain); origin = (side, block
        index, warp chain): where the points came from (for the chain rule of input gradients); the
        first path that registers an array names it."""
        k = (id(X), chain)
        if k not in self.index:
            self.index[k] = len(self.arrays)
            self.arrays.append(X if not chain else _kernels.apply_chain(chain, X))
            self.origin.append(origin + (chain, X) if origin is not None else None)
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
                        ri = table.get(p.X, s, ("row", I, p.chain))
                        ci = table.get(q.X, s, ("row" if symmetric else "col", J, q.chain))
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
    spec.input_origin = table.origin   # per spec input: (side, block, warp chain, kernel input chain, raw points)
    spec.block_shapes = ([_leaf_shape(v) for _, v in rows], [_leaf_shape(v) for _, v in cols])
    return spec, rows, cols


def _leaf_shape(v):
    while isinstance(v, GPPPInput):   # nested programmes index a process of the inner GPPP
        v = v.x
    return as_matrix(v).shape


def zero_spec(n):
    """A spec with no terms: K == 0 exactly (used to feed an explicit covariance as dense noise)."""
    return _lib.Spec([n], [n], [], {}, True)


# ---- chain rule of input gradients through the host-side transformations ------------------------
def _warp_vjp(g, x_in, gout):
    """Cotangent of the warp's input given the cotangent `gout` (dim_out x n) of its output."""
    if isinstance(g, _gp.Stretch):
        if np.ndim(g.l) == 0:
            return float(g.l) * gout
        return np.asarray(g.l, dtype=np.float64).T @ gout
    if isinstance(g, _gp.Shift):
        return gout
    if isinstance(g, _gp.Select):
        X = as_matrix(x_in)
        gin = np.zeros(X.shape)
        if isinstance(g.idx, (int, np.integer)):
            gin[g.idx, :] += gout.reshape(-1)
        else:
            np.add.at(gin, np.asarray(g.idx), gout)
        return gin
    if isinstance(g, _gp.Periodic):
        t = (2.0 * np.pi * g.f) * as_matrix(x_in)          # 1 x n; output rows are [cos t; sin t]
        return (2.0 * np.pi * g.f) * (-np.sin(t) * gout[0:1, :] + np.cos(t) * gout[1:2, :])
    raise NotImplementedError("input gradients through an arbitrary point-wise warp: supply its Jacobian yourself")


def chain_input_gradients(spec, grads):
    """Map gradients w.r.t. the spec inputs (the transformed points the terms read, as returned by
    sgp_*_grad_x) back onto the blocks the spec was built from: the kernel's input scale
    (ScaleTransform / with_lengthscale) and the Stretch / Select / Periodic / Shift warps of the
    model are undone in reverse.  Returns (row_block_grads, col_block_grads): lists of (D, n) arrays
    aligned with the blocks of x (and of x2 for a cross spec; for a symmetric spec the second list
    is the first)."""
    rshapes, cshapes = spec.block_shapes
    rows = [np.zeros(sh) for sh in rshapes]
    cols = rows if spec.symmetric else [np.zeros(sh) for sh in cshapes]
    for k, g in enumerate(grads):
        org = spec.input_origin[k]
        if org is None or g is None:
            continue
        side, I, chain, kchain, X_raw = org
        gg = _kernels.chain_vjp(kchain, X_raw, g) if kchain else np.asarray(g, dtype=np.float64)
        for (w, x_in) in reversed(chain):
            gg = _warp_vjp(w, x_in, gg)
        (rows if side == "row" else cols)[I] += gg.reshape((rows if side == "row" else cols)[I].shape)
    return rows, cols
