"""GP node types and affine transformations of the drop-in surface (host side).

Same names and semantics as the reference's Julia surface:
  GP(mean, kernel)                [EXT] AbstractGPs.GP
  GPC, atomic, AtomicGP            /root/reference/src/gp/util.jl:18-25, gp/atomic_gp.jl:11-22
  DerivedGP                        src/gp/derived_gp.jl:7-29
  +, -, *                          src/affine_transformations/addition.jl, product.jl
  compose (the Julia `∘`), stretch, select, periodic, shift   compose.jl:8-127
  additive_gp                      additive_gp.jl:10-29
  cross                            cross.jl:37-45

Nothing here touches a covariance matrix: the tree is only *described* on the host.  Means are
evaluated on the host (O(N), as SURVEY.md Appendix B prescribes); covariances are flattened
into kernel terms (flatten.py) and evaluated by the HIP library.
"""
from __future__ import annotations

import numpy as np

from .inputs import BlockData, ColVecs, GPPPInput, as_matrix, blocks, is_pair_vector, regroup_pairs
from .kernels import Kernel


# ---- leaf GP ([EXT] AbstractGPs.GP) -----------------------------------------------------------
class GP:
    """GP(kernel) | GP(c::Real, kernel) | GP(meanfunction, kernel)."""

    def __init__(self, *args):
        if len(args) == 1:
            self.mean_spec, self.kernel = None, args[0]
        elif len(args) == 2:
            self.mean_spec, self.kernel = args
        else:
            raise TypeError("GP(kernel) or GP(mean, kernel)")
        if not isinstance(self.kernel, Kernel):
            raise TypeError("GP needs a Kernel")

    def mean_vector(self, x):
        n = len(x)
        m = self.mean_spec
        if m is None:
            return np.zeros(n)
        if isinstance(m, (int, float, np.integer, np.floating)):
            return np.full(n, float(m))
        return _map_points(m, x)


def _map_points(g, x):
    """g.(x): g maps one input (a float, or a D-vector for ColVecs) to a float."""
    if isinstance(x, ColVecs):
        return np.array([float(g(x.X[:, i])) for i in range(len(x))], dtype=np.float64)
    return np.array([float(g(float(v))) for v in np.asarray(x)], dtype=np.float64)


def _is_real(s):
    return isinstance(s, (int, float, np.integer, np.floating))


# ---- bookkeeping (gp/util.jl:18-25) -------------------------------------------------------------
class GPC:
    """GP collection: hands out creation indices; all nodes of one model share one GPC."""

    def __init__(self):
        self.n = 0


class SthenoAbstractGP:
    gpc: GPC
    n: int

    def __call__(self, x, noise=1e-18):
        from .finite_gp import FiniteGP
        return FiniteGP(self, x, noise)

    def __add__(self, other):
        if isinstance(other, SthenoAbstractGP):
            assert self.gpc is other.gpc, "GPs must share a GPC"
            return DerivedGP(("+", self, other), self.gpc)
        return DerivedGP(("+known", other, self), self.gpc)

    def __radd__(self, other):
        return DerivedGP(("+known", other, self), self.gpc)

    def __neg__(self):
        return DerivedGP(("*", -1.0, self), self.gpc)       # product.jl:73

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return other + (-self)

    def __mul__(self, s):
        if isinstance(s, SthenoAbstractGP):
            raise ValueError("Cannot multiply two GPs together.")  # product.jl:13
        return DerivedGP(("*", s, self), self.gpc)

    __rmul__ = __mul__


class AtomicGP(SthenoAbstractGP):
    def __init__(self, gp, gpc):
        self.gp, self.gpc = gp, gpc
        self.n = gpc.n + 1
        gpc.n += 1


def atomic(gp, gpc):
    return AtomicGP(gp, gpc)


class DerivedGP(SthenoAbstractGP):
    def __init__(self, args, gpc):
        self.args, self.gpc = args, gpc
        self.n = gpc.n + 1
        gpc.n += 1


# ---- input warps (compose.jl:36-127) -----------------------------------------------------------
class Stretch:
    def __init__(self, l):
        self.l = l

    def __call__(self, x):
        if isinstance(x, ColVecs):
            if np.ndim(self.l) == 0:
                return ColVecs(self.l * x.X)
            return ColVecs(np.asarray(self.l, dtype=np.float64) @ x.X)
        return self.l * np.asarray(x, dtype=np.float64)


class Select:
    def __init__(self, idx):
        self.idx = idx

    def __call__(self, x):
        if isinstance(self.idx, (int, np.integer)):
            return np.array(x.X[self.idx, :], dtype=np.float64)
        return ColVecs(x.X[np.asarray(self.idx), :])


class Periodic:
    def __init__(self, f):
        self.f = float(f)

    def __call__(self, x):
        t = (2.0 * np.pi * self.f) * np.asarray(x, dtype=np.float64)
        return ColVecs(np.vstack([np.cos(t), np.sin(t)]))


class Shift:
    def __init__(self, a):
        self.a = a

    def __call__(self, x):
        if isinstance(x, ColVecs):
            a = np.asarray(self.a, dtype=np.float64)
            return ColVecs(x.X - (a[:, None] if a.ndim == 1 else a))
        return np.asarray(x, dtype=np.float64) - self.a


def warp(g, x):
    """g.(x) for the structured warps (fast broadcasts) or any point-wise function."""
    if isinstance(g, (Stretch, Select, Periodic, Shift)):
        return g(x)
    if isinstance(x, ColVecs):
        vals = [g(x.X[:, i]) for i in range(len(x))]
    else:
        vals = [g(float(v)) for v in np.asarray(x)]
    if np.ndim(vals[0]) == 0:
        return np.array(vals, dtype=np.float64)
    return ColVecs(np.stack([np.asarray(v, dtype=np.float64) for v in vals], axis=1))


def compose(f, g):
    """f ∘ g : the DerivedGP f'(x) := f(g(x))."""
    return DerivedGP(("o", f, g), f.gpc)


def stretch(f, l):
    if np.ndim(l) == 1:
        l = np.diag(np.asarray(l, dtype=np.float64))
    return compose(f, Stretch(l))


def select(f, idx):
    return compose(f, Select(idx))


def periodic(f, freq):
    return compose(f, Periodic(freq))


def shift(f, a):
    return compose(f, Shift(a))


def additive_gp(fs, indices=None):
    if indices is None:
        indices = list(range(len(fs)))
    proj = [compose(f, Select(idx)) for f, idx in zip(fs, indices)]
    out = proj[0]
    for p in proj[1:]:
        out = out + p
    return out


def cross(fs):
    fs = list(fs)
    assert len(fs) >= 1 and all(f.gpc is fs[0].gpc for f in fs)
    return DerivedGP(("cross", fs), fs[0].gpc)


# ---- prior mean (host, O(N) per node) ----------------------------------------------------------
def mean_vector(f, x):
    """mean(f, x) by the reference's recursion (addition.jl:26,73-74; product.jl:25,54;
    compose.jl:16; cross.jl:54-57; atomic_gp.jl:28)."""
    from .gppp import GPPP, extract_components
    if isinstance(f, GPPP):
        node, v = extract_components(f, x)
        return mean_vector(node, v)
    if is_pair_vector(x):     # pairs on their way to a nested programme: regrouped by key (gppp.jl:32-43), as flatten.block_list does
        x = regroup_pairs(x)
    if isinstance(x, BlockData) and not (isinstance(f, DerivedGP) and f.args[0] == "cross"):
        # BlockData is an ordinary AbstractVector for any GP: the same process on each block
        return np.concatenate([mean_vector(f, b) for b in blocks(x)]) if len(blocks(x)) else np.zeros(0)
    if isinstance(f, AtomicGP):
        if isinstance(f.gp, GP):
            return f.gp.mean_vector(x)
        return mean_vector(f.gp, x)          # a wrapped GPPP / other Stheno GP
    op = f.args[0]
    if op == "+":
        return mean_vector(f.args[1], x) + mean_vector(f.args[2], x)
    if op == "+known":
        b = f.args[1]
        return (float(b) if _is_real(b) else _map_points(b, x)) + mean_vector(f.args[2], x)
    if op == "*":
        s = f.args[1]
        return (float(s) if _is_real(s) else _map_points(s, x)) * mean_vector(f.args[2], x)
    if op == "o":
        return mean_vector(f.args[1], warp(f.args[2], x))
    if op == "cross":
        return np.concatenate([mean_vector(g, b) for g, b in zip(f.args[1], blocks(x))])
    raise ValueError(op)
