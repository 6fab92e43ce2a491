"""Kernel descriptions of the drop-in surface (the KernelFunctions.jl names Stheno re-exports,
/root/reference/src/Stheno.jl:4-6).  Pure descriptions: no arithmetic happens here -- a kernel
expands into (kind, coef, param, input_chain) leaf terms that the HIP assembly kernel
evaluates (stheno.jl_amd/csrc/kernelmatrix.hip).  `input_chain` is the kernel-level input
transformation (KernelFunctions `TransformedKernel`): a tuple of steps applied to the raw points in
order -- ("scale", s) for ScaleTransform(s) / with_lengthscale, ("periodic", f) for
PeriodicTransform(f) (/root/reference/examples/extended_mauna_loa/script.jl:129) -- evaluated on the
host (O(N D)) when the spec is built.
"""
from __future__ import annotations

import numpy as np

from . import lib as _lib


def _push(step, chain):
    """chain with `step` applied FIRST (outer transforms act on the raw input before inner ones);
    adjacent scalings are merged so that equal maps get equal chains."""
    if chain and step[0] == "scale" and chain[0][0] == "scale":
        s = step[1] * chain[0][1]
        return ((("scale", s),) if s != 1.0 else ()) + tuple(chain[1:])
    if step[0] == "scale" and step[1] == 1.0:
        return tuple(chain)
    return (step,) + tuple(chain)


def apply_chain(chain, X):
    """the D x n (ColVecs-layout) points the kernel reads, from the raw D x n points X"""
    for kind, v in chain:
        if kind == "scale":
            X = v * X
        elif kind == "periodic":
            if X.shape[0] != 1:
                raise ValueError("PeriodicTransform acts on 1-D inputs")
            t = (2.0 * np.pi * v) * X
            X = np.vstack([np.sin(t), np.cos(t)])      # KernelFunctions order: [sin, cos]
        else:
            raise ValueError(kind)
    return np.asfortranarray(X)


def chain_vjp(chain, X, gout):
    """cotangent of the raw points given the cotangent of apply_chain(chain, X)"""
    stack = [X]
    for kind, v in chain[:-1]:
        stack.append(apply_chain(((kind, v),), stack[-1]))
    g = np.asarray(gout, dtype=np.float64)
    for (kind, v), xin in zip(reversed(chain), reversed(stack)):
        if kind == "scale":
            g = v * g
        else:
            t = (2.0 * np.pi * v) * xin
            g = (2.0 * np.pi * v) * (np.cos(t) * g[0:1, :] - np.sin(t) * g[1:2, :])
    return g


def chain_scale(chain):
    """the scale of a pure-scaling chain (1.0 for the empty chain); None if the chain is not a scaling"""
    if not chain:
        return 1.0
    if len(chain) == 1 and chain[0][0] == "scale":
        return chain[0][1]
    return None


class Kernel:
    def __add__(self, other):
        return KernelSum([self, other])

    def __mul__(self, s):
        return ScaledKernel(self, float(s))

    __rmul__ = __mul__

    def __matmul__(self, transform):
        """k @ t  ==  k ∘ t  (TransformedKernel)"""
        return TransformedKernel(self, transform)

    def leaf_terms(self):
        """list of (kind, coef, param, input_chain)"""
        raise NotImplementedError


class _Simple(Kernel):
    kind = None

    def leaf_terms(self):
        return [(self.kind, 1.0, 0.0, ())]


class SEKernel(_Simple):
    kind = _lib.SE


SqExponentialKernel = SEKernel


class Matern12Kernel(_Simple):
    kind = _lib.MATERN12


ExponentialKernel = Matern12Kernel


class Matern32Kernel(_Simple):
    kind = _lib.MATERN32


class Matern52Kernel(_Simple):
    kind = _lib.MATERN52


class WhiteKernel(_Simple):
    kind = _lib.WHITE


class ConstantKernel(Kernel):
    def __init__(self, c=1.0):
        self.c = float(c)

    def leaf_terms(self):
        return [(_lib.CONST, 1.0, self.c, ())]


class ScaledKernel(Kernel):
    def __init__(self, kernel, s2):
        self.kernel, self.s2 = kernel, float(s2)

    def leaf_terms(self):
        return [(k, c * self.s2, p, s) for (k, c, p, s) in self.kernel.leaf_terms()]


class KernelSum(Kernel):
    def __init__(self, kernels):
        self.kernels = list(kernels)

    def leaf_terms(self):
        out = []
        for k in self.kernels:
            out.extend(k.leaf_terms())
        return out


class ScaleTransformedKernel(Kernel):
    """k o ScaleTransform(s)"""

    def __init__(self, kernel, s):
        self.kernel, self.s = kernel, float(s)

    def leaf_terms(self):
        return [(k, c, p, _push(("scale", self.s), ch)) for (k, c, p, ch) in self.kernel.leaf_terms()]


class ScaleTransform:
    def __init__(self, s):
        self.s = float(s)

    def step(self):
        return ("scale", self.s)


class PeriodicTransform:
    """x -> [sin(2 pi f x), cos(2 pi f x)] for 1-D inputs (KernelFunctions.PeriodicTransform [EXT];
    Stheno's own `periodic(f, freq)` warp uses [cos, sin]: same kernel values)."""

    def __init__(self, f):
        self.f = float(f)

    def step(self):
        return ("periodic", self.f)


class TransformedKernel(Kernel):
    """k ∘ t for t a ScaleTransform or PeriodicTransform (written `k @ t` here)."""

    def __init__(self, kernel, transform):
        self.kernel, self.transform = kernel, transform

    def leaf_terms(self):
        st = self.transform.step()
        return [(k, c, p, _push(st, ch)) for (k, c, p, ch) in self.kernel.leaf_terms()]


def with_lengthscale(kernel, l):
    return ScaleTransformedKernel(kernel, 1.0 / float(l))
