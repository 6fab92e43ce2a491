"""Kernel descriptions of the drop-in surface (the KernelFunctions.jl names Stheno re-exports,
/root/reference/src/Stheno.jl:4-6).  Pure descriptions: no arithmetic happens here -- a kernel
expands into (kind, coef, param, input_scale) leaf terms that the HIP assembly kernel
evaluates (stheno.jl_amd/csrc/kernelmatrix.hip).
"""
from __future__ import annotations

from . import lib as _lib


class Kernel:
    def __add__(self, other):
        return KernelSum([self, other])

    def __mul__(self, s):
        return ScaledKernel(self, float(s))

    __rmul__ = __mul__

    def leaf_terms(self):
        """list of (kind, coef, param, input_scale)"""
        raise NotImplementedError


class _Simple(Kernel):
    kind = None

    def leaf_terms(self):
        return [(self.kind, 1.0, 0.0, 1.0)]


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
        return [(_lib.CONST, 1.0, self.c, 1.0)]


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
        return [(k, c, p, s * self.s) for (k, c, p, s) in self.kernel.leaf_terms()]


def with_lengthscale(kernel, l):
    return ScaleTransformedKernel(kernel, 1.0 / float(l))
