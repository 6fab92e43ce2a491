"""Oracle restatement of the KernelFunctions.jl / Distances.jl arithmetic the reference
delegates to.  NumPy fp64.  TEST INFRASTRUCTURE.

[EXT] KernelFunctions.jl compat "0.9.6, 0.10" (/root/reference/Project.toml:19) is not
vendored; the formulas below are its published SimpleKernel definitions.  Call sites in the
reference: src/gp/atomic_gp.jl:28-34 (through `GP` -> kernelmatrix), src/gp/util.jl:5-7
(kernelmatrix_diag), test/gp/atomic_gp.jl:15 (cov(f, x) == kernelmatrix(k, x)).

Input convention: a "vector of inputs" is either a 1-D float array (scalar inputs) or a
`ColVecs` wrapping a D x N matrix (docs/src/input_types.md:48-55).
"""
from __future__ import annotations

import numpy as np


class ColVecs:
    """KernelFunctions.ColVecs: the columns of X (D x N) are the inputs."""

    def __init__(self, X):
        self.X = np.asarray(X, dtype=np.float64)
        assert self.X.ndim == 2

    def __len__(self):
        return self.X.shape[1]

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.X[:, idx]
        return ColVecs(self.X[:, idx])


def as_matrix(x):
    """D x N matrix view of an input vector (1-D inputs become 1 x N)."""
    if isinstance(x, ColVecs):
        return x.X
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        return x.reshape(1, -1)
    raise TypeError("inputs must be a 1-D array or ColVecs")


def n_inputs(x):
    return as_matrix(x).shape[1]


# ---- Distances.pairwise (GEMM trick, reference-faithful) and the direct form ("truth") ----
def pairwise_sqeuclidean(X, Y=None, faithful=True):
    """d2[i, j] = |x_i - y_j|^2.

    faithful=True follows Distances.jl: |a|^2 + |b|^2 - 2 a'b, clamped at 0; the symmetric
    call forces an exactly-zero diagonal and mirrors one triangle (App. A.1).
    faithful=False sums (a_d - b_d)^2 directly (what the HIP kernel does)."""
    sym = Y is None
    Y = X if sym else Y
    if faithful:
        G = X.T @ Y
        d2 = (X * X).sum(0)[:, None] + (Y * Y).sum(0)[None, :] - 2.0 * G
        d2 = np.maximum(d2, 0.0)
        if sym:
            d2 = np.tril(d2) + np.tril(d2, -1).T
            np.fill_diagonal(d2, 0.0)
        return d2
    d2 = np.zeros((X.shape[1], Y.shape[1]))
    for d in range(X.shape[0]):
        df = X[d][:, None] - Y[d][None, :]
        d2 += df * df
    return d2


# ---- kernels ---------------------------------------------------------------------------------
class Kernel:
    def __matmul__(self, transform):
        return TransformedKernel(self, transform)

    def __add__(self, other):
        return KernelSum([self, other])

    def __rmul__(self, s):
        return ScaledKernel(self, float(s))

    def __mul__(self, s):
        return ScaledKernel(self, float(s))


class SimpleKernel(Kernel):
    metric = "sqeuclidean"

    def kappa(self, d):
        raise NotImplementedError

    def matrix(self, X, Y=None, faithful=True):
        d2 = pairwise_sqeuclidean(X, Y, faithful)
        return self.kappa(d2 if self.metric == "sqeuclidean" else np.sqrt(d2))

    def diag(self, X, Y=None):
        if Y is None:
            d2 = np.zeros(X.shape[1])
        else:
            d2 = ((X - Y) ** 2).sum(0)
        return self.kappa(d2 if self.metric == "sqeuclidean" else np.sqrt(d2))


class SEKernel(SimpleKernel):
    """SqExponentialKernel: kappa(d2) = exp(-d2 / 2), metric SqEuclidean."""
    metric = "sqeuclidean"

    def kappa(self, d2):
        return np.exp(-d2 / 2.0)


SqExponentialKernel = SEKernel


class Matern12Kernel(SimpleKernel):
    """Matern12Kernel == ExponentialKernel: exp(-d), metric Euclidean."""
    metric = "euclidean"

    def kappa(self, d):
        return np.exp(-d)


ExponentialKernel = Matern12Kernel


class Matern32Kernel(SimpleKernel):
    metric = "euclidean"

    def kappa(self, d):
        return (1.0 + np.sqrt(3.0) * d) * np.exp(-np.sqrt(3.0) * d)


class Matern52Kernel(SimpleKernel):
    metric = "euclidean"

    def kappa(self, d):
        lam = np.sqrt(5.0) * d
        return (1.0 + lam + lam * lam / 3.0) * np.exp(-lam)


class WhiteKernel(Kernel):
    """delta(x, y): 1 when the inputs are identical."""

    def matrix(self, X, Y=None, faithful=True):
        Y = X if Y is None else Y
        return np.all(X[:, :, None] == Y[:, None, :], axis=0).astype(np.float64)

    def diag(self, X, Y=None):
        if Y is None:
            return np.ones(X.shape[1])
        return np.all(X == Y, axis=0).astype(np.float64)


class ConstantKernel(Kernel):
    def __init__(self, c=1.0):
        self.c = float(c)

    def matrix(self, X, Y=None, faithful=True):
        Y = X if Y is None else Y
        return np.full((X.shape[1], Y.shape[1]), self.c)

    def diag(self, X, Y=None):
        return np.full(X.shape[1], self.c)


class ScaledKernel(Kernel):
    """sigma2 * k  (KernelFunctions.ScaledKernel; `s * k` sugar)."""

    def __init__(self, kernel, s2):
        self.kernel, self.s2 = kernel, float(s2)

    def matrix(self, X, Y=None, faithful=True):
        return self.s2 * self.kernel.matrix(X, Y, faithful)

    def diag(self, X, Y=None):
        return self.s2 * self.kernel.diag(X, Y)


class KernelSum(Kernel):
    def __init__(self, kernels):
        self.kernels = []
        for k in kernels:
            self.kernels.extend(k.kernels if isinstance(k, KernelSum) else [k])

    def matrix(self, X, Y=None, faithful=True):
        out = self.kernels[0].matrix(X, Y, faithful)
        for k in self.kernels[1:]:
            out = out + k.matrix(X, Y, faithful)
        return out

    def diag(self, X, Y=None):
        out = self.kernels[0].diag(X, Y)
        for k in self.kernels[1:]:
            out = out + k.diag(X, Y)
        return out


class ScaleTransformedKernel(Kernel):
    """k o ScaleTransform(s): k(s x, s y).  with_lengthscale(k, l) == this with s = 1 / l."""

    def __init__(self, kernel, s):
        self.kernel, self.s = kernel, float(s)

    def matrix(self, X, Y=None, faithful=True):
        return self.kernel.matrix(self.s * X, None if Y is None else self.s * Y, faithful)

    def diag(self, X, Y=None):
        return self.kernel.diag(self.s * X, None if Y is None else self.s * Y)


def with_lengthscale(kernel, l):
    return ScaleTransformedKernel(kernel, 1.0 / float(l))


class ScaleTransform:
    """KernelFunctions.ScaleTransform(s) [EXT]: x -> s x."""

    def __init__(self, s):
        self.s = float(s)

    def __call__(self, X):
        return self.s * X


class PeriodicTransform:
    """KernelFunctions.PeriodicTransform(f) [EXT]: 1-D x -> [sin(2 pi f x), cos(2 pi f x)]
    (/root/reference/examples/extended_mauna_loa/script.jl:129)."""

    def __init__(self, f):
        self.f = float(f)

    def __call__(self, X):
        assert X.shape[0] == 1
        t = 2.0 * np.pi * self.f * X
        return np.vstack([np.sin(t), np.cos(t)])


class TransformedKernel(Kernel):
    """k o t: k(t(x), t(y)) (KernelFunctions.TransformedKernel [EXT])."""

    def __init__(self, kernel, transform):
        self.kernel, self.transform = kernel, transform

    def matrix(self, X, Y=None, faithful=True):
        return self.kernel.matrix(self.transform(X), None if Y is None else self.transform(Y), faithful)

    def diag(self, X, Y=None):
        return self.kernel.diag(self.transform(X), None if Y is None else self.transform(Y))


def kernelmatrix(k, x, y=None, faithful=True):
    X = as_matrix(x)
    Y = None if y is None else as_matrix(y)
    return k.matrix(X, Y, faithful)


def kernelmatrix_diag(k, x, y=None):
    X = as_matrix(x)
    Y = None if y is None else as_matrix(y)
    return k.diag(X, Y)
