"""Input collection types of the drop-in surface.

Mirrors /root/reference/src/input_collection_types.jl (GPPPInput :24-33, BlockData :61-95,
blocks :82) and KernelFunctions.ColVecs ([EXT], docs/src/input_types.md:48-55): a ColVecs
wraps a D x N matrix whose *columns* are the inputs -- the exact HBM layout the library
consumes (each point D contiguous doubles).
"""
from __future__ import annotations

import numpy as np


class ColVecs:
    def __init__(self, X):
        # Float32 inputs keep their element type as a tag (type stability of the reference in Float32:
        # test/gp/util.jl:76-88 -- the FiniteGP operators then run the fp32 device path); the host algebra
        # (warps, flattening) works on the exactly converted fp64 copy.
        self.eltype = np.float32 if getattr(X, "dtype", None) == np.float32 else np.float64
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("ColVecs needs a D x N matrix")
        self.X = X

    def __len__(self):
        return self.X.shape[1]

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.X[:, idx]
        return ColVecs(self.X[:, idx])

    def __eq__(self, other):
        return isinstance(other, ColVecs) and np.array_equal(self.X, other.X)


def is_pair_vector(x):
    """a generic vector of (process key, input) pairs -- the form gppp.jl:32-43 regroups by key"""
    return isinstance(x, (list, tuple)) and len(x) > 0 and all(
        isinstance(e, tuple) and len(e) == 2 and isinstance(e[0], (str, bytes)) for e in x)


def regroup_pairs(x):
    """BlockData of one GPPPInput per distinct key, in order of first appearance (gppp.jl:32-43: like the reference this
    CHANGES the element order)."""
    uniq = []
    for k, _ in x:
        if k not in uniq:
            uniq.append(k)
    blks = []
    for k in uniq:
        sel = [v for kk, v in x if kk == k]
        if np.ndim(sel[0]) == 0:
            blks.append(GPPPInput(k, np.array(sel, dtype=np.float64)))
        else:
            blks.append(GPPPInput(k, ColVecs(np.stack(sel, axis=1))))
    return BlockData(blks)


class GPPPInput:
    """GPPPInput(p, x): the inputs `x`, to be read from process `p` of a GPPP."""

    def __init__(self, p, x):
        self.p = p
        self.eltype = np.float32 if getattr(x, "dtype", None) == np.float32 else None   # raw 1-D vectors only
        if is_pair_vector(x):    # inputs of a NESTED programme given as (key, value) pairs (gppp.jl:32-43): kept as they are
            self.x = list(x)
        else:
            self.x = x if isinstance(x, (ColVecs, GPPPInput, BlockData)) else np.asarray(x, dtype=np.float64)

    def __len__(self):
        return len(self.x)

    def __iter__(self):
        for i in range(len(self)):
            yield (self.p, self.x[i])


class BlockData:
    """A strictly ordered collection of input vectors (a ragged array of data)."""

    def __init__(self, *xs):
        if len(xs) == 1 and isinstance(xs[0], (list, tuple)):
            xs = tuple(xs[0])
        self.X = list(xs)

    def __len__(self):
        return sum(len(b) for b in self.X)

    def __iter__(self):
        for b in self.X:
            yield from b

    def __eq__(self, other):
        return isinstance(other, BlockData) and len(self.X) == len(other.X) and all(
            _same_inputs(a, b) for a, b in zip(self.X, other.X))


def _same_inputs(a, b):
    if isinstance(a, GPPPInput) and isinstance(b, GPPPInput):
        return a.p == b.p and _same_inputs(a.x, b.x)
    if isinstance(a, ColVecs) or isinstance(b, ColVecs):
        return a == b
    return np.array_equal(np.asarray(a), np.asarray(b))


def blocks(x):
    return x.X


def vcat(*xs):
    """Base.vcat(x::GPPPInput...) = BlockData([...]) (input_collection_types.jl:93-95)."""
    return BlockData(list(xs))


def split(x, Y):
    """Base.split(x::BlockData, Y) (gaussian_process_probabilistic_programme.jl:121-135)."""
    Y = np.asarray(Y)
    if len(x) != Y.shape[0]:
        raise ValueError("Expected length(x) == size(Y, 1)")
    out, o = [], 0
    for b in x.X:
        out.append(Y[o:o + len(b)])
        o += len(b)
    return out


def as_matrix(x):
    """D x N float64 matrix of an input vector (1-D inputs are D = 1)."""
    if isinstance(x, ColVecs):
        return x.X
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 1:
        return a.reshape(1, -1)
    raise TypeError("inputs must be a 1-D real vector or ColVecs")


def eltype(x):
    """np.float32 if every block of the input collection was given in Float32, else np.float64."""
    if isinstance(x, BlockData):
        ts = [eltype(b) for b in x.X]
        return np.float32 if ts and all(t == np.float32 for t in ts) else np.float64
    if isinstance(x, GPPPInput):
        return np.float32 if x.eltype == np.float32 else eltype(x.x)
    if isinstance(x, ColVecs):
        return x.eltype
    return np.float32 if getattr(x, "dtype", None) == np.float32 else np.float64
