"""Block orders that keep the Cholesky factor of a programme's covariance sparse.

The factorisation skips the tile products that independent components make exactly zero (DESIGN.md section 3.3c,
include/sthenomi.h: sgp_ctx_factor_work) -- but WHICH tiles of the factor are zero depends on the order of the blocks in the
caller's BlockData, exactly as for a sparse direct solver: f3 = f1 + f2 observed in the order (f1, f2, f3) keeps the (f2, f1)
block of the factor zero, the order (f3, f1, f2) fills it in.  The library never permutes (the factor's layout is part of
what `posterior` keeps and what `rand` multiplies a draw with); this module tells the caller a good order and applies it:
logpdf, posterior moments and the ELBO do not depend on the order of the observations beyond rounding.

    perm = fill_reducing_order(f, x)                      # a permutation of the blocks of x
    x2, (y2,), noise2 = permute_blocks(x, perm, y, noise=noise)  # the same observations, reordered
    logpdf(f(x2, noise2), y2)

No reference analogue: Stheno builds the dense matrix and LAPACK factors it whatever the order."""
import numpy as np

from . import flatten as _fl
from .inputs import BlockData, blocks


def block_atoms(f, x):
    """Per block of x: the set of independent atoms (identity keys of the flattener's paths) its process depends on.  Two
    blocks have a non-zero covariance block exactly when their sets intersect."""
    mat = _fl._MatCache()
    return [frozenset(p.key for p in _fl._merge_paths(_fl._paths(n, v, 1.0, None, (), mat))) for n, v in _fl.block_list(f, x)]


def fill_reducing_order(f, x):
    """A permutation of the blocks of x under which symbolic elimination of the block graph creates the least fill, weighted
    by block sizes (greedy minimum fill; ties: the smaller block first, then the caller's order).  Blocks are the units
    block_list(f, x) yields -- for a GPPP indexed by a BlockData of GPPPInputs, the GPPPInputs."""
    atoms = block_atoms(f, x)
    sizes = [len(v) for _, v in _fl.block_list(f, x)]
    n = len(atoms)
    adj = [set(j for j in range(n) if j != i and atoms[i] & atoms[j]) for i in range(n)]
    left, order = set(range(n)), []
    while left:
        best, best_key = None, None
        for v in sorted(left):
            nb = sorted(adj[v] & left)
            fill = sum(sizes[a] * sizes[b] for ia, a in enumerate(nb) for b in nb[ia + 1:] if b not in adj[a])
            key = (fill, sizes[v], v)
            if best_key is None or key < best_key:
                best, best_key = v, key
        nb = sorted(adj[best] & left)
        for ia, a in enumerate(nb):          # eliminating `best` couples its remaining neighbours
            for b in nb[ia + 1:]:
                adj[a].add(b)
                adj[b].add(a)
        left.remove(best)
        order.append(best)
    return order


_NO_NOISE = object()


def suggest_order_capi(f, x):
    """The same suggestion through the C-ABI (sgp_cov_spec_suggest_order: what a host without this Python mirror calls --
    julia/SthenoMI355X.jl does): (perm, changes) with changes = True when the suggested order skips more than the given one.
    Host-only arithmetic inside the library: no GPU, no context."""
    import ctypes as C

    from . import lib as _lib
    from .finite_gp import _prior_spec
    spec = _prior_spec(f, x)
    nb = len(spec.row_len)
    perm = (C.c_int32 * nb)()
    ch = C.c_int32()
    _lib.check(_lib.load().sgp_cov_spec_suggest_order(spec.ref(), perm, C.byref(ch)), "sgp_cov_spec_suggest_order")
    return [int(v) for v in perm], bool(ch.value)


def permute_blocks(x, perm, *vectors, noise=_NO_NOISE):
    """x with its blocks in the order `perm`, and every vector of per-observation values (observations y, a diagonal of
    noise variances; scalars and None pass through; an N x S matrix is permuted by rows) reordered with it.

    `noise=` takes Sigma_y in any of the forms f(x, Sigma_y) does -- scalar, length-N diagonal or dense N x N -- and permutes
    a dense matrix by rows AND columns; with it the result is (x2, vectors2, noise2).  A positional square N x N argument is
    refused: whether it is N right-hand sides (rows only) or a covariance (both sides) cannot be told from its shape."""
    bl = blocks(x) if isinstance(x, BlockData) else None
    if bl is None or sorted(perm) != list(range(len(bl))):
        raise ValueError("permute_blocks: x must be a BlockData and perm a permutation of its blocks")
    lens = [len(b) for b in bl]
    offs = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in perm]) if perm else np.zeros(0, dtype=int)
    n = int(offs[-1])
    out = []
    for v in vectors:
        if v is None or np.ndim(v) == 0:
            out.append(v)
            continue
        a = np.asarray(v)
        if a.shape[0] != n:
            raise ValueError("permute_blocks: a vector's length is not the number of observations")
        if a.ndim == 2 and a.shape[1] == n and n > 1:
            raise ValueError("permute_blocks: a square N x N argument is ambiguous -- pass a dense Sigma_y as noise=, "
                             "N right-hand sides column by column")
        out.append(a[idx] if a.ndim == 1 else np.asfortranarray(a[idx, :]))
    x2 = BlockData([bl[i] for i in perm])
    if noise is _NO_NOISE:
        return x2, tuple(out)
    if noise is None or np.ndim(noise) == 0:
        noise2 = noise
    else:
        a = np.asarray(noise)
        if a.shape[0] != n or (a.ndim == 2 and a.shape[1] != n) or a.ndim > 2:
            raise ValueError("permute_blocks: noise must be a scalar, a length-N diagonal or an N x N matrix")
        noise2 = a[idx] if a.ndim == 1 else np.asfortranarray(a[np.ix_(idx, idx)])
    return x2, tuple(out), noise2
