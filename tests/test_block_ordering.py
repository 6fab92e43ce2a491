"""Host logic (no GPU): stheno.jl_amd/ordering.py -- the block order under which the factor of a programme's covariance
keeps its structural zeros, and the permutation of the observations that goes with it.  The
values are checked on the NumPy double of the C-ABI; what the order buys on the device is in
tests/test_gpu_struct_zeros.py::test_reordering_the_blocks_restores_the_skipping."""
import numpy as np
import pytest

import np_capi
import stheno_jl_amd as P


def _fill(atoms, order):
    """Fill-in (pairs of blocks coupled only by elimination) of eliminating the blocks in `order`."""
    n = len(atoms)
    adj = [set(j for j in range(n) if j != i and atoms[i] & atoms[j]) for i in range(n)]
    left, fill = set(range(n)), 0
    for v in order:
        nb = sorted(adj[v] & left)
        for ia, a in enumerate(nb):
            for b in nb[ia + 1:]:
                if b not in adj[a]:
                    fill += 1
                    adj[a].add(b)
                    adj[b].add(a)
        left.remove(v)
    return fill


def test_sum_model_orders_without_fill_from_any_starting_order():
    F = P.gppp_sum_model()
    rng = np.random.default_rng(0)
    xs = {k: P.ColVecs(np.asfortranarray(rng.standard_normal((2, n)))) for k, n in (("f1", 30), ("f2", 40), ("f3", 25))}
    for order in (("f1", "f2", "f3"), ("f3", "f1", "f2"), ("f2", "f3", "f1"), ("f3", "f2", "f1")):
        x = P.BlockData([P.GPPPInput(k, xs[k]) for k in order])
        atoms = P.block_atoms(F, x)
        assert sorted(len(a) for a in atoms) == [1, 1, 2]
        perm = P.fill_reducing_order(F, x)
        assert sorted(perm) == [0, 1, 2]
        assert _fill(atoms, perm) == 0
        assert order[perm[-1]] != "f3" or _fill(atoms, perm) == 0
    # the caller's order (f3, f1, f2) itself fills in: f1 and f2 are coupled through f3's column
    x = P.BlockData([P.GPPPInput(k, xs[k]) for k in ("f3", "f1", "f2")])
    assert _fill(P.block_atoms(F, x), [0, 1, 2]) == 1


def test_a_star_of_sums_is_ordered_leaves_first():
    gpc = P.GPC()
    atoms = [P.atomic(P.GP(P.SEKernel()), gpc) for _ in range(4)]
    total = atoms[0] + atoms[1] + atoms[2] + atoms[3]
    F = P.GPPP({"t": total, "a0": atoms[0], "a1": atoms[1], "a2": atoms[2], "a3": atoms[3]}, gpc)
    rng = np.random.default_rng(1)
    x = P.BlockData([P.GPPPInput(k, rng.standard_normal(10 + 3 * i)) for i, k in enumerate(("t", "a0", "a1", "a2", "a3"))])
    perm = P.fill_reducing_order(F, x)
    assert perm[-1] == 0 or _fill(P.block_atoms(F, x), perm) == 0      # the hub goes last (or at least nothing fills)
    assert _fill(P.block_atoms(F, x), perm) == 0
    assert _fill(P.block_atoms(F, x), [0, 1, 2, 3, 4]) == 6            # hub first: every pair of leaves fills in


def test_values_do_not_depend_on_the_order(monkeypatch):
    np_capi.install(monkeypatch)
    F = P.gppp_sum_model()
    rng = np.random.default_rng(2)
    order = ("f3", "f1", "f2")
    xs = {k: P.ColVecs(np.asfortranarray(rng.standard_normal((2, n)))) for k, n in (("f1", 40), ("f2", 55), ("f3", 33))}
    x = P.BlockData([P.GPPPInput(k, xs[k]) for k in order])
    N = len(x)
    y = rng.standard_normal(N)
    noise = 0.2 + rng.random(N)
    Y = np.asfortranarray(rng.standard_normal((N, 3)))
    perm = P.fill_reducing_order(F, x)
    x2, (y2, noise2, Y2, s2, none) = P.permute_blocks(x, perm, y, noise, Y, 0.3, None)
    assert s2 == 0.3 and none is None and len(x2) == N
    assert [b.p for b in P.blocks(x2)] == [order[i] for i in perm]
    lp, lp2 = P.logpdf(F(x, noise), y), P.logpdf(F(x2, noise2), y2)
    assert abs(lp - lp2) <= 1e-11 * abs(lp)
    np.testing.assert_allclose(P.logpdf(F(x, 0.3), Y), P.logpdf(F(x2, 0.3), Y2), rtol=1e-11)
    t = P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((2, 9)))))
    m, v = P.posterior(F(x, noise), y).mean_and_var(t)
    m2, v2 = P.posterior(F(x2, noise2), y2).mean_and_var(t)
    np.testing.assert_allclose(m, m2, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v, v2, rtol=1e-9, atol=1e-11)
    with pytest.raises(ValueError):
        P.permute_blocks(x, [0, 0, 1], y)
    with pytest.raises(ValueError):
        P.permute_blocks(x, perm, y[:-1])
    # a dense Sigma_y goes through noise= and is permuted on BOTH sides (advisor, round 4: rows only was a silently wrong
    # model); a positional square argument is refused
    Bm = rng.standard_normal((N, N))
    S = np.asfortranarray(0.05 * Bm @ Bm.T / N + np.diag(noise))
    x3, (y3,), S3 = P.permute_blocks(x, perm, y, noise=S)
    lp3 = P.logpdf(F(x3, S3), y3)
    assert abs(P.logpdf(F(x, S), y) - lp3) <= 1e-10 * abs(lp3)
    x4, (y4,), d4 = P.permute_blocks(x, perm, y, noise=noise)
    np.testing.assert_array_equal(d4, noise2)
    assert P.permute_blocks(x, perm, noise=0.3)[2] == 0.3
    with pytest.raises(ValueError):
        P.permute_blocks(x, perm, S)


def test_the_c_abi_suggests_the_same_order_as_the_python_mirror():
    """sgp_cov_spec_suggest_order (round 5: the helper lives in the library, where every host gets it) against
    fill_reducing_order on the sum model from every starting order, the star of sums, and random programmes of sums of
    random subsets of atoms with ragged block sizes.  Host-only arithmetic: runs without a GPU."""
    F = P.gppp_sum_model()
    rng = np.random.default_rng(3)
    xs = {k: P.ColVecs(np.asfortranarray(rng.standard_normal((2, n)))) for k, n in (("f1", 30), ("f2", 40), ("f3", 25))}
    for order in (("f1", "f2", "f3"), ("f3", "f1", "f2"), ("f2", "f3", "f1"), ("f3", "f2", "f1")):
        x = P.BlockData([P.GPPPInput(k, xs[k]) for k in order])
        perm, changes = P.suggest_order_capi(F, x)
        assert perm == P.fill_reducing_order(F, x)
        assert changes == (_fill(P.block_atoms(F, x), list(range(3))) > 0)
    for trial in range(25):
        gpc = P.GPC()
        na = int(rng.integers(2, 5))
        atoms = [P.atomic(P.GP(P.SEKernel()), gpc) for _ in range(na)]
        procs = {}
        for j in range(int(rng.integers(2, 7))):
            pick = [a for a in atoms if rng.random() < 0.5] or [atoms[int(rng.integers(na))]]
            g = pick[0]
            for a in pick[1:]:
                g = g + a
            procs[f"p{j}"] = g
        Fm = P.GPPP(procs, gpc)
        x = P.BlockData([P.GPPPInput(k, rng.standard_normal(int(rng.integers(3, 30)))) for k in procs])
        perm, changes = P.suggest_order_capi(Fm, x)
        ref = P.fill_reducing_order(Fm, x)
        assert perm == ref, (trial, perm, ref)
        at = P.block_atoms(Fm, x)
        assert _fill(at, perm) <= _fill(at, list(range(len(at))))
