"""sgp_logpdf_batch (round 6): B independent models in one call -- equally sized members are factored by ONE launch of the
dataflow kernel as a single task pool (chol_df.hip: ids dealt round robin over the matrices, progress counters per matrix).
The contract: every member's value is BIT-EQUAL to its own `logpdf` call (same assembly, same k-ascending contractions, same
reductions) and agrees with the CPU oracle; a member that is not positive definite does not lose the others.  The loop it
serves is the multi-restart / cross-validation loop around /root/reference/examples/getting_started/script.jl:154-213."""
import numpy as np
import pytest

import stheno_jl_amd as P
from oracle import reference_model as orm

pytestmark = pytest.mark.gpu


def _members(B, N, D=3, seed=5, kern=None):
    rng = np.random.default_rng(seed)
    fxs, ys, raw = [], [], []
    for b in range(B):
        ell, s2 = 0.5 + rng.random(), 0.05 + 0.2 * rng.random()
        k = kern(ell) if kern else P.with_lengthscale(P.Matern52Kernel(), ell)
        f = P.atomic(P.GP(k), P.GPC())
        x = np.asfortranarray(rng.standard_normal((D, N)))
        y = rng.standard_normal(N)
        fxs.append(f(P.ColVecs(x), s2))
        ys.append(y)
        raw.append((x, y, ell, s2))
    return fxs, ys, raw


@pytest.mark.parametrize("B,N", [(2, 300), (3, 1000), (8, 1536), (16, 640), (5, 4096), (19, 512)])
def test_batch_members_are_bit_equal_to_their_own_calls(B, N):
    fxs, ys, _ = _members(B, N)
    single = np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)])
    got = P.logpdf_batch(fxs, ys)
    assert got.shape == (B,)
    assert np.array_equal(got, single), (got - single)
    assert np.array_equal(P.logpdf_batch(fxs, ys), single)          # repeated call: the pool is reused


def test_batch_against_the_oracle_and_with_diagonal_noise_and_means():
    rng = np.random.default_rng(11)
    F = P.gppp_sum_model()
    fxs, ys, refs = [], [], []
    for b in range(4):
        N = 900
        n1 = N // 3
        xs = [np.asfortranarray(rng.standard_normal((3, n))) for n in (n1, n1, N - 2 * n1)]
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
        y = rng.standard_normal(N)
        s2 = 0.1 + 0.05 * b
        fxs.append(F(x, s2))
        ys.append(y)
        refs.append(orm.gppp_sum_logpdf(xs, y, s2))
    got = P.logpdf_batch(fxs, ys)
    np.testing.assert_allclose(got, refs, rtol=1e-10)
    assert np.array_equal(got, np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)]))
    # diagonal noise
    fxd = [F(fx.x, 0.05 + np.random.default_rng(i).random(len(fx))) for i, fx in enumerate(fxs)]
    assert np.array_equal(P.logpdf_batch(fxd, ys), np.array([P.logpdf(fx, y) for fx, y in zip(fxd, ys)]))


def test_members_of_different_sizes_and_mixed_noise_run_member_by_member():
    f1, y1, _ = _members(2, 500, seed=1)
    f2, y2, _ = _members(2, 700, seed=2)
    fxs, ys = f1 + f2, y1 + y2
    assert np.array_equal(P.logpdf_batch(fxs, ys), np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)]))
    fxs[1] = fxs[1].f(fxs[1].x, 0.1 + np.random.default_rng(0).random(500))      # a diagonal-noise member among scalar ones
    assert np.array_equal(P.logpdf_batch(fxs, ys), np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)]))


def test_folds_that_differ_by_a_few_points_pool_like_equal_sizes():
    """Round 6: members pool when their PADDED size agrees -- the folds of a cross-validation differ by a point or two, not by a
    128-column tile.  Every member keeps the bits of its own call (its own N in the assembly, the row sums and the constant),
    and the call takes a fraction of the member-by-member time (8 chains hiding each other: 0.13 -> ~0.5 of the peak at this
    size; the bound here is loose)."""
    import time
    rng = np.random.default_rng(17)
    sizes = [3968, 3967, 3966, 3967, 3968, 3950, 3900, 3845]          # 31 tiles of 128 columns each
    fxs, ys = [], []
    for b, N in enumerate(sizes):
        f = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), 0.6 + 0.1 * b)), P.GPC())
        fxs.append(f(P.ColVecs(np.asfortranarray(rng.standard_normal((3, N)))), 0.1 + 0.01 * b))
        ys.append(rng.standard_normal(N))
    single = np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)])
    assert np.array_equal(P.logpdf_batch(fxs, ys), single)

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    t_batch = best(lambda: P.logpdf_batch(fxs, ys))
    t_single = best(lambda: [P.logpdf(fx, y) for fx, y in zip(fxs, ys)])
    assert t_batch < 0.7 * t_single, (t_batch, t_single)
    # one more tile in one member: that batch runs member by member, same values
    fxs[0] = fxs[0].f(P.ColVecs(np.asfortranarray(rng.standard_normal((3, 3969)))), 0.1)
    ys[0] = rng.standard_normal(3969)
    assert np.array_equal(P.logpdf_batch(fxs, ys), np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)]))


@pytest.mark.parametrize("N", [640, 4096])
def test_one_bad_member_does_not_lose_the_others(N):
    fxs, ys, _ = _members(4, N, seed=9)
    good = np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)])
    fxs[2] = fxs[2].f(fxs[2].x, -3.0)                     # K - 3 I: not positive definite
    with pytest.raises(P.PosDefException) as e:
        P.logpdf(fxs[2], ys[2])
    vals, infos = P.logpdf_batch(fxs, ys, return_infos=True)
    assert np.isnan(vals[2]) and infos[2] == e.value.info and infos[2] >= 1
    keep = [0, 1, 3]
    assert np.array_equal(vals[keep], good[keep]) and not infos[keep].any()


def test_batch_switched_off_and_under_a_forced_timeout(monkeypatch):
    """SGP_BATCH_MAX_N=0: member by member.  SGP_DF_TIMEOUT_S tiny: the pooled launch runs into its wait bound, the entry
    point reruns member by member on the launch-based schedule (with_df_fallback) -- same bits either way."""
    fxs, ys, _ = _members(6, 1024, seed=3)
    good = np.array([P.logpdf(fx, y) for fx, y in zip(fxs, ys)])
    for env in ({"SGP_BATCH_MAX_N": "0"}, {"SGP_DF_TIMEOUT_S": "1e-7"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = P.lib.Context(0)
        prev = P.lib.set_default_context(ctx)
        try:
            assert np.array_equal(P.logpdf_batch(fxs, ys), good), env
        finally:
            P.lib.set_default_context(prev)
            ctx.close()
        for k in env:
            monkeypatch.delenv(k)
