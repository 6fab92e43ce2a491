"""Multi-rank orchestration of the sharded logpdf (stheno.jl_amd/dist.py) on CPU over gloo,
world_size 2 and 3, with the NumPy test double tests/np_ops.py standing in for the HIP
building blocks.  Checks the result against the CPU oracle and the communication structure
(ownership, look-ahead order, reductions, PosDef propagation)."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(N, bad=False):
    import stheno_jl_amd as P
    rng = np.random.default_rng(123456)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [rng.standard_normal((2, n)) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(np.asfortranarray(v))) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(N)
    spec, _, _ = P.build_spec(F, x)
    return spec, xs, y, (-5.0 if bad else 0.1)


def _worker(rank, world, port, N, W, bad, q, subpanel=None):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import __graft_entry__ as entry
        entry.load_package()
        from stheno_jl_amd import dist as sdist
        from stheno_jl_amd import lib as slib
        import np_ops
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        spec, xs, y, s2 = _problem(N, bad)
        ops = np_ops.NumpyOps()
        stats = {}
        if subpanel:
            sdist.SUBPANEL = subpanel
        try:
            val = sdist.dist_logpdf(ops, spec, y, None, s2, world=world, rank=rank, W=W, stats=stats)
            q.put((rank, "ok", val, stats, ops.calls))
        except slib.PosDefException as e:
            q.put((rank, "posdef", e.info, stats, ops.calls))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc(), {}, []))


def _run(world, N, W, bad=False, subpanel=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, W, bad, q, subpanel)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize("world,N,W", [(2, 700, 128), (2, 1000, 256), (3, 900, 128)])
def test_sharded_logpdf_matches_oracle(world, N, W):
    from oracle import reference_model as orm
    res = _run(world, N, W)
    for r in res:
        assert r[1] == "ok", r[2]
    spec, xs, y, s2 = _problem(N)
    ref = orm.gppp_sum_logpdf(xs, y, s2)
    vals = [r[2] for r in res]
    assert all(v == vals[0] for v in vals)                 # every rank returns the same number
    assert abs(vals[0] - ref) <= 1e-10 * abs(ref)
    # structure: block-cyclic ownership, every panel factored exactly once, by its owner
    n_panels = res[0][3]["n_panels"]
    factored = {}
    for rank, _, _, stats, calls in res:
        for c in calls:
            if c[0] == "factor":
                J = c[1] // stats["W"]
                assert J % world == rank and J not in factored
                factored[J] = rank
        # look-ahead: an owner updates panel J+1 with panel J before it factors J+1
        order = [(c[0], c[1], c[2]) for c in calls if c[0] in ("factor", "update")]
        for i, c in enumerate(order):
            if c[0] == "factor" and c[1] > 0:
                J0 = c[1]
                prev = [o for o in order[:i] if o[0] == "update" and o[2] == J0]
                assert len(prev) == J0 // stats["W"]       # one update from every earlier panel
    assert sorted(factored) == list(range(n_panels))


@pytest.mark.parametrize("world,N,W,sub", [(2, 1500, 512, 128), (3, 1300, 384, 256)])
def test_sub_panel_pipeline_matches_oracle(world, N, W, sub):
    """Round 4: a panel wider than SUBPANEL is factored, broadcast and applied to the next panel sub-panel by sub-panel
    (dist.py: factor_and_send) -- same value as the oracle, every sub-panel factored once, in order, by the panel's owner,
    each followed by the update of the panel's remaining columns, and the owner of the next panel applies them one by one."""
    from oracle import reference_model as orm
    res = _run(world, N, W, subpanel=sub)
    for r in res:
        assert r[1] == "ok", r[2]
    spec, xs, y, s2 = _problem(N)
    ref = orm.gppp_sum_logpdf(xs, y, s2)
    vals = [r[2] for r in res]
    assert all(v == vals[0] for v in vals)
    assert abs(vals[0] - ref) <= 1e-10 * abs(ref)
    n_pad = (N + 127) // 128 * 128
    starts = []
    for rank, _, _, stats, calls in res:
        assert stats["W"] == W
        for i, c in enumerate(calls):
            if c[0] == "factor":
                J0, w = c[1], c[2]
                assert (J0 // W) % world == rank and w <= sub
                starts.append(J0)
                if (J0 % W) + w < min(W, n_pad - J0 // W * W):     # not the panel's last sub-panel: the rest of the panel follows
                    nxt = [o for o in calls[i + 1:] if o[0] == "update"][0]
                    assert nxt[1] == J0 // W * W and nxt[2] == J0 + w
    assert sorted(starts) == [c for J in range(0, n_pad, W) for c in range(J, min(J + W, n_pad), sub)]


def test_single_rank_path_and_layout():
    sys.path.insert(0, HERE)
    import np_ops
    from oracle import reference_model as orm
    from stheno_jl_amd import dist as sdist
    lay = sdist.PanelLayout(1024, 256, 3, 1, 1152)
    assert lay.n_panels == 4 and lay.mine == [1] and lay.owner(3) == 0 and lay.local_index(3) == 1
    assert lay.ld(1) == 1152 - 256 and lay.offset(1) == 0 and lay.n_local_doubles() == (1152 - 256) * 256
    lay0 = sdist.PanelLayout(1024, 256, 3, 0, 1152)          # owns panels 0 and 3, packed back to back
    assert lay0.mine == [0, 3] and lay0.offset(3) == 1152 * 256 and lay0.ld(3) == 1152 - 768
    assert sdist.geometry(1000, 1) == (1024, 1152) and sdist.geometry(128, 0) == (128, 128)
    spec, xs, y, s2 = _problem(300)
    val = sdist.dist_logpdf(np_ops.NumpyOps(), spec, y, None, s2, world=1, rank=0, W=128)
    ref = orm.gppp_sum_logpdf(xs, y, s2)
    assert abs(val - ref) <= 1e-10 * abs(ref)


def test_posdef_failure_reaches_every_rank():
    res = _run(2, 400, 128, bad=True)
    assert all(r[1] == "posdef" for r in res), res
    assert len({r[2] for r in res}) == 1 and res[0][2] >= 1


# ---- posterior on the sharded factor: test points ride along as bordered rows, one all-reduce ---------
def _post_problem(N, ns):
    import stheno_jl_amd as P
    spec, xs, y, s2 = _problem(N)
    rng = np.random.default_rng(97531)
    Xs = np.asfortranarray(rng.standard_normal((2, ns)))
    F = P.gppp_sum_model()
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(np.asfortranarray(v))) for k, v in zip(("f1", "f2", "f3"), xs)])
    xq = P.GPPPInput("f3", P.ColVecs(Xs))
    cross, _, _ = P.build_spec(F, xq, F, x)
    kss, _, _ = P.build_spec(F, xq)
    return spec, cross, kss, xs, y, s2, Xs


def _post_worker(rank, world, port, N, ns, W, want_cov, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import __graft_entry__ as entry
        entry.load_package()
        from stheno_jl_amd import dist as sdist
        import np_ops
        import np_terms
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        spec, cross, kss, xs, y, s2, Xs = _post_problem(N, ns)
        Kss = np_terms.dense_from_spec(kss)
        Kss = np.tril(Kss) + np.tril(Kss, -1).T
        ops = np_ops.NumpyOps()
        lp, m, v, c = sdist.dist_posterior_predict(ops, spec, cross, y, None, s2, np.zeros(ns), np.diag(Kss).copy(),
                                                   Kss if want_cov else None, world=world, rank=rank, W=W)
        # the FiniteGP-level wrapper builds the same specs and prior moments itself
        import stheno_jl_amd as P
        F = P.gppp_sum_model()
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(np.asfortranarray(u))) for k, u in zip(("f1", "f2", "f3"), xs)])
        m2, v2, c2 = sdist.dist_posterior(ops, F(x, s2), y, P.GPPPInput("f3", P.ColVecs(Xs)), want_cov=want_cov,
                                          world=world, rank=rank, W=W)
        assert np.array_equal(m2, m) and np.array_equal(v2, v) and (c is None or np.array_equal(c2, c))
        q.put((rank, "ok", (lp, m, v, c), {}, ops.calls))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc(), {}, []))


@pytest.mark.parametrize("world,N,ns,W,want_cov", [(2, 700, 37, 128, True), (3, 900, 130, 256, False), (2, 300, 1, 128, True)])
def test_sharded_posterior_matches_oracle(world, N, ns, W, want_cov):
    import oracle.abstractgps as oagp
    import oracle.kernelfunctions as okf
    import oracle.stheno as ost
    from oracle import reference_model as orm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_post_worker, args=(r, world, port, N, ns, W, want_cov, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    spec, cross, kss, xs, y, s2, Xs = _post_problem(N, ns)
    Fo = orm.gppp_sum()
    post = oagp.posterior(Fo(orm._blockdata(xs), s2), y)
    xq = ost.GPPPInput("f3", okf.ColVecs(Xs))
    m_ref, v_ref = post.mean(xq), post.var(xq)
    lp_ref = orm.gppp_sum_logpdf(xs, y, s2)
    lp0, m0, v0, c0 = res[0][2]
    for r in res[1:]:                                       # the all-reduce leaves the same numbers everywhere
        assert r[2][0] == lp0 and np.array_equal(r[2][1], m0) and np.array_equal(r[2][2], v0)
    assert abs(lp0 - lp_ref) <= 1e-10 * abs(lp_ref)
    assert np.max(np.abs(m0 - m_ref)) <= 1e-9 and np.max(np.abs(v0 - v_ref)) <= 1e-9
    if want_cov:
        c_ref = post.cov(xq)
        assert np.max(np.abs(c0 - c_ref)) <= 1e-9 and np.allclose(np.diag(c0), v0, atol=1e-12)
    # every rank assembled the cross rows of exactly its own panels
    for rank, _, _, _, calls in res:
        assert [c[1] for c in calls if c[0] == "cross"] == [c[1] for c in calls if c[0] == "assemble"]


# ---- sparse ELBO sharded over the data points (one all-reduce of the partial sums) --------------------
def _elbo_problem(N, M, diag_noise):
    import stheno_jl_amd as P
    rng = np.random.default_rng(2468)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((2, n))) for n in (n1, 0, N - n1)]     # one empty block
    Z = np.asfortranarray(rng.standard_normal((2, M)))
    y = rng.standard_normal(N)
    noise = (0.05 + rng.random(N)) if diag_noise else 0.1
    return F, xs, Z, y, noise


def _elbo_worker(rank, world, port, N, M, diag_noise, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import __graft_entry__ as entry
        P = entry.load_package()
        from stheno_jl_amd import dist as sdist
        import np_ops
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        F, xs, Z, y, noise = _elbo_problem(N, M, diag_noise)
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f3", "f1", "f3"), xs)])
        fx, fz = F(x, noise), F(P.GPPPInput("f3", P.ColVecs(Z)), 1e-6)
        ops = np_ops.NumpyOps()
        val = sdist.dist_elbo(ops, P.VFE(fz), fx, y, world=world, rank=rank)
        q.put((rank, "ok", val, {}, ops.calls))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc(), {}, []))


@pytest.mark.parametrize("world,N,M,diag_noise", [(2, 301, 17, False), (3, 500, 40, True), (3, 2, 5, False)])
def test_sharded_elbo_matches_oracle(world, N, M, diag_noise):
    import oracle.abstractgps as oagp
    import oracle.kernelfunctions as okf
    import oracle.stheno as ost
    from oracle import reference_model as orm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_elbo_worker, args=(r, world, port, N, M, diag_noise, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    F, xs, Z, y, noise = _elbo_problem(N, M, diag_noise)
    Fo = orm.gppp_sum()
    xo = ost.BlockData([ost.GPPPInput(k, okf.ColVecs(v)) for k, v in zip(("f3", "f1", "f3"), xs)])
    ref = oagp.elbo(oagp.VFE(Fo(ost.GPPPInput("f3", okf.ColVecs(Z)), 1e-6)), Fo(xo, noise), y)
    vals = [r[2] for r in res]
    assert all(v == vals[0] for v in vals)                    # the all-reduce leaves identical sums everywhere
    assert abs(vals[0] - ref) <= 1e-9 * abs(ref)
    # every data point is in exactly one rank's slice
    assert sum(c[1] for r in res for c in r[4] if c[0] == "elbo_partial") == N


def test_slice_inputs_and_row_shards():
    import stheno_jl_amd as P
    from stheno_jl_amd import dist as sdist
    x = P.BlockData([P.GPPPInput("f1", np.arange(5.0)), P.GPPPInput("f2", P.ColVecs(np.arange(12.0).reshape(2, 6)))])
    s = sdist.slice_inputs(x, 3, 8)
    assert [len(b) for b in P.blocks(s)] == [2, 3]
    assert np.array_equal(P.blocks(s)[0].x, [3.0, 4.0]) and np.array_equal(P.blocks(s)[1].x.X, np.arange(12.0).reshape(2, 6)[:, :3])
    cover = [sdist.shard_rows(10, 3, r) for r in range(3)]
    assert cover == [(0, 3), (3, 6), (6, 10)]
