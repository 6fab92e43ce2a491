"""GPU tests of the sharded (multi-rank) logpdf path with the real HIP building blocks.
The GPU box has one MI355X, so the 2-rank case runs both ranks on cuda:0 over gloo (RCCL
refuses two ranks on one device); the 8-GPU RCCL run differs only in the backend string."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(N, D=4):
    import stheno_jl_amd as P
    rng = np.random.default_rng(123456)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(N)
    return F, x, xs, y


def test_sharded_path_world1_matches_single_gpu_driver_and_oracle():
    import stheno_jl_amd as P
    from oracle import reference_model as orm
    from stheno_jl_amd import dist as sdist
    for N, W in [(500, 128), (3000, 512), (4500, 1024)]:
        F, x, xs, y = _problem(N)
        spec, _, _ = P.build_spec(F, x)
        ops = sdist.HipOps()
        v1 = sdist.dist_logpdf(ops, spec, y, None, 0.1, world=1, rank=0, W=W)
        v0 = P.logpdf(F(x, 0.1), y)
        ref = orm.gppp_sum_logpdf(xs, y, 0.1)
        assert abs(v1 - ref) <= 1e-10 * abs(ref) and abs(v0 - ref) <= 1e-10 * abs(ref)


def _worker(rank, world, port, N, W, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as entry
        P = entry.load_package()
        from stheno_jl_amd import dist as sdist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        F, x, xs, y = _problem(N)
        spec, _, _ = P.build_spec(F, x)
        ops = sdist.HipOps(P.lib.Context(0))
        val = sdist.dist_logpdf(ops, spec, y, None, 0.1, world=world, rank=rank, W=W)
        q.put((rank, "ok", val))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("N,W", [(2000, 256), (5000, 512)])
def test_two_ranks_on_one_gpu_over_gloo(N, W):
    import torch.multiprocessing as mp
    from oracle import reference_model as orm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    F, x, xs, y = _problem(N)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert res[0][2] == res[1][2]
    assert abs(res[0][2] - ref) <= 1e-10 * abs(ref)


def _nccl_worker(port, N, W, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as entry
        P = entry.load_package()
        from stheno_jl_amd import dist as sdist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        # the collectives the sharded driver issues, on HIP tensors, through RCCL
        t = torch.arange(1024, dtype=torch.float64, device="cuda")
        dist.broadcast(t, src=0)
        w = dist.broadcast(t, src=0, async_op=True)
        w.wait()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        assert float(t[5]) == 5.0
        F, x, xs, y = _problem(N)
        spec, _, _ = P.build_spec(F, x)
        ops = sdist.HipOps(P.lib.Context(0))
        val = sdist.dist_logpdf(ops, spec, y, None, 0.1, world=1, rank=0, W=W, always_collective=True)
        q.put((0, "ok", val))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((0, "error", traceback.format_exc()))


def test_rccl_backend_executes_the_sharded_drivers_collectives():
    """backend "nccl" == RCCL: a one-rank communicator on the single GPU runs every collective the
    sharded driver issues (async panel broadcasts on the panel stream, final SUM / MIN all-reduces)
    with HIP tensors -- the same calls an 8-GPU run makes, with a trivial communicator."""
    import torch.multiprocessing as mp
    from oracle import reference_model as orm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    N, W = 3000, 512
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), N, W, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[1] == "ok", res[2]
    F, x, xs, y = _problem(N)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(res[2] - ref) <= 1e-10 * abs(ref)


# ---- posterior on the sharded factor (test points as bordered rows) -----------------------------------
def _post_inputs(ns, D=4):
    import stheno_jl_amd as P
    rng = np.random.default_rng(8642)
    return P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((D, ns)))))


def test_sharded_posterior_world1_matches_single_gpu_posterior_and_oracle():
    import stheno_jl_amd as P
    import oracle.abstractgps as oagp
    import oracle.kernelfunctions as okf
    import oracle.stheno as ost
    from oracle import reference_model as orm
    from stheno_jl_amd import dist as sdist
    for N, ns, W in [(500, 1, 128), (3000, 200, 512), (4500, 129, 1024)]:
        F, x, xs, y = _problem(N)
        xq = _post_inputs(ns)
        fx = F(x, 0.1)
        m1, v1, c1 = sdist.dist_posterior(sdist.HipOps(), fx, y, xq, want_cov=True, world=1, rank=0, W=W)
        post = P.posterior(fx, y)
        m0, v0 = post.mean_and_var(xq)
        c0 = post.cov(xq)
        assert np.max(np.abs(m1 - m0)) <= 1e-10 and np.max(np.abs(v1 - v0)) <= 1e-10 and np.max(np.abs(c1 - c0)) <= 1e-10
        if N <= 3000:
            po = oagp.posterior(orm.gppp_sum()(orm._blockdata(xs), 0.1), y)
            xo = ost.GPPPInput("f3", okf.ColVecs(xq.x.X))
            assert np.max(np.abs(m1 - po.mean(xo))) <= 1e-9 and np.max(np.abs(v1 - po.var(xo))) <= 1e-9
            assert np.max(np.abs(c1 - po.cov(xo))) <= 1e-9


def _post_worker(rank, world, port, N, ns, W, backend, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as entry
        P = entry.load_package()
        from stheno_jl_amd import dist as sdist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        F, x, xs, y = _problem(N)
        m, v, c = sdist.dist_posterior(sdist.HipOps(P.lib.Context(0)), F(x, 0.1), y, _post_inputs(ns), want_cov=True,
                                       world=world, rank=rank, W=W, always_collective=True)
        q.put((rank, "ok", (m, v, c)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_sharded_posterior_two_ranks_on_one_gpu_and_rccl_world1(world, backend):
    import torch.multiprocessing as mp
    import stheno_jl_amd as P
    N, ns, W = 2600, 150, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_post_worker, args=(r, world, port, N, ns, W, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] == "ok", r[2]
    F, x, xs, y = _problem(N)
    post = P.posterior(F(x, 0.1), y)
    xq = _post_inputs(ns)
    m0, v0 = post.mean_and_var(xq)
    c0 = post.cov(xq)
    for r in res:
        m, v, c = r[2]
        assert np.array_equal(m, res[0][2][0]) and np.array_equal(v, res[0][2][1])
        assert np.max(np.abs(m - m0)) <= 1e-10 and np.max(np.abs(v - v0)) <= 1e-10 and np.max(np.abs(c - c0)) <= 1e-10


# ---- sparse ELBO sharded over the data points ----------------------------------------------------------
def _elbo_problem(N, M, D=3):
    import stheno_jl_amd as P
    rng = np.random.default_rng(97531)
    F = P.gppp_sum_model()
    n1 = N // 2
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, N - n1)]
    Z = np.asfortranarray(rng.standard_normal((D, M)))
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f3", "f1"), xs)])
    y = rng.standard_normal(N)
    noise = 0.05 + rng.random(N)
    return F, x, Z, y, noise


def test_sharded_elbo_world1_equals_host_elbo():
    import stheno_jl_amd as P
    from stheno_jl_amd import dist as sdist
    for N, M in [(700, 60), (3000, 256)]:
        F, x, Z, y, noise = _elbo_problem(N, M)
        fx, fz = F(x, noise), F(P.GPPPInput("f3", P.ColVecs(Z)), 1e-6)
        e0 = P.elbo(P.VFE(fz), fx, y)
        e1 = sdist.dist_elbo(sdist.HipOps(), P.VFE(fz), fx, y, world=1, rank=0)
        assert abs(e1 - e0) <= 1e-11 * abs(e0), (e0, e1)


def _elbo_worker(rank, world, port, N, M, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as entry
        P = entry.load_package()
        from stheno_jl_amd import dist as sdist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        F, x, Z, y, noise = _elbo_problem(N, M)
        fx, fz = F(x, noise), F(P.GPPPInput("f3", P.ColVecs(Z)), 1e-6)
        val = sdist.dist_elbo(sdist.HipOps(P.lib.Context(0)), P.VFE(fz), fx, y, world=world, rank=rank)
        q.put((rank, "ok", val))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def test_sharded_elbo_two_ranks_on_one_gpu_over_gloo():
    import torch.multiprocessing as mp
    import stheno_jl_amd as P
    N, M = 2500, 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_elbo_worker, args=(r, 2, port, N, M, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r[2]
    F, x, Z, y, noise = _elbo_problem(N, M)
    e0 = P.elbo(P.VFE(F(P.GPPPInput("f3", P.ColVecs(Z)), 1e-6)), F(x, noise), y)
    assert res[0][2] == res[1][2]
    assert abs(res[0][2] - e0) <= 1e-10 * abs(e0)
