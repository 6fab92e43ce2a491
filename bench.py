#!/usr/bin/env python
"""bench.py -- one "step" = one full logpdf(f(X, s2), y) of the dense-GP hot path on MI355X:
covariance assembly over ColVecs inputs + `+ s2 I` + blocked fp64 Cholesky (with the forward
substitution L^-1 (y - m) riding along as a bordered row) + logdet, inputs resident in HBM.

Workload (BASELINE.json metric "logpdf/sec and Cholesky TFLOPS (fp64) at N=64k", configs[4]):
single GP, Matern-5/2, N = 65536, D = 8, lengthscale sqrt(D) applied as stretch(f, 1/sqrt(D)),
sigma^2 = 0.1, zero mean, X, y ~ N(0,1) from numpy default_rng(123456).
It fits one GPU (34.4 GB), so the same problem is timed at every N: with --gpus N > 1 the
covariance is sharded in column panels over N ranks (stheno.jl_amd/dist.py) -- strong scaling.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29501 bench.py --gpus 8 --steps 3 --warmup 1

Prints ONE JSON line on rank 0 (contract in the round prompt), with `roofline` for the dominant
kernel (the fp64-MFMA trailing-update GEMM, timed with HIP events on its own streams inside
libsthenomi) and `cpu_baseline` (the NumPy/SciPy/OpenBLAS oracle restatement -- NOT Julia --
timed on this host's cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix rate; measured
                               # instruction ceilings are in DESIGN.md section 5
CONFIGS = {
    # name: (kernel, N, D)        BASELINE.json configs[0..4] = c1..c5
    "c5": ("matern52", 65536, 8),
    "c2": ("se", 16384, 8),
    "c1": ("se", 2048, 2),
    "c3": ("gppp3", 32768, 4),     # @gppp f3 = f1 + f2 over BlockData (:f1,10923),(:f2,10923),(:f3,10922)
    "c4": ("elbo", 262144, 8),     # sparse ELBO, M = 4096 inducing points (host-buffer C-ABI call)
    "n32k": ("matern52", 32768, 8),
    "n4k": ("matern52", 4096, 8),
}


def make_inputs(N, D):
    rng = np.random.default_rng(123456)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    y = rng.standard_normal(N)
    return X, y


def build_model(pkg, kernel):
    k = {"se": pkg.SEKernel, "matern52": pkg.Matern52Kernel}[kernel]()
    return pkg.atomic(pkg.GP(k), pkg.GPC())


def cpu_baseline(kernel, D, n_sample, N_target):
    """Oracle (CPU restatement) timed on a bounded sample; extrapolated to N_target with the
    measured stage split: assembly ~ N^2, Cholesky ~ N^3."""
    from oracle import reference_model as orm
    import scipy.linalg as sla
    from threadpoolctl import threadpool_limits
    # OpenBLAS with every hardware thread is far from its best on these sizes: calibrate the
    # thread count on a small Cholesky and give the CPU its best configuration.
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, 128, ncpu) if c <= ncpu})
    r0 = np.random.default_rng(0)
    B = r0.standard_normal((3072, 3072))
    S = B @ B.T + 3072 * np.eye(3072)
    best, threads = None, cands[0]
    for c in cands:
        with threadpool_limits(limits=c):
            sla.cholesky(S, lower=True, check_finite=False)
            t0 = time.perf_counter()
            sla.cholesky(S, lower=True, check_finite=False)
            dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, threads = dt, c
    X, y = make_inputs(n_sample, D)
    with threadpool_limits(limits=threads):
        val, t_tot, t_chol = orm.cpu_logpdf_timed(kernel, math.sqrt(D), X, y, 0.1)
    t_other = t_tot - t_chol
    r = N_target / n_sample
    t_target = t_other * r ** 2 + t_chol * r ** 3
    return {
        "value": 1.0 / t_target, "unit": "logpdf/s", "cores": int(threads), "kind": "port",
        "sample": (f"oracle restatement (NumPy/SciPy/OpenBLAS, not Julia) measured at N={n_sample}, D={D}: "
                   f"{t_tot:.2f} s total, {t_chol:.2f} s dpotrf ({n_sample**3 / 3 / t_chol / 1e9:.0f} GFLOP/s); "
                   f"extrapolated to N={N_target} as N^2 (assembly+solve) + N^3 (Cholesky)"),
        "measured_s": t_tot, "measured_cholesky_gflops": n_sample ** 3 / 3 / t_chol / 1e9,
        "logpdf_at_sample": val,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("SGP_BENCH_CONFIG", "c5"), choices=sorted(CONFIGS))
    ap.add_argument("--cpu-sample", type=int, default=8192, help="N of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--panel", type=int, default=1024, help="column-panel width of the multi-GPU path")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (multi-GPU) driver even at 1 GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "SGP_FORCE_DEVICE" in os.environ:      # test hook: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["SGP_FORCE_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SGP_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank) if backend == "nccl" else None)

    import __graft_entry__ as entry
    pkg = entry.load_package()
    L = pkg.lib
    from stheno_jl_amd import dist as sdist

    kernel, N, D = CONFIGS[args.config]
    X, y = make_inputs(N, D)
    elbo_step = None
    if kernel == "gppp3":
        F = pkg.gppp_sum_model()
        n1 = (N + 2) // 3
        cuts = [0, n1, 2 * n1, N]
        xb = pkg.BlockData([pkg.GPPPInput(k, pkg.ColVecs(np.asfortranarray(X[:, cuts[i]:cuts[i + 1]] / math.sqrt(D))))
                            for i, k in enumerate(("f1", "f2", "f3"))])
        spec, _, _ = pkg.build_spec(F, xb)
    elif kernel == "elbo":
        if world > 1:
            raise SystemExit("config c4 (ELBO) is a single-GPU bench line")
        f = pkg.stretch(build_model(pkg, "se"), 1.0 / math.sqrt(D))
        M = 4096
        Z = np.asfortranarray(X[:, np.random.default_rng(7).permutation(N)[:M]])
        fx, fz = f(pkg.ColVecs(X), 0.1), f(pkg.ColVecs(Z), 1e-6)
        elbo_step = lambda: pkg.elbo(pkg.VFE(fz), fx, y)   # noqa: E731
        spec, _, _ = pkg.build_spec(f, pkg.ColVecs(Z))
    else:
        f = pkg.stretch(build_model(pkg, kernel), 1.0 / math.sqrt(D))
        spec, _, _ = pkg.build_spec(f, pkg.ColVecs(X))
    ctx = L.Context(local_rank)
    lib = ctx.lib
    sigma2 = 0.1

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timings = np.zeros(8)
    if elbo_step is not None:
        def step(tm=None):
            return elbo_step()
    elif world == 1 and not args.force_dist:
        ds = C.c_void_p()
        L.check(lib.sgp_dspec_create(ctx.handle, spec.ref(), C.byref(ds)), "sgp_dspec_create")
        npad, mtot = C.c_int64(), C.c_int64()
        lib.sgp_geometry(N, 1, C.byref(npad), C.byref(mtot))
        A = torch.empty(npad.value * mtot.value, dtype=torch.float64, device="cuda")
        dY = torch.from_numpy(y).cuda()
        out = np.zeros(1)
        nz = np.array([sigma2])

        def step(tm=None):
            L.check(lib.sgp_dev_logpdf(ctx.handle, ds, A.data_ptr(), None, L.NOISE_SCALAR, L.dptr(nz), None,
                                       dY.data_ptr(), N, 1, L.dptr(out), L.dptr(tm) if tm is not None else None),
                    "sgp_dev_logpdf")
            return float(out[0])
    else:
        ops = sdist.HipOps(ctx)
        n_pad, m_tot = sdist.geometry(N, 1)
        lay = sdist.PanelLayout(n_pad, min(args.panel, n_pad), world, rank)
        A = ops.empty(max(1, m_tot * lay.n_local_cols()))

        def step(tm=None):
            return sdist.dist_logpdf(ops, spec, y, None, sigma2, world=world, rank=rank, W=args.panel, A=A)

    val = None
    for _ in range(args.warmup):
        val = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # one extra (untimed) instrumented step on 1 GPU: stage split + per-launch GEMM timing
    roofline = None
    stages = None
    if world == 1 and elbo_step is None and not args.force_dist:
        step(timings)
        upd_ms, n_launch, upd_flops = timings[3], int(timings[4]), timings[5]
        achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        roofline = {
            "kernel": "sgp::gemm_nt_dma_kernel<1> (fp64 MFMA trailing update of the blocked Cholesky, v_mfma_f64_4x4x4_4b_f64; <0> = the same code in its auxiliary uses)",
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
            "launches": n_launch, "avg_launch_ms": upd_ms / max(1, n_launch),
            "algorithmic_flops_per_launch_avg": upd_flops / max(1, n_launch),
            # The look-ahead keeps update launches of two streams (and the panel kernels) on the chip at
            # once, so the per-launch durations above overlap and include the CU slots lent to the panel
            # stream.  busy_ms = union of the launch intervals (same HIP events); achieved_while_busy =
            # the launches' algorithmic flops / busy_ms = what the kernel delivers while it is running.
            "busy_ms": timings[6],
            "achieved_while_busy": (upd_flops / (timings[6] * 1e-3) / 1e12) if timings[6] > 0 else None,
        }
        # HBM traffic cannot be read without rocprofv3; the committed PMC passes of one
        # representative launch of this kernel (tools/gpu_gemm_one.py, separate --pmc runs) are
        # attached for reference -- `traffic` itself stays null in the live line.
        pmc = os.path.join(ROOT, "profiles", "r01_gemm_pmc.json")
        if os.path.exists(pmc):
            roofline["traffic_profiled"] = json.load(open(pmc))
        stages = {"assemble_ms": timings[0], "cholesky_ms": timings[1], "finalize_ms": timings[2],
                  "kernelmatrix_GBps": (8.0 * N * (N + 1) / 2 + 8.0 * D * N) / (timings[0] * 1e-3) / 1e9}
    chol_tflops = (N ** 3 / 3.0) / (ms_per_step * 1e-3) / 1e12
    if elbo_step is not None:   # 2 M^2 N + 2 M^3/3 + 2 D M N flops
        M = 4096
        chol_tflops = (2.0 * M * M * N + 2.0 * M ** 3 / 3.0 + 2.0 * D * M * N) / (ms_per_step * 1e-3) / 1e12

    if rank == 0:
        cpu = None
        if args.cpu_sample > 0 and world == 1 and kernel in ("se", "matern52"):
            cpu = cpu_baseline(kernel, D, min(args.cpu_sample, N), N)
        line = {
            "metric": "elbo_per_sec" if elbo_step is not None else "logpdf_per_sec",
            "value": 1e3 / ms_per_step, "unit": "elbo/s" if elbo_step is not None else "logpdf/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ({"gppp3": f"@gppp f3=f1+f2 (SE + Matern52) over 3 BlockData blocks, total N={N}, D={D}",
                                     "elbo": f"sparse ELBO, SE, M=4096 inducing points, N={N}, D={D}, host-buffer C-ABI"}
                                    .get(kernel, f"single GP, {kernel}, N={N}, D={D}") +
                                    f", lengthscale sqrt(D), sigma2=0.1 (BASELINE config '{args.config}')"),
                       "N": N, "D": D, "kernel": kernel, "parallelism": f"column-panel x{world}" if world > 1 else "1 GPU",
                       "panel_width": args.panel if world > 1 else None},
            "cholesky_tflops_whole_step": chol_tflops,  # (c4: ELBO flops of SURVEY 8d)
            "cholesky_frac_of_fp64_matrix_peak": chol_tflops / (PEAK_FP64_MFMA_TFLOPS * world),
            "logpdf": val, "stages": stages, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
