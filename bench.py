#!/usr/bin/env python
"""bench.py -- one "step" = one full logpdf(f(X, s2), y) of the dense-GP hot path on MI355X:
covariance assembly over ColVecs / BlockData inputs + `+ s2 I` + blocked fp64 Cholesky (with the
forward substitution L^-1 (y - m) riding along as a bordered row) + logdet, inputs resident in HBM.

Default workload (BASELINE.json metric "logpdf/sec and Cholesky TFLOPS (fp64) at N=64k", configs[4]):
single GP, Matern-5/2, N = 65536, D = 8, lengthscale sqrt(D), sigma^2 = 0.1, zero mean, X, y ~ N(0,1)
from numpy default_rng(123456) (bench_configs.py).  It fits one GPU (34.4 GB), so the
same problem is timed at every N: with --gpus N > 1 the covariance is sharded in column panels over
N ranks (stheno.jl_amd/dist.py) -- strong scaling.  --config picks the other BASELINE workloads
(c1..c4, n4k) and `target`, the north-star model (@gppp f3 = f1 + f2 over three BlockData blocks,
N = 65536, D = 8).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus 8 --steps 3 --warmup 1          # ONE process, 8 GPUs: the in-library multi-GPU context
                                                         # (sgp_ctx_create_multi: what the Julia `ccall` reaches)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29501 bench.py --gpus 8 --steps 3 --warmup 1     # one process per GPU (stheno.jl_amd/dist.py)

--gpus N never runs on fewer than N GPUs: without WORLD_SIZE it builds the N-device context and exits non-zero when
fewer devices are visible; under torch.distributed.run WORLD_SIZE must equal N.

Prints ONE JSON line on rank 0 (contract in the round prompt) carrying, next to the contract's keys:
  parity_rel   |value - CPU golden| / |golden| for the configuration that ran (tests/golden/)
  roofline     the dominant kernel against its bound (fp64-MFMA trailing-update GEMM timed with HIP
               events on its own streams inside libsthenomi; for the ELBO also the HBM-bound
               K(x,z) assembly stage)
  host_api     the same step through the host-buffer entry point the Julia `ccall` binds
               (sgp_logpdf / sgp_elbo: uploads X, y; workspace from the context's cache)
  cpu_baseline the NumPy/SciPy/OpenBLAS oracle restatement -- NOT Julia -- timed on this host's
               cores on a bounded sample.
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

import bench_configs as bc  # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix rate; measured
                               # instruction ceilings are in DESIGN.md section 5
PEAK_HBM_GBS = 8000.0


def cpu_baseline(name, n_sample):
    from oracle import cpu_baseline as cb  # test infrastructure: the checker / baseline only
    kind, N, D = bc.CONFIGS[name]
    # (runs in a subprocess of its own: threading environment set before BLAS loads, inputs regenerated from the same seed)
    return cb.measure(kind, D, N, bc.GPPP_BLOCKS.get(name), None, None, bc.SIGMA2, n_sample,
                      elbo_m=bc.ELBO_M, elbo_znoise=bc.ELBO_ZNOISE)


def update_launches(N, schedule=None):
    """(rows m, columns nc, depth K) of every trailing-update launch of one logpdf, mirroring the driver's panel rule
    (capi.hip: chol_bordered / panel_factor_mid): outer panels of W with the look-ahead split (next panel's columns, then
    the rest) -- W = 2048 under the hybrid schedule (the default from 24576 columns on: the panels themselves are
    factored by the dataflow kernel); the launch-only schedules: W = 512 / 1024 below 65536 columns, from 65536 on the
    serial schedule with outer panels of 4096 columns factored by recursive halving down to 1024 (one update of the right
    half per level)."""
    n_pad = (N + 127) // 128 * 128
    m_tot = n_pad + 128
    out = []
    hybrid = schedule == "hybrid"
    deep = n_pad >= 65536 and not hybrid
    W = n_pad if n_pad <= 4096 else (1024 if n_pad <= 8192 else (4096 if deep else (1024 if n_pad >= 32768 else 512)))
    if hybrid:
        W = min(int(os.environ.get("SGP_HYBRID_W", 2048)) // 128 * 128, n_pad)

    def mid(c0, w):          # recursive halving inside the panel starting at global column c0
        if not deep or w <= 1024:
            return
        wl = max(1024, (w // 2 + 1023) // 1024 * 1024)
        mid(c0, wl)
        if w - wl > 0:
            out.append((m_tot - (c0 + wl), w - wl, wl))
            mid(c0 + wl, w - wl)

    J0 = 0
    while J0 < n_pad:
        wj = min(W, n_pad - J0)
        mid(J0, wj)
        c0 = J0 + wj
        if c0 >= n_pad:
            break
        w1 = min(W, n_pad - c0)
        split = ((c0, n_pad - c0),) if deep else ((c0, w1), (c0 + w1, n_pad - c0 - w1))
        for (c, nc) in split:
            if nc > 0:
                out.append((m_tot - c, nc, wj))
        J0 += W
    return out


def update_bytes_avg(N, schedule=None):
    """Algorithmic bytes of an average trailing-update launch of one logpdf: every launch reads and writes the
    lower 128-tiles of its C block once and reads its panel rows once."""
    ls = update_launches(N, schedule)
    tot = 0.0
    for (m, nc, k) in ls:
        entries = nc * (nc + 128) / 2 + (m - nc) * nc          # lower tiles of the square part + rows below
        tot += 2 * 8.0 * entries + 8.0 * m * k
    return tot / max(1, len(ls))


def lookup_traffic(config, schedule):
    """the committed counter record of `config` (profiles/r06_traffic.json, else the round-5 / round-4 files), if it was collected
    under `schedule`.  A LOOKUP of a committed rocprofv3 --pmc collection, not something this run measures: the record
    carries the sha1 of the library it was collected with and `measured_in_this_run: false`."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json"):
        f = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(f):
            continue
        rec = json.load(open(f)).get(config)
        if not rec or (rec.get("schedule") and schedule and rec["schedule"] != schedule):
            continue
        return dict(rec, file="profiles/" + name, measured_in_this_run=False)
    return None


def resolve_devices(gpus, devices_arg, device_count, world):
    """Which GPUs an in-process (WORLD_SIZE unset) run uses: `--devices a,b,..` verbatim (a device may repeat: several
    ranks on one GPU, the 1-GPU test hook), else 0 .. gpus-1.  Raises SystemExit -- never falls back to fewer GPUs."""
    if world > 1:
        if devices_arg:
            raise SystemExit("--devices is for the in-process multi-GPU context; under torch.distributed.run each rank "
                             "takes LOCAL_RANK")
        if gpus != world:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")
        return None
    if devices_arg:
        devs = [int(d) for d in devices_arg.split(",") if d.strip() != ""]
        if len(devs) != gpus:
            raise SystemExit(f"--gpus {gpus} but --devices lists {len(devs)} entries")
    else:
        devs = list(range(gpus))
    if device_count <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    bad = [d for d in devs if d < 0 or d >= device_count]
    if bad:
        raise SystemExit(f"--gpus {gpus} needs devices {devs} but only {device_count} GPU(s) are visible: refusing to "
                         f"run on fewer GPUs than asked for (launch on a node with {max(devs) + 1} GPUs)")
    return devs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("SGP_BENCH_CONFIG", "c5"), choices=sorted(bc.CONFIGS))
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="N of the bounded CPU-baseline sample (0 = skip; -1 = automatic: min(N, 32768) -- at the headline size the "
                         "Cholesky of the sample is 1/8 of the workload's, 8 s of LAPACK dpotrf; round 5 sampled 16384 = 1/64; "
                         "ELBO: 2x this many data points, automatic 16384)")
    ap.add_argument("--panel", type=int, default=1024, help="column-panel width of the multi-GPU path")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded (multi-GPU) driver even at 1 GPU")
    ap.add_argument("--devices", default=os.environ.get("SGP_BENCH_DEVICES", ""),
                    help="in-process multi-GPU context over exactly these devices, e.g. 0,1,2,3 (a repeated device = "
                         "several ranks on one GPU: loopback test configuration)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-buffer C-ABI leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="default c5 line only: skip the untimed-by-the-headline extras (north-star `target` model, N = 4096 / "
                         "16384 size sweep) that ride in the same JSON line")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="f32: the fp32 instantiation (sgp_logpdf_f32, host-buffer entry point; 1 GPU, dense configs)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries may write there too (RCCL prints a
    # version banner on rank 0): route file descriptor 1 to stderr for the whole run and keep the real
    # stdout for the result line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "SGP_FORCE_DEVICE" in os.environ:      # test hook: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["SGP_FORCE_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    devs = resolve_devices(args.gpus, args.devices, torch.cuda.device_count(), world)
    inproc = devs is not None and (len(devs) > 1 or bool(args.devices))   # one process, sgp_ctx_create_multi
    if inproc:
        local_rank = devs[0]
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if inproc and (use_dist or args.dtype != "f64"):
        raise SystemExit("--gpus N > 1 without torch.distributed.run is the in-process fp64 context: not with "
                         "--force-dist / --dtype f32")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("SGP_DIST_BACKEND", "nccl")
        # (a collective that never completes -- a rank that died, a link that is down -- fails the run after this bound instead
        # of the backend's default ten minutes: the first multi-GPU run must not hang)
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("SGP_DIST_TIMEOUT_S", "300"))),
                                device_id=torch.device("cuda", local_rank) if backend == "nccl" else None)

    import __graft_entry__ as entry
    pkg = entry.load_package()
    L = pkg.lib
    from stheno_jl_amd import dist as sdist

    kind, N, D = bc.CONFIGS[args.config]
    w = bc.build(pkg, args.config)
    y = w["y"]
    is_elbo = kind == "elbo"
    if is_elbo and use_dist:
        raise SystemExit("config c4 (ELBO) is a single-GPU bench line")
    transport_probe = None
    if (inproc and (len(set(devs)) > 2 or os.environ.get("SGP_BENCH_FORCE_PROBE")) and "SGP_MULTI_TRANSPORT" not in os.environ
            and kind != "elbo"):       # (SGP_BENCH_FORCE_PROBE: test hook -- run the probe on loopback ranks too)
        # Which panel transport this node runs faster is not knowable without the node (no multi-GPU hardware was
        # available while this was written): peer copies as scatter + all-gather use every xGMI link of a receiver,
        # RCCL's grouped broadcast is the conservative choice.  One UNTIMED logpdf per transport decides (part of the
        # warm-up, reported in the line); a transport that fails to initialise or run is skipped.
        transport_probe = {}
        spec_p = pkg.build_spec(w["f"], w["x"])[0]
        yp, op_, nzp = np.ascontiguousarray(y), np.zeros(1), np.array([bc.SIGMA2])
        for tr_name in ("p2p", "rccl"):
            os.environ["SGP_MULTI_TRANSPORT"] = tr_name
            try:
                c_ = L.Context(devices=devs)
                for rep in range(2):        # first call allocates; the second is the probe
                    t0p = time.perf_counter()
                    L.check(c_.lib.sgp_logpdf(c_.handle, spec_p.ref(), None, L.NOISE_SCALAR, L.dptr(nzp), L.dptr(yp), N, 1,
                                              L.dptr(op_)), "sgp_logpdf (transport probe)")
                    transport_probe[tr_name] = (time.perf_counter() - t0p) * 1e3
                c_.close()
            except Exception as e:      # noqa: BLE001 -- e.g. peer access missing, librccl not loadable
                transport_probe[tr_name + "_error"] = str(e)[:200]
        ok = {k: v for k, v in transport_probe.items() if not k.endswith("_error")}
        if not ok:
            raise SystemExit(f"no multi-GPU transport works on this node: {transport_probe}")
        os.environ["SGP_MULTI_TRANSPORT"] = min(ok, key=ok.get)
    if inproc:
        ctx = L.Context(devices=devs)
        if ctx.ndev != len(devs):
            raise SystemExit(f"multi-GPU context reports {ctx.ndev} ranks, asked for {len(devs)}")
        L.set_default_context(ctx)
    else:
        ctx = L.default_context() if is_elbo else L.Context(local_rank)
    lib = ctx.lib
    sigma2 = bc.SIGMA2
    spec = None if is_elbo else pkg.build_spec(w["f"], w["x"])[0]

    def sync():
        if use_dist:
            dist.barrier()
        if inproc:
            for d in sorted(set(devs)):
                torch.cuda.synchronize(d)
        torch.cuda.synchronize()

    timings = np.zeros(8)
    host_step = None
    if is_elbo:
        # the ELBO has no device-resident entry point: the timed step is the host-buffer C-ABI call
        # itself (X, Z, y uploaded every step; everything N x M stays on the device)
        zz, xz, mean_x, nk, nbuf, zk, zbuf = pkg.finite_gp._vfe_args(w["vfe"], w["fx"])
        var_x = np.ascontiguousarray(pkg.prior_var(w["f"], w["x"]))
        out = np.zeros(1)

        def step(tm=None):
            L.check(lib.sgp_elbo(ctx.handle, zz.ref(), xz.ref(), L.dptr(var_x), L.dptr(mean_x), nk, L.dptr(nbuf), zk,
                                 L.dptr(zbuf), L.dptr(y), L.dptr(out)), "sgp_elbo")
            return float(out[0])
    elif args.dtype == "f32":
        if use_dist:
            raise SystemExit("--dtype f32 is a single-GPU line")
        yh = np.ascontiguousarray(y)
        out = np.zeros(1)
        nz = np.array([sigma2])

        def step(tm=None):
            L.check(lib.sgp_logpdf_f32(ctx.handle, spec.ref(), None, L.NOISE_SCALAR, L.dptr(nz), L.dptr(yh), L.dptr(out)),
                    "sgp_logpdf_f32")
            return float(out[0])
    elif inproc:
        # the host-buffer entry point on the multi-GPU context: exactly the call the Julia shim makes
        yh = np.ascontiguousarray(y)
        out = np.zeros(1)
        nz = np.array([sigma2])

        def step(tm=None):
            L.check(lib.sgp_logpdf(ctx.handle, spec.ref(), None, L.NOISE_SCALAR, L.dptr(nz), L.dptr(yh), N, 1, L.dptr(out)),
                    "sgp_logpdf (multi-GPU context)")
            return float(out[0])
    elif not use_dist:
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

        yh = np.ascontiguousarray(y)
        out_h = np.zeros(1)

        def host_step():
            # what julia/SthenoMI355X.jl's logpdf(fx, y) ccalls: host spec + host y in, one double out
            L.check(lib.sgp_logpdf(ctx.handle, spec.ref(), None, L.NOISE_SCALAR, L.dptr(nz), L.dptr(yh), N, 1,
                                   L.dptr(out_h)), "sgp_logpdf")
            return float(out_h[0])
    else:
        ops = sdist.HipOps(ctx)
        n_pad, m_tot = sdist.geometry(N, 1)
        lay = sdist.PanelLayout(n_pad, min(args.panel, n_pad), world, rank, m_tot)
        A = ops.empty(lay.n_local_doubles())

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

    # ---- untimed extras on 1 GPU: instrumented step (stage split + per-launch GEMM timing), host API
    roofline = stages = host_api = None
    M = bc.ELBO_M
    if is_elbo:
        flops = 2.0 * M * M * N + 2.0 * M ** 3 / 3.0 + 2.0 * D * M * N   # 8.86 TFLOP
        L.check(lib.sgp_ctx_stage_timing(ctx.handle, 1))
        step()
        st = np.zeros(16)
        L.check(lib.sgp_ctx_stage_ms(ctx.handle, L.dptr(st)))
        L.check(lib.sgp_ctx_stage_timing(ctx.handle, 0))
        kxz_bytes = 8.0 * M * N + 8.0 * D * (N + M)
        mfma_ms = st[2] + st[4]                       # row solve + Gram product: 2 M^2 N flops
        mfma_tf = 2.0 * M * M * N / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
        asm_gbs = kxz_bytes / (st[1] * 1e-3) / 1e9 if st[1] > 0 else 0.0
        stages = {"kzz_assemble_factor_ms": st[0], "kxz_assemble_ms": st[1], "row_solve_ms": st[2],
                  "reductions_transpose_ms": st[3], "gram_ms": st[4], "final_factor_ms": st[5],
                  "device_total_ms": float(st[:6].sum())}
        roofline = {
            "kernel": "sgp::gemm_nt_dma_kernel<0> (fp64 MFMA: deep-K row solve against Lz + split-K Gram product A A')",
            "bound": "mfma", "achieved": mfma_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": mfma_tf / PEAK_FP64_MFMA_TFLOPS,
            # (per STEP of the bound, all launches of the kernel: it runs in many shapes)
            "traffic": (lookup_traffic("c4", None) or {}).get("hbm_bytes") if args.config == "c4" else None,
            "traffic_source": lookup_traffic("c4", None) if args.config == "c4" else None,
            "traffic_measured_in_this_run": False,
            "algorithmic_flops": 2.0 * M * M * N, "stage_ms": mfma_ms,
            "whole_step_frac_on_8.86TF": flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            # BASELINE.json calls c4 "HBM-bound"; SURVEY 8d asks for both figures and which one binds
            "hbm_stage": {"kernel": "sgp::assemble_block_kernel<8> (K(x,z), written once)", "bound": "hbm",
                          "achieved": asm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": asm_gbs / PEAK_HBM_GBS,
                          "algorithmic_bytes": kxz_bytes, "stage_ms": st[1]},
            "binding": ("mfma: the two M^2 N products take %.0f%% of the device time, the K(x,z) assembly %.0f%%"
                        % (100 * mfma_ms / max(st[:6].sum(), 1e-9), 100 * st[1] / max(st[:6].sum(), 1e-9))),
        }
        whole_tflops = flops / (ms_per_step * 1e-3) / 1e12
    else:
        # Models with independent components have exact zero blocks in their covariance and in its Cholesky factor; the
        # factorisation skips the tile products with a structurally zero operand (sthenomi.h: sgp_ctx_factor_work).  Rates
        # and roofline fractions are quoted on the flops that RAN (work_frac x N^3 / 3); the dense-equivalent rate -- what a
        # dense factorisation would need to deliver for the same step time -- is reported beside them.
        work_frac = 1.0
        if not use_dist and args.dtype == "f64":
            ex, de = ctx.factor_work()
            if de > 0:
                work_frac = ex / de
        whole_tflops = work_frac * (N ** 3 / 3.0) / (ms_per_step * 1e-3) / 1e12
    if args.dtype == "f32" and not is_elbo:
        PEAK_F32 = 157.3   # v_mfma_f32_32x32x2_f32 / fp32 vector peak (guide section 3)
        roofline = {"kernel": "gemm_nt_f32_dma_kernel (v_mfma_f32_32x32x2_f32 trailing updates; second instantiation, LDS-DMA operand "
                              "path, 4 workgroups per CU)",
                    "bound": "mfma", "achieved": whole_tflops, "peak": PEAK_F32, "unit": "TFLOP/s",
                    "frac": whole_tflops / PEAK_F32, "traffic": None,
                    "note": "whole-step N^3/3 rate of the fp32 path (host-buffer entry point; panel chain in fp64 arithmetic "
                            "on fp32 storage; the kernel alone: profiles/r04_bench_c5_f32_kernel_stats.csv)"}
    multi = None
    if inproc:
        st = np.zeros(11 + 4 * len(devs))
        nst = C.c_int64()
        L.check(lib.sgp_ctx_multi_stats(ctx.handle, L.dptr(st), len(st), C.byref(nst)), "sgp_ctx_multi_stats")
        P_ = len(devs)
        per_rank = [{"rank": i, "device": devs[i], "update_flops": st[8 + 4 * i], "update_span_ms": st[9 + 4 * i],
                     "update_tflops": (st[8 + 4 * i] / (st[9 + 4 * i] * 1e-3) / 1e12) if st[9 + 4 * i] > 0 else None,
                     "panels_factored": int(st[10 + 4 * i]), "bytes_received": st[11 + 4 * i]} for i in range(P_)]
        multi = {"driver": "in-process: sgp_ctx_create_multi + sgp_logpdf / sgp_elbo (one host thread, one C-ABI call)",
                 "devices": devs, "ranks": int(st[0]), "transport": ctx.transport,
                 "rccl_ranks": int(st[3]) if st[3] >= 0 else None,
                 "peer_copy_form": ("scatter + all-gather" if st[6] else "direct") if ctx.transport in ("p2p", "loopback") else None,
                 "panel_width": int(st[4]), "panels": int(st[5]), "panels_per_update_group": int(st[7]),
                 "last_call_ms": st[1], "host_enqueue_ms": st[8 + 4 * P_], "per_rank": per_rank,
                 "panel_ownership": {0: "cyclic", 1: "balanced table (own_table.h: from the symbolic tile pattern)",
                                     2: "explicit list (SGP_MULTI_OWNERS)"}.get(int(st[9 + 4 * P_]), "?"),
                 "late_bound_event_waits": int(st[10 + 4 * P_]),
                 "transport_probe_ms": transport_probe}
        if not is_elbo:
            tf = [r["update_tflops"] for r in per_rank if r["update_tflops"]]
            per_gpu = whole_tflops / len(set(devs))
            roofline = {"kernel": "sgp::gemm_nt_seg_kernel (batched trailing updates of the in-library column-panel driver: one launch "
                                  "per rank, step and class, far panels with K = a whole group of received panels)",
                        "bound": "mfma", "achieved": per_gpu, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": per_gpu / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                        "per_rank_update_tflops": [r["update_tflops"] for r in per_rank],
                        "note": "achieved = whole-step N^3/3 rate per physical GPU of the sharded run (panel factorisations "
                                "and transport included); per_rank_update_tflops = each rank's trailing-update flops / the "
                                "span from its first to its last update (HIP events)" + ("" if tf else " (no updates ran)")}
    if not use_dist and not inproc and not is_elbo and args.dtype == "f64":
        step(timings)
        upd_ms, n_launch, upd_flops = timings[3], int(timings[4]), timings[5]
        achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        schedule = ctx.factor_schedule(N)
        dataflow = schedule.startswith("dataflow")
        if dataflow:
            # one launch of persistent workgroups does the whole factorisation (chol_df.hip): the dominant kernel IS the
            # Cholesky stage; its algorithmic flops are the factorisation's N^3 / 3 (+ the bordered row)
            upd_ms, n_launch, upd_flops = timings[1], 1, work_frac * N ** 3 / 3.0 + 1.0 * N * N
            achieved = upd_flops / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        elif n_launch == 0:
            # N <= 4096 is factored as ONE outer panel (no outer trailing update): the step is the latency
            # chain panel_solve -> K = 128 update with the next diagonal block's potrf_diag fused in, per 128
            # columns; report the whole step
            achieved = whole_tflops
        n_pad = (N + 127) // 128 * 128
        fused = n_pad < 32768     # capi.hip: fuse_mode -- the look-ahead column updates carry the next diagonal block
        hybrid = schedule == "hybrid"
        serial = n_pad >= 65536 and not hybrid   # capi.hip: chol_bordered -- launch-only schedule: no look-ahead from 65536 columns on
        if dataflow:
            kname = ("sgp::chol_dataflow_fat_kernel" if schedule == "dataflow-fat" else "sgp::chol_dataflow_kernel") + (
                " (the whole blocked Cholesky in one launch of persistent workgroups: 128 x 128 tile tasks in topological "
                "order, contraction = the v_mfma_f64_4x4x4_4b tile program of sgp::gemm_nt_dma_kernel, diagonal tiles through "
                "potrf_diag_body, off-diagonal tiles through the refined 16-row substitution; tile-row progress counters with "
                "agent-scope release / acquire instead of kernel boundaries" +
                ("; one workgroup per CU, 256 VGPRs)" if schedule == "dataflow-fat" else "; two workgroups per CU)"))
        elif hybrid:
            kname = ("sgp::gemm_nt_dma_kernel<1> (fp64 MFMA trailing update of the blocked Cholesky, v_mfma_f64_4x4x4_4b_f64; hybrid "
                     "schedule: outer panels of 2048 columns, each factored by ONE launch of sgp::chol_dataflow_fat_kernel on the "
                     "panel -- diagonal chain and row solves as tile tasks -- beside the previous panel's K = 2048 trailing update; "
                     "two update launches per panel: the next panel's columns, then the rest)")
        elif serial:
            kname = ("sgp::gemm_nt_dma_potrf_kernel<1, true> (fp64 MFMA trailing update of the blocked Cholesky, "
                     "v_mfma_f64_4x4x4_4b_f64: the tile program of sgp::gemm_nt_dma_kernel, whose tile (0, 0) workgroup goes on "
                     "to factor the next diagonal block; serial schedule: outer panels of 4096 columns, one K = 4096 launch per "
                     "panel + the K = 2048 / 1024 updates of the recursive halving inside a panel, no look-ahead at this size)")
        else:
            kname = ("sgp::gemm_nt_dma_kernel<1> (fp64 MFMA trailing update of the blocked Cholesky, v_mfma_f64_4x4x4_4b_f64; "
                     "<0> = the same code in its auxiliary uses)"
                     + ("; below N = 32768 the look-ahead half of the launches is sgp::gemm_nt_dma_potrf_kernel<1, true>: the same "
                        "tile program, whose tile (0, 0) workgroup goes on to factor the next diagonal block" if fused else ""))
        roofline = {
            "kernel": kname,
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
            "note": ("single outer panel: no trailing-update launches; achieved = whole-step N^3/3 rate, latency-bound on the "
                     "chain sgp::panel_solve_kernel (14 us) -> sgp::gemm_nt_dma_potrf_kernel<0, true> (K = 128 tile update + "
                     "potrf_diag of the next diagonal block, 50 us) per 128 columns") if n_launch == 0 else
                    ("achieved = (N^3 / 3 + N^2) flops / the Cholesky stage (HIP events around the one launch); SGP_DF_STATS=1 "
                     "prints the per-workgroup time split and the per-column chain (profiles/archive/r03_dataflow.md)") if dataflow else None,
            "schedule": schedule,
            "launches": n_launch, "avg_launch_ms": upd_ms / max(1, n_launch),
            "algorithmic_flops_per_launch_avg": upd_flops / max(1, n_launch),
            # The look-ahead keeps update launches of two streams (and the panel kernels) on the chip at
            # once, so the per-launch durations above overlap and include the CU slots lent to the panel
            # stream.  busy_ms = union of the launch intervals (same HIP events); achieved_while_busy =
            # the launches' algorithmic flops / busy_ms = what the kernel delivers while it is running.
            "busy_ms": timings[6],
            "achieved_while_busy": (upd_flops / (timings[6] * 1e-3) / 1e12) if timings[6] > 0 else None,
        }
        # HBM traffic cannot be read from inside the process (it needs rocprofv3 --pmc passes): `traffic` is the committed
        # per-launch average of exactly this command under the FETCH_SIZE / WRITE_SIZE passes of the round's collection
        # (tools/collect_r04.sh -> tools/make_r04_summary.py -> profiles/r04_traffic.json; every record carries the sha1 of
        # the library it was collected with and the schedule it ran under), null for configurations not profiled or
        # profiled under another schedule.
        rec = lookup_traffic(args.config, schedule)
        if rec:
            roofline["traffic"] = rec["hbm_bytes"]
            roofline["traffic_source"] = rec
        if hybrid and not args.no_extras:
            # Under the hybrid schedule the HIP events around an update launch include the time its grid waits while the
            # panel kernel (one 256-VGPR workgroup per CU: it does not share a CU) holds the chip, and the two update
            # streams overlap: `frac` above is the launches' IN SITU rate.  The same launches with the chip to themselves:
            # one more instrumented step on a second context whose schedule issues everything on ONE stream
            # (SGP_HYBRID_SERIAL=1 is read when a context is created; same launches, same bits, no overlap).
            prev_env = os.environ.get("SGP_HYBRID_SERIAL")
            os.environ["SGP_HYBRID_SERIAL"] = "1"
            try:
                sctx = L.Context(ctx.device if hasattr(ctx, "device") else 0)
                try:
                    sds = C.c_void_p()
                    L.check(lib.sgp_dspec_create(sctx.handle, spec.ref(), C.byref(sds)), "sgp_dspec_create")
                    st, so = np.zeros(8), np.zeros(1)
                    for tm_ in (None, st):     # one warm-up, one instrumented
                        t0s = time.perf_counter()
                        L.check(lib.sgp_dev_logpdf(sctx.handle, sds, A.data_ptr(), None, L.NOISE_SCALAR, L.dptr(nz), None,
                                                   dY.data_ptr(), N, 1, L.dptr(so), L.dptr(tm_) if tm_ is not None else None),
                                "sgp_dev_logpdf (serial)")
                        ser_ms = (time.perf_counter() - t0s) * 1e3
                    lib.sgp_dspec_destroy(sds)
                finally:
                    sctx.close()
            finally:
                if prev_env is None:
                    os.environ.pop("SGP_HYBRID_SERIAL", None)
                else:
                    os.environ["SGP_HYBRID_SERIAL"] = prev_env
            s_ach = st[5] / (st[3] * 1e-3) / 1e12 if st[3] > 0 else 0.0
            roofline["uncontended"] = {
                "how": "the same schedule issued on ONE stream (SGP_HYBRID_SERIAL=1, second context, one instrumented step): "
                       "no kernel shares the chip, HIP events around every update launch",
                "launches": int(st[4]), "avg_launch_ms": st[3] / max(1, int(st[4])), "achieved": s_ach,
                "frac": s_ach / PEAK_FP64_MFMA_TFLOPS, "ms_per_step": ser_ms, "same_bits": float(so[0]) == float(out[0])}
        if hybrid and not args.no_extras:
            # The round-4 schedule (SGP_HYBRID=0: serial-deep launches at 65536 columns, the whole-matrix dataflow kernel below)
            # on the SAME box, same buffers, second context: separates schedule from box in the driver's record (round-5
            # verdict: the driver's fresh boxes saw -1.2 % where the same-box A/B said -4.5 %).  Same bits.
            prev_env = os.environ.get("SGP_HYBRID")
            os.environ["SGP_HYBRID"] = "0"
            try:
                hctx = L.Context(ctx.device if hasattr(ctx, "device") else 0)
                try:
                    hds = C.c_void_p()
                    L.check(lib.sgp_dspec_create(hctx.handle, spec.ref(), C.byref(hds)), "sgp_dspec_create")
                    ho = np.zeros(1)

                    def hstep():
                        L.check(lib.sgp_dev_logpdf(hctx.handle, hds, A.data_ptr(), None, L.NOISE_SCALAR, L.dptr(nz), None,
                                                   dY.data_ptr(), N, 1, L.dptr(ho), None), "sgp_dev_logpdf (SGP_HYBRID=0)")
                    hstep()
                    torch.cuda.synchronize()
                    nrep_h = 3 if N >= 32768 else 10
                    t0h = time.perf_counter()
                    for _ in range(nrep_h):
                        hstep()
                    h_ms = (time.perf_counter() - t0h) / nrep_h * 1e3
                    h_sched = hctx.factor_schedule(N)
                    lib.sgp_dspec_destroy(hds)
                finally:
                    hctx.close()
            finally:
                if prev_env is None:
                    os.environ.pop("SGP_HYBRID", None)
                else:
                    os.environ["SGP_HYBRID"] = prev_env
            roofline["round4_schedule_same_box"] = {
                "how": "SGP_HYBRID=0 on a second context, same device buffers, after the timed region: " + h_sched,
                "schedule": h_sched, "steps": nrep_h, "ms_per_step": h_ms, "hybrid_over_this": ms_per_step / h_ms,
                "same_bits": float(ho[0]) == float(out[0])}
        if hybrid:
            roofline["note"] = ("hybrid schedule: `frac` is the update launches' rate IN SITU (their HIP-event durations include the "
                                "time the panel kernel holds the chip and the overlap of the two update streams); "
                                "`achieved_while_busy` = their flops / the union of their intervals; `uncontended` = the same "
                                "launches alone on the chip")
        roofline["algorithmic_bytes_per_launch_avg"] = (8.0 * N * (N + 1) if dataflow else update_bytes_avg(N, schedule))   # dataflow: the
        # lower triangle read once and written once
        roofline["traffic_measured_in_this_run"] = False   # (counters need separate rocprofv3 --pmc passes: see traffic_source)
        asm_bytes = 8.0 * N * (N + 1) / 2 + 8.0 * D * N
        stages = {"assemble_ms": timings[0], "cholesky_ms": timings[1], "finalize_ms": timings[2],
                  "kernelmatrix_GBps": asm_bytes / (timings[0] * 1e-3) / 1e9,
                  "kernelmatrix_frac_of_hbm_peak": asm_bytes / (timings[0] * 1e-3) / 1e9 / PEAK_HBM_GBS}
    if use_dist and not inproc and not is_elbo:
        # the per-launch HIP-event instrument lives in the 1-GPU driver; the sharded run reports its whole-step rate
        per_gpu = whole_tflops / world
        roofline = {"kernel": "sgp::gemm_nt_dma_kernel<1> (per-panel trailing updates of the column-panel driver)",
                    "bound": "mfma", "achieved": per_gpu, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": per_gpu / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                    "note": "whole-step N^3/3 rate per GPU of the sharded run (panel broadcasts and factorisations included)"}
    if host_step is not None and not args.no_host_api:
        # the real boundary: host buffers in, workspace from the context's grow-only cache
        hv = host_step()          # first call sizes the cache
        torch.cuda.synchronize()
        nrep = max(1, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(nrep):
            hv = host_step()
        host_ms = (time.perf_counter() - t0) / nrep * 1e3
        host_api = {"entry": "sgp_logpdf (host spec + host y; X, y uploaded per call; cached workspace)",
                    "ms_per_call": host_ms, "calls": nrep, "vs_device_resident": host_ms / ms_per_step, "logpdf": hv}

    # ---- extras of the DEFAULT line (round-3 verdict: the driver only ever timed c5): the north-star model (`target`: the
    # 3-process @gppp over BlockData, N = 65536, D = 8) and the N = 4096 / 16384 points of the size sweep, each through the
    # host-buffer entry point the Julia `ccall` binds, on the same context, AFTER the timed region of the headline.
    north_star = sizes = None
    if (args.config == "c5" and not args.no_extras and not use_dist and not inproc and args.dtype == "f64" and rank == 0):
        def time_config(name, steps, warmup, ctx=ctx):
            kd, Nn, Dd = bc.CONFIGS[name]
            ww = bc.build(pkg, name)
            sp = pkg.build_spec(ww["f"], ww["x"])[0]
            yy, oo, nzz = np.ascontiguousarray(ww["y"]), np.zeros(1), np.array([sigma2])

            def one():
                L.check(lib.sgp_logpdf(ctx.handle, sp.ref(), None, L.NOISE_SCALAR, L.dptr(nzz), L.dptr(yy), Nn, 1, L.dptr(oo)),
                        "sgp_logpdf (" + name + ")")
            for _ in range(warmup):
                one()
            torch.cuda.synchronize()
            t0x = time.perf_counter()
            for _ in range(steps):
                one()
            ms = (time.perf_counter() - t0x) / steps * 1e3
            gg = bc.golden(name)
            exx, dee = ctx.factor_work()
            wf = exx / dee if dee > 0 else 1.0
            tf = wf * (Nn ** 3 / 3.0) / (ms * 1e-3) / 1e12
            return {"config": name, "workload": bc.describe(name), "entry": "sgp_logpdf (host buffers)", "steps": steps,
                    "ms_per_step": ms, "cholesky_tflops_whole_step": tf, "frac": tf / PEAK_FP64_MFMA_TFLOPS,
                    "executed_work_fraction": wf, "dense_equivalent_tflops": (Nn ** 3 / 3.0) / (ms * 1e-3) / 1e12,
                    "schedule": ctx.factor_schedule(Nn), "logpdf": float(oo[0]),
                    "parity_rel": None if gg is None else abs(float(oo[0]) - gg["logpdf"]) / abs(gg["logpdf"])}
        north_star = time_config("target", 3, 1)
        # ... and under the DENSE schedule (SGP_STRUCT_ZEROS=0 is read when a context is created): the same model with every
        # structurally zero tile product multiplied out, as the reference's LAPACK path does -- same bits, the time the
        # skipping saves.  (A second context on the same device; released before the size sweep.)
        if north_star.get("executed_work_fraction", 1.0) < 0.999:
            prev_env = os.environ.get("SGP_STRUCT_ZEROS")
            os.environ["SGP_STRUCT_ZEROS"] = "0"
            try:
                dense_ctx = L.Context(ctx.device if hasattr(ctx, "device") else 0)
                try:
                    dn = time_config("target", 2, 1, dense_ctx)
                finally:
                    dense_ctx.close()
            finally:
                if prev_env is None:
                    os.environ.pop("SGP_STRUCT_ZEROS", None)
                else:
                    os.environ["SGP_STRUCT_ZEROS"] = prev_env
            north_star["dense_schedule"] = {"ms_per_step": dn["ms_per_step"], "frac": dn["frac"], "logpdf": dn["logpdf"],
                                            "same_bits": dn["logpdf"] == north_star["logpdf"], "steps": dn["steps"],
                                            "how": "SGP_STRUCT_ZEROS=0 on a second context"}
        sizes = {"n4k": time_config("n4k", 30, 3), "c2": time_config("c2", 10, 2)}

        # Small N fills the chip through sgp_logpdf_batch (round 6; the round-5 verdict: one N = 4096 factorisation runs at 0.13
        # of the MFMA peak, its diagonal chain leaves ~220 CUs idle): B members of the n4k workload with DIFFERENT
        # hyper-parameters (member 0 is the n4k configuration itself: its value must be bit-equal to the single call above
        # and matches the golden) factored as ONE task pool of the dataflow kernel.  Aggregate rate = B N^3 / 3 / call time.
        def time_batch(name, B, steps, warmup):
            kd, Nn, Dd = bc.CONFIGS[name]
            Xb, yb = bc.make_inputs(Nn, Dd)
            ker = {"se": pkg.SEKernel, "matern52": pkg.Matern52Kernel}[kd]
            mem = []
            for b in range(B):
                fb = pkg.stretch(pkg.atomic(pkg.GP(ker()), pkg.GPC()), 1.0 / (math.sqrt(Dd) * (1.0 + 0.04 * b)))
                mem.append((pkg.build_spec(fb, pkg.ColVecs(Xb))[0], np.ascontiguousarray(yb), np.array([sigma2 * (1.0 + 0.1 * b)])))
            specs = (C.POINTER(L.sgp_cov_spec) * B)(*[C.pointer(mm[0].c) for mm in mem])
            means = (C.POINTER(C.c_double) * B)(*[L.dptr(None) for _ in mem])
            noises = (C.POINTER(C.c_double) * B)(*[L.dptr(mm[2]) for mm in mem])
            ysb = (C.POINTER(C.c_double) * B)(*[L.dptr(mm[1]) for mm in mem])
            ob, ib = np.zeros(B), np.zeros(B, dtype=np.int32)

            def one():
                L.check(lib.sgp_logpdf_batch(ctx.handle, B, specs, means, L.NOISE_SCALAR, noises, ysb, L.dptr(ob),
                                             ib.ctypes.data_as(C.POINTER(C.c_int))), "sgp_logpdf_batch")
            for _ in range(warmup):
                one()
            t0x = time.perf_counter()
            for _ in range(steps):
                one()
            ms = (time.perf_counter() - t0x) / steps * 1e3
            o1 = np.zeros(1)
            singles = []
            for sp_, y_, nz_ in mem:        # every member through its own sgp_logpdf call: the bits must agree
                L.check(lib.sgp_logpdf(ctx.handle, sp_.ref(), None, L.NOISE_SCALAR, L.dptr(nz_), L.dptr(y_), Nn, 1, L.dptr(o1)),
                        "sgp_logpdf (batch member)")
                singles.append(float(o1[0]))
            gg = bc.golden(name)
            tf = B * (Nn ** 3 / 3.0) / (ms * 1e-3) / 1e12
            return {"config": name, "members": B, "entry": "sgp_logpdf_batch (host buffers; one task pool of the dataflow kernel)",
                    "steps": steps, "ms_per_call": ms, "ms_per_member": ms / B, "aggregate_logpdf_per_s": B / (ms * 1e-3),
                    "aggregate_cholesky_tflops": tf, "frac": tf / PEAK_FP64_MFMA_TFLOPS,
                    "every_member_bit_equal_to_its_own_call": bool(np.array_equal(ob, np.array(singles))),
                    "member0_parity_rel": None if gg is None else abs(float(ob[0]) - gg["logpdf"]) / abs(gg["logpdf"])}
        sizes["n4k_batch8"] = time_batch("n4k", 8, 10, 2)
        sizes["n4k_batch16"] = time_batch("n4k", 16, 5, 1)

    # ---- the gradient (the reference's main use, examples/getting_started/script.jl:154-213) under the same evidence
    # regime as logpdf (round-4 verdict): sgp_logpdf_grad through the host API at n4k / c2 / n32k -- ms, the fraction of the fp64
    # MFMA peak on its N^3 flops (factorisation N^3/3 + inverse 2N^3/3: LAPACK dpotri economics), and parity of d/d sigma^2
    # and d/d inscale against the standalone CPU goldens (tests/golden/grad_configs.json; n32k: logpdf only).
    grad = None
    if (args.config == "c5" and not args.no_extras and not use_dist and not inproc and args.dtype == "f64" and rank == 0):
        gg_all = {}
        gpath = os.path.join(ROOT, "tests", "golden", "grad_configs.json")
        if os.path.exists(gpath):
            gg_all = json.load(open(gpath)).get("cases", {})
        L.set_default_context(ctx)
        grad = {}
        for name, steps in (("n4k", 10), ("c2", 5), ("n32k", 2), ("c3", 2)):
            ww = bc.build(pkg, name)
            Nn = bc.CONFIGS[name][1]
            r = pkg.logpdf_and_gradient(ww["fx"], ww["y"])          # warm-up (sizes the cache)
            torch.cuda.synchronize()
            t0x = time.perf_counter()
            for _ in range(steps):
                r = pkg.logpdf_and_gradient(ww["fx"], ww["y"])
            ms = (time.perf_counter() - t0x) / steps * 1e3
            gex, gde = ctx.factor_work()      # (structured models: the gradient's bordered factorisation skips its zeros too)
            # logpdf alone on the same context, same entry level
            pkg.logpdf(ww["fx"], ww["y"])
            t0x = time.perf_counter()
            for _ in range(steps):
                pkg.logpdf(ww["fx"], ww["y"])
            ms_lp = (time.perf_counter() - t0x) / steps * 1e3
            tf = (Nn ** 3) / (ms * 1e-3) / 1e12
            gref = gg_all.get(name)
            term = r["terms"][0]
            rec = {"entry": "sgp_logpdf_grad (host buffers)", "N": Nn, "steps": steps, "ms_per_call": ms, "logpdf_ms": ms_lp,
                   "ratio_to_logpdf": ms / ms_lp, "tflops_on_N3": tf,
                   # (structured models skip part of the N^3: the figure is then a dense-EQUIVALENT rate, not a roofline fraction)
                   "frac": (tf / PEAK_FP64_MFMA_TFLOPS) if (gde <= 0 or gex / gde > 0.999) else None,
                   "dense_equivalent_frac": tf / PEAK_FP64_MFMA_TFLOPS,
                   "executed_work_fraction_of_the_bordered_factorisation": (gex / gde) if gde > 0 else 1.0,
                   "d_sigma2": float(np.ravel(r["noise"])[0]), "d_inscale": float(term["d_inscale"]), "logpdf": float(r["logpdf"])}
            if gref:
                rec["parity_rel_d_sigma2"] = abs(rec["d_sigma2"] - gref["d_sigma2"]) / abs(gref["d_sigma2"])
                rec["parity_rel_d_inscale"] = abs(rec["d_inscale"] - gref["d_inscale"]) / abs(gref["d_inscale"])
                rec["parity_rel_logpdf"] = abs(rec["logpdf"] - gref["logpdf"]) / abs(gref["logpdf"])
            else:
                gl = bc.golden(name)
                rec["parity_rel_logpdf"] = None if gl is None else abs(rec["logpdf"] - gl["logpdf"]) / abs(gl["logpdf"])
            grad[name] = rec

    if rank == 0:
        g = bc.golden(args.config)
        gval = None if g is None else g.get("elbo" if is_elbo else "logpdf")
        parity = None if gval is None else abs(val - gval) / abs(gval)   # (fp32 lines: fp32 accuracy, ~1e-6)
        cpu = None
        cpu_n = args.cpu_sample if args.cpu_sample >= 0 else (16384 if is_elbo else min(N, 32768))
        if cpu_n > 0 and world == 1 and not inproc:
            # (a reported baseline must never cost the line: a failing CPU leg is retried on half the sample, then recorded)
            for attempt, n_try in enumerate((cpu_n, max(cpu_n // 2, 1024))):
                try:
                    cpu = cpu_baseline(args.config, n_try * (2 if is_elbo else 1))
                    break
                except Exception as e:  # noqa: BLE001
                    cpu = {"value": None, "unit": "elbo/s" if is_elbo else "logpdf/s", "cores": 0, "kind": "port",
                           "sample": f"CPU baseline failed on this host at sample N={n_try}", "error": str(e)[-400:]}
        line = {
            "metric": "elbo_per_sec" if is_elbo else "logpdf_per_sec",
            "value": 1e3 / ms_per_step, "unit": "elbo/s" if is_elbo else "logpdf/s",
            # physical GPUs (advisor, round 3: several loopback ranks on one GPU are ranks, not GPUs)
            "n_gpus": len(set(devs)) if inproc else world,
            "ranks": len(devs) if inproc else world,
            "loopback": bool(inproc and len(set(devs)) < len(devs)),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": bc.describe(args.config) + (", host-buffer C-ABI" if is_elbo else ""),
                       # what `value` times (the round prompt: inputs resident in HBM when the timed region starts); the same step
                       # through the host-buffer entry point the `ccall` binds -- SURVEY 8d's definition, <= 4 MB of copies
                       # included -- is `host_api` beside it (1.002 x in round 4)
                       "timed_entry": ("sgp_elbo (host buffers: the ELBO has no device-resident entry point)" if is_elbo else
                                       "sgp_logpdf_f32 (host buffers)" if args.dtype == "f32" else
                                       "sgp_logpdf on the multi-GPU context (host buffers)" if inproc else
                                       "dist_logpdf (process per GPU, device-resident panels)" if use_dist else
                                       "sgp_dev_logpdf (X, y device-resident; the host-buffer sgp_logpdf is timed in host_api)"),
                       "N": N, "D": D, "kernel": kind,
                       "parallelism": (f"in-process multi-GPU context x{len(devs)} ({'data points' if is_elbo else 'column panels'} "
                                       f"sharded inside libsthenomi, transport {ctx.transport})" if inproc
                                       else f"column-panel x{world}" if use_dist else "1 GPU"),
                       "panel_width": (multi or {}).get("panel_width") if inproc else (args.panel if use_dist else None)},
            "cholesky_tflops_whole_step": whole_tflops,  # (c4: ELBO flops of SURVEY 8d; structured models: the flops that ran)
            "executed_work_fraction": None if is_elbo else (work_frac if args.dtype == "f64" else 1.0),
            "dense_equivalent_tflops": None if is_elbo else (N ** 3 / 3.0) / (ms_per_step * 1e-3) / 1e12,
            "cholesky_frac_of_fp64_matrix_peak": (whole_tflops / (PEAK_FP64_MFMA_TFLOPS * (len(set(devs)) if inproc else world))
                                                  if args.dtype == "f64" else None),
            "multi_gpu": multi,
            "logpdf": val, "golden": gval, "parity_rel": parity,
            "stages": stages, "roofline": roofline, "host_api": host_api, "cpu_baseline": cpu,
            "north_star_target": north_star, "sizes": sizes, "grad": grad,
        }
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
