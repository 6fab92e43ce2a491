"""Multi-GPU logpdf: one process per GPU, the N x N covariance sharded in column panels
(block-cyclic, width W) across ranks, right-looking Cholesky with one-panel look-ahead;
panels travel between ranks with torch.distributed broadcast (backend "nccl" == RCCL over xGMI
on the MI355X node; "gloo" in the CPU tests), scalars (logdet, |L^-1 (y - m)|^2, info) with one
all-reduce at the end.  SURVEY.md 8e / DESIGN.md section 6.

Round 4 (the schedule of csrc/multi.hip, same building blocks): a rank's trailing panels are updated by ONE launch
per step (sgp_dev_panel_update_batch), and the chain factorisation -> broadcast -> look-ahead update is pipelined in
sub-panels of `SUBPANEL` columns: a panel is factored sub-panel by sub-panel, each broadcast as soon as it is final
while the next is being factored, and the owner of the next panel applies them as they land.

The reference has no distributed path at all: this is the MI355X-side
scaling of `logpdf(f(X, s2), y)` (AbstractGPs.logpdf [EXT], Appendix A.3).

All numerics are the C-ABI building blocks of include/sthenomi.h (sgp_dev_assemble_cols,
sgp_dev_panel_factor, sgp_dev_panel_update, sgp_dev_rowsumsq) driven through an `ops` object;
`HipOps` is the product implementation.  The orchestration below is backend-agnostic so that
the CPU test-suite can drive it with a NumPy test double over gloo (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import lib as _lib

LOG2PI = math.log(2.0 * math.pi)
TILE = 128
SUBPANEL = 512   # sub-panel width of the factor / broadcast / look-ahead pipeline (multiple of 128; >= W: whole panels)


def geometry(N, ncols):
    n_pad = max(TILE, (N + TILE - 1) // TILE * TILE)
    m_tot = n_pad + ((ncols + TILE - 1) // TILE * TILE if ncols > 0 else 0)
    return n_pad, m_tot


class PanelLayout:
    """Block-cyclic ownership of the column panels of the bordered matrix.  A rank stores its panels
    PACKED: panel J holds rows J0 .. m_tot only, as a contiguous (m_tot - J0) x W column-major block
    (leading dimension ld(J)), one after the other -- a factored panel is broadcast straight out of
    its storage, with no packing copy."""

    def __init__(self, n_pad, W, world, rank, m_tot=None):
        assert W % TILE == 0
        self.n_pad, self.W, self.world, self.rank = n_pad, W, world, rank
        self.m_tot = n_pad if m_tot is None else m_tot
        self.n_panels = (n_pad + W - 1) // W
        self.mine = [J for J in range(self.n_panels) if J % world == rank]
        self._off, tot = {}, 0
        for J in self.mine:
            self._off[J] = tot
            tot += self.ld(J) * self.width(J)
        self.n_local = tot

    def owner(self, J):
        return J % self.world

    def col0(self, J):
        return J * self.W

    def width(self, J):
        return min(self.W, self.n_pad - J * self.W)

    def ld(self, J):
        return self.m_tot - J * self.W

    def offset(self, J):
        """offset (in doubles) of owned panel J in the rank's storage"""
        return self._off[J]

    def count(self, J):
        return self.ld(J) * self.width(J)

    def local_index(self, J):
        return J // self.world

    def n_local_cols(self):
        return len(self.mine) * self.W

    def n_local_doubles(self):
        return max(1, self.n_local)


class HipOps:
    """Product backend: torch CUDA tensors for HBM, libsthenomi.so for every kernel."""

    def __init__(self, ctx=None):
        import torch
        self.torch = torch
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.device = torch.device("cuda", self.ctx.device)
        # every kernel of this rank (torch copies, libsthenomi kernels, collectives' stream
        # dependencies) is ordered on ONE explicit non-default stream: the HIP null stream (handle
        # 0) cannot be named through the C-ABI, where NULL means "the ctx's own stream".
        self._stream = torch.cuda.Stream(self.device)                  # trailing updates
        self._pstream = torch.cuda.Stream(self.device, priority=-1)   # panel factorisation (look-ahead)
        self._events = {}

    def stream_context(self):
        return self.torch.cuda.stream(self._stream)

    def panel_context(self):
        """High-priority stream for the look-ahead panel (update + factor + pack + broadcast)."""
        return self.torch.cuda.stream(self._pstream)

    # A rank's trailing panels are updated by one launch each; issuing them round-robin on a few
    # streams lets the tail of one launch overlap the head of the next (a panel always uses the
    # same stream, so successive updates of one panel stay ordered).
    N_POOL = 3

    def fork_updates(self):
        if not hasattr(self, "_pool"):
            self._pool = [self.torch.cuda.Stream(self.device) for _ in range(self.N_POOL)]
            self._pool_ev = [self.torch.cuda.Event() for _ in range(self.N_POOL)]
            self._fork_ev = self.torch.cuda.Event()
        self._fork_ev.record(self.torch.cuda.current_stream(self.device))
        for st in self._pool:
            st.wait_event(self._fork_ev)

    def pool_context(self, idx):
        return self.torch.cuda.stream(self._pool[idx % self.N_POOL])

    def join_updates(self):
        cur = self.torch.cuda.current_stream(self.device)
        for st, ev in zip(self._pool, self._pool_ev):
            ev.record(st)
            cur.wait_event(ev)

    def record(self, name):
        ev = self._events.get(name)
        if ev is None:
            ev = self._events[name] = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))

    def wait(self, name):
        ev = self._events.get(name)
        if ev is not None:
            self.torch.cuda.current_stream(self.device).wait_event(ev)

    # -- memory ---------------------------------------------------------------------------
    def empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device=self.device)

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def izeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.int32, device=self.device)

    def stream(self):
        h = self.torch.cuda.current_stream(self.device).cuda_stream
        if not h:
            raise _lib.SthenoMIError("HipOps kernels must run inside ops.stream_context()")
        return h

    def to_host(self, t):
        return t.cpu().numpy()

    # -- model ----------------------------------------------------------------------------
    def make_dspec(self, spec):
        h = C.c_void_p()
        _lib.check(self.lib.sgp_dspec_create(self.ctx.handle, spec.ref(), C.byref(h)), "sgp_dspec_create")
        return h

    def free_dspec(self, h):
        self.lib.sgp_dspec_destroy(h)

    # -- kernels --------------------------------------------------------------------------
    # Panels are packed (PanelLayout): `off` is the offset of the panel's first stored element (its row
    # J0) in the rank's storage tensor, `ld` its leading dimension m_tot - J0.  The C-ABI building blocks
    # index rows globally, so they get the (virtual) address of global row 0 = first stored row - J0.
    def assemble_cols(self, ds, N, c0, nc, A, off, ld, m_tot, mean, sigma2, Y, ncols):
        nz = np.array([sigma2], dtype=np.float64)
        rc = self.lib.sgp_dev_assemble_cols(self.ctx.handle, ds, N, c0, nc, A.data_ptr() + 8 * (off - c0), ld, m_tot,
                                            mean.data_ptr() if mean is not None else None,
                                            _lib.NOISE_SCALAR, _lib.dptr(nz), None,
                                            Y.data_ptr() if Y is not None else None, N, ncols, self.stream())
        _lib.check(rc, "sgp_dev_assemble_cols")

    def panel_factor(self, A, off, ld, m, J0, w, logdet, info):
        rc = self.lib.sgp_dev_panel_factor(self.ctx.handle, A.data_ptr() + 8 * off, ld, m, w, J0, logdet.data_ptr(),
                                           info.data_ptr(), self.stream())
        _lib.check(rc, "sgp_dev_panel_factor")

    def panel_update(self, Pt, p_off, ldp, J0, w, A, off, ld, c0, nc, m_tot):
        """panel at A[off..] (first stored row c0, leading dimension ld) -= P[rows >= c0] P[rows c0..c0+nc]',
        P = the factored panel J at Pt[p_off..] (first stored row J0, leading dimension ldp)"""
        rc = self.lib.sgp_dev_panel_update(self.ctx.handle, Pt.data_ptr() + 8 * p_off, ldp, J0, w,
                                           A.data_ptr() + 8 * (off - c0), ld, c0, nc, m_tot, self.stream())
        _lib.check(rc, "sgp_dev_panel_update")

    def panel_update_batch(self, srcs, dsts, m_tot):
        """ONE launch: every destination (tensor, offset, ld, c0, w, src_first, src_count) -= its rows of the sources
        (tensor, offset, ld, row0, w) of its range, applied in order, times their rows of its diagonal block (the lower
        trapezoid: tile rows >= tile columns).  Offsets address the first STORED element of a packed panel."""
        if not srcs or not dsts:
            return
        sa = (_lib.sgp_panel_src * len(srcs))()
        for q, (t, off, ld, row0, w) in enumerate(srcs):
            sa[q] = _lib.sgp_panel_src(t.data_ptr() + 8 * off, ld, row0, w)
        da = (_lib.sgp_panel_dst * len(dsts))()
        for d, (t, off, ld, c0, w, s0, sn) in enumerate(dsts):
            da[d] = _lib.sgp_panel_dst(t.data_ptr() + 8 * off, ld, c0, w, s0, sn)
        rc = self.lib.sgp_dev_panel_update_batch(self.ctx.handle, sa, len(srcs), da, len(dsts), m_tot, self.stream())
        _lib.check(rc, "sgp_dev_panel_update_batch")

    def rowsumsq(self, A, off, ld, nc, nrows, out):
        rc = self.lib.sgp_dev_rowsumsq(self.ctx.handle, A.data_ptr() + 8 * off, ld, nc, nrows, out.data_ptr(),
                                       self.stream())
        _lib.check(rc, "sgp_dev_rowsumsq")

    # -- posterior on the sharded factor -----------------------------------------------------
    def assemble_cross_rows(self, dx, c0, nc, A, off, ld, row0):
        rc = self.lib.sgp_dev_assemble_cross_rows(self.ctx.handle, dx, c0, nc, A.data_ptr() + 8 * (off - c0), ld, row0,
                                                  self.stream())
        _lib.check(rc, "sgp_dev_assemble_cross_rows")

    def rows_dot(self, A, r_off, ld, nrows, nc, z_off, pv, ns_pad):
        """pv[:ns_pad] += rowsumsq, pv[ns_pad:] += rows . z   (rows at A[r_off..], z row at A[z_off..])"""
        rc = self.lib.sgp_dev_rows_dot(self.ctx.handle, A.data_ptr() + 8 * r_off, ld, nrows, nc,
                                       A.data_ptr() + 8 * z_off, pv.data_ptr(), pv.data_ptr() + 8 * ns_pad,
                                       self.stream())
        _lib.check(rc, "sgp_dev_rows_dot")

    def rows_gram(self, A, r_off, ld, nrows_pad, nc, G):
        rc = self.lib.sgp_dev_rows_gram(self.ctx.handle, A.data_ptr() + 8 * r_off, ld, nrows_pad, nc, G.data_ptr(),
                                        nrows_pad, self.stream())
        _lib.check(rc, "sgp_dev_rows_gram")

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    # -- sharded ELBO -----------------------------------------------------------------------
    def prior_var(self, f, x):
        from . import finite_gp as _fg
        return np.ascontiguousarray(_fg.prior_var(f, x)) if len(x) else np.zeros(0)

    def prior_cov(self, f, x):
        from . import finite_gp as _fg
        return np.ascontiguousarray(_fg.prior_cov(f, x))

    def elbo_part(self, M):
        n = C.c_int64()
        _lib.check(self.lib.sgp_elbo_part_len(M, C.byref(n)), "sgp_elbo_part_len")
        return self.torch.zeros(n.value, dtype=self.torch.float64, device=self.device)

    # sgp_dev_elbo_partial / _finish take no stream argument: they run on the context's own stream, which is not
    # ordered with torch's.  `part` is zero-filled (and, between the two calls, all-reduced) by work enqueued on the
    # current torch stream, so that stream is drained before the library touches the buffer -- otherwise a zero-fill
    # (or a still running all-reduce) that lands late overwrites / precedes what the library wrote / reads.  The
    # library drains its own streams before it returns (CtxScope), so torch-side consumers need nothing more.
    def _drain_current_stream(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def elbo_partial(self, zz, xz, var_x, mean_x, nk, nbuf, zk, zbuf, y, part):
        self._drain_current_stream()
        rc = self.lib.sgp_dev_elbo_partial(self.ctx.handle, zz.ref(), xz.ref(), _lib.dptr(var_x), _lib.dptr(mean_x), nk,
                                           _lib.dptr(nbuf), zk, _lib.dptr(zbuf), _lib.dptr(y), part.data_ptr(),
                                           part.numel())
        _lib.check(rc, "sgp_dev_elbo_partial")

    def elbo_finish(self, M, N_total, part):
        out = np.zeros(1)
        self._drain_current_stream()
        _lib.check(self.lib.sgp_dev_elbo_finish(self.ctx.handle, M, N_total, part.data_ptr(), _lib.dptr(out)),
                   "sgp_dev_elbo_finish")
        return float(out[0])


def dist_logpdf(ops, spec, y, mean, sigma2, world=1, rank=0, group=None, W=1024, A=None, stats=None,
                always_collective=False):
    """logpdf(f(X, sigma2), y) with the covariance sharded over `world` ranks.

    spec : lib.Spec (symmetric) of the prior covariance, identical on every rank
    y    : (N,) observations, mean: (N,) prior mean or None
    A    : optional preallocated local panel storage (PanelLayout(...).n_local_doubles() doubles)
    always_collective : issue the panel broadcasts / final all-reduces even when world == 1 (a
           one-rank RCCL communicator executes them as self-copies: exercises the backend calls)
    Every rank returns the same float.  Raises lib.PosDefException like the single-GPU path."""
    ds = ops.make_dspec(spec)
    try:
        with ops.stream_context():
            return _dist_logpdf(ops, ds, spec.N, y, mean, sigma2, world, rank, group, W, A, stats, always_collective)
    finally:
        ops.free_dspec(ds)


def dist_posterior_predict(ops, spec, cross, y, mean, sigma2, prior_mean_s, prior_var_s, prior_cov_s=None,
                           world=1, rank=0, group=None, W=1024, always_collective=False):
    """Posterior marginals at test points x* from the SHARDED factorisation: what
    `mean_and_var(posterior(f(x, s2), y)(x*))` (and `cov(...)` when prior_cov_s is given) returns on the
    reference path (AbstractGPs.posterior / PosteriorGP mean, var, cov [EXT], App. A.5; reference call sites
    /root/reference/test/gp/util.jl, /root/reference/src/gp/sparse_finite_gp.jl:60-62 for the dispatch).

    K(x*, x) is appended to the bordered matrix as extra rows below the observation row; the column-panel
    factorisation that computes logpdf turns them into V' = K(x*, x) L^-T on every rank's own columns, so
        mean* = m* + V' z,   var* = k** - sum_c V'^2,   cov* = K** - V' V      (z = L^-1 (y - m))
    are sums over columns: ONE all-reduce of 2 n* (+ n*^2) doubles, no distributed triangular solve.

    spec  : lib.Spec (symmetric) of cov(f, x);  cross : lib.Spec of cov(f, x*, x) (rows x*, columns x)
    prior_mean_s, prior_var_s (, prior_cov_s) : m(x*), diag K(x*, x*) (, K(x*, x*)) as host arrays
    Returns (logpdf, mean*, var*, cov* or None), identical on every rank."""
    ds = ops.make_dspec(spec)
    dx = ops.make_dspec(cross)
    try:
        with ops.stream_context():
            out = {}
            lp = _dist_logpdf(ops, ds, spec.N, y, mean, sigma2, world, rank, group, W, None, None, always_collective,
                              cross=(dx, cross.N, prior_cov_s is not None), post=out)
    finally:
        ops.free_dspec(ds)
        ops.free_dspec(dx)
    ns = cross.N
    mean_s = np.asarray(prior_mean_s, dtype=np.float64).reshape(-1) + out["dot"][:ns]
    var_s = np.asarray(prior_var_s, dtype=np.float64).reshape(-1) - out["sumsq"][:ns]
    cov_s = None
    if prior_cov_s is not None:
        cov_s = np.asarray(prior_cov_s, dtype=np.float64) - out["gram"][:ns, :ns]
    return lp, mean_s, var_s, cov_s


def dist_posterior(ops, fx, y, xs, want_cov=False, world=1, rank=0, group=None, W=1024, always_collective=False):
    """`mean_and_var(posterior(fx, y)(xs))` (and `cov` when want_cov) with the factorisation sharded over
    `world` ranks: fx = f(x, s2) a FiniteGP of a prior process with isotropic noise, xs the test inputs (any
    input collection of the host mirror).  Builds the two specs and the prior moments at xs and calls
    dist_posterior_predict.  Returns (mean*, var*, cov* or None); every rank gets the same arrays."""
    from .flatten import build_spec
    from .gp import mean_vector
    noise = np.asarray(fx.noise, dtype=np.float64)
    if noise.ndim != 0:
        raise ValueError("the sharded posterior takes isotropic observation noise")
    spec, _, _ = build_spec(fx.f, fx.x)
    cross, _, _ = build_spec(fx.f, xs, fx.f, fx.x)
    mean_x = mean_vector(fx.f, fx.x)
    mean_x = None if not np.any(mean_x) else np.ascontiguousarray(mean_x)
    with ops.stream_context():
        var_s = ops.prior_var(fx.f, xs)
        cov_s = ops.prior_cov(fx.f, xs) if want_cov else None
    _, m, v, c = dist_posterior_predict(ops, spec, cross, y, mean_x, float(noise), mean_vector(fx.f, xs), var_s, cov_s,
                                        world=world, rank=rank, group=group, W=W, always_collective=always_collective)
    return m, v, c


def _dist_logpdf(ops, ds, N, y, mean, sigma2, world, rank, group, W, A, stats, always_collective=False,
                 cross=None, post=None):
    import torch
    import torch.distributed as dist

    n_pad, m_tot = geometry(N, 1)
    ns = ns_pad = 0
    if cross is not None:   # test points as extra bordered rows (dist_posterior_predict)
        ns = cross[1]
        ns_pad = (ns + TILE - 1) // TILE * TILE
        row_s = m_tot
        m_tot += ns_pad
    W = min(W, n_pad)
    lay = PanelLayout(n_pad, W, world, rank, m_tot)
    if A is None:
        A = ops.empty(lay.n_local_doubles())
    dY = ops.from_host(np.asarray(y, dtype=np.float64))
    dmean = ops.from_host(mean) if mean is not None else None
    logdet = ops.zeros(1)
    sq = ops.zeros(1)
    info = ops.izeros(1)
    need_bufs = world > 1 or always_collective
    bufs = [ops.empty(m_tot * W), ops.empty(m_tot * W)] if need_bufs else [None, None]

    # 1. every rank assembles its own column panels (no communication)
    for J in lay.mine:
        ops.assemble_cols(ds, N, lay.col0(J), lay.width(J), A, lay.offset(J), lay.ld(J), m_tot, dmean, sigma2, dY, 1)
        if ns:
            ops.assemble_cross_rows(cross[0], lay.col0(J), lay.width(J), A, lay.offset(J), lay.ld(J), row_s)

    def panel(J):
        """(tensor, offset) of factored panel J on this rank: its own storage, or the receive buffer"""
        if lay.owner(J) == rank:
            return A, lay.offset(J)
        return bufs[J % 2], 0

    sub_w = max(TILE, SUBPANEL // TILE * TILE)

    def subs(J):
        """sub-panels (first column inside the panel, width) of panel J"""
        w = lay.width(J)
        return [(c, min(sub_w, w - c)) for c in range(0, w, sub_w)]

    def bcast_sub(J, c, wq):
        """columns [c, c + wq) of panel J: a contiguous slice of the packed panel (all its stored rows)"""
        if not need_bufs:
            return None
        lo, hi = c * lay.ld(J), (c + wq) * lay.ld(J)
        if lay.owner(J) == rank:     # straight out of the packed storage: no packing copy
            t = A[lay.offset(J) + lo: lay.offset(J) + hi]
        else:
            t = bufs[J % 2][lo:hi]
        return dist.broadcast(t, src=_global_rank(group, lay.owner(J)), group=group, async_op=True)

    def receive(J):
        return [bcast_sub(J, c, wq) for c, wq in subs(J)]

    def factor_and_send(J):
        """(owner, panel stream) factor panel J sub-panel by sub-panel; every finished sub-panel is broadcast at once and
        the panel's remaining columns are updated with it (K = the sub-panel's width; a tile sees k ascending exactly as
        in the 128-column steps of an unsplit panel)"""
        works = []
        ld, w, J0, off = lay.ld(J), lay.width(J), lay.col0(J), lay.offset(J)
        for c, wq in subs(J):
            ops.panel_factor(A, off + c + c * ld, ld, ld - c, J0 + c, wq, logdet, info)
            works.append(bcast_sub(J, c, wq))
            r = c + wq
            if r < w:
                ops.panel_update_batch([(A, off + c * ld, ld, J0, wq)],
                                       [(A, off + r + r * ld, ld, J0 + r, w - r, 0, 1)], m_tot)
        return works

    def source(J, c=0, wq=None):
        Pt, p_off = panel(J)
        return (Pt, p_off + c * lay.ld(J), lay.ld(J), lay.col0(J), lay.width(J) if wq is None else wq)

    def dest(Jp):
        return (A, lay.offset(Jp), lay.ld(Jp), lay.col0(Jp), lay.width(Jp), 0, 1)

    def wait_all(works):
        for wk in works:
            if wk is not None:
                wk.wait()

    # 2. right-looking factorisation with one-panel look-ahead.  Two streams per rank: the owner of
    # the next panel updates + factors + broadcasts it on the panel stream while its other
    # trailing panels are still being updated with the current panel on the update stream.
    #   "upd"   : update-stream work of the previous step (and the assembly) is complete
    #   "panel" : the panel this rank just factored is final in its (packed) storage
    ops.record("upd")
    if lay.owner(0) == rank:
        with ops.panel_context():
            ops.wait("upd")
            works = factor_and_send(0)
            ops.record("panel")
    else:
        works = receive(0)
    for J in range(lay.n_panels):
        nxt = J + 1
        # (a) look-ahead on the owner of the next panel: the update follows panel J sub-panel by sub-panel
        works_next = None
        if nxt < lay.n_panels:
            if lay.owner(nxt) == rank:
                with ops.panel_context():
                    ops.wait("upd")            # step J-1's updates of panel nxt are done
                    if lay.owner(J) == rank:
                        ops.wait("panel")
                    for (c, wq), wk in zip(subs(J), works):
                        if wk is not None:
                            wk.wait()          # this sub-panel has landed
                        ops.panel_update_batch([source(J, c, wq)], [dest(nxt)], m_tot)
                    works_next = factor_and_send(nxt)
                    ops.record("panel")
            else:
                works_next = receive(nxt)      # ordered after step J-1's readers of that buffer
        # (b) the update stream needs all of panel J (own storage on its owner, bufs[J % 2] elsewhere); the rest of this
        # rank's trailing panels in ONE launch
        wait_all(works)
        if lay.owner(J) == rank:
            ops.wait("panel")
        rest = [dest(Jp) for Jp in lay.mine if Jp > nxt]
        if rest:
            ops.panel_update_batch([source(J)], rest, m_tot)
        ops.record("upd")
        works = works_next
    with ops.panel_context():
        pass
    ops.wait("panel")   # join: everything the panel stream did is visible to the update stream

    # 3. scalars: |L^-1 (y - m)|^2 from the bordered row, logdet, info
    for J in lay.mine:
        nc = min(lay.width(J), max(0, N - lay.col0(J)))
        if nc > 0:
            ops.rowsumsq(A, lay.offset(J) + (n_pad - lay.col0(J)), lay.ld(J), nc, 1, sq)
    if ns:
        # posterior sums over this rank's columns: V' z, rowsumsq(V') (and V' V), then one all-reduce
        pv = ops.zeros(2 * ns_pad)
        G = ops.zeros(ns_pad * ns_pad) if cross[2] else None
        for J in lay.mine:
            nc = min(lay.width(J), max(0, N - lay.col0(J)))
            if nc > 0:
                r_off = lay.offset(J) + (row_s - lay.col0(J))
                ops.rows_dot(A, r_off, lay.ld(J), ns, nc, lay.offset(J) + (n_pad - lay.col0(J)), pv, ns_pad)
                if G is not None:
                    ops.rows_gram(A, r_off, lay.ld(J), ns_pad, lay.width(J), G)
        if world > 1 or always_collective:
            ops.synchronize()
            dist.all_reduce(pv, op=dist.ReduceOp.SUM, group=group)
            if G is not None:
                dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
        pv_h = ops.to_host(pv)
        post["sumsq"], post["dot"] = pv_h[:ns_pad].copy(), pv_h[ns_pad:].copy()
        if G is not None:
            post["gram"] = ops.to_host(G).reshape(ns_pad, ns_pad).copy()   # symmetric: storage order irrelevant
    red = torch.stack([logdet.reshape(()), sq.reshape(())])
    inf = info.to(torch.float64)
    big = float(2 ** 52)
    inf = torch.where(inf > 0, inf, torch.full_like(inf, big))
    if world > 1 or always_collective:
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(inf, op=dist.ReduceOp.MIN, group=group)
    red_h = ops.to_host(red)
    inf_h = float(ops.to_host(inf)[0])
    if stats is not None:
        stats.update(n_pad=n_pad, m_tot=m_tot, W=W, n_panels=lay.n_panels, local_panels=len(lay.mine),
                     logdet=float(red_h[0]), sqmahal=float(red_h[1]))
    if inf_h < big:
        raise _lib.PosDefException(int(inf_h), "distributed Cholesky")
    return -0.5 * (N * LOG2PI + float(red_h[0]) + float(red_h[1]))


# ---- sparse ELBO, sharded over the data points ------------------------------------------------------
def slice_inputs(x, lo, hi):
    """x[lo:hi] for the input collections of the host mirror (1-D vector, ColVecs, GPPPInput, BlockData);
    a BlockData keeps its block structure (blocks outside the range become empty)."""
    from .inputs import BlockData, ColVecs, GPPPInput, blocks
    if isinstance(x, BlockData):
        out, off = [], 0
        for b in blocks(x):
            n = len(b)
            a, e = min(max(lo - off, 0), n), min(max(hi - off, 0), n)
            out.append(slice_inputs(b, a, e))
            off += n
        return BlockData(out)
    if isinstance(x, GPPPInput):
        return GPPPInput(x.p, slice_inputs(x.x, lo, hi))
    if isinstance(x, ColVecs):
        return ColVecs(np.asfortranarray(x.X[:, lo:hi]))
    return np.asarray(x)[lo:hi]


def shard_rows(N, world, rank):
    """contiguous, balanced slice of the N data points owned by `rank`"""
    return (N * rank) // world, (N * (rank + 1)) // world


def dist_elbo(ops, vfe, fx, y, world=1, rank=0, group=None, always_collective=False):
    """elbo(VFE(fz), fx, y) (AbstractGPs.elbo [EXT], App. A.6; the reference reaches it through
    /root/reference/src/gp/sparse_finite_gp.jl:52-58) with the N data points sharded over `world`
    ranks -- SURVEY.md 8e: every rank builds its slice of K(x,z), solves it against its own copy of
    chol(K(z,z) + Sigma_z) and forms its partial sums; ONE all-reduce of the "part" (M^2 + M + 2
    meaningful doubles) follows; every rank finishes on the reduced sums, so all return the same float
    with no further exchange.  `ops` supplies elbo_part / elbo_partial / elbo_finish (HipOps: the C-ABI
    entry points sgp_dev_elbo_partial / sgp_dev_elbo_finish)."""
    import torch.distributed as dist
    from . import finite_gp as _fg
    from .gp import mean_vector
    fz = vfe.fz
    if fz.f is not fx.f:
        raise AssertionError("VFE requires fz.f === fx.f")
    N, M = len(fx), len(fz)
    lo, hi = shard_rows(N, world, rank)
    xs = slice_inputs(fx.x, lo, hi)
    noise = np.asarray(fx.noise, dtype=np.float64)
    if noise.ndim > 1:
        raise ValueError("elbo needs isotropic or diagonal observation noise")
    noise_s = noise if noise.ndim == 0 else noise[lo:hi]
    fxs = _fg.FiniteGP(fx.f, xs, noise_s)
    zz, xz, mean_x, nk, nbuf, zk, zbuf = _fg._vfe_args(vfe, fxs)
    yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).ravel()[lo:hi])
    with ops.stream_context():
        var_x = ops.prior_var(fx.f, xs)
        part = ops.elbo_part(M)
        ops.elbo_partial(zz, xz, var_x, mean_x, nk, nbuf, zk, zbuf, yv, part)
        if world > 1 or always_collective:
            ops.synchronize()
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
        return ops.elbo_finish(M, N, part)


def _global_rank(group, r):
    import torch.distributed as dist
    if group is None:
        return r
    return dist.get_global_rank(group, r)
