// In-process multi-GPU operators behind the C-ABI: sgp_ctx_create_multi(devices, ndev, &ctx) returns an ordinary
// sgp_ctx whose sgp_logpdf / sgp_posterior_create + _predict / sgp_rand / sgp_elbo shard their work over the
// listed GPUs, so that ONE `ccall` from the Julia host reaches the whole node.  The reference
// has no distributed code; the contract is BASELINE.json's north_star.
//
// Dense operators (logpdf, posterior, rand) -- layout: outer column panels of W columns, block-cyclic over the
// ranks (= devices); every rank keeps its panels PACKED (panel J holds rows J0 .. m_tot only, leading dimension
// m_tot - J0), so a factored panel is one contiguous block that travels as is -- no packing copy.
// Right-looking blocked Cholesky, driven by one host thread that only enqueues.  Round 4 rebuilt the schedule around what
// the round-3 profile showed -- at 8 ranks the CHAIN (panel factorisation -> transport -> look-ahead update of the next
// panel -> its factorisation ...: 74 + 65 + 74 ms at N = 65536) is as long as a rank's whole share of the trailing updates
// (182 ms), and those ran as 8 small launches per rank and step:
//  * ONE update launch per rank and step (gemm_nt.hip: gemm_nt_seg_kernel -- a list of destination panels, each contracted
//    over a range of source panels): 8 loopback ranks on one GPU 1893 -> 1635 ms, per-rank updates at the single-GPU
//    kernel's rate;
//  * the chain is PIPELINED in sub-panels (SGP_MULTI_SUBPANEL, 512 columns): a panel is factored sub-panel by sub-panel
//    (each followed by one update of the panel's remaining columns), every finished sub-panel is sent while the next is
//    being factored, and the owner of the next panel applies them as they land -- only the LAST sub-panel's transport
//    and update sit between the end of one factorisation and the start of the next (projected 8-GPU step 234 -> 216 ms);
//  * optional (both measured, neither pays at N = 65536, off by default): panel GROUPS (SGP_MULTI_GROUP = G > 1) -- with
//    g the group of the newest factored panel J a rank's panels fall into
//      current group g  ("near A", stream s_near)  updated with panel J as soon as it arrives          K = one panel
//      next group g + 1 ("near B", stream s_upd)   the same, behind whatever s_upd still has queued    K = one panel
//      groups >= g + 2  ("far",    stream s_upd)   updated ONCE per group with the whole group          K = G panels
//    (the single-GPU schedule's deep blocking; the far launches reach the same 64 - 66 TFLOP/s as the batched K = one
//    panel launches, and the near classes cost what the far class saves) -- and MIXED panel widths (SGP_MULTI_PANEL_TAIL
//    < SGP_MULTI_PANEL for the last 1 - SGP_MULTI_TAIL_FRAC of the columns: the panel work is dominated by the tall
//    early panels, so a narrow tail does not shorten it).  With G = 1 every trailing panel is "far" and every step ends a
//    group: one launch per rank and step.  A tile sees k ascending through the same tile program whatever the grouping:
//    for one panel layout the results are bit-identical for every G.
// Receive buffers: a ring of 2 G + 2 per rank.
// Transport (SGP_MULTI_TRANSPORT=rccl|p2p|auto): RCCL ncclBroadcast in one group call per panel over communicators
// from ncclCommInitAll (librccl is dlopen'ed here, not linked), or peer copies.  xGMI is point to point (one link
// per GPU pair), so a plain owner -> receiver copy is bound by ONE link per receiver: the peer-copy transport moves
// a panel as scatter + all-gather -- the owner sends each of the P - 1 peers a different 1 / (P - 1) slab, the
// peers then exchange slabs among themselves -- so that all P - 1 ingress links of every receiver carry traffic
// (SGP_MULTI_BCAST=direct restores the one-link form).  A device listed more than once gives several ranks on one
// GPU ("loopback": same-device copies) -- that is how the 1-GPU test box exercises the multi-rank orchestration,
// both broadcast forms included, with the real kernels.
// Scalars (logdet, |L^-1 (y - m)|^2 per column) are all-reduced (ncclAllReduce) or summed on the host in rank order.
//
// posterior: the sharded factor is KEPT (with the inverse diagonal blocks of every panel) so that repeated
// predictions run against it: K(x*, x) is assembled column-sharded like the factor, V' = K(x*, x) L^-T by a
// left-looking sweep -- every rank forms the partial sums of its own panels, the owner of panel J collects them
// (P - 1 messages of n* x W doubles), subtracts in rank order and solves against its diagonal block -- and
// mean* - m* = V' z, var*, cov* are sums over columns, i.e. one small reduction over ranks.  alpha = L^-T z by a
// panel-wise back substitution (one W-vector broadcast per panel).
// rand: m + L Z = sum over panels of L[:, J] Z[J, :]: every rank multiplies its own panels, one reduction.
// elbo: the data points are sharded (contiguous slices), every rank turns its slice into a "part" (see
// sgp_dev_elbo_partial) on a host thread of its own, ONE reduction of M^2 + M + 2 doubles, rank 0 finishes.
#include "driver.h"
#include "own_table.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <unordered_map>

using namespace sgp;

#define M_CHECK_ARG(cond, msg)   \
  do {                           \
    if (!(cond)) {               \
      sgp::set_error(msg);       \
      return -1;                 \
    }                            \
  } while (0)
#define M_RC(expr)            \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

namespace {

// ---- the few RCCL entry points used, resolved at run time -------------------------------------------
typedef void* ncclComm_p;
struct Rccl {
  void* h = nullptr;
  int (*CommInitAll)(ncclComm_p*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_p) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(ncclComm_p, int*) = nullptr;
  int (*CommAbort)(ncclComm_p) = nullptr;
  bool load() {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
    AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
    GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    CommCount = (decltype(CommCount))dlsym(h, "ncclCommCount");
    CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
    return CommInitAll && CommDestroy && Broadcast && AllReduce && GroupStart && GroupEnd;
  }
};
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;  // rccl.h: ncclFloat64 = 8, ncclSum = 0

enum { TR_LOOPBACK = 0, TR_P2P = 1, TR_RCCL = 2 };

struct Rank {
  int dev = 0;
  sgp_ctx* ctx = nullptr;        // child context (kernels + scratch of this rank)
  hipStream_t s_upd = nullptr, s_panel = nullptr, s_comm = nullptr;
  hipStream_t s_near = nullptr;   // updates of the current group's panels (the ones the chain needs next)
  hipEvent_t ev_upd = nullptr, ev_fact = nullptr, ev_done = nullptr;
  hipEvent_t ev_A = nullptr;      // s_near: this step's "near A" updates are done
  hipEvent_t ev_B = nullptr;      // s_upd: this step's "near B" updates are done (recorded BEFORE a far update)
  static constexpr int NGEV = 8;
  hipEvent_t ev_far[NGEV] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // s_upd: the far update with group g (slot g % NGEV)
  std::vector<hipEvent_t> ev_recv;   // [ring slot] the panel in this receive buffer has landed (all of it)
  static constexpr int NSUB = 8;     // a panel is factored and sent in up to NSUB sub-panels (sgp_multi::sub columns each)
  hipEvent_t ev_sub[NSUB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // s_panel: sub-panel q of the panel being factored is final
  std::vector<std::vector<hipEvent_t>> ev_recv_sub;   // [ring slot][q] sub-panel q has landed
  // scatter + all-gather panel transport: one incoming stream per source rank, one "piece landed" event per
  // (receive buffer, source), one "buffer free" event per receive buffer
  std::vector<hipStream_t> s_in;
  std::vector<std::vector<hipEvent_t>> ev_in;   // [ring slot][source rank]
  std::vector<hipEvent_t> ev_free;              // [ring slot]
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // timing: first / last trailing update of a call (sgp_ctx_multi_stats)
  bool t0_set = false;
  double upd_flops = 0.0, upd_span_ms = 0.0, recv_bytes = 0.0;
  long n_factored = 0;
  const sz_word* d_sz = nullptr;   // this rank's device copy of the factor's tile pattern (structural zeros), or nullptr
  double* store = nullptr;       // owned panels of a transient factorisation, packed (grow-only)
  size_t store_cap = 0;
  std::vector<double*> buf;      // ring of receive buffers (sgp_multi::ring of them), m_tot x widest panel each
  size_t buf_cap = 0;
  double* d_small = nullptr;     // Y | mean | noise diag | scalars
  size_t small_cap = 0;
  double* d_work = nullptr;      // grow-only scratch of the operators on top of a factor (predict, rand, elbo)
  size_t work_cap = 0;
  double* d_work2 = nullptr;
  size_t work2_cap = 0;
  int* d_info = nullptr;
  ncclComm_p comm = nullptr;
};

}  // namespace

struct sgp_multi {
  std::vector<Rank> r;
  int transport = TR_LOOPBACK;
  bool allgather = true;    // peer-copy transport: scatter + all-gather (false: one copy owner -> receiver)
  bool peer_ok = true;      // every distinct device pair has peer access enabled (else copies stage through the host)
  long W = 1024;            // panel width of the first part of the columns
  long W_tail = 0;          // ... of the last part (SGP_MULTI_PANEL_TAIL); 0 / >= W: one width (the default: measured, round 4)
  double tail_frac = 2.0 / 3.0;   // the first tail_frac of the columns get W
  int group = 1;            // panels per group (see the head of this file); 1 = every update at K = one panel
  long sub = 512;           // sub-panel width of the factorisation / transport / look-ahead pipeline (0: whole panels)
  // Round 6 -- the hybrid schedule in the sharded sweep (SGP_MULTI_PANEL_DF, default 1; 0 = the launch-based chain of rounds 2 - 5):
  // a sub-panel is factored by ONE launch of the dataflow kernel (chol_df.hip) that also updates the panel's remaining columns
  // with it, and (SGP_MULTI_FUSE_LA, default 1) the look-ahead update with the LAST sub-panel of the previous panel rides in the
  // first of those launches as an external source: the diagonal chain starts as soon as the first tile column has seen it.
  // Off together with the primary context's hybrid switch (SGP_HYBRID=0; the dataflow time-out fallback reruns that way).
  int panel_df = 1, fuse_la = 1;
  int compact = 1;          // SGP_MULTI_COMPACT: compacted live-tile ids in the far update launches of a structured model (2: at any size)
  sgp_ctx* primary = nullptr;
  // One enqueue thread per rank for the sweep of the sharded factorisation (SGP_MULTI_THREADS: 1 / 0; -1 = automatic: on
  // with more than one rank).  Every thread walks the SAME schedule and issues only the HIP calls of its own rank's
  // streams; an event recorded by one thread and waited for by another is ordered through a per-event sequence number
  // (Exec::rec / Exec::wait below): the waiter knows, from the schedule alone, which record it needs.
  int threads = -1;
  std::unordered_map<hipEvent_t, std::atomic<long>> seq;
  std::atomic<int> abort_flag{0};
  bool broken = false;      // an enqueue thread failed under the RCCL transport: the communicators are gone
  // Round 6 -- so that the first run on a real node can neither hang nor lie:
  //  * every cross-thread host spin (Exec::wait) is bounded by wall-clock time (SGP_MULTI_SPIN_TIMEOUT_S, default 30 s): past it
  //    the waiter raises the abort flag and the call fails with rc < 0 and a text naming the rank it waited for;
  //  * ncclCommInitAll runs under a bound of its own (SGP_MULTI_INIT_TIMEOUT_S, default 180 s);
  //  * a TEST-ONLY fault hook (SGP_MULTI_FAULT=rank:step, sgp_bench_multi_fault): the enqueue thread of `rank` (the one thread
  //    when the sweep is not threaded) fails at panel `step` of the next sharded factorisation, mid-schedule, after which the
  //    hook disarms itself.
  double spin_timeout_s = 30.0, init_timeout_s = 180.0;
  int fault_rank = -1;
  long fault_step = -1;
  double fault_stall_s = 0.0;   // > 0: the thread does not fail but SLEEPS that long at the step (sgp_bench_multi_stall): what
                                // the other threads' spin bound is for
  int ring = 10;            // receive buffers per rank: 2 * group + 2
  // panel ownership (own_table.h; make_geometry): balanced from the model's tile pattern unless SGP_MULTI_OWNERS says
  // "cyclic" or gives an explicit list; own_mode = what the last geometry used (0 cyclic, 1 balanced table, 2 list)
  bool own_balanced = true;
  std::vector<int> own_list;
  int own_mode = 0;
  std::vector<int> last_own;   // owner of every panel of the last geometry (sgp_ctx_multi_owners)
  std::atomic<long> late_binds{0};   // Exec::wait: records that had been re-recorded by the time the wait was enqueued
  Rccl rccl;
  double last_ms = 0.0;
  int sz_words = 0;               // structural zeros: words per pattern row of the current call (0: dense)
  double sz_frac = 1.0;           // executed / dense tile products of its contractions
  double last_enqueue_ms = 0.0;   // host time the one enqueue thread spent issuing the last sharded factorisation
  long last_npan = 0;
  // profile mode (sgp_ctx_multi_profile): the factorisation runs serialised, every group of launches timed alone
  int profile = 0;
  std::vector<double> prof;   // per panel J: factor_ms, lookahead_update_ms, panel bytes, rest_update_ms[rank 0..P)
  std::vector<double> prof_pieces;   // per panel J, Rank::NSUB slots: ms of each sub-panel's launches, alone on the hardware
                                     // (sgp_bench_multi_profile_pieces; tools/multi_projection.py prices uneven pieces with it)
};

namespace {

int hipfail(hipError_t e, const char* what) {
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return -2;
}
#define M_HIP(expr)                                  \
  do {                                               \
    hipError_t _e = (expr);                          \
    if (_e != hipSuccess) return hipfail(_e, #expr); \
  } while (0)

int grow(double** p, size_t* cap, size_t need) {
  if (need <= *cap) return 0;
  if (*p) hipFree(*p);
  *p = nullptr;
  *cap = 0;
  if (hipMalloc(p, sizeof(double) * need) != hipSuccess) {
    (void)hipGetLastError();
    set_error("multi: hipMalloc failed (" + std::to_string(need * 8) + " bytes)");
    return -2;
  }
  *cap = need;
  return 0;
}

inline long rup(long x, long m) { return (x + m - 1) / m * m; }

struct Geometry {
  long N = 0, n_pad = 0, m_tot = 0, W = 0, npan = 0, P = 1, S = 0;   // W: the widest panel
  std::vector<long> c0s;   // first column of every panel, and n_pad
  // who owns which panel (own_table.h): cyclic, or balanced from the symbolic tile pattern of a structured model.  Every
  // loop over "rank i's panels" walks mine[i] (ascending); local(J) = position of J among its owner's panels.
  std::vector<int> own;
  std::vector<long> loc;
  std::vector<std::vector<long>> mine;
  long col0(long J) const { return c0s[J]; }
  long width(long J) const { return c0s[J + 1] - c0s[J]; }
  long ldp(long J) const { return m_tot - c0s[J]; }          // packed leading dimension of panel J
  int owner(long J) const { return own[(size_t)J]; }
  long local(long J) const { return loc[(size_t)J]; }
  long ncols_owned(int i) const {
    long t = 0;
    for (long J : mine[(size_t)i]) t += width(J);
    return t;
  }
  void set_owners(const std::vector<int>& o) {
    own = o;
    loc.assign((size_t)npan, 0);
    mine.assign((size_t)P, {});
    for (long J = 0; J < npan; ++J) {
      loc[(size_t)J] = (long)mine[(size_t)own[(size_t)J]].size();
      mine[(size_t)own[(size_t)J]].push_back(J);
    }
  }
};

// spec / noise_kind: the model whose factor is about to be sharded (nullptr: unknown -- the cyclic deal)
Geometry make_geometry(sgp_multi* m, long N, long S, const sgp_cov_spec* spec = nullptr, int noise_kind = SGP_NOISE_SCALAR) {
  Geometry g;
  int64_t n_pad, m_tot;
  sgp_geometry(N, S, &n_pad, &m_tot);
  g.N = N;
  g.S = S;
  g.n_pad = n_pad;
  g.m_tot = m_tot;
  g.W = std::min<long>(m->W, n_pad);
  // wide panels up to the first panel boundary at or beyond tail_frac of the columns, narrow ones from there on
  const long wt = (m->W_tail >= TILE && m->W_tail < g.W) ? m->W_tail : g.W;
  const long switch_col = (long)(m->tail_frac * (double)n_pad);
  for (long c = 0; c < n_pad;) {
    g.c0s.push_back(c);
    c += std::min((c < switch_col ? g.W : wt), n_pad - c);
  }
  g.npan = (long)g.c0s.size();
  g.c0s.push_back(n_pad);
  g.P = (long)m->r.size();
  // ---- ownership (own_table.h).  SGP_MULTI_OWNERS = balanced (default) | cyclic | r0,r1,... (a list dealt out cyclically:
  // tests).  Balanced: panel costs from the symbolic tile pattern of the model (rank 0's context holds the switch for the
  // structural zeros), one panel per rank and round, heaviest panel to the least loaded rank; a dense model with one panel
  // width gets the cyclic deal itself (the balanced table is within 1 % of it there and the cyclic one is what every
  // earlier measurement ran on).
  std::vector<int> own((size_t)g.npan);
  for (long J = 0; J < g.npan; ++J) own[(size_t)J] = (int)(J % g.P);
  m->own_mode = 0;
  if (g.P > 1 && !m->own_list.empty()) {
    for (long J = 0; J < g.npan; ++J) own[(size_t)J] = m->own_list[(size_t)J % m->own_list.size()] % (int)g.P;
    m->own_mode = 2;
  } else if (g.P > 1 && m->own_balanced && spec) {
    std::vector<double> col_work;
    if (drv_sz_col_work(m->r[0].ctx, spec, noise_kind, n_pad, m_tot, col_work) == 0 && !col_work.empty()) {
      const std::vector<double> cost = panel_costs(g.c0s, TILE, m_tot / TILE, col_work);
      own = balanced_owners(cost, (int)g.P);
      m->own_mode = 1;
    }
  }
  g.set_owners(own);
  m->last_own = own;
  return g;
}

// one sharded factorisation: per rank the base of its packed panels, their offsets and (kept factors only) the
// inverse diagonal blocks of every owned panel
struct Fact {
  Geometry g;
  std::vector<double*> base;
  std::vector<std::vector<size_t>> off;
  std::vector<double*> inv;   // [P] or empty
  long inv_per_panel = 0;
  double* panel(int i, long J) const { return base[i] + off[i][g.local(J)]; }
  double* invp(int i, long J) const { return inv.empty() ? nullptr : inv[i] + g.local(J) * inv_per_panel; }
};

size_t plan_offsets(const Geometry& g, int i, std::vector<size_t>& off) {
  off.clear();
  size_t tot = 0;
  for (long J : g.mine[(size_t)i]) {
    off.push_back(tot);
    tot += (size_t)g.ldp(J) * g.width(J);
  }
  return tot;
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int sync_all(sgp_multi* m) {
  for (auto& k : m->r) {
    M_HIP(hipSetDevice(k.dev));
    M_HIP(hipDeviceSynchronize());
  }
  return 0;
}

double update_flops(long mrows, long nc, long k) {
  return (double)k * (double)nc * (double)(nc + 1) + 2.0 * (double)k * (double)(mrows - nc) * (double)nc;
}

}  // namespace

extern "C" int sgp_ctx_ndev(sgp_ctx* ctx) {
  if (!ctx) return 0;
  return ctx->multi ? (int)ctx->multi->r.size() : 1;
}

void sgp_multi_destroy(sgp_multi* m) {
  if (!m) return;
  for (auto& k : m->r) {
    hipSetDevice(k.dev);
    if (k.s_upd) hipStreamSynchronize(k.s_upd);
    if (k.s_panel) hipStreamSynchronize(k.s_panel);
    if (k.s_comm) hipStreamSynchronize(k.s_comm);
    if (k.comm && m->rccl.CommDestroy) m->rccl.CommDestroy(k.comm);
    if (k.store) hipFree(k.store);
    for (auto b : k.buf)
      if (b) hipFree(b);
    if (k.d_small) hipFree(k.d_small);
    if (k.d_work) hipFree(k.d_work);
    if (k.d_work2) hipFree(k.d_work2);
    if (k.d_info) hipFree(k.d_info);
    for (hipEvent_t e : {k.ev_upd, k.ev_fact, k.ev_done, k.ev_A, k.ev_B, k.ev_far[0], k.ev_far[1], k.ev_far[2], k.ev_far[3],
                         k.ev_far[4], k.ev_far[5], k.ev_far[6], k.ev_far[7], k.ev_t0, k.ev_t1})
      if (e) hipEventDestroy(e);
    for (auto e : k.ev_recv)
      if (e) hipEventDestroy(e);
    for (auto e : k.ev_sub)
      if (e) hipEventDestroy(e);
    for (auto& v : k.ev_recv_sub)
      for (auto e : v)
        if (e) hipEventDestroy(e);
    for (auto e : k.ev_free)
      if (e) hipEventDestroy(e);
    for (auto& v : k.ev_in)
      for (auto e : v)
        if (e) hipEventDestroy(e);
    if (k.s_near) {
      hipStreamSynchronize(k.s_near);
      hipStreamDestroy(k.s_near);
    }
    for (auto st : k.s_in)
      if (st) {
        hipStreamSynchronize(st);
        hipStreamDestroy(st);
      }
    if (k.s_comm) hipStreamDestroy(k.s_comm);
    if (k.ctx) sgp_ctx_destroy(k.ctx);   // owns s_upd / s_panel
  }
  delete m;
}

// Run-time switches of a multi-GPU context: one table, read once at creation (docs/03_kernels.md section 3.5 lists every entry
// with the test that exercises it).  Layout and schedule switches select between forms that give the same bits (for one panel
// layout); the time bounds and the fault hook belong to the failure path (tests/test_gpu_multi_faults.py).
namespace {
struct MultiKnob {
  const char* name;
  void (*set)(sgp_multi*, const char*);
};
const MultiKnob kMultiKnobs[] = {
    {"SGP_MULTI_TRANSPORT", [](sgp_multi*, const char*) {}},      // rccl | p2p | auto (read by sgp_ctx_create_multi itself)
    {"SGP_MULTI_ALLOW_STAGED", [](sgp_multi*, const char*) {}},   // accept peer copies staged through the host (ditto)
    {"SGP_MULTI_BCAST", [](sgp_multi* m, const char* v) { m->allgather = strcmp(v, "direct") != 0; }},
    {"SGP_MULTI_PANEL", [](sgp_multi* m, const char* v) { if (atol(v) >= TILE) m->W = atol(v) / TILE * TILE; }},
    {"SGP_MULTI_PANEL_TAIL", [](sgp_multi* m, const char* v) { m->W_tail = atol(v) / TILE * TILE; }},
    {"SGP_MULTI_TAIL_FRAC", [](sgp_multi* m, const char* v) { if (atof(v) >= 0.0 && atof(v) <= 1.0) m->tail_frac = atof(v); }},
    {"SGP_MULTI_GROUP", [](sgp_multi* m, const char* v) { if (atoi(v) >= 1 && atoi(v) <= SEG_MAX_SRC) m->group = atoi(v); }},
    {"SGP_MULTI_SUBPANEL", [](sgp_multi* m, const char* v) { m->sub = atol(v) / TILE * TILE; }},
    {"SGP_MULTI_OWNERS",                                            // balanced (default) | cyclic | r0,r1,... (dealt out cyclically)
     [](sgp_multi* m, const char* ow) {
       if (!strcmp(ow, "cyclic")) {
         m->own_balanced = false;
       } else if (strcmp(ow, "balanced") != 0) {
         for (const char* c = ow; *c;) {
           char* e = nullptr;
           const long v = strtol(c, &e, 10);
           if (e == c) break;
           m->own_list.push_back((int)std::max<long>(0, v));
           c = (*e == ',') ? e + 1 : e;
         }
       }
     }},
    {"SGP_MULTI_THREADS", [](sgp_multi* m, const char* v) { m->threads = atoi(v); }},       // 0: one enqueue thread, 1: one per rank
    {"SGP_MULTI_PANEL_DF", [](sgp_multi* m, const char* v) { m->panel_df = atoi(v); }},     // 0: the launch-based panel chain
    {"SGP_MULTI_FUSE_LA", [](sgp_multi* m, const char* v) { m->fuse_la = atoi(v); }},       // 0: the last look-ahead piece as a launch
    {"SGP_MULTI_COMPACT", [](sgp_multi* m, const char* v) { m->compact = atoi(v); }},       // 0 off, 2 at any size (tests)
    {"SGP_MULTI_SPIN_TIMEOUT_S", [](sgp_multi* m, const char* v) { m->spin_timeout_s = std::max(0.01, atof(v)); }},
    {"SGP_MULTI_INIT_TIMEOUT_S", [](sgp_multi* m, const char* v) { m->init_timeout_s = std::max(1.0, atof(v)); }},
    {"SGP_MULTI_FAULT",                                             // rank:step (tests)
     [](sgp_multi* m, const char* v) {
       const char* c = strchr(v, ':');
       m->fault_rank = atoi(v);
       m->fault_step = c ? atol(c + 1) : 0;
     }},
};
void apply_multi_knobs(sgp_multi* m) {
  for (const MultiKnob& k : kMultiKnobs)
    if (const char* v = getenv(k.name)) k.set(m, v);
}
}  // namespace

extern "C" int sgp_ctx_create_multi(const int* devices, int ndev, sgp_ctx** out) {
  M_CHECK_ARG(devices && out && ndev >= 1 && ndev <= 64, "sgp_ctx_create_multi: bad argument");
  sgp_ctx* primary = nullptr;
  M_RC(sgp_ctx_create(devices[0], &primary));
  sgp_multi* m = new sgp_multi();
  primary->multi = m;
  primary->multi_nranks = ndev;
  m->primary = primary;
  auto fail = [&](int rc) {
    sgp_ctx_destroy(primary);   // destroys m as well
    return rc;
  };
  bool distinct = true;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devices[i] == devices[j]) distinct = false;
  std::string want = "auto";
  bool allow_staged = false;
  for (const MultiKnob& k : kMultiKnobs)          // the two switches that steer the creation itself
    if (const char* v = getenv(k.name)) {
      if (!strcmp(k.name, "SGP_MULTI_TRANSPORT")) want = v;
      if (!strcmp(k.name, "SGP_MULTI_ALLOW_STAGED")) allow_staged = atoi(v) != 0;
    }
  if (!distinct) {
    m->transport = TR_LOOPBACK;   // several ranks on one GPU: same-device copies (test configuration)
  } else if (want == "p2p" || (ndev == 1 && want != "rccl")) {
    m->transport = TR_P2P;   // (one device: nothing to exchange unless RCCL is asked for explicitly)
  } else {
    if (m->rccl.load())
      m->transport = TR_RCCL;
    else if (want == "rccl") {
      set_error("sgp_ctx_create_multi: SGP_MULTI_TRANSPORT=rccl but librccl could not be loaded");
      return fail(-3);
    } else
      m->transport = TR_P2P;
  }
  apply_multi_knobs(m);
  m->ring = 2 * m->group + 2;
  m->r.resize(ndev);
  for (int i = 0; i < ndev; ++i) {
    Rank& k = m->r[i];
    k.dev = devices[i];
    int rc = sgp_ctx_create(devices[i], &k.ctx);
    if (rc) return fail(rc);
    k.s_panel = k.ctx->stream;    // high priority
    k.s_upd = k.ctx->stream2;
    if (hipSetDevice(k.dev) != hipSuccess) return fail(-2);
    if (hipStreamCreateWithFlags(&k.s_comm, hipStreamNonBlocking) != hipSuccess) return fail(-2);
    {
      // the near stream between the panel stream (highest) and the update stream (lowest)
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return fail(-2);
      if (hipStreamCreateWithPriority(&k.s_near, hipStreamNonBlocking, hi < lo ? hi + 1 : hi) != hipSuccess) return fail(-2);
    }
    for (hipEvent_t* e : {&k.ev_upd, &k.ev_fact, &k.ev_done, &k.ev_A, &k.ev_B, &k.ev_far[0], &k.ev_far[1], &k.ev_far[2],
                          &k.ev_far[3], &k.ev_far[4], &k.ev_far[5], &k.ev_far[6], &k.ev_far[7]})
      if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(-2);
    const int R = m->ring;
    for (auto& e : k.ev_sub)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(-2);
    k.ev_recv_sub.assign(R, std::vector<hipEvent_t>(Rank::NSUB, nullptr));
    for (int b = 0; b < R; ++b)
      for (auto& e : k.ev_recv_sub[b])
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(-2);
    k.buf.assign(R, nullptr);
    k.ev_recv.assign(R, nullptr);
    k.ev_free.assign(R, nullptr);
    k.ev_in.assign(R, std::vector<hipEvent_t>(ndev, nullptr));
    for (int b = 0; b < R; ++b) {
      if (hipEventCreateWithFlags(&k.ev_recv[b], hipEventDisableTiming) != hipSuccess) return fail(-2);
      if (hipEventCreateWithFlags(&k.ev_free[b], hipEventDisableTiming) != hipSuccess) return fail(-2);
    }
    k.s_in.assign(ndev, nullptr);
    for (int q = 0; q < ndev; ++q) {
      if (q == i) continue;
      if (hipStreamCreateWithFlags(&k.s_in[q], hipStreamNonBlocking) != hipSuccess) return fail(-2);
      for (int b = 0; b < R; ++b)
        if (hipEventCreateWithFlags(&k.ev_in[b][q], hipEventDisableTiming) != hipSuccess) return fail(-2);
    }
    if (hipEventCreate(&k.ev_t0) != hipSuccess || hipEventCreate(&k.ev_t1) != hipSuccess) return fail(-2);
    if (hipMalloc(&k.d_info, sizeof(int)) != hipSuccess) return fail(-2);
  }
  if (m->transport == TR_P2P || m->transport == TR_RCCL) {
    for (int i = 0; i < ndev; ++i)
      for (int j = 0; j < ndev; ++j) {
        if (i == j) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess) can = 0;
        if (can) {
          hipSetDevice(devices[i]);
          hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
          (void)hipGetLastError();
        }
        if (!can) m->peer_ok = false;
      }
    // a node without peer access between its GPUs would run every panel through host memory: say so
    // instead of silently being an order of magnitude slower (SGP_MULTI_ALLOW_STAGED=1 accepts it)
    if (!m->peer_ok && m->transport == TR_P2P) {
      if (!allow_staged) {
        set_error("sgp_ctx_create_multi: peer access between the listed GPUs is not available and RCCL could not be "
                  "loaded: panel copies would be staged through host memory (SGP_MULTI_ALLOW_STAGED=1 to accept)");
        return fail(-3);
      }
    }
  }
  if (m->transport == TR_RCCL) {
    // ncclCommInitAll on its own thread, bounded: a node whose fabric / driver state is bad has been seen to sit in the
    // bootstrap for ever.  On a time-out the thread is left behind (it owns its copies of everything it touches).
    struct InitJob {
      std::vector<ncclComm_p> comms;
      std::vector<int> devs;
      std::atomic<int> done{0};
      int rc = 0;
    };
    auto job = std::make_shared<InitJob>();
    job->comms.assign(ndev, nullptr);
    job->devs.assign(devices, devices + ndev);
    auto init_fn = m->rccl.CommInitAll;
    std::thread([job, init_fn, ndev]() {
      job->rc = init_fn(job->comms.data(), ndev, job->devs.data());
      job->done.store(1, std::memory_order_release);
    }).detach();
    const double t0 = now_ms();
    while (!job->done.load(std::memory_order_acquire) && now_ms() - t0 < 1e3 * m->init_timeout_s)
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    if (!job->done.load(std::memory_order_acquire)) {
      set_error("sgp_ctx_create_multi: ncclCommInitAll did not return within " + std::to_string((long)m->init_timeout_s) +
                " s (SGP_MULTI_INIT_TIMEOUT_S); SGP_MULTI_TRANSPORT=p2p runs the panel copies without RCCL");
      return fail(-4);
    }
    std::vector<ncclComm_p> comms = job->comms;
    int rc = job->rc;
    if (rc != 0) {
      set_error(std::string("ncclCommInitAll failed: ") +
                (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "?"));
      return fail(-4);
    }
    for (int i = 0; i < ndev; ++i) m->r[i].comm = comms[i];
  }
  // the sequence numbers of every event the per-rank enqueue threads may order themselves by (Exec)
  for (auto& k : m->r) {
    auto reg = [&](hipEvent_t e) {
      if (e) m->seq.emplace(std::piecewise_construct, std::forward_as_tuple(e), std::forward_as_tuple(0));
    };
    for (hipEvent_t e : {k.ev_upd, k.ev_fact, k.ev_done, k.ev_A, k.ev_B}) reg(e);
    for (auto e : k.ev_far) reg(e);
    for (auto e : k.ev_sub) reg(e);
    for (auto e : k.ev_recv) reg(e);
    for (auto e : k.ev_free) reg(e);
    for (auto& v : k.ev_recv_sub)
      for (auto e : v) reg(e);
    for (auto& v : k.ev_in)
      for (auto e : v) reg(e);
  }
  hipSetDevice(devices[0]);
  *out = primary;
  return 0;
}

extern "C" const char* sgp_ctx_transport(sgp_ctx* ctx) {
  if (!ctx || !ctx->multi) return "single";
  switch (ctx->multi->transport) {
    case TR_RCCL: return "rccl";
    case TR_P2P: return ctx->multi->peer_ok ? "p2p" : "p2p-staged";
    default: return "loopback";
  }
}

// out[0] = ranks, [1] = wall ms of the last sharded factorisation (enqueue to completion), [2] = transport
// (0 loopback, 1 peer copies, 2 RCCL), [3] = ranks the RCCL communicator reports (-1: none), [4] = panel width,
// [5] = panels, [6] = 1 if the peer-copy transport runs scatter + all-gather, [7] = panels per update group; then per rank 4
// doubles: algorithmic flops of its trailing updates, ms from its first update's start to its last update's end,
// panels factored, bytes received.
extern "C" int sgp_ctx_multi_stats(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out) {
  M_CHECK_ARG(ctx && out && n_out, "sgp_ctx_multi_stats: NULL argument");
  M_CHECK_ARG(ctx->multi, "sgp_ctx_multi_stats: not a multi-GPU context");
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  M_CHECK_ARG(cap >= 8 + 4 * P, "sgp_ctx_multi_stats: buffer too small (8 + 4 * ranks doubles)");
  out[0] = P;
  out[1] = m->last_ms;
  out[2] = m->transport;
  out[3] = -1;
  if (m->transport == TR_RCCL && m->rccl.CommCount && m->r[0].comm) {
    int c = -1;
    if (m->rccl.CommCount(m->r[0].comm, &c) == 0) out[3] = c;
  }
  out[4] = (double)m->W;
  out[5] = (double)m->last_npan;
  out[6] = m->allgather ? 1 : 0;
  out[7] = (double)m->group;
  for (int i = 0; i < P; ++i) {
    out[8 + 4 * i] = m->r[i].upd_flops * m->sz_frac;   // (the launches are priced dense; the skipped share comes off here)
    out[9 + 4 * i] = m->r[i].upd_span_ms;
    out[10 + 4 * i] = (double)m->r[i].n_factored;
    out[11 + 4 * i] = m->r[i].recv_bytes;
  }
  *n_out = 8 + 4 * P;
  if (cap >= 9 + 4 * P) {   // one more figure for callers that leave room: the host-side enqueue time of that call
    out[8 + 4 * P] = m->last_enqueue_ms;
    *n_out = 9 + 4 * P;
  }
  if (cap >= 11 + 4 * P) {   // ... and two more: the ownership of the last geometry (0 cyclic, 1 balanced table, 2 list) and
                             // the event waits that bound to a later record than the schedule named (Exec::wait), cumulative
    out[9 + 4 * P] = (double)m->own_mode;
    out[10 + 4 * P] = (double)m->late_binds.load();
    *n_out = 11 + 4 * P;
  }
  return 0;
}

// state behind the test hooks of sthenomi_bench.h (the extern "C" wrappers live in libsthenomi_bench.so: bench_hooks.hip).
// Fault hook: the next sharded factorisation fails at panel `step` on rank `rank`'s enqueue thread -- or, with stall_s > 0,
// sleeps there without failing (the other threads then run into their spin bound).
int sgp_multi_set_fault(sgp_multi* m, int rank, long step, double stall_s) {
  m->fault_rank = rank;
  m->fault_step = step;
  m->fault_stall_s = stall_s;
  return 0;
}
int sgp_multi_is_broken(sgp_multi* m) { return m->broken ? 1 : 0; }
const std::vector<double>& sgp_multi_profile_pieces(sgp_multi* m) { return m->prof_pieces; }

extern "C" int sgp_ctx_multi_owners(sgp_ctx* ctx, int32_t* out, int64_t cap, int64_t* n_out) {
  M_CHECK_ARG(ctx && ctx->multi && n_out, "sgp_ctx_multi_owners: not a multi-GPU context");
  const auto& v = ctx->multi->last_own;
  *n_out = (int64_t)v.size();
  if (out) {
    M_CHECK_ARG(cap >= (int64_t)v.size(), "sgp_ctx_multi_owners: buffer too small");
    for (size_t J = 0; J < v.size(); ++J) out[J] = v[J];
  }
  return 0;
}

// Profile mode: the next sharded factorisations run SERIALISED -- every group of launches (the look-ahead update
// and factorisation of a panel on its owner; each rank's remaining trailing updates) alone on the hardware and
// timed with the host clock around a device synchronisation.  With several ranks on one GPU (loopback) these are
// the times each GPU of a real node would see for its own share; tools/multi_projection.py turns them into a
// critical-path projection.  sgp_ctx_multi_profile_get: per panel J  {factor_ms, lookahead_update_ms, panel bytes, then per
// rank the three update classes of step J: near_a_ms, near_b_ms, far_ms}  (3 + 3 P doubles).
extern "C" int sgp_ctx_multi_profile(sgp_ctx* ctx, int enable) {
  M_CHECK_ARG(ctx && ctx->multi, "sgp_ctx_multi_profile: not a multi-GPU context");
  ctx->multi->profile = enable ? 1 : 0;
  ctx->multi->prof.clear();
  return 0;
}
extern "C" int sgp_ctx_multi_profile_get(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out) {
  M_CHECK_ARG(ctx && ctx->multi && n_out, "sgp_ctx_multi_profile_get: bad argument");
  const auto& v = ctx->multi->prof;
  *n_out = (int64_t)v.size();
  if (out) {
    M_CHECK_ARG(cap >= (int64_t)v.size(), "sgp_ctx_multi_profile_get: buffer too small");
    std::copy(v.begin(), v.end(), out);
  }
  return 0;
}

namespace {

// ---- panel transport ----------------------------------------------------------------------------------
// Who executes the schedule.  me < 0: the caller's thread issues everything (profile mode, one rank, SGP_MULTI_THREADS=0).
// me = i: one of P threads that all walk the same schedule; this one issues the calls that go to rank i's streams.
// rec / wait count the records of every event in schedule order (identically in every thread), so a wait knows which
// record it refers to: the owner of the recording stream publishes the count after hipEventRecord, the waiting thread
// spins (host side, microseconds) until that record has been issued and only then enqueues hipStreamWaitEvent.  A wait
// only ever refers to a record EARLIER in the schedule and records never block, so the threads cannot deadlock.
struct Exec {
  sgp_multi* m = nullptr;
  int me = -1;
  std::unordered_map<hipEvent_t, long> cnt;
  std::vector<char> factored_once;   // per rank, as the schedule sees it (every thread keeps its own copy)
  bool mine(int i) const { return me < 0 || me == i; }
  bool threaded() const { return me >= 0; }
  int dev(int i) const {
    if (hipSetDevice(m->r[i].dev) != hipSuccess) return hipfail(hipGetLastError(), "hipSetDevice");
    return 0;
  }
  int rec(int i, hipEvent_t e, hipStream_t s) {
    const long c = ++cnt[e];
    if (!mine(i)) return 0;
    M_RC(dev(i));
    M_HIP(hipEventRecord(e, s));
    if (threaded()) m->seq.at(e).store(c, std::memory_order_release);
    return 0;
  }
  int wait(int i, hipStream_t s, hipEvent_t e) {
    if (!mine(i)) return 0;
    if (threaded()) {
      auto it = cnt.find(e);
      const long c = it == cnt.end() ? 0 : it->second;
      if (c == 0) return 0;   // never recorded in this call: whatever it guarded was drained with the previous call
      std::atomic<long>& q = m->seq.at(e);
      const double t0 = now_ms();
      for (unsigned spins = 0; q.load(std::memory_order_acquire) < c; ++spins) {
        if (m->abort_flag.load(std::memory_order_relaxed)) {
          set_error("multi: another rank's enqueue thread failed");
          return -5;
        }
        if ((spins & 1023) == 1023 && now_ms() - t0 > 1e3 * m->spin_timeout_s) {
          // the producer never issued the record this wait refers to (a stuck HIP call on its thread, a diverged schedule):
          // stop everybody instead of spinning for ever
          m->abort_flag.store(1);
          set_error("multi: rank " + std::to_string(i) + "'s enqueue thread waited more than " +
                    std::to_string(m->spin_timeout_s) + " s for an event record of another rank (SGP_MULTI_SPIN_TIMEOUT_S)");
          return -7;   // (a root cause, unlike -5: the threads that leave because of the flag)
        }
        std::this_thread::yield();
      }
      // hipStreamWaitEvent binds to the LATEST record: if the producer has already re-recorded the event this waits for a
      // later point of the same stream (still correct -- a stream is in order -- and, by the schedule's construction, never a
      // cycle); counted so that a run can tell whether it ever happened (sgp_ctx_multi_stats)
      if (q.load(std::memory_order_acquire) > c) m->late_binds.fetch_add(1, std::memory_order_relaxed);
    }
    M_RC(dev(i));
    M_HIP(hipStreamWaitEvent(s, e, 0));
    return 0;
  }
};

// `st` of rank i waits until ring slot J % ring may be overwritten with panel J: every launch that read the slot's previous
// panel (J - ring) is done -- its near updates (s_near: the latest ev_A is at least that late), the far update with its
// group (s_upd: ev_far of that group, which also orders everything s_upd did before it, the near-B update included), the
// look-ahead that used it (s_panel: ev_fact)
int wait_slot_free(Exec& x, int i, long J, hipStream_t st) {
  sgp_multi* m = x.m;
  Rank& k = m->r[i];
  M_RC(x.wait(i, st, k.ev_A));
  M_RC(x.wait(i, st, k.ev_upd));   // (the assembly, before the first panel)
  if (J >= m->ring) M_RC(x.wait(i, st, k.ev_far[((J - m->ring) / m->group) % Rank::NGEV]));
  if (x.factored_once[i]) M_RC(x.wait(i, st, k.ev_fact));
  return 0;
}

// Move sub-panel q -- columns [c, c + wq) -- of factored panel J from its owner to every other rank's receive buffer (ring
// slot J % ring), as soon as the owner's panel stream has recorded ev_sub[q].  A panel travels sub-panel by sub-panel while
// the owner is still factoring its later columns; `last`: the whole panel has then landed (ev_recv).
int broadcast_panel(Exec& x, const Fact& F, long J, int q, long c, long wq, bool last) {
  sgp_multi* m = x.m;
  const Geometry& g = F.g;
  const int o = g.owner(J);
  const size_t off = (size_t)c * g.ldp(J);
  const size_t count = (size_t)g.ldp(J) * wq;
  const int P = (int)m->r.size();
  const int b = (int)(J % m->ring);
  if (P == 1 && m->transport != TR_RCCL) return 0;
  Rank& root = m->r[o];
  double* src = F.panel(o, J) + off;
  hipEvent_t ev_final = root.ev_sub[q];
  for (int i = 0; i < P; ++i)
    if (i != o && x.mine(i)) m->r[i].recv_bytes += 8.0 * (double)count;
  auto landed = [&](int i) -> int {   // on rank i's s_comm, after its copies of this sub-panel
    Rank& k = m->r[i];
    M_RC(x.rec(i, k.ev_recv_sub[b][q], k.s_comm));
    if (last) M_RC(x.rec(i, k.ev_recv[b], k.s_comm));
    return 0;
  };
  if (m->transport == TR_RCCL) {
    for (int i = 0; i < P; ++i) {   // order the communicator streams behind the data / the buffer's readers
      Rank& k = m->r[i];
      if (i == o) {
        M_RC(x.wait(i, k.s_comm, ev_final));
      } else if (q == 0) {
        M_RC(wait_slot_free(x, i, J, k.s_comm));
      }
    }
    // one grouped call from the one enqueue thread, or every rank's own call from its own thread
    int rc = x.threaded() ? 0 : m->rccl.GroupStart();
    for (int i = 0; i < P && rc == 0; ++i) {
      if (!x.mine(i)) continue;
      Rank& k = m->r[i];
      hipSetDevice(k.dev);
      void* buf = (i == o) ? (void*)src : (void*)(k.buf[b] + off);
      rc = m->rccl.Broadcast(buf, buf, count, NCCL_DOUBLE, o, k.comm, k.s_comm);
    }
    int rc2 = x.threaded() ? 0 : m->rccl.GroupEnd();
    if (rc || rc2) {
      set_error(std::string("ncclBroadcast failed: ") +
                (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc ? rc : rc2) : "?"));
      return -4;
    }
    for (int i = 0; i < P; ++i) M_RC(landed(i));
    return 0;
  }
  auto copy = [&](int di, double* d, int fi, const double* sp, size_t n, hipStream_t st) -> int {
    if (n == 0 || !x.mine(di)) return 0;
    Rank &dst = m->r[di], &from = m->r[fi];
    M_RC(x.dev(di));
    if (dst.dev == from.dev)
      M_HIP(hipMemcpyAsync(d, sp, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    else
      M_HIP(hipMemcpyPeerAsync(d, dst.dev, sp, from.dev, sizeof(double) * n, st));
    return 0;
  };
  if (!m->allgather || P <= 2) {
    // one copy owner -> receiver (one xGMI link per receiver)
    for (int i = 0; i < P; ++i) {
      if (i == o) continue;
      Rank& k = m->r[i];
      if (q == 0) M_RC(wait_slot_free(x, i, J, k.s_comm));      // readers of the panel this slot held a ring ago
      M_RC(x.wait(i, k.s_comm, ev_final));                       // the sub-panel is final
      M_RC(copy(i, k.buf[b] + off, o, src, count, k.s_comm));
      M_RC(landed(i));
    }
    return 0;
  }
  // scatter + all-gather: peer q_c (c = 0 .. P - 2, the ranks != o in order) gets slab c from the owner, then every
  // other peer copies slab c from q_c's receive buffer -- P - 1 concurrent incoming copies per receiver, one per link
  std::vector<int> peers;
  for (int i = 0; i < P; ++i)
    if (i != o) peers.push_back(i);
  const int np = (int)peers.size();
  const size_t align = 512;   // 4 KB slab boundaries
  const size_t per = ((count + np - 1) / np + align - 1) / align * align;
  auto slab = [&](int cc, size_t& beg, size_t& n) {
    beg = std::min(count, (size_t)cc * per);
    n = std::min(count - beg, per);
  };
  // (1) the receive buffer of every peer is free: the readers of the panel it held a ring ago are done -- this rank's own
  //     updates and, since slabs are forwarded out of it, the incoming copies the other peers made FROM it.  Once per
  //     panel: the later sub-panels follow the first on the same in-order streams.
  if (q == 0)
    for (int cc = 0; cc < np; ++cc) {
      const int pi = peers[cc];
      Rank& k = m->r[pi];
      M_RC(wait_slot_free(x, pi, J, k.s_comm));
      if (J >= m->ring)
        for (int d = 0; d < P; ++d) {
          if (d == pi) continue;
          M_RC(x.wait(pi, k.s_comm, m->r[d].ev_in[b][pi]));   // (never recorded == complete)
        }
      M_RC(x.rec(pi, k.ev_free[b], k.s_comm));
    }
  // (2) scatter: owner -> q_c, slab c
  for (int cc = 0; cc < np; ++cc) {
    const int pi = peers[cc];
    Rank& k = m->r[pi];
    size_t beg, n;
    slab(cc, beg, n);
    hipStream_t st = k.s_in[o];
    M_RC(x.wait(pi, st, k.ev_free[b]));
    M_RC(x.wait(pi, st, ev_final));
    M_RC(copy(pi, k.buf[b] + off + beg, o, src + beg, n, st));
    M_RC(x.rec(pi, k.ev_in[b][o], st));
  }
  // (3) all-gather: q_d <- q_c, slab c
  for (int d = 0; d < np; ++d) {
    const int pd = peers[d];
    Rank& k = m->r[pd];
    for (int cc = 0; cc < np; ++cc) {
      if (cc == d) continue;
      const int pc = peers[cc];
      Rank& from = m->r[pc];
      size_t beg, n;
      slab(cc, beg, n);
      hipStream_t st = k.s_in[pc];
      M_RC(x.wait(pd, st, k.ev_free[b]));
      M_RC(x.wait(pd, st, from.ev_in[b][o]));
      M_RC(copy(pd, k.buf[b] + off + beg, pc, from.buf[b] + off + beg, n, st));
      M_RC(x.rec(pd, k.ev_in[b][pc], st));
    }
    // (4) join: the whole sub-panel has landed
    for (int qq = 0; qq < P; ++qq)
      if (qq != pd) M_RC(x.wait(pd, k.s_comm, k.ev_in[b][qq]));
    M_RC(landed(pd));
  }
  return 0;
}

// panel J as seen by rank i (pointer to its row J0, leading dimension ldp(J))
const double* panel_on(sgp_multi* m, const Fact& F, long J, int i) {
  return (F.g.owner(J) == i) ? F.panel(i, J) : m->r[i].buf[J % m->ring];
}

int wait_panel(Exec& x, const Fact& F, long J, int i, hipStream_t s) {
  Rank& k = x.m->r[i];
  if (F.g.owner(J) == i) return x.wait(i, s, k.ev_fact);
  return x.wait(i, s, k.ev_recv[J % x.m->ring]);
}

// One launch: every panel of `dsts` (owned by rank i) -= its rows of the factored panels J_first .. J_last times their
// rows of the panel's diagonal block, in that order (gemm_nt.hip: gemm_nt_seg_kernel).  Returns the flops through *fl.
// sub_c >= 0: ONE source, the sub_w columns from sub_c on of panel J_first (== J_last).
// compact: a big structured launch on the rank's UPDATE stream may compact its live tile ids first (the child context's map
// scratch; launches of one stream are ordered, so one map is enough -- the near / look-ahead launches are small and keep theirs)
int update_panels(sgp_multi* m, const Fact& F, long J_first, long J_last, const std::vector<long>& dsts, int i,
                  hipStream_t s, double* fl, long sub_c = -1, long sub_w = 0, bool compact = false) {
  const Geometry& g = F.g;
  if (dsts.empty() || J_last < J_first) return 0;
  if (J_last - J_first + 1 > SEG_MAX_SRC) {
    set_error("multi: more source panels than one update launch takes");
    return -1;
  }
  SegBatch b;
  b.m_tot = g.m_tot;
  b.nz = m->sz_words > 0 ? m->r[i].d_sz : nullptr;
  b.nz_words = m->sz_words;
  if (compact && b.nz && m->compact && m->r[i].ctx->d_szmap) {
    b.map_scratch = m->r[i].ctx->d_szmap;
    b.map_ints = m->r[i].ctx->n_szmap;
    if (m->compact >= 2) b.map_min_ids = 8;   // (tests: every structured far launch)
  }
  long ksum = 0;
  for (int q = 0; q < SEG_MAX_SRC; ++q) b.src[q] = SegSrc{nullptr, 0, 0, 0};
  for (long J = J_first; J <= J_last; ++J) {
    b.src[J - J_first] = SegSrc{panel_on(m, F, J, i), g.ldp(J), g.col0(J), (int)g.width(J)};
    if (sub_c >= 0) {   // columns [sub_c, sub_c + sub_w) of the packed panel: same first stored row, same leading dimension
      b.src[0].base += (size_t)sub_c * g.ldp(J);
      b.src[0].w = (int)sub_w;
      b.src[0].k0 = g.col0(J) + sub_c;
    }
    ksum += b.src[J - J_first].w;
  }
  for (size_t d0 = 0; d0 < dsts.size(); d0 += SEG_MAX_DST) {
    b.n_dst = 0;
    for (size_t d = d0; d < std::min(dsts.size(), d0 + (size_t)SEG_MAX_DST); ++d) {
      const long Jp = dsts[d];
      b.dst[b.n_dst++] = SegDst{F.panel(i, Jp), g.ldp(Jp), g.col0(Jp), (int)g.width(Jp), 0, (int)(J_last - J_first + 1), 0u};
      if (fl) *fl += update_flops(g.m_tot - g.col0(Jp), g.width(Jp), ksum);
    }
    M_RC(launch_gemm_nt_seg(b, s));
  }
  return 0;
}

// ---- host-side inputs of one call, uploaded to every rank ----------------------------------------------
struct SmallLayout {
  long N, S, npan;
  size_t y, mean, noise, scal, pan, red, total;
  // pan: per panel J 1 + max(S, 1) doubles -- the panel's logdet contribution and the |L^-1 (Y - m)|^2 contributions of its
  // columns, written by the panel's owner only.  The host adds them in PANEL order (reduce_scalars), so logpdf does not
  // depend on who owns which panel (round 4 added per-rank sums in rank order: the value moved in the last bits with the
  // ownership table).
  long per() const { return 1 + std::max<long>(S, 1); }
  SmallLayout(long N_, long S_, long npan_) : N(N_), S(S_), npan(npan_) {
    y = 0;
    mean = y + (size_t)N * std::max<long>(S, 1);
    noise = mean + N;
    scal = noise + N;
    pan = scal + 16 + 2 * (size_t)std::max<long>(S, 1);
    red = pan + (size_t)npan * (size_t)per();     // receive buffer of the RCCL all-reduce of the slots (advisor, round 5: an
                                                  // in-place reduction would return P times the values on a second call)
    total = red + (size_t)npan * (size_t)per();
  }
};

// Assemble every rank's owned panels of K + Sigma_y (+ the bordered rows (Y - m)') and run the sharded
// factorisation.  d_small of rank i afterwards holds at `pan`, per OWNED panel: [0] logdet contribution, [1 .. 1 + S) the
// |L^-1 (Y - m)|^2 contributions of its columns (reduce_scalars).
int factorize(sgp_multi* m, Fact& F, const sgp_cov_spec* spec, const double* mean, int noise_kind, const double* noise,
              const double* Y, long ldy, std::vector<sgp_dspec*>& ds) {
  const Geometry& g = F.g;
  const int P = (int)g.P;
  const long N = g.N, S = g.S;
  if (m->broken) {
    set_error("multi: this context's RCCL communicators were aborted after a failed call: create a new context");
    return -4;
  }
  const double s2 = noise_kind == SGP_NOISE_SCALAR ? noise[0] : 0.0;
  const bool dense_noise = noise_kind == SGP_NOISE_DENSE;   // noise: host N x N, column-major, ld = N
  const int asm_kind = dense_noise ? SGP_NOISE_SCALAR : noise_kind;   // (dense: assembled with s2 = 0, the slabs added below)
  const SmallLayout L(N, S, g.npan);
  const bool prof = m->profile != 0;
  if (prof) m->prof.assign((size_t)g.npan * (3 + 3 * P), 0.0);
  if (prof) m->prof_pieces.assign((size_t)g.npan * Rank::NSUB, 0.0);
  const double t_begin = now_ms();
  Exec x0;                       // the caller's thread: buffers, uploads, assembly -- and the sweep when it is not threaded
  x0.m = m;
  x0.factored_once.assign(P, 0);
  for (int i = 0; i < P; ++i) {
    Rank& k = m->r[i];
    M_HIP(hipSetDevice(k.dev));
    if (P > 1 || m->transport == TR_RCCL) {
      size_t bc = (size_t)g.m_tot * g.W;
      if (bc > k.buf_cap || k.buf.empty() || !k.buf[0]) {
        M_HIP(hipDeviceSynchronize());
        for (auto& b : k.buf) {
          if (b) hipFree(b);
          b = nullptr;
        }
        k.buf_cap = 0;
        for (auto& b : k.buf)
          if (hipMalloc(&b, sizeof(double) * bc) != hipSuccess) {
            (void)hipGetLastError();
            set_error("multi: hipMalloc failed (panel receive buffer)");
            return -2;
          }
        k.buf_cap = bc;
      }
    }
    M_RC(grow(&k.d_small, &k.small_cap, L.total));
    k.t0_set = false;
    k.upd_flops = k.upd_span_ms = k.recv_bytes = 0.0;
    k.n_factored = 0;
    M_RC(drv_dspec_create(k.ctx, spec, &ds[i]));
    // structural zeros (common.h): rank 0's context derives the tile pattern of the factor, every rank keeps a device copy
    // (uploaded ahead of the assembly on the stream every update waits for)
    if (i == 0) {
      m->sz_words = 0;
      M_RC(drv_sz_pattern(k.ctx, ds[0], noise_kind, g.n_pad, g.m_tot, &m->sz_words));
      double ex = 0, de = 0;
      sgp_ctx_factor_work(k.ctx, &ex, &de);
      m->sz_frac = (m->sz_words > 0 && de > 0) ? ex / de : 1.0;
    }
    k.d_sz = nullptr;
    if (m->sz_words > 0) M_RC(drv_sz_upload(k.ctx, m->r[0].ctx, m->sz_words, k.s_upd, &k.d_sz));
    double* dY = k.d_small + L.y;
    double* dM = k.d_small + L.mean;
    double* dNz = k.d_small + L.noise;
    double* dSc = k.d_small + L.scal;
    if (S > 0) {
      if (ldy == N)
        M_HIP(hipMemcpyAsync(dY, Y, sizeof(double) * N * S, hipMemcpyHostToDevice, k.s_upd));
      else
        M_HIP(hipMemcpy2DAsync(dY, sizeof(double) * N, Y, sizeof(double) * ldy, sizeof(double) * N, (size_t)S,
                               hipMemcpyHostToDevice, k.s_upd));
    }
    if (mean) M_HIP(hipMemcpyAsync(dM, mean, sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
    if (noise_kind == SGP_NOISE_DIAG)
      M_HIP(hipMemcpyAsync(dNz, noise, sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
    M_HIP(hipMemsetAsync(dSc, 0, sizeof(double) * (L.total - L.scal), k.s_upd));   // scalars + the per-panel slots
    M_HIP(hipMemsetAsync(k.d_info, 0, sizeof(int), k.s_upd));
    // ---- assembly of the owned panels: no communication.  A dense Sigma_y (round 4) is added slab by slab: the owner of
    // panel J stages rows J0 .. N of ITS columns of the host matrix (one strided copy) and adds them on the tiles the
    // factorisation reads -- every rank touches 1 / P of the matrix, nothing travels between ranks.
    if (dense_noise) M_RC(grow(&k.d_work, &k.work_cap, (size_t)N * std::min<long>(g.W, N)));
    for (long J : g.mine[(size_t)i]) {
      double* base = F.panel(i, J);
      M_RC(sgp_dev_assemble_cols(k.ctx, ds[i], N, g.col0(J), g.width(J), base - g.col0(J), g.ldp(J), g.m_tot,
                                 mean ? dM : nullptr, asm_kind, &s2, noise_kind == SGP_NOISE_DIAG ? dNz : nullptr,
                                 S > 0 ? dY : nullptr, N, S, (void*)k.s_upd));
      const long c0 = g.col0(J), wv = std::min(g.width(J), N - c0);
      if (dense_noise && wv > 0) {
        const long rows = N - c0;
        M_HIP(hipSetDevice(k.dev));
        M_HIP(hipMemcpy2DAsync(k.d_work, sizeof(double) * rows, noise + c0 + c0 * N, sizeof(double) * N, sizeof(double) * rows,
                               (size_t)wv, hipMemcpyHostToDevice, k.s_upd));
        M_RC(launch_add_dense_cols(base, g.ldp(J), k.d_work, rows, c0, wv, N, k.s_upd));
      }
    }
    M_RC(x0.rec(i, k.ev_upd, k.s_upd));
  }
  // Factor panel J on its owner's panel stream in sub-panels of `sub` columns, each sent on its way as soon as it is final
  // (broadcast_panel) while the later columns are still being factored: sub-panel q = its own factorisation (128-column
  // steps inside: drv_panel_factor) + ONE update of the panel's remaining columns with it (K = sub; the same ascending-k
  // accumulation per tile as the 128-column steps of an unsplit panel: bit-identical).
  const long SUB = (m->sub >= TILE) ? m->sub : (1L << 40);
  // (Round 6 also measured UNEVEN pieces -- a long first piece, a short last one, so that less transport and look-ahead work sits
  // between two factorisations: profiles/r06_experiments/sharded_chain.md.  Every split lost to equal halves: the receiver
  // side is bound by transport + look-ahead update of ALL pieces, which only equal pieces pipeline.  Removed.)
  auto n_sub = [&](long J) { return (int)std::min<long>(Rank::NSUB, (g.width(J) + SUB - 1) / SUB); };
  auto sub_range = [&](long J, int q, long& c, long& wq) {
    const int ns = n_sub(J);
    c = (long)q * SUB;
    wq = (q == ns - 1) ? g.width(J) - c : SUB;   // (more than NSUB sub-panels: the last takes the rest)
  };
  const long G = m->group;
  auto group_of = [&](long J) { return J / G; };
  // the panel kernel (see sgp_multi::panel_df): with the primary context's hybrid switch, which the time-out fallback clears
  const bool df_panels = m->panel_df != 0 && m->primary && m->primary->hybrid != 0;
  const bool fuse_la = df_panels && m->fuse_la != 0;
  // Ranks that SHARE a GPU (a device listed several times: the test configuration) run the panel kernel's two-workgroups-per-CU
  // instantiation: the one-per-CU form waits for whole CUs to drain from the other ranks' update launches (8 ranks on one GPU:
  // c5 1869 vs 1667 ms, north-star model 1167 vs 1021; profiles/r06_experiments/sharded_chain.md).  On distinct devices -- and in
  // profile mode, whose launches run alone as they would on a node -- the one-per-CU form is faster (projection 142 vs 154 ms).
  const int lean_panels = (m->transport == TR_LOOPBACK && P > 1 && !prof) ? 1 : 0;
  const int fault_rank = std::min<int>(m->fault_rank, P - 1);   // the hook fires once: disarmed before the sweep starts
  const long fault_step = m->fault_rank >= 0 ? std::min<long>(m->fault_step, g.npan - 1) : -1;
  const double fault_stall_s = m->fault_stall_s;
  m->fault_rank = -1;
  m->fault_step = -1;
  m->fault_stall_s = 0.0;
  // ---- the schedule from the first panel's factorisation to the last row sums, as ONE function of who executes it
  auto run = [&](Exec& x) -> int {
    auto factor = [&](long J) -> int {
      const int o = g.owner(J);
      Rank& k = m->r[o];
      double* Pj = F.panel(o, J);
      const long ldp = g.ldp(J), w = g.width(J), J0 = g.col0(J);
      const int ns = n_sub(J);
      for (int q = 0; q < ns; ++q) {
        long c, wq;
        sub_range(J, q, c, wq);
        double* d_logdet = k.d_small + L.pan + (size_t)J * L.per();
        double* invq = F.invp(o, J) ? F.invp(o, J) + (c / TILE) * drv_invd_stride() : nullptr;
        double tq0 = 0;
        if (prof) {
          M_RC(sync_all(m));
          tq0 = now_ms();
        }
        auto piece_done = [&]() -> int {   // profile mode: this piece's launches, alone on the hardware
          if (!prof) return 0;
          M_RC(sync_all(m));
          m->prof_pieces[(size_t)J * Rank::NSUB + q] = now_ms() - tq0;
          return 0;
        };
        const sz_word* nzp = m->sz_words > 0 ? k.d_sz : nullptr;   // structural zeros inside the panel's own factorisation
        if (df_panels) {
          // ONE launch: [the look-ahead update with the previous panel's last sub-panel -- an external source, q == 0 only]
          // + factorisation of sub-panel q + update of the panel's remaining columns with it (update-only tile columns)
          if (x.mine(o)) {
            M_RC(x.dev(o));   // (broadcast_panel leaves another rank's device current in the one-thread mode)
            DfPanel px;
            px.n_fact = wq;
            px.n_ext = 0;
            if (q == 0 && fuse_la && J > 0) {
              long cl, wl;
              sub_range(J - 1, n_sub(J - 1) - 1, cl, wl);
              const long ldq = g.ldp(J - 1), Jm = g.col0(J - 1);
              px.ext[0] = DfExt{panel_on(m, F, J - 1, o) + (J0 - Jm) + (size_t)cl * ldq, ldq, (int)((Jm + cl) / TILE), (int)(wl / TILE)};
              px.n_ext = 1;
              k.upd_flops += update_flops(g.m_tot - J0, w, wl);
            }
            M_RC(drv_panel_factor(k.ctx, Pj + c + c * ldp, ldp, ldp - c, w - c, J0 + c, d_logdet, k.d_info, invq, k.s_panel, 1, nzp,
                                  m->sz_words, &px, lean_panels));
          }
          M_RC(x.rec(o, k.ev_sub[q], k.s_panel));
          if (q == ns - 1) M_RC(x.rec(o, k.ev_fact, k.s_panel));
          if (!prof) M_RC(broadcast_panel(x, F, J, q, c, wq, q == ns - 1));
          M_RC(piece_done());
          continue;
        }
        if (x.mine(o)) {
          M_RC(x.dev(o));
          // (structural zeros, round 5: for the columns of an independent block a third of the rows below are exact zeros --
          // the inner K = 128 updates and the panel solves skip those tiles)
          M_RC(drv_panel_factor(k.ctx, Pj + c + c * ldp, ldp, ldp - c, wq, J0 + c, d_logdet, k.d_info, invq, k.s_panel, 0, nzp,
                                m->sz_words, nullptr));
        }
        M_RC(x.rec(o, k.ev_sub[q], k.s_panel));
        if (q == ns - 1) M_RC(x.rec(o, k.ev_fact, k.s_panel));
        if (!prof) M_RC(broadcast_panel(x, F, J, q, c, wq, q == ns - 1));
        if (c + wq < w && x.mine(o)) {   // the rest of the panel -= (its rows of sub-panel q) (sub-panel q's rows of the rest)'
          M_RC(x.dev(o));
          SegBatch b;
          b.m_tot = g.m_tot;
          for (int t = 0; t < SEG_MAX_SRC; ++t) b.src[t] = SegSrc{nullptr, 0, 0, 0};
          b.src[0] = SegSrc{Pj + (size_t)c * ldp, ldp, J0, (int)wq};
          b.src[0].k0 = J0 + c;
          b.nz = m->sz_words > 0 ? k.d_sz : nullptr;
          b.nz_words = m->sz_words;
          const long r = c + wq;
          b.n_dst = 1;
          b.dst[0] = SegDst{Pj + r + (size_t)r * ldp, ldp, J0 + r, (int)(w - r), 0, 1, 0u};
          M_RC(launch_gemm_nt_seg(b, k.s_panel));
        }
        M_RC(piece_done());
      }
      x.factored_once[o] = 1;
      if (x.mine(o)) k.n_factored += 1;
      return 0;
    };
    // (profile mode times the factorisation alone: the sub-panels are sent afterwards)
    auto broadcast_all = [&](long J) -> int {
      const int ns = n_sub(J);
      for (int q = 0; q < ns; ++q) {
        long c, wq;
        sub_range(J, q, c, wq);
        M_RC(broadcast_panel(x, F, J, q, c, wq, q == ns - 1));
      }
      return 0;
    };
    // ---- panel 0
    {
      const int o0 = g.owner(0);
      Rank& k = m->r[o0];
      M_RC(x.wait(o0, k.s_panel, k.ev_upd));
      double t0 = 0;
      if (prof) {
        M_RC(sync_all(m));
        t0 = now_ms();
      }
      M_RC(factor(0));
      if (prof) {
        M_RC(sync_all(m));
        m->prof[0] = now_ms() - t0;
        m->prof[2] = 8.0 * (double)g.ldp(0) * (double)g.width(0);
        M_RC(broadcast_all(0));
      }
    }
    for (int i = 0; i < P; ++i) {   // the near stream starts behind the assembly
      Rank& k = m->r[i];
      M_RC(x.wait(i, k.s_near, k.ev_upd));
      M_RC(x.rec(i, k.ev_A, k.s_near));
      M_RC(x.rec(i, k.ev_B, k.s_upd));
    }
    // ---- right-looking sweep (the schedule at the head of this file)
    std::vector<long> near_a, near_b, far, la(1);
    for (long J = 0; J < g.npan; ++J) {
      const long nxt = J + 1, gj = group_of(J);
      const bool group_ends = (J % G == G - 1) || J == g.npan - 1;
      if (J == fault_step && (x.me == fault_rank || x.me < 0)) {   // test-only (sgp_multi::fault_rank)
        if (fault_stall_s > 0.0) {
          std::this_thread::sleep_for(std::chrono::milliseconds((long)(1e3 * fault_stall_s)));
        } else {
          set_error("multi: injected fault at panel " + std::to_string(J) + " (SGP_MULTI_FAULT / sgp_bench_multi_fault)");
          return -6;
        }
      }
      if (nxt < g.npan) {             // (a) look-ahead on the owner of the next panel
        const int o1 = g.owner(nxt);
        Rank& k = m->r[o1];
        // what updated panel nxt in the previous steps: near-A launches (s_near) while it was in the current group beyond
        // the look-ahead, near-B launches (s_upd) before that -- the last of them at the previous step if nxt is the first
        // or second panel of its group (ev_B is recorded BEFORE a far update: this never waits for one)
        M_RC(x.wait(o1, k.s_panel, k.ev_A));
        if (nxt % G == 0 || J % G == 0) M_RC(x.wait(o1, k.s_panel, k.ev_B));
        // ... and, before those, the far update with the group two before its own (long done unless G = 1, where that is
        // the previous step's launch)
        if (group_of(nxt) >= 2) M_RC(x.wait(o1, k.s_panel, k.ev_far[(group_of(nxt) - 2) % Rank::NGEV]));
        double t0 = 0, t1 = 0;
        if (prof) {
          M_RC(sync_all(m));
          t0 = now_ms();
        }
        // the look-ahead update follows panel J sub-panel by sub-panel as they land (K = sub each): only the last one sits
        // between the end of J's factorisation and the start of nxt's
        la[0] = nxt;
        {
          const int ns = n_sub(J);
          const int bj = (int)(J % m->ring);
          for (int q = 0; q < ns; ++q) {
            long c, wq;
            sub_range(J, q, c, wq);
            if (g.owner(J) == o1)
              M_RC(x.wait(o1, k.s_panel, k.ev_sub[q]));   // (one rank: its own panel)
            else
              M_RC(x.wait(o1, k.s_panel, k.ev_recv_sub[bj][q]));
            if (fuse_la && q == ns - 1) break;   // the last piece rides in nxt's first panel launch (factor: external source)
            if (x.mine(o1)) {
              M_RC(x.dev(o1));
              M_RC(update_panels(m, F, J, J, la, o1, k.s_panel, &k.upd_flops, c, wq));
            }
          }
        }
        if (prof) {
          M_RC(sync_all(m));
          t1 = now_ms();
          m->prof[(size_t)nxt * (3 + 3 * P) + 1] = t1 - t0;
        }
        M_RC(factor(nxt));
        if (prof) {
          M_RC(sync_all(m));
          m->prof[(size_t)nxt * (3 + 3 * P) + 0] = now_ms() - t1;
          m->prof[(size_t)nxt * (3 + 3 * P) + 2] = 8.0 * (double)g.ldp(nxt) * (double)g.width(nxt);
          M_RC(broadcast_all(nxt));
          M_RC(sync_all(m));
        }
      }
      for (int i = 0; i < P; ++i) {   // (b) every rank's trailing panels: near A, near B, and -- at the end of a group -- far
        Rank& k = m->r[i];
        near_a.clear();
        near_b.clear();
        far.clear();
        for (long Jp : g.mine[(size_t)i]) {
          if (Jp <= nxt) continue;
          const long gp = group_of(Jp);
          if (gp == gj) near_a.push_back(Jp);
          else if (gp == gj + 1) near_b.push_back(Jp);
          else if (group_ends) far.push_back(Jp);
        }
        double t0 = 0;
        auto lap = [&](int slot) -> int {   // profile mode: the class just enqueued, alone on the hardware
          if (!prof) return 0;
          M_RC(sync_all(m));
          const double t = now_ms();
          m->prof[(size_t)J * (3 + 3 * P) + 3 + 3 * i + slot] = t - t0;
          t0 = t;
          return 0;
        };
        if (prof) {
          M_RC(sync_all(m));
          t0 = now_ms();
        }
        const bool any_upd = !near_a.empty() || !near_b.empty() || !far.empty();
        if (any_upd && x.mine(i) && !k.t0_set) {
          M_RC(x.dev(i));
          M_HIP(hipEventRecord(k.ev_t0, k.s_upd));
          k.t0_set = true;
        }
        // near A: on the near stream.  A panel enters the current group out of the next one: its last near-B update (s_upd,
        // previous step) must be done -- ev_B is recorded BEFORE a far update, so this never waits for one.
        if (!near_a.empty()) {
          M_RC(wait_panel(x, F, J, i, k.s_near));
          if (J % G == 0) {
            M_RC(x.wait(i, k.s_near, k.ev_B));
            if (gj >= 2) M_RC(x.wait(i, k.s_near, k.ev_far[(gj - 2) % Rank::NGEV]));
          }
          if (x.mine(i)) {
            M_RC(x.dev(i));
            M_RC(update_panels(m, F, J, J, near_a, i, k.s_near, &k.upd_flops));
          }
        }
        M_RC(x.rec(i, k.ev_A, k.s_near));
        M_RC(lap(0));
        // near B, then far: on the update stream, in order
        M_RC(wait_panel(x, F, J, i, k.s_upd));
        if (!near_b.empty() && x.mine(i)) {
          M_RC(x.dev(i));
          M_RC(update_panels(m, F, J, J, near_b, i, k.s_upd, &k.upd_flops));
        }
        M_RC(x.rec(i, k.ev_B, k.s_upd));
        M_RC(lap(1));
        if (group_ends) {
          if (!far.empty() && x.mine(i)) {
            M_RC(x.dev(i));
            M_RC(update_panels(m, F, gj * G, J, far, i, k.s_upd, &k.upd_flops, -1, 0, true));
          }
          M_RC(x.rec(i, k.ev_far[gj % Rank::NGEV], k.s_upd));
        }
        if (any_upd && x.mine(i)) {
          M_RC(x.dev(i));
          M_HIP(hipEventRecord(k.ev_t1, k.s_upd));
        }
        M_RC(lap(2));
      }
    }
    // ---- |L^-1 (Y - m)|^2 from the bordered rows of the owned panels
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      if (x.factored_once[i]) M_RC(x.wait(i, k.s_upd, k.ev_fact));
      M_RC(x.wait(i, k.s_upd, k.ev_A));
      if (S > 0 && x.mine(i)) {
        M_RC(x.dev(i));
        for (long J : g.mine[(size_t)i]) {
          long nc = std::min(g.width(J), std::max<long>(0, N - g.col0(J)));
          if (nc > 0)
            M_RC(sgp_dev_rowsumsq(k.ctx, F.panel(i, J) + (g.n_pad - g.col0(J)), g.ldp(J), nc, S,
                                  k.d_small + L.pan + (size_t)J * L.per() + 1, (void*)k.s_upd));
        }
      }
    }
    return 0;
  };
  // A failed sweep: part of the schedule is enqueued, part is not.  Peer-copy / loopback work only waits for events that were
  // recorded (a wait is enqueued after its record has been issued), so it drains; RCCL broadcasts whose partners never
  // enqueued their side cannot complete -- the communicators are aborted (ncclCommAbort) and the context refuses further
  // sharded calls (`broken`; advisor, round 4).  Under the one-thread enqueue a grouped broadcast is issued for all ranks or
  // for none, so the communicators survive.
  auto failed = [&](int rc, const std::string& text, bool comm_partial) -> int {
    if (m->transport == TR_RCCL && comm_partial) {
      for (auto& k : m->r) {
        if (k.comm && m->rccl.CommAbort) m->rccl.CommAbort(k.comm);
        k.comm = nullptr;
      }
      m->broken = true;
    } else {
      for (auto& k : m->r) {
        hipSetDevice(k.dev);
        hipDeviceSynchronize();
      }
      (void)hipGetLastError();
    }
    set_error(text + (m->broken ? " (the context's RCCL communicators were aborted: create a new context)" : ""));
    return rc;
  };
  const bool threaded = !prof && ((P > 1 && m->threads < 0) || m->threads == 1);   // (threads == 1 with ONE rank: tests of the failure path)
  if (!threaded) {
    const int rc = run(x0);
    if (rc) return failed(rc, sgp_last_error(), false);
  } else {
    // one enqueue thread per rank (see Exec): the sequence numbers start from what the prologue recorded
    for (auto& kv : m->seq) kv.second.store(0, std::memory_order_relaxed);
    for (auto& kv : x0.cnt) m->seq.at(kv.first).store(kv.second, std::memory_order_relaxed);
    m->abort_flag.store(0);
    std::vector<int> rcs(P, 0);
    std::vector<std::string> errs(P);
    std::vector<std::thread> th;
    for (int i = 0; i < P; ++i)
      th.emplace_back([&, i]() {
        Exec x = x0;
        x.me = i;
        const int rc = run(x);
        if (rc) {
          errs[i] = sgp_last_error();
          m->abort_flag.store(1);
        }
        rcs[i] = rc;
      });
    for (auto& t : th) t.join();
    // report the root cause: a thread that stopped because ANOTHER one failed (-5) only says so
    int bad = -1;
    for (int i = 0; i < P; ++i)
      if (rcs[i] && (bad < 0 || (rcs[bad] == -5 && rcs[i] != -5))) bad = i;
    if (bad >= 0) return failed(rcs[bad], errs[bad], true);
  }
  m->last_enqueue_ms = now_ms() - t_begin;
  // completion of the factorisation proper (statistics; the reductions follow in the caller)
  for (int i = 0; i < P; ++i) {
    Rank& k = m->r[i];
    M_HIP(hipSetDevice(k.dev));
    M_HIP(hipStreamSynchronize(k.s_near));
    M_HIP(hipStreamSynchronize(k.s_upd));
    M_HIP(hipStreamSynchronize(k.s_panel));
    M_HIP(hipStreamSynchronize(k.s_comm));
    for (auto st : k.s_in)
      if (st) M_HIP(hipStreamSynchronize(st));
    if (k.t0_set) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, k.ev_t0, k.ev_t1) == hipSuccess) k.upd_span_ms = ms;
      (void)hipGetLastError();
    }
  }
  m->last_ms = now_ms() - t_begin;
  m->last_npan = g.npan;
  return 0;
}

// info of every rank -> LAPACK's (smallest failing leading minor); > 0 sets the error text
int collect_info(sgp_multi* m) {
  int info = 0;
  for (auto& k : m->r) {
    int inf = 0;
    M_HIP(hipSetDevice(k.dev));
    M_HIP(hipMemcpy(&inf, k.d_info, sizeof(int), hipMemcpyDeviceToHost));
    if (inf == SGP_DF_TIMEOUT) {
      // a panel launch of the dataflow kernel ran into its wait bound (a preempted queue, a profiler serialising kernels): the
      // entry point reruns the operator with the launch-based panel chain (capi.hip: with_df_fallback clears `hybrid`)
      if (m->primary) m->primary->df_timed_out = true;
      set_error("multi: a panel launch of the dataflow kernel ran into its wait bound (SGP_DF_TIMEOUT_S)");
      return -3;
    }
    if (inf > 0 && (info == 0 || inf < info)) info = inf;
  }
  if (info > 0)
    set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
              std::to_string(info));
  return info;
}

void drain(sgp_multi* m, std::vector<sgp_dspec*>& ds, int restore_dev) {
  for (size_t i = 0; i < m->r.size(); ++i) {
    Rank& k = m->r[i];
    hipSetDevice(k.dev);
    hipStreamSynchronize(k.s_upd);
    hipStreamSynchronize(k.s_panel);
    hipStreamSynchronize(k.s_comm);
    hipStreamSynchronize(k.s_near);
    for (auto st : k.s_in)
      if (st) hipStreamSynchronize(st);
    if (i < ds.size() && ds[i]) {
      drv_dspec_free(ds[i]);
      ds[i] = nullptr;
    }
  }
  hipSetDevice(restore_dev);
}

long spec_rows(const sgp_cov_spec* spec) {
  long N = 0;
  for (int i = 0; i < spec->n_row_blocks; ++i) N += spec->row_len[i];
  return N;
}

// logdet and |L^-1 (Y - m)|^2 per column from the per-panel slots of every rank (SmallLayout::pan), added in panel order:
// red[0] = logdet, red[1 + s] = column s.  RCCL: one ncclAllReduce of the slot array (every slot is non-zero on its owner
// only, so the sum is exact), then rank 0's copy; else every rank's slots are fetched and the owner's taken.
int reduce_scalars(sgp_multi* m, const Geometry& g, const SmallLayout& L, std::vector<double>& red) {
  const int P = (int)m->r.size();
  const long per = L.per(), nslot = g.npan * per;
  std::vector<double> all((size_t)nslot, 0.0), tmp((size_t)nslot);
  if (m->transport == TR_RCCL) {
    int rc = m->rccl.GroupStart();
    for (int i = 0; i < P && rc == 0; ++i) {
      Rank& k = m->r[i];
      hipSetDevice(k.dev);
      rc = m->rccl.AllReduce(k.d_small + L.pan, k.d_small + L.red, (size_t)nslot, NCCL_DOUBLE, NCCL_SUM, k.comm, k.s_upd);
    }
    int rc2 = m->rccl.GroupEnd();
    if (rc || rc2) {
      set_error("ncclAllReduce failed");
      return -4;
    }
    Rank& k0 = m->r[0];
    M_HIP(hipSetDevice(k0.dev));
    M_HIP(hipMemcpyAsync(all.data(), k0.d_small + L.red, sizeof(double) * nslot, hipMemcpyDeviceToHost, k0.s_upd));
    for (auto& k : m->r) {
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipStreamSynchronize(k.s_upd));
    }
  } else {
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipMemcpy(tmp.data(), k.d_small + L.pan, sizeof(double) * nslot, hipMemcpyDeviceToHost));
      for (long J : g.mine[(size_t)i])
        for (long q = 0; q < per; ++q) all[(size_t)(J * per + q)] = tmp[(size_t)(J * per + q)];
    }
  }
  red.assign((size_t)per, 0.0);
  for (long J = 0; J < g.npan; ++J)
    for (long q = 0; q < per; ++q) red[(size_t)q] += all[(size_t)(J * per + q)];
  return 0;
}

// transient factor in the ranks' grow-only stores
int transient_fact(sgp_multi* m, Fact& F) {
  const int P = (int)F.g.P;
  F.base.assign(P, nullptr);
  F.off.assign(P, {});
  for (int i = 0; i < P; ++i) {
    Rank& k = m->r[i];
    M_HIP(hipSetDevice(k.dev));
    size_t tot = plan_offsets(F.g, i, F.off[i]);
    if (tot > k.store_cap) M_HIP(hipDeviceSynchronize());
    M_RC(grow(&k.store, &k.store_cap, std::max<size_t>(tot, 1)));
    F.base[i] = k.store;
  }
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// logpdf(fx, y) / logpdf(fx, Y) over the ranks of ctx->multi (called from sgp_logpdf with the primary context held)
// ---------------------------------------------------------------------------------------------------------
int sgp_multi_logpdf(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                     const double* noise, const double* Y, int64_t ldy, int64_t ncols, double* out) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  const long N = spec_rows(spec);
  M_CHECK_ARG(N >= 1 && ncols >= 1 && ldy >= N, "sgp_logpdf (multi): bad sizes");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG || noise_kind == SGP_NOISE_DENSE,
              "sgp_logpdf (multi): bad noise kind");
  Fact F;
  F.g = make_geometry(m, N, ncols, spec, noise_kind);
  const SmallLayout L(N, ncols, F.g.npan);
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto body = [&]() -> int {
    M_RC(transient_fact(m, F));
    M_RC(factorize(m, F, spec, mean, noise_kind, noise, Y, ldy, ds));
    std::vector<double> red;
    M_RC(reduce_scalars(m, F.g, L, red));
    int info = collect_info(m);
    if (info) return info;
    for (long s = 0; s < ncols; ++s) out[s] = -0.5 * ((double)N * 1.8378770664093453 + red[0] + red[1 + s]);
    return 0;
  };
  int rc = body();
  drain(m, ds, ctx->device);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
// rand(rng, fx, S): m + L Z with the factor sharded; every rank multiplies its own column panels
// ---------------------------------------------------------------------------------------------------------
int sgp_multi_rand(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind, const double* noise,
                   const double* Z, int64_t ldz, int64_t S, double* out, int64_t ldo) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  const long N = spec_rows(spec);
  M_CHECK_ARG(N >= 1 && S >= 1 && ldz >= N && ldo >= N, "sgp_rand (multi): bad sizes");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG || noise_kind == SGP_NOISE_DENSE,
              "sgp_rand (multi): bad noise kind");
  Fact F;
  F.g = make_geometry(m, N, 0, spec, noise_kind);
  const Geometry& g = F.g;
  const long s_pad = rup(S, TILE), n_pad = g.n_pad;
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto body = [&]() -> int {
    M_RC(transient_fact(m, F));
    M_RC(factorize(m, F, spec, nullptr, noise_kind, noise, nullptr, N, ds));
    int info = collect_info(m);
    if (info) return info;
    std::vector<double> acc((size_t)N * S), part((size_t)N * S);
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      // work: Z (N x S) | Zt (s_pad x n_pad) ; work2: Out (n_pad x s_pad)
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)N * S + (size_t)s_pad * n_pad));
      M_RC(grow(&k.d_work2, &k.work2_cap, (size_t)n_pad * s_pad));
      double* dZ = k.d_work;
      double* dZt = k.d_work + (size_t)N * S;
      double* dOut = k.d_work2;
      hipStream_t s = k.s_upd;
      M_HIP(hipMemcpy2DAsync(dZ, sizeof(double) * N, Z, sizeof(double) * ldz, sizeof(double) * N, (size_t)S,
                             hipMemcpyHostToDevice, s));
      M_HIP(hipMemsetAsync(dZt, 0, sizeof(double) * s_pad * n_pad, s));
      M_HIP(hipMemsetAsync(dOut, 0, sizeof(double) * n_pad * s_pad, s));
      M_RC(launch_transpose_add(dZ, N, N, S, dZt, s_pad, nullptr, s));
      for (long J : g.mine[(size_t)i]) {
        const long J0 = g.col0(J);
        M_RC(launch_gemm_nt_lz_k(F.panel(i, J), g.ldp(J), dZt + J0 * s_pad, s_pad, dOut + J0, n_pad, n_pad - J0, s_pad,
                                 g.width(J), 1.0, s));
      }
    }
    for (int i = 0; i < P; ++i) {   // fixed rank order: deterministic
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipStreamSynchronize(k.s_upd));
      M_HIP(hipMemcpy2D(part.data(), sizeof(double) * N, k.d_work2, sizeof(double) * n_pad, sizeof(double) * N,
                        (size_t)S, hipMemcpyDeviceToHost));
      if (i == 0)
        acc = part;
      else
        for (size_t q = 0; q < acc.size(); ++q) acc[q] += part[q];
    }
    for (long s = 0; s < S; ++s)
      for (long r = 0; r < N; ++r) out[r + s * ldo] = (mean ? mean[r] : 0.0) + acc[(size_t)r + (size_t)s * N];
    return 0;
  };
  int rc = body();
  drain(m, ds, ctx->device);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
// posterior(fx, y): a KEPT sharded factor
// ---------------------------------------------------------------------------------------------------------
struct sgp_mpost {
  sgp_ctx* ctx = nullptr;
  std::vector<int> devs;        // device of every rank (the object may outlive its context: freed without it)
  Fact F;
  std::vector<double*> zloc;    // [P] z = L^-1 (y - m) of the rank's own columns, contiguous (panel order)
  std::vector<double*> stage;   // [P] receive slots of the left-looking row solve: (P - 1) x ns_pad x W, grow-only
  std::vector<size_t> stage_cap;
};

void sgp_multi_posterior_destroy(sgp_mpost* mp) {
  if (!mp) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (size_t i = 0; i < mp->F.base.size(); ++i) {
    if (i < mp->devs.size()) hipSetDevice(mp->devs[i]);
    if (mp->F.base[i]) hipFree(mp->F.base[i]);
    if (i < mp->F.inv.size() && mp->F.inv[i]) hipFree(mp->F.inv[i]);
    if (i < mp->zloc.size() && mp->zloc[i]) hipFree(mp->zloc[i]);
    if (i < mp->stage.size() && mp->stage[i]) hipFree(mp->stage[i]);
  }
  hipSetDevice(cur);
  delete mp;
}

int sgp_multi_posterior_create(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                               const double* noise, const double* y, double* alpha_out, sgp_mpost** out) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  const long N = spec_rows(spec);
  M_CHECK_ARG(N >= 1, "sgp_posterior_create (multi): empty data");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG || noise_kind == SGP_NOISE_DENSE,
              "sgp_posterior_create (multi): bad noise kind");
  sgp_mpost* mp = new sgp_mpost();
  mp->ctx = ctx;
  for (auto& k : m->r) mp->devs.push_back(k.dev);
  Fact& F = mp->F;
  F.g = make_geometry(m, N, 1, spec, noise_kind);
  const Geometry& g = F.g;
  F.base.assign(P, nullptr);
  F.inv.assign(P, nullptr);
  F.off.assign(P, {});
  F.inv_per_panel = (g.W / TILE) * drv_invd_stride();
  mp->zloc.assign(P, nullptr);
  mp->stage.assign(P, nullptr);
  mp->stage_cap.assign(P, 0);
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto body = [&]() -> int {
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      size_t tot = plan_offsets(g, i, F.off[i]);
      size_t cap = 0;
      M_RC(grow(&F.base[i], &cap, std::max<size_t>(tot, 1)));
      cap = 0;
      M_RC(grow(&F.inv[i], &cap, std::max<size_t>((size_t)F.off[i].size() * F.inv_per_panel, 1)));
      cap = 0;
      M_RC(grow(&mp->zloc[i], &cap, (size_t)std::max<long>(g.ncols_owned(i), 1)));
    }
    M_RC(factorize(m, F, spec, mean, noise_kind, noise, y, N, ds));
    int info = collect_info(m);
    if (info) return info;
    // z of the owned columns, contiguous
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      long at = 0;
      for (long J : g.mine[(size_t)i]) {
        M_RC(drv_copy_strided(F.panel(i, J) + (g.n_pad - g.col0(J)), g.ldp(J), g.width(J), mp->zloc[i] + at, k.s_upd));
        at += g.width(J);
      }
    }
    if (alpha_out) {
      // alpha = L^-T z, panel by panel from the last: the owner subtracts what the later panels contribute
      // (its own columns of L times the alpha tail, which every rank holds) and solves its diagonal block;
      // the W new entries go to every rank.  work: zg (n_pad, global index) | alpha (n_pad)
      for (int i = 0; i < P; ++i) {
        Rank& k = m->r[i];
        M_HIP(hipSetDevice(k.dev));
        M_RC(grow(&k.d_work, &k.work_cap, (size_t)2 * g.n_pad));
        M_HIP(hipMemsetAsync(k.d_work, 0, sizeof(double) * 2 * g.n_pad, k.s_upd));
        long at = 0;
        for (long J : g.mine[(size_t)i]) {
          M_HIP(hipMemcpyAsync(k.d_work + g.col0(J), mp->zloc[i] + at, sizeof(double) * g.width(J),
                               hipMemcpyDeviceToDevice, k.s_upd));
          at += g.width(J);
        }
        M_HIP(hipEventRecord(k.ev_upd, k.s_upd));
        M_HIP(hipStreamWaitEvent(k.s_panel, k.ev_upd, 0));
      }
      for (long J = g.npan - 1; J >= 0; --J) {
        const int o = g.owner(J);
        Rank& k = m->r[o];
        const long J0 = g.col0(J), w = g.width(J);
        M_HIP(hipSetDevice(k.dev));
        // virtual bases: factor element (r, c) at Lv[r + c * ldp], inverse blocks of global 128-block b at
        // wall_v + b * stride
        const double* Lv = F.panel(o, J) - J0 - J0 * g.ldp(J);
        const double* wall_v = F.invp(o, J) - (J0 / TILE) * drv_invd_stride();
        M_RC(drv_back_substitute(Lv, g.ldp(J), wall_v, J0 + w - TILE, J0, g.n_pad, k.d_work, k.d_work + g.n_pad,
                                 k.s_panel));
        M_HIP(hipEventRecord(k.ev_fact, k.s_panel));
        for (int i = 0; i < P; ++i) {   // the W new entries of alpha -> every other rank (written once: nothing to order)
          if (i == o) continue;
          Rank& q = m->r[i];
          M_HIP(hipSetDevice(q.dev));
          M_HIP(hipStreamWaitEvent(q.s_panel, k.ev_fact, 0));
          if (q.dev == k.dev)
            M_HIP(hipMemcpyAsync(q.d_work + g.n_pad + J0, k.d_work + g.n_pad + J0, sizeof(double) * w,
                                 hipMemcpyDeviceToDevice, q.s_panel));
          else
            M_HIP(hipMemcpyPeerAsync(q.d_work + g.n_pad + J0, q.dev, k.d_work + g.n_pad + J0, k.dev, sizeof(double) * w,
                                     q.s_panel));
        }
      }
      Rank& k0 = m->r[0];
      M_HIP(hipSetDevice(k0.dev));
      M_HIP(hipStreamSynchronize(k0.s_panel));
      M_HIP(hipMemcpy(alpha_out, k0.d_work + g.n_pad, sizeof(double) * N, hipMemcpyDeviceToHost));
    }
    return 0;
  };
  int rc = body();
  drain(m, ds, ctx->device);
  if (rc) {
    sgp_multi_posterior_destroy(mp);
    return rc;
  }
  *out = mp;
  return 0;
}

// R <- R L^-T for ns_pad rows held column-sharded like the factor (rank i: ns_pad x its columns in d_work, panel after panel,
// at[i][l] = column offset of its l-th panel; the partial-sum scratch S behind them; mp->stage sized by the caller): the
// left-looking sweep over the panels of the kept factor -- every rank forms the partial sums of its own earlier panels, the
// owner collects them in rank order and solves against its diagonal block.
static int sweep_rows(sgp_multi* m, sgp_mpost* mp, long ns_pad, const std::vector<std::vector<long>>& at) {
  const Fact& F = mp->F;
  const Geometry& g = F.g;
  const int P = (int)g.P;
  std::vector<char> stage_used(P, 0);
  for (long J = 0; J < g.npan; ++J) {
    const int o = g.owner(J);
    const long J0 = g.col0(J), w = g.width(J);
    Rank& ko = m->r[o];
    double* Tj = ko.d_work + (size_t)at[o][g.local(J)] * ns_pad;   // R_J, becomes V_J
    bool used_now = false;
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const long nco = g.ncols_owned(i);
      double* Si = k.d_work + (size_t)ns_pad * std::max<long>(nco, 1);
      bool first = true, any = false;
      for (long kk : g.mine[(size_t)i]) {
        if (kk >= J) break;
        // V_kk (ns_pad x w_kk) times rows J0 .. J0 + w of panel kk
        const double* Vk = k.d_work + (size_t)at[i][g.local(kk)] * ns_pad;
        const double* Ljk = F.panel(i, kk) + (J0 - g.col0(kk));
        if (i == o)
          M_RC(launch_gemm_nt(Vk, ns_pad, Ljk, g.ldp(kk), Tj, ns_pad, ns_pad, w, g.width(kk), -1.0, 1.0, -(1L << 40), 0,
                              0, k.s_upd));
        else
          M_RC(launch_gemm_nt(Vk, ns_pad, Ljk, g.ldp(kk), Si, ns_pad, ns_pad, w, g.width(kk), 1.0, first ? 0.0 : 1.0,
                              -(1L << 40), 0, 0, k.s_upd));
        first = false;
        any = true;
      }
      if (i != o && any) {
        double* slot = mp->stage[o] + (size_t)(i < o ? i : i - 1) * ns_pad * g.W;
        if (stage_used[o]) M_HIP(hipStreamWaitEvent(k.s_upd, ko.ev_done, 0));   // the slot's last content was consumed
        if (k.dev == ko.dev)
          M_HIP(hipMemcpyAsync(slot, Si, sizeof(double) * ns_pad * w, hipMemcpyDeviceToDevice, k.s_upd));
        else
          M_HIP(hipMemcpyPeerAsync(slot, ko.dev, Si, k.dev, sizeof(double) * ns_pad * w, k.s_upd));
        M_HIP(hipEventRecord(k.ev_upd, k.s_upd));
        M_HIP(hipSetDevice(ko.dev));
        M_HIP(hipStreamWaitEvent(ko.s_upd, k.ev_upd, 0));
        M_RC(drv_axpy_block(Tj, ns_pad, slot, ns_pad, ns_pad, w, -1.0, ko.s_upd));   // rank order: deterministic
        used_now = true;
      }
    }
    M_HIP(hipSetDevice(ko.dev));
    if (used_now) {
      M_HIP(hipEventRecord(ko.ev_done, ko.s_upd));
      stage_used[o] = 1;
    }
    M_RC(drv_row_trsm(ko.ctx, Tj, ns_pad, ns_pad, F.panel(o, J), g.ldp(J), F.invp(o, J), w, ko.s_upd));
  }
  return 0;
}

int sgp_multi_posterior_predict(sgp_mpost* mp, const sgp_cov_spec* cross, const sgp_cov_spec* prior_ss,
                                const double* mean_s, double* mean_out, double* var_out, double* cov_out,
                                int64_t ldcov) {
  sgp_ctx* ctx = mp->ctx;
  sgp_multi* m = ctx->multi;
  const Fact& F = mp->F;
  const Geometry& g = F.g;
  const int P = (int)g.P;
  const long Ns = spec_rows(cross);
  long Mx = 0;
  for (int j = 0; j < cross->n_col_blocks; ++j) Mx += cross->col_len[j];
  M_CHECK_ARG(Mx == g.N, "sgp_posterior_predict: cross spec columns != training size");
  M_CHECK_ARG(!prior_ss || spec_rows(prior_ss) == Ns, "sgp_posterior_predict: prior_ss size != number of x*");
  M_CHECK_ARG(!cov_out || ldcov >= Ns, "sgp_posterior_predict: ldcov < Ns");
  M_CHECK_ARG((!var_out && !cov_out) || prior_ss, "sgp_posterior_predict: prior_ss spec required for var / cov");
  if (Ns == 0) return 0;
  const long ns_pad = rup(Ns, TILE);
  std::vector<sgp_dspec*> ds(P, nullptr);
  sgp_dspec* dprior = nullptr;
  auto body = [&]() -> int {
    // ---- K(x*, x), column-sharded like the factor: rank i holds ns_pad x (its columns), panel after panel
    std::vector<std::vector<long>> at(P);   // column offset of local panel l inside the rank's block
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const long nco = g.ncols_owned(i);
      // work: R / V' (ns_pad x nco) | partial sums S (ns_pad x W);  work2: results (dot, sumsq, zero, Gram)
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)ns_pad * std::max<long>(nco, 1) + (size_t)ns_pad * g.W));
      M_RC(grow(&k.d_work2, &k.work2_cap, (size_t)3 * ns_pad + (cov_out ? (size_t)ns_pad * ns_pad : 0)));
      if (P > 1) {
        size_t need = (size_t)(P - 1) * ns_pad * g.W;
        if (need > mp->stage_cap[i]) M_RC(sync_all(m));   // peers copy into this buffer: all ranks idle before it is freed
        M_RC(grow(&mp->stage[i], &mp->stage_cap[i], need));
      }
      M_RC(drv_dspec_create(k.ctx, cross, &ds[i]));
      M_HIP(hipMemsetAsync(k.d_work, 0, sizeof(double) * ns_pad * std::max<long>(nco, 1), k.s_upd));
      long a = 0;
      for (long J : g.mine[(size_t)i]) {
        at[i].push_back(a);
        const long c0 = g.col0(J), w = g.width(J);
        // element (r, c) of K(x*, x) at Kv[r + c * ns_pad] with Kv = block - c0 * ns_pad
        double* blk = k.d_work + (size_t)a * ns_pad;
        M_RC(drv_assemble(ds[i], blk - c0 * ns_pad, ns_pad, 0, ns_pad / TILE, c0 / TILE, (c0 + w) / TILE, 0, -1, 0.0,
                          nullptr, k.s_upd));
        a += w;
      }
    }
    M_RC(sweep_rows(m, mp, ns_pad, at));
    // ---- sums over columns = over ranks
    std::vector<double> dot(Ns, 0.0), ssq(Ns, 0.0), tmp(Ns), prior(Ns, 0.0);
    std::vector<double> G, Gt;
    if (cov_out) {
      G.assign((size_t)Ns * Ns, 0.0);
      Gt.resize((size_t)Ns * Ns);
    }
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const long nco = g.ncols_owned(i);
      if (nco == 0) continue;
      double* r_dot = k.d_work2;
      double* r_ssq = k.d_work2 + ns_pad;
      double* r_zero = k.d_work2 + 2 * ns_pad;
      double* r_G = k.d_work2 + 3 * ns_pad;
      hipStream_t s = k.s_upd;
      M_HIP(hipMemsetAsync(r_zero, 0, sizeof(double) * ns_pad, s));
      if (mean_out) M_RC(launch_gemv_rows(k.d_work, ns_pad, Ns, nco, mp->zloc[i], 1, nullptr, r_dot, s));
      if (var_out) M_RC(launch_colsumsq_sub(k.d_work, ns_pad, Ns, nco, r_zero, r_ssq, 1.0, s));
      if (cov_out)
        M_RC(launch_gemm_nt(k.d_work, ns_pad, k.d_work, ns_pad, r_G, ns_pad, ns_pad, ns_pad, nco, 1.0, 0.0, -(1L << 40), 0,
                            0, s));
    }
    for (int i = 0; i < P; ++i) {   // fixed rank order: deterministic
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipStreamSynchronize(k.s_upd));
      if (g.ncols_owned(i) == 0) continue;
      if (mean_out) {
        M_HIP(hipMemcpy(tmp.data(), k.d_work2, sizeof(double) * Ns, hipMemcpyDeviceToHost));
        for (long q = 0; q < Ns; ++q) dot[q] += tmp[q];
      }
      if (var_out) {
        M_HIP(hipMemcpy(tmp.data(), k.d_work2 + ns_pad, sizeof(double) * Ns, hipMemcpyDeviceToHost));
        for (long q = 0; q < Ns; ++q) ssq[q] += tmp[q];
      }
      if (cov_out) {
        M_HIP(hipMemcpy2D(Gt.data(), sizeof(double) * Ns, k.d_work2 + 3 * ns_pad, sizeof(double) * ns_pad,
                          sizeof(double) * Ns, (size_t)Ns, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < G.size(); ++q) G[q] += Gt[q];
      }
    }
    if (mean_out)
      for (long q = 0; q < Ns; ++q) mean_out[q] = (mean_s ? mean_s[q] : 0.0) + dot[q];
    if (var_out || cov_out) {   // the prior at x*: rank 0
      Rank& k = m->r[0];
      M_HIP(hipSetDevice(k.dev));
      M_RC(drv_dspec_create(k.ctx, prior_ss, &dprior));
      hipStream_t s = k.s_upd;
      if (var_out) {
        M_RC(drv_diag_of_spec(k.ctx, dprior, k.d_work2, s));
        M_HIP(hipStreamSynchronize(s));
        M_HIP(hipMemcpy(prior.data(), k.d_work2, sizeof(double) * Ns, hipMemcpyDeviceToHost));
        for (long q = 0; q < Ns; ++q) var_out[q] = prior[q] - ssq[q];
      }
      if (cov_out) {
        double* r_G = k.d_work2 + 3 * ns_pad;
        M_HIP(hipMemsetAsync(r_G, 0, sizeof(double) * ns_pad * ns_pad, s));
        M_RC(drv_assemble(dprior, r_G, ns_pad, 0, ns_pad / TILE, 0, ns_pad / TILE, 0, -1, 0.0, nullptr, s));
        M_HIP(hipStreamSynchronize(s));
        M_HIP(hipMemcpy2D(Gt.data(), sizeof(double) * Ns, r_G, sizeof(double) * ns_pad, sizeof(double) * Ns, (size_t)Ns,
                          hipMemcpyDeviceToHost));
        for (long c = 0; c < Ns; ++c)
          for (long r = 0; r < Ns; ++r)
            cov_out[r + c * ldcov] = Gt[(size_t)r + (size_t)c * Ns] - G[(size_t)r + (size_t)c * Ns];
      }
    }
    return 0;
  };
  int rc = body();
  if (dprior) {
    hipSetDevice(m->r[0].dev);
    hipStreamSynchronize(m->r[0].s_upd);
    drv_dspec_free(dprior);
  }
  drain(m, ds, ctx->device);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
// logpdf and its reverse-mode gradient over the ranks (round 4; single-GPU: capi.hip: logpdf_grad_core) -- what Zygote
// derives through logpdf(f(x, s2), y) for hyper-parameter learning (examples/getting_started/script.jl:154-213) at sizes
// that need the node.  With alpha = C^-1 (y - m), G = (alpha alpha' - C^-1) / 2:
//   1. the sharded factorisation, kept (sgp_multi_posterior_create: packed panels + inverse diagonal blocks, z, alpha);
//   2. X = L^-T, column-sharded like the factor: the identity rides through the left-looking row sweep of the posterior
//      (sweep_rows) -- rank i ends up with the columns of L^-T that belong to its panels;
//   3. C^-1 = X X' = sum over ranks of X_i X_i': every rank forms its partial N x N product (one MFMA GEMM, K = its columns),
//      then a reduce-scatter by column slabs with peer copies -- rank j adds, in rank order, the slabs of ITS columns;
//   4. every rank contracts its column slabs of G with the kernel derivatives of every term (grad.hip: the single-GPU
//      contraction kernel on a column window); the per-term sums are added over ranks in rank order on the host;
//   5. d / d noise from diag C^-1 (each rank reports the diagonal entries of its columns), d / d y = -alpha, d / d m = alpha.
// Per rank ~ (1/3 + 1/3 + 1) N^3 / P flops and (P - 1) / P N^2 doubles received; N^2 doubles of scratch per rank.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ void set_identity_cols_kernel(double* blk, long ld, long c0, long w, long N) {
  const long lc = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lc < w && c0 + lc < N) blk[(c0 + lc) + lc * ld] = 1.0;
}
}  // namespace

// grad_inputs / grad_rowscale (round 6; sgp_logpdf_grad_x / _xs): the input-point and function-scale gradients on the sharded
// result as well -- every rank contracts ITS column slabs of G = (alpha alpha' - C^-1) / 2 with the kernel derivatives row-side
// (the single-GPU kernel, grad.hip: launch_grad_inputs, on a column window of the block pair), the per-rank sums are added on
// the host in rank order.  "Row side x 2" covers the column side as on one GPU (the spec and G are symmetric).
int sgp_multi_logpdf_grad(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind, const double* noise,
                          const double* y, double* logpdf_out, double* grad_y, double* grad_mean, double* grad_noise,
                          double* grad_coef, double* grad_inscale, double* const* grad_inputs, double* const* grad_rowscale) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  const long N = spec_rows(spec);
  M_CHECK_ARG(N >= 1, "sgp_logpdf_grad (multi): empty data");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG || noise_kind == SGP_NOISE_DENSE,
              "sgp_logpdf_grad (multi): bad noise kind");
  const bool dense_noise = noise_kind == SGP_NOISE_DENSE;   // (round 6) d / d Sigma_y = G itself, N x N: every rank returns its columns
  M_CHECK_ARG(m->W <= 4096, "sgp_logpdf_grad (multi): panels wider than 4096 columns");
  std::vector<double> alpha(N);
  sgp_mpost* mp = nullptr;
  M_RC(sgp_multi_posterior_create(ctx, spec, mean, noise_kind, noise, y, alpha.data(), &mp));
  const Fact& F = mp->F;
  const Geometry& g = F.g;
  const long n_pad = g.n_pad;
  const SmallLayout L(N, 1, g.npan);
  std::vector<sgp_dspec*> ds(P, nullptr);
  size_t nterms = 0;
  std::vector<double> gc_sum, gs_sum, kdiag(N, 0.0);
  std::vector<std::vector<double*>> gx_dev((size_t)P), gr_dev((size_t)P);   // [rank][input] / [rank][term], freed below
  auto body = [&]() -> int {
    // ---- logpdf from the scalars the factorisation left on every rank (rank order: deterministic)
    std::vector<double> red;
    M_RC(reduce_scalars(m, g, L, red));
    *logpdf_out = -0.5 * ((double)N * 1.8378770664093453 + red[0] + red[1]);
    // ---- X = L^-T: identity rows through the row sweep
    std::vector<std::vector<long>> at(P);
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const long nco = g.ncols_owned(i);
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)n_pad * std::max<long>(nco, 1) + (size_t)n_pad * g.W));
      M_RC(grow(&k.d_work2, &k.work2_cap, (size_t)n_pad * n_pad + (size_t)n_pad + 4 * 4096));
      if (P > 1) {
        size_t need = (size_t)(P - 1) * n_pad * g.W;
        if (need > mp->stage_cap[i]) M_RC(sync_all(m));
        M_RC(grow(&mp->stage[i], &mp->stage_cap[i], need));
      }
      M_RC(drv_dspec_create(k.ctx, spec, &ds[i]));
      M_HIP(hipMemsetAsync(k.d_work, 0, sizeof(double) * n_pad * std::max<long>(nco, 1), k.s_upd));
      long a = 0;
      for (long J : g.mine[(size_t)i]) {
        at[i].push_back(a);
        const long w = g.width(J);
        hipLaunchKernelGGL(set_identity_cols_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, k.s_upd,
                           k.d_work + (size_t)a * n_pad, n_pad, g.col0(J), w, N);
        M_HIP(hipGetLastError());
        a += w;
      }
    }
    M_RC(sweep_rows(m, mp, n_pad, at));
    // ---- partial C^-1 of every rank, then the reduce-scatter by column slabs
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const long nco = g.ncols_owned(i);
      if (nco > 0)
        M_RC(launch_gemm_nt(k.d_work, n_pad, k.d_work, n_pad, k.d_work2, n_pad, n_pad, n_pad, nco, 1.0, 0.0, -(1L << 40), 0, 0,
                            k.s_upd));
      else
        M_HIP(hipMemsetAsync(k.d_work2, 0, sizeof(double) * n_pad * n_pad, k.s_upd));
      M_HIP(hipEventRecord(k.ev_done, k.s_upd));
    }
    for (int j = 0; j < P; ++j) {
      Rank& kj = m->r[j];
      M_HIP(hipSetDevice(kj.dev));
      for (int i = 0; i < P; ++i)
        if (i != j) M_HIP(hipStreamWaitEvent(kj.s_upd, m->r[i].ev_done, 0));
      for (long J : g.mine[(size_t)j]) {
        const long c0 = g.col0(J), w = g.width(J);
        double* dst = kj.d_work2 + (size_t)c0 * n_pad;
        for (int i = 0; i < P; ++i) {
          if (i == j) continue;
          Rank& ki = m->r[i];
          double* slot = mp->stage[j] + (size_t)(i < j ? i : i - 1) * n_pad * g.W;
          const double* src = ki.d_work2 + (size_t)c0 * n_pad;
          if (ki.dev == kj.dev)
            M_HIP(hipMemcpyAsync(slot, src, sizeof(double) * n_pad * w, hipMemcpyDeviceToDevice, kj.s_upd));
          else
            M_HIP(hipMemcpyPeerAsync(slot, kj.dev, src, ki.dev, sizeof(double) * n_pad * w, kj.s_upd));
          M_RC(drv_axpy_block(dst, n_pad, slot, n_pad, n_pad, w, 1.0, kj.s_upd));   // rank order: deterministic
        }
      }
    }
    // (a rank's partial product is read by the others until their slabs are complete: nobody frees / reuses it before)
    // ---- contraction of every rank's column slabs with the kernel derivatives, diag C^-1 of its columns
    nterms = ds[0]->h_terms.size();
    gc_sum.assign(std::max<size_t>(1, nterms), 0.0);
    gs_sum.assign(std::max<size_t>(1, nterms), 0.0);
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      const sgp_dspec* d = ds[i];
      const long nco = g.ncols_owned(i);
      double* d_alpha = k.d_work2 + (size_t)n_pad * n_pad;
      double* d_gc = d_alpha + n_pad;          // nterms <= 4096 each (checked below)
      double* d_gs = d_gc + 4096;
      double* d_diag = d_gs + 4096;             // (4096 doubles: one panel's diagonal at a time)
      M_CHECK_ARG(nterms <= 4096, "sgp_logpdf_grad (multi): more than 4096 terms");
      M_HIP(hipMemsetAsync(d_alpha, 0, sizeof(double) * n_pad, k.s_upd));
      M_HIP(hipMemcpyAsync(d_alpha, alpha.data(), sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
      M_HIP(hipMemsetAsync(d_gc, 0, sizeof(double) * 2 * 4096, k.s_upd));
      // partial-sum scratch of the contraction kernel: 16 doubles per tile of the largest window
      double* d_part = k.d_work;   // (X is no longer needed)
      M_CHECK_ARG((size_t)(n_pad / TILE) * (g.W / TILE) * 16 <= k.work_cap, "sgp_logpdf_grad (multi): scratch too small");
      // per-rank accumulators of the input-point / row-scale gradients (device; summed over ranks on the host below)
      if (grad_inputs) {
        gx_dev[i].assign((size_t)spec->n_inputs, nullptr);
        for (int a = 0; a < spec->n_inputs; ++a) {
          const size_t cnt = (size_t)std::max<long>(1, (long)d->in_dim[a] * d->in_n[a]);
          M_HIP(hipMalloc(&gx_dev[i][(size_t)a], sizeof(double) * cnt));
          M_HIP(hipMemsetAsync(gx_dev[i][(size_t)a], 0, sizeof(double) * cnt, k.s_upd));
        }
      }
      if (grad_rowscale) {
        gr_dev[i].assign(nterms, nullptr);
        for (int I = 0; I < d->nrb; ++I)
          for (int Jb = 0; Jb < d->ncb; ++Jb)
            for (int t = d->term_ptr[I * d->ncb + Jb]; t < d->term_ptr[I * d->ncb + Jb + 1]; ++t)
              if (grad_rowscale[t] && d->h_terms[t].rs && d->row_len[I] > 0) {
                M_HIP(hipMalloc(&gr_dev[i][(size_t)t], sizeof(double) * d->row_len[I]));
                M_HIP(hipMemsetAsync(gr_dev[i][(size_t)t], 0, sizeof(double) * d->row_len[I], k.s_upd));
              }
      }
      for (long J : g.mine[(size_t)i]) {
        const long pc0 = g.col0(J), pw = std::min(g.width(J), N - pc0);
        if (pw <= 0) continue;
        if (grad_coef || grad_inscale)
          for (int I = 0; I < d->nrb; ++I) {
            if (d->row_len[I] == 0) continue;
            for (int Jb = 0; Jb < d->ncb; ++Jb) {
              if (d->col_len[Jb] == 0) continue;
              // columns of block Jb inside this panel
              const long bc0 = d->col_off[Jb], bc1 = bc0 + d->col_len[Jb];
              const long lo = std::max(bc0, pc0), hi = std::min(bc1, pc0 + pw);
              if (lo >= hi) continue;
              const long r0 = d->row_off[I], nr = d->row_len[I];
              const long trf = r0 / TILE, trl = (r0 + nr - 1) / TILE + 1, tcf = lo / TILE, tcl = (hi - 1) / TILE + 1;
              const int p = I * d->ncb + Jb;
              const int t0 = d->term_ptr[p], t1 = d->term_ptr[p + 1];
              const int dmax = d->pair_dmax[p];
              const int per = std::min(8, std::max(1, 64 / dmax));
              for (int t = t0; t < t1; t += per) {
                const int cnt = std::min(per, t1 - t);
                // (the kernel masks entries outside rows r0 .. r0 + nr and columns lo .. hi of the block pair; the term's
                // column points are addressed relative to the block's first column: the window starts at lo)
                M_RC(launch_grad_block(k.d_work2, n_pad, d_alpha, r0, nr, bc0, bc1 - bc0, d->d_terms + t, cnt, dmax, trf, tcf,
                                       trl - trf, tcl - tcf, d_part, d_gc + t, d_gs + t, k.s_upd, 1, lo, hi));
              }
            }
          }
        if (grad_inputs || grad_rowscale)
          for (int I = 0; I < d->nrb; ++I) {
            if (d->row_len[I] == 0) continue;
            for (int Jb = 0; Jb < d->ncb; ++Jb) {
              const long bc0 = d->col_off[Jb], bc1 = bc0 + d->col_len[Jb];
              const long lo = std::max(bc0, pc0), hi = std::min(bc1, pc0 + pw);
              if (lo >= hi) continue;
              const int p = I * d->ncb + Jb;
              for (int t = d->term_ptr[p]; t < d->term_ptr[p + 1]; ++t) {
                double* gsv = (grad_rowscale && grad_rowscale[t] && d->h_terms[t].rs) ? gr_dev[i][(size_t)t] : nullptr;
                double* gxa = grad_inputs ? gx_dev[i][(size_t)d->term_row_input[t]] : nullptr;
                if (!gxa && !gsv) continue;
                DevTerm Tw = d->h_terms[t];          // the term's column data from column `lo` of its block on
                Tw.xc += (lo - bc0) * Tw.ldc;
                if (Tw.cs) Tw.cs += (lo - bc0);
                M_RC(launch_grad_inputs(k.d_work2, 1, n_pad, d_alpha, d->row_off[I], d->row_len[I], lo, hi - lo, Tw, d->pair_dmax[p],
                                        2.0, gxa, k.s_upd, gsv));
              }
            }
          }
        if (dense_noise && grad_noise) {
          // the panel's columns of G = (alpha alpha' - C^-1) / 2, straight into the caller's N x N matrix (ld N)
          double* slab = k.d_work + (size_t)n_pad * std::max<long>(nco, 1);   // (the n_pad x W tail of the scratch)
          M_RC(launch_grad_noise_dense_cols(k.d_work2, n_pad, d_alpha, N, pc0, pw, slab, N, k.s_upd));
          M_HIP(hipMemcpyAsync(grad_noise + (size_t)pc0 * N, slab, sizeof(double) * N * pw, hipMemcpyDeviceToHost, k.s_upd));
        }
        M_RC(drv_copy_strided(k.d_work2 + pc0 + (size_t)pc0 * n_pad, n_pad + 1, pw, d_diag, k.s_upd));
        M_HIP(hipMemcpyAsync(kdiag.data() + pc0, d_diag, sizeof(double) * pw, hipMemcpyDeviceToHost, k.s_upd));
        M_HIP(hipStreamSynchronize(k.s_upd));   // (d_diag is reused by the next panel; kdiag is pageable)
      }
    }
    std::vector<double> tmpv(std::max<size_t>(1, nterms));
    for (int i = 0; i < P; ++i) {   // rank order: deterministic
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipStreamSynchronize(k.s_upd));
      if (nterms == 0) continue;
      double* d_gc = k.d_work2 + (size_t)n_pad * n_pad + n_pad;
      M_HIP(hipMemcpy(tmpv.data(), d_gc, sizeof(double) * nterms, hipMemcpyDeviceToHost));
      for (size_t t = 0; t < nterms; ++t) gc_sum[t] += tmpv[t];
      M_HIP(hipMemcpy(tmpv.data(), d_gc + 4096, sizeof(double) * nterms, hipMemcpyDeviceToHost));
      for (size_t t = 0; t < nterms; ++t) gs_sum[t] += tmpv[t];
    }
    // ---- input-point / row-scale gradients: per-rank sums added in rank order
    std::vector<double> hb;
    if (grad_inputs)
      for (int a = 0; a < spec->n_inputs; ++a) {
        const sgp_input& in = spec->inputs[a];
        if (!grad_inputs[a] || in.n == 0) continue;
        const size_t cnt = (size_t)(in.dim * in.n);
        std::fill(grad_inputs[a], grad_inputs[a] + cnt, 0.0);
        hb.resize(cnt);
        for (int i = 0; i < P; ++i) {
          M_HIP(hipSetDevice(m->r[i].dev));
          M_HIP(hipMemcpy(hb.data(), gx_dev[i][(size_t)a], sizeof(double) * cnt, hipMemcpyDeviceToHost));
          for (size_t q = 0; q < cnt; ++q) grad_inputs[a][q] += hb[q];
        }
      }
    if (grad_rowscale) {
      const sgp_dspec* d = ds[0];
      for (int I = 0; I < d->nrb; ++I)
        for (int Jb = 0; Jb < d->ncb; ++Jb)
          for (int t = d->term_ptr[I * d->ncb + Jb]; t < d->term_ptr[I * d->ncb + Jb + 1]; ++t) {
            if (!grad_rowscale[t] || !d->h_terms[t].rs || d->row_len[I] <= 0) continue;
            const size_t cnt = (size_t)d->row_len[I];
            std::fill(grad_rowscale[t], grad_rowscale[t] + cnt, 0.0);
            hb.resize(cnt);
            for (int i = 0; i < P; ++i) {
              M_HIP(hipSetDevice(m->r[i].dev));
              M_HIP(hipMemcpy(hb.data(), gr_dev[i][(size_t)t], sizeof(double) * cnt, hipMemcpyDeviceToHost));
              for (size_t q = 0; q < cnt; ++q) grad_rowscale[t][q] += hb[q];
            }
          }
    }
    return 0;
  };
  int rc = body();
  drain(m, ds, ctx->device);
  for (int i = 0; i < P; ++i) {
    hipSetDevice(m->r[i].dev);
    for (double* q : gx_dev[(size_t)i])
      if (q) hipFree(q);
    for (double* q : gr_dev[(size_t)i])
      if (q) hipFree(q);
  }
  hipSetDevice(ctx->device);
  sgp_multi_posterior_destroy(mp);
  if (rc) return rc;
  if (grad_y)
    for (long i = 0; i < N; ++i) grad_y[i] = -alpha[i];
  if (grad_mean)
    for (long i = 0; i < N; ++i) grad_mean[i] = alpha[i];
  if (grad_noise && !dense_noise) {
    if (noise_kind == SGP_NOISE_DIAG) {
      for (long i = 0; i < N; ++i) grad_noise[i] = 0.5 * (alpha[i] * alpha[i] - kdiag[i]);
    } else {
      double acc = 0.0;
      for (long i = 0; i < N; ++i) acc += 0.5 * (alpha[i] * alpha[i] - kdiag[i]);
      grad_noise[0] = acc;
    }
  }
  if (grad_coef)
    for (size_t t = 0; t < nterms; ++t) grad_coef[t] = gc_sum[t];
  if (grad_inscale)
    for (size_t t = 0; t < nterms; ++t) grad_inscale[t] = gs_sum[t];
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// cov(f, x) / cov(f, x, x') and var(f, x) on a multi-GPU context (round 6; they used to run on devices[0]): the assembly the
// sharded factorisation already does per rank, for its own sake.  kernelmatrix: the M columns in P contiguous tile-aligned
// chunks, rank i assembles chunk i for all rows and copies it straight into the caller's matrix (one host thread per rank: P
// PCIe links at once); no communication.  A symmetric spec keeps its guarantee of an EXACTLY symmetric result: a rank builds
// the lower trapezoid of its chunk, mirrors the diagonal square, and obtains the rows ABOVE its chunk's diagonal as the
// transpose of the row slab [chunk rows, columns left of the chunk] -- lower tiles it assembles itself (twice the kernel
// evaluations of the one-GPU path over the strictly upper part, still 1 / P of the matrix per rank, no exchange).
// kernelmatrix_diag: every block's points in P slices, the terms' device pointers advanced to the slice.
int sgp_multi_kernelmatrix(sgp_ctx* ctx, const sgp_cov_spec* spec, double* K, int64_t ldk) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  long N = 0, M = 0;
  for (int i = 0; i < spec->n_row_blocks; ++i) N += spec->row_len[i];
  for (int j = 0; j < spec->n_col_blocks; ++j) M += spec->col_len[j];
  M_CHECK_ARG(ldk >= N, "sgp_kernelmatrix (multi): ldk < N");
  if (N == 0 || M == 0) return 0;
  const bool sym = spec->symmetric != 0;
  const long T_r = rup(N, TILE) / TILE, T_c = rup(M, TILE) / TILE;
  const long per = (T_c + P - 1) / P;   // tile columns per rank
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto body = [&]() -> int {
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      const long tc0 = std::min<long>(T_c, (long)i * per), tc1 = std::min<long>(T_c, tc0 + per);
      if (tc0 >= tc1) continue;
      const long c0 = tc0 * TILE, wv = std::min(M, tc1 * TILE) - c0;
      M_HIP(hipSetDevice(k.dev));
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)N * (size_t)(tc1 - tc0) * TILE));
      if (sym && c0 > 0) M_RC(grow(&k.d_work2, &k.work2_cap, (size_t)(tc1 - tc0) * TILE * (size_t)c0));
      M_RC(drv_dspec_create(k.ctx, spec, &ds[i]));
      double* Kc = k.d_work;                      // element (r, chunk column c - c0) at Kc[r + (c - c0) * N]
      double* Kv = Kc - (size_t)c0 * N;            // ... i.e. global (r, c) at Kv[r + c * N]
      if (!sym) {
        M_RC(drv_assemble(ds[i], Kv, N, 0, T_r, tc0, tc1, 0, -1, 0.0, nullptr, k.s_upd));
      } else {
        M_RC(drv_assemble(ds[i], Kv, N, tc0, T_r, tc0, tc1, 1, -1, 0.0, nullptr, k.s_upd));
        M_RC(launch_mirror_lower(Kc + c0, N, wv, k.s_upd));
        if (c0 > 0) {
          const long ldS = (tc1 - tc0) * TILE;
          double* S = k.d_work2;                   // global (r, c), r in the chunk's rows, c < c0, at S[(r - c0) + c * ldS]
          M_RC(drv_assemble(ds[i], S - c0, ldS, tc0, tc1, 0, tc0, 1, -1, 0.0, nullptr, k.s_upd));
          M_RC(launch_transpose_add(S, ldS, wv, c0, Kc, N, nullptr, k.s_upd));
        }
      }
    }
    std::vector<std::thread> th;
    std::vector<int> rcs(P, 0);
    for (int i = 0; i < P; ++i) {
      const long tc0 = std::min<long>(T_c, (long)i * per), tc1 = std::min<long>(T_c, tc0 + per);
      if (tc0 >= tc1) continue;
      th.emplace_back([&, i, tc0, tc1]() {
        Rank& k = m->r[i];
        const long c0 = tc0 * TILE, wv = std::min(M, tc1 * TILE) - c0;
        if (hipSetDevice(k.dev) != hipSuccess || hipStreamSynchronize(k.s_upd) != hipSuccess ||
            hipMemcpy2D(K + (size_t)c0 * ldk, sizeof(double) * ldk, k.d_work, sizeof(double) * N, sizeof(double) * N, (size_t)wv,
                        hipMemcpyDeviceToHost) != hipSuccess)
          rcs[i] = -2;
      });
    }
    for (auto& t : th) t.join();
    for (int i = 0; i < P; ++i)
      if (rcs[i]) {
        set_error("sgp_kernelmatrix (multi): device -> host copy failed");
        return rcs[i];
      }
    return 0;
  };
  const int rc = body();
  drain(m, ds, ctx->device);
  return rc;
}

int sgp_multi_kernelmatrix_diag(sgp_ctx* ctx, const sgp_cov_spec* spec, double* out) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  M_CHECK_ARG(spec->n_row_blocks == spec->n_col_blocks, "kernelmatrix_diag: row / col block counts differ");
  long N = 0;
  for (int i = 0; i < spec->n_row_blocks; ++i) {
    M_CHECK_ARG(spec->row_len[i] == spec->col_len[i], "kernelmatrix_diag: block lengths differ");
    N += spec->row_len[i];
  }
  if (N == 0) return 0;
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto body = [&]() -> int {
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_RC(drv_dspec_create(k.ctx, spec, &ds[i]));
      const sgp_dspec* d = ds[i];
      const size_t nt = d->h_terms.size();
      const size_t term_doubles = (nt * sizeof(DevTerm) + 7) / 8 + 1;
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)N + term_doubles));
      std::vector<DevTerm> tw(d->h_terms);
      std::vector<std::pair<long, long>> sl((size_t)d->nrb);   // this rank's slice of every block
      for (int I = 0; I < d->nrb; ++I) {
        const long len = d->row_len[I], lo = len * i / P, hi = len * (i + 1) / P;
        sl[(size_t)I] = {lo, hi};
        const int p = I * d->ncb + I;
        for (int t = d->term_ptr[p]; t < d->term_ptr[p + 1]; ++t) {
          DevTerm& T = tw[(size_t)t];
          T.xr += lo * T.ldr;
          T.xc += lo * T.ldc;
          if (T.rs) T.rs += lo;
          if (T.cs) T.cs += lo;
        }
      }
      DevTerm* d_tw = reinterpret_cast<DevTerm*>(k.d_work + N);
      if (nt) M_HIP(hipMemcpyAsync(d_tw, tw.data(), nt * sizeof(DevTerm), hipMemcpyHostToDevice, k.s_upd));
      M_HIP(hipStreamSynchronize(k.s_upd));   // (tw is a local)
      for (int I = 0; I < d->nrb; ++I) {
        const long lo = sl[(size_t)I].first, hi = sl[(size_t)I].second;
        if (hi <= lo) continue;
        const int p = I * d->ncb + I;
        const int t0 = d->term_ptr[p], t1 = d->term_ptr[p + 1];
        M_RC(launch_diag_terms(k.d_work + d->row_off[I] + lo, hi - lo, d_tw + t0, t1 - t0, k.s_upd));
        M_HIP(hipMemcpyAsync(out + d->row_off[I] + lo, k.d_work + d->row_off[I] + lo, sizeof(double) * (hi - lo),
                             hipMemcpyDeviceToHost, k.s_upd));
      }
    }
    return 0;
  };
  const int rc = body();
  drain(m, ds, ctx->device);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------
int sgp_multi_factor_work(sgp_multi* m, double* executed, double* dense) {
  if (!m || m->r.empty()) {
    *executed = *dense = 0.0;
    return 0;
  }
  return sgp_ctx_factor_work(m->r[0].ctx, executed, dense);   // rank 0's context derived the pattern (factorize)
}

// elbo(VFE(fz), fx, y): data points sharded, one reduction (reference entry: src/gp/sparse_finite_gp.jl:52-58)
// ---------------------------------------------------------------------------------------------------------
namespace {

// rows [lo, hi) of a cross spec (rows = data blocks): new row lengths, fresh input entries for the sliced row
// inputs (an input may also serve as a column input somewhere: never edited in place), row scales offset
struct SlicedSpec {
  sgp_cov_spec c;
  std::vector<int64_t> row_len;
  std::vector<sgp_input> inputs;
  std::vector<sgp_term> terms;
  std::vector<long> row_a;                        // per row block: its first row inside the slice's view of the block
  std::vector<std::pair<int, long>> origin;       // per input of the sliced spec: (input of the whole spec, first point)
  SlicedSpec(const sgp_cov_spec* sp, long lo, long hi) {
    c = *sp;
    const int nrb = sp->n_row_blocks, ncb = sp->n_col_blocks;
    inputs.assign(sp->inputs, sp->inputs + sp->n_inputs);
    terms.assign(sp->terms, sp->terms + sp->term_ptr[nrb * ncb]);
    row_len.resize(nrb);
    row_a.resize(nrb);
    for (int k = 0; k < sp->n_inputs; ++k) origin.emplace_back(k, 0L);
    std::map<std::pair<int, int>, int> made;
    long off = 0;
    for (int I = 0; I < nrb; ++I) {
      const long n = sp->row_len[I];
      const long a = std::min(std::max(lo - off, 0L), n), e = std::min(std::max(hi - off, 0L), n);
      row_len[I] = e - a;
      row_a[I] = a;
      for (int J = 0; J < ncb; ++J) {
        const int p = I * ncb + J;
        for (int t = sp->term_ptr[p]; t < sp->term_ptr[p + 1]; ++t) {
          sgp_term& T = terms[t];
          auto key = std::make_pair((int)T.row_input, I);
          auto it = made.find(key);
          if (it == made.end()) {
            sgp_input in = sp->inputs[T.row_input];
            in.x = in.x + a * in.ld;
            in.n = e - a;
            inputs.push_back(in);
            origin.emplace_back((int)T.row_input, a);
            it = made.emplace(key, (int)inputs.size() - 1).first;
          }
          T.row_input = it->second;
          if (T.row_scale) T.row_scale = T.row_scale + a;
        }
      }
      off += n;
    }
    c.row_len = row_len.data();
    c.n_inputs = (int32_t)inputs.size();
    c.inputs = inputs.data();
    c.terms = terms.data();
  }
};

}  // namespace

int sgp_multi_vfe(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                  const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                  const double* z_noise, const double* y, double* h6, double* dLz, double* d_wz, double* d_part0,
                  double* d_wg) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  const long N = spec_rows(xz), M = spec_rows(zz);
  M_CHECK_ARG(N >= 1 && M >= 1, "sgp_elbo (multi): empty inputs");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
              "vfe: Sigma_y must be isotropic or diagonal (as in AbstractGPs.elbo)");
  int64_t len = 0;
  M_RC(sgp_elbo_part_len(M, &len));
  std::vector<double*> part(P, nullptr);
  for (int i = 0; i < P; ++i) {
    Rank& k = m->r[i];
    M_HIP(hipSetDevice(k.dev));
    if (i == 0 && d_part0) {
      part[i] = d_part0;
    } else {
      M_RC(grow(&k.d_work, &k.work_cap, (size_t)len));
      part[i] = k.d_work;
    }
    if (i == 0 && P > 1 && m->transport != TR_RCCL) M_RC(grow(&k.d_work2, &k.work2_cap, (size_t)len));
  }
  // every rank's slice on a host thread of its own (the pipeline synchronises inside: Lz's info, chunk buffers)
  std::vector<int> rcs(P, 0);
  std::vector<std::string> errs(P);
  std::vector<std::thread> th;
  const double t0 = now_ms();
  for (int i = 0; i < P; ++i) {
    th.emplace_back([&, i]() {
      const long lo = N * i / P, hi = N * (i + 1) / P;
      SlicedSpec sl(xz, lo, hi);
      Rank& k = m->r[i];
      hipSetDevice(k.dev);
      rcs[i] = drv_vfe_partial(k.ctx, zz, &sl.c, var_x ? var_x + lo : nullptr, mean_x ? mean_x + lo : nullptr, noise_kind,
                               noise_kind == SGP_NOISE_DIAG ? noise_x + lo : noise_x, z_noise_kind, z_noise, y + lo,
                               i == 0 ? dLz : nullptr, i == 0 ? d_wz : nullptr, part[i], len);
      if (rcs[i]) errs[i] = sgp_last_error();
    });
  }
  for (auto& t : th) t.join();
  for (int i = 0; i < P; ++i)
    if (rcs[i]) {
      set_error(errs[i]);
      hipSetDevice(ctx->device);
      return rcs[i];
    }
  // ---- ONE reduction of the parts
  if (P > 1) {
    if (m->transport == TR_RCCL) {
      int rc = m->rccl.GroupStart();
      for (int i = 0; i < P && rc == 0; ++i) {
        Rank& k = m->r[i];
        hipSetDevice(k.dev);
        rc = m->rccl.AllReduce(part[i], part[i], (size_t)len, NCCL_DOUBLE, NCCL_SUM, k.comm, k.s_upd);
      }
      int rc2 = m->rccl.GroupEnd();
      if (rc || rc2) {
        set_error("ncclAllReduce failed (elbo parts)");
        hipSetDevice(ctx->device);
        return -4;
      }
      for (auto& k : m->r) {
        M_HIP(hipSetDevice(k.dev));
        M_HIP(hipStreamSynchronize(k.s_upd));
      }
    } else {
      Rank& k0 = m->r[0];
      for (int i = 1; i < P; ++i) {   // rank order: deterministic
        Rank& k = m->r[i];
        M_HIP(hipSetDevice(k0.dev));
        if (k.dev == k0.dev)
          M_HIP(hipMemcpyAsync(k0.d_work2, part[i], sizeof(double) * len, hipMemcpyDeviceToDevice, k0.s_upd));
        else
          M_HIP(hipMemcpyPeerAsync(k0.d_work2, k0.dev, part[i], k.dev, sizeof(double) * len, k0.s_upd));
        M_RC(drv_axpy_block(part[0], len, k0.d_work2, len, len, 1, 1.0, k0.s_upd));
      }
      M_HIP(hipStreamSynchronize(k0.s_upd));
    }
  }
  int rc = drv_vfe_finish(m->r[0].ctx, M, part[0], d_wg, h6);
  m->last_ms = now_ms() - t0;
  hipSetDevice(ctx->device);
  return rc;
}

// elbo + its gradient (sgp_elbo_grad / _x / _xs; formulas: capi.hip elbo_grad_core): the data points are sharded as in
// sgp_multi_vfe, every rank runs the whole pipeline on its slice on a host thread of its own, and the sums over data points
// (A A', A delta, four scalars) meet in ONE reduction between the two factorisations -- the second factorisation and the
// M x M stage then run replicated on identical numbers.  Per-point results (d/dy, d/dmean, a diagonal d/dSigma_y, the x
// points, the function scales at x) land in the caller's arrays slice by slice; sums over data points (kernel parameters of
// the xz spec, the z points, the scales at z, an isotropic d/dsigma^2) are added on the host in rank order; the zz side is
// rank 0's.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct PartBarrier {   // the ranks' host threads meet here; the last one in performs the reduction for all
  std::mutex mu;
  std::condition_variable cv;
  int P = 0, arrived = 0;
  long gen = 0;
  bool failed = false;
  int rc = 0;
  std::vector<double*> part;
  long len = 0;
  std::function<int()> work;
  int arrive(int i, double* d_part, long n) {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return -5;
    part[i] = d_part;
    len = n;
    if (++arrived == P) {
      rc = work();
      arrived = 0;
      ++gen;
      cv.notify_all();
      return rc;
    }
    const long g = gen;
    cv.wait(lk, [&] { return gen != g || failed; });
    return gen != g ? rc : -5;
  }
  void fail() {   // a rank left the pipeline early: nobody may wait for it
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
};

}  // namespace

int sgp_multi_elbo_grad(sgp_ctx* ctx, const ElboGradArgs& a) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  M_CHECK_ARG(a.zz && a.xz && a.var_x && a.noise_x && a.z_noise && a.y && a.elbo_out, "sgp_elbo_grad: NULL argument");
  M_CHECK_ARG(a.noise_kind == SGP_NOISE_SCALAR || a.noise_kind == SGP_NOISE_DIAG,
              "sgp_elbo_grad: Sigma_y must be isotropic or diagonal (as in AbstractGPs.elbo)");
  const long N = spec_rows(a.xz);
  if (P == 1 || N < (long)P * TILE) return drv_elbo_grad(m->r[0].ctx, a, nullptr);   // nothing to shard
  const sgp_cov_spec* xz = a.xz;
  const int nrb = xz->n_row_blocks, ncb = xz->n_col_blocks;
  const int nt = xz->term_ptr[nrb * ncb];
  const bool diag = a.noise_kind == SGP_NOISE_DIAG;
  // ---- per rank: the slice, its arguments, host buffers for what is summed over the ranks
  struct PerRank {
    std::unique_ptr<SlicedSpec> sl;
    ElboGradArgs b;
    double elbo = 0.0, noise_sum = 0.0;
    std::vector<double> coef, inscale;
    std::vector<std::vector<double>> in_buf, cs_buf;
    std::vector<double*> in_ptr, rs_ptr, cs_ptr;
  };
  std::vector<PerRank> pr(P);
  for (int i = 0; i < P; ++i) {
    PerRank& q = pr[i];
    const long lo = N * i / P, hi = N * (i + 1) / P;
    q.sl.reset(new SlicedSpec(xz, lo, hi));
    const sgp_cov_spec& c = q.sl->c;
    ElboGradArgs& b = q.b;
    b = a;
    b.xz = &c;
    b.var_x = a.var_x + lo;
    b.mean_x = a.mean_x ? a.mean_x + lo : nullptr;
    b.noise_x = diag ? a.noise_x + lo : a.noise_x;
    b.y = a.y + lo;
    b.elbo_out = i == 0 ? a.elbo_out : &q.elbo;
    b.grad_y = a.grad_y ? a.grad_y + lo : nullptr;
    b.grad_mean = a.grad_mean ? a.grad_mean + lo : nullptr;
    b.grad_var_x = a.grad_var_x ? a.grad_var_x + lo : nullptr;
    b.grad_noise = a.grad_noise ? (diag ? a.grad_noise + lo : &q.noise_sum) : nullptr;
    if (i != 0) {   // the zz side: rank 0
      b.grad_z_noise = b.grad_coef_zz = b.grad_inscale_zz = nullptr;
      b.grad_inputs_zz = nullptr;
      b.grad_rowscale_zz = nullptr;
    }
    if (a.grad_coef_xz) q.coef.assign((size_t)std::max(1, nt), 0.0), b.grad_coef_xz = q.coef.data();
    if (a.grad_inscale_xz) q.inscale.assign((size_t)std::max(1, nt), 0.0), b.grad_inscale_xz = q.inscale.data();
    if (a.grad_inputs_xz) {
      q.in_buf.resize((size_t)c.n_inputs);
      q.in_ptr.assign((size_t)c.n_inputs, nullptr);
      for (int k = 0; k < c.n_inputs; ++k) {
        if (!a.grad_inputs_xz[q.sl->origin[(size_t)k].first] || c.inputs[k].n <= 0) continue;
        q.in_buf[(size_t)k].assign((size_t)(c.inputs[k].dim * c.inputs[k].n), 0.0);
        q.in_ptr[(size_t)k] = q.in_buf[(size_t)k].data();
      }
      b.grad_inputs_xz = q.in_ptr.data();
    }
    if (a.grad_rowscale_xz || a.grad_colscale_xz) {
      q.rs_ptr.assign((size_t)std::max(1, nt), nullptr);
      q.cs_ptr.assign((size_t)std::max(1, nt), nullptr);
      q.cs_buf.resize((size_t)std::max(1, nt));
      for (int I = 0; I < nrb; ++I)
        for (int J = 0; J < ncb; ++J)
          for (int t = xz->term_ptr[I * ncb + J]; t < xz->term_ptr[I * ncb + J + 1]; ++t) {
            if (a.grad_rowscale_xz && a.grad_rowscale_xz[t]) q.rs_ptr[(size_t)t] = a.grad_rowscale_xz[t] + q.sl->row_a[(size_t)I];
            if (a.grad_colscale_xz && a.grad_colscale_xz[t]) {
              q.cs_buf[(size_t)t].assign((size_t)std::max<long>(1, (long)xz->col_len[J]), 0.0);
              q.cs_ptr[(size_t)t] = q.cs_buf[(size_t)t].data();
            }
          }
      b.grad_rowscale_xz = a.grad_rowscale_xz ? q.rs_ptr.data() : nullptr;
      b.grad_colscale_xz = a.grad_colscale_xz ? q.cs_ptr.data() : nullptr;
    }
  }
  // ---- the reduction the ranks meet in
  PartBarrier bar;
  bar.P = P;
  bar.part.assign((size_t)P, nullptr);
  bar.work = [&]() -> int {   // (runs on the host thread of the last rank to arrive, every part complete)
    const long len = bar.len;
    if (m->transport == TR_RCCL) {
      int rc = m->rccl.GroupStart();
      for (int i = 0; i < P && rc == 0; ++i) {
        Rank& k = m->r[i];
        hipSetDevice(k.dev);
        rc = m->rccl.AllReduce(bar.part[(size_t)i], bar.part[(size_t)i], (size_t)len, NCCL_DOUBLE, NCCL_SUM, k.comm, k.s_upd);
      }
      const int rc2 = m->rccl.GroupEnd();
      if (rc || rc2) {
        set_error("ncclAllReduce failed (elbo gradient parts)");
        return -4;
      }
      for (auto& k : m->r) {
        M_HIP(hipSetDevice(k.dev));
        M_HIP(hipStreamSynchronize(k.s_upd));
      }
      return 0;
    }
    Rank& k0 = m->r[0];
    M_HIP(hipSetDevice(k0.dev));
    M_RC(grow(&k0.d_work2, &k0.work2_cap, (size_t)len));
    for (int i = 1; i < P; ++i) {   // rank order: deterministic
      Rank& k = m->r[i];
      if (k.dev == k0.dev)
        M_HIP(hipMemcpyAsync(k0.d_work2, bar.part[(size_t)i], sizeof(double) * len, hipMemcpyDeviceToDevice, k0.s_upd));
      else
        M_HIP(hipMemcpyPeerAsync(k0.d_work2, k0.dev, bar.part[(size_t)i], k.dev, sizeof(double) * len, k0.s_upd));
      M_RC(drv_axpy_block(bar.part[0], len, k0.d_work2, len, len, 1, 1.0, k0.s_upd));
    }
    for (int i = 1; i < P; ++i) {   // and back: every rank continues on the same numbers
      Rank& k = m->r[i];
      if (k.dev == k0.dev)
        M_HIP(hipMemcpyAsync(bar.part[(size_t)i], bar.part[0], sizeof(double) * len, hipMemcpyDeviceToDevice, k0.s_upd));
      else
        M_HIP(hipMemcpyPeerAsync(bar.part[(size_t)i], k.dev, bar.part[0], k0.dev, sizeof(double) * len, k0.s_upd));
    }
    M_HIP(hipStreamSynchronize(k0.s_upd));
    return 0;
  };
  std::vector<int> rcs(P, 0);
  std::vector<std::string> errs(P);
  const double t0 = now_ms();
  auto run_all = [&]() {
    bar.failed = false;
    bar.arrived = 0;
    std::vector<std::thread> th;
    for (int i = 0; i < P; ++i) {
      th.emplace_back([&, i]() {
        Rank& k = m->r[i];
        hipSetDevice(k.dev);
        ElboGradShard sh;
        sh.primary = i == 0;
        sh.n_total = N;
        sh.reduce = [&, i](double* d_part, long len) -> int {
          const int rc = bar.arrive(i, d_part, len);
          hipSetDevice(m->r[i].dev);   // (the reduction visits every rank's device on this thread)
          if (rc == -5) set_error("sgp_elbo_grad (multi): another rank failed before the reduction");
          return rc;
        };
        rcs[i] = drv_elbo_grad(k.ctx, pr[i].b, &sh);
        if (rcs[i]) {
          errs[i] = sgp_last_error();
          bar.fail();
        }
      });
    }
    for (auto& t : th) t.join();
  };
  run_all();
  {   // a dataflow launch that timed out on some rank: once more, every rank on the launch-based schedule (capi.hip:
      // with_df_fallback -- a rank cannot rerun on its own, the reduction is collective)
    bool timed_out = false;
    for (int i = 0; i < P; ++i) timed_out = timed_out || (rcs[i] == -3 && m->r[i].ctx->df_timed_out);
    if (timed_out) {
      std::vector<std::pair<int, int>> keep;
      for (auto& k : m->r) {
        keep.emplace_back(k.ctx->dataflow, k.ctx->hybrid);
        k.ctx->dataflow = k.ctx->hybrid = 0;
        k.ctx->df_timed_out = false;
        k.ctx->df_fallbacks += 1;
      }
      run_all();
      for (int i = 0; i < P; ++i) m->r[i].ctx->dataflow = keep[(size_t)i].first, m->r[i].ctx->hybrid = keep[(size_t)i].second;
    }
  }
  hipSetDevice(ctx->device);
  {   // the root cause first: a rank that merely saw another one fail (-5) is not it
    int first = -1;
    for (int i = 0; i < P; ++i)
      if (rcs[i] && rcs[i] != -5 && first < 0) first = i;
    for (int i = 0; i < P && first < 0; ++i)
      if (rcs[i]) first = i;
    if (first >= 0) {
      set_error(errs[(size_t)first]);
      return rcs[(size_t)first];
    }
  }
  // ---- what is summed over the ranks, in rank order
  if (a.grad_noise && !diag) {
    double acc = 0.0;
    for (int i = 0; i < P; ++i) acc += pr[i].noise_sum;
    a.grad_noise[0] = acc;
  }
  for (int t = 0; t < nt; ++t) {
    if (a.grad_coef_xz) {
      double acc = 0.0;
      for (int i = 0; i < P; ++i) acc += pr[i].coef[(size_t)t];
      a.grad_coef_xz[t] = acc;
    }
    if (a.grad_inscale_xz) {
      double acc = 0.0;
      for (int i = 0; i < P; ++i) acc += pr[i].inscale[(size_t)t];
      a.grad_inscale_xz[t] = acc;
    }
  }
  if (a.grad_inputs_xz) {
    for (int k = 0; k < xz->n_inputs; ++k)
      if (a.grad_inputs_xz[k] && xz->inputs[k].n > 0)
        std::fill(a.grad_inputs_xz[k], a.grad_inputs_xz[k] + xz->inputs[k].dim * xz->inputs[k].n, 0.0);
    for (int i = 0; i < P; ++i) {
      const SlicedSpec& sl = *pr[i].sl;
      for (int k = 0; k < sl.c.n_inputs; ++k) {
        const std::vector<double>& src = pr[i].in_buf[(size_t)k];
        if (src.empty()) continue;
        double* dst = a.grad_inputs_xz[sl.origin[(size_t)k].first] + sl.origin[(size_t)k].second * sl.c.inputs[k].dim;
        for (size_t e = 0; e < src.size(); ++e) dst[e] += src[e];
      }
    }
  }
  if (a.grad_colscale_xz)
    for (int I = 0; I < nrb; ++I)
      for (int J = 0; J < ncb; ++J)
        for (int t = xz->term_ptr[I * ncb + J]; t < xz->term_ptr[I * ncb + J + 1]; ++t) {
          if (!a.grad_colscale_xz[t] || !xz->terms[t].col_scale) continue;
          for (long e = 0; e < (long)xz->col_len[J]; ++e) {
            double acc = 0.0;
            for (int i = 0; i < P; ++i) acc += pr[i].cs_buf[(size_t)t][(size_t)e];
            a.grad_colscale_xz[t][e] = acc;
          }
        }
  m->last_ms = now_ms() - t0;
  return 0;
}
