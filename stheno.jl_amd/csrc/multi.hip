// In-process multi-GPU logpdf behind the C-ABI: sgp_ctx_create_multi(devices, ndev, &ctx) returns an
// ordinary sgp_ctx whose sgp_logpdf shards the N x N covariance over the listed GPUs,
// so that ONE `ccall` from the Julia host reaches the whole node.  The reference has no distributed
// code; the contract is BASELINE.json's north_star.
//
// Layout: outer column panels of W columns, block-cyclic over the ranks (= devices); every rank keeps
// its panels PACKED (panel J holds rows J0 .. m_tot only, leading dimension m_tot - J0), so a factored
// panel is one contiguous block that travels as is -- no packing copy.  Right-looking blocked Cholesky
// with one-panel look-ahead, driven by one host thread that only enqueues (three streams per rank:
// trailing updates / panel factorisation at high priority / panel receive):
//   step J:  every rank updates its panels > J + 1 with panel J (update stream) while the owner of
//            panel J + 1 updates + factors it (panel stream) and the transport moves it to the others'
//            receive buffers (double-buffered), overlapping the updates of step J.
// Transport (SGP_MULTI_TRANSPORT=rccl|p2p|auto): RCCL ncclBroadcast in one group call per panel over
// communicators from ncclCommInitAll (xGMI rings; librccl is dlopen'ed here, not linked), or plain
// peer copies (hipMemcpyPeerAsync: xGMI point to point, one link per receiver).  A device listed more
// than once gives several ranks on one GPU ("loopback": same-device copies) -- that is how the 1-GPU
// test box exercises the multi-rank orchestration with the real kernels.
// Scalars (logdet, |L^-1 (y - m)|^2) are all-reduced (ncclAllReduce) or summed on the host.
#include "ctx.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

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
    return CommInitAll && CommDestroy && Broadcast && AllReduce && GroupStart && GroupEnd;
  }
};
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;  // rccl.h: ncclFloat64 = 8, ncclSum = 0

enum { TR_LOOPBACK = 0, TR_P2P = 1, TR_RCCL = 2 };

struct Rank {
  int dev = 0;
  sgp_ctx* ctx = nullptr;        // child context (kernels + scratch of this rank)
  hipStream_t s_upd = nullptr, s_panel = nullptr, s_comm = nullptr;
  // A rank's trailing panels are separate (packed) matrices, one update launch each; issued round-robin
  // on a small pool of streams the tail of one launch overlaps the head of the next (a panel always uses
  // the same pool stream, so its successive updates stay ordered).
  static constexpr int NPOOL = 3;
  hipStream_t s_pool[NPOOL] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_pool[NPOOL] = {nullptr, nullptr, nullptr}, ev_fork = nullptr;
  hipEvent_t ev_upd = nullptr, ev_fact = nullptr, ev_recv[2] = {nullptr, nullptr}, ev_done = nullptr;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // timing events: first / last trailing update of a call (sgp_ctx_multi_stats)
  double upd_flops = 0.0, upd_span_ms = 0.0, recv_bytes = 0.0;
  long n_factored = 0;
  bool factored_once = false;
  double* store = nullptr;       // owned panels, packed
  size_t store_cap = 0;
  double* buf[2] = {nullptr, nullptr};
  size_t buf_cap = 0;
  double* d_small = nullptr;     // y (N) | mean (N) | noise diag (N) | scal: logdet, sq
  size_t small_cap = 0;
  int* d_info = nullptr;
  ncclComm_p comm = nullptr;
  std::vector<size_t> off;       // offset of local panel i in store
};

}  // namespace

struct sgp_multi {
  std::vector<Rank> r;
  int transport = TR_LOOPBACK;
  long W = 1024;
  Rccl rccl;
  double last_ms = 0.0;
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

struct Geometry {
  long N, n_pad, m_tot, W, npan, P;
  long col0(long J) const { return J * W; }
  long width(long J) const { return std::min(W, n_pad - J * W); }
  long ldp(long J) const { return m_tot - J * W; }          // packed leading dimension of panel J
  int owner(long J) const { return (int)(J % P); }
};

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
    if (k.d_info) hipFree(k.d_info);
    for (hipEvent_t e : {k.ev_upd, k.ev_fact, k.ev_recv[0], k.ev_recv[1], k.ev_done, k.ev_fork, k.ev_pool[0], k.ev_pool[1],
                         k.ev_pool[2], k.ev_t0, k.ev_t1})
      if (e) hipEventDestroy(e);
    for (auto st : k.s_pool)
      if (st) {
        hipStreamSynchronize(st);
        hipStreamDestroy(st);
      }
    if (k.s_comm) hipStreamDestroy(k.s_comm);
    if (k.ctx) sgp_ctx_destroy(k.ctx);   // owns s_upd / s_panel
  }
  delete m;
}

extern "C" int sgp_ctx_create_multi(const int* devices, int ndev, sgp_ctx** out) {
  M_CHECK_ARG(devices && out && ndev >= 1 && ndev <= 64, "sgp_ctx_create_multi: bad argument");
  sgp_ctx* primary = nullptr;
  M_RC(sgp_ctx_create(devices[0], &primary));
  sgp_multi* m = new sgp_multi();
  primary->multi = m;
  auto fail = [&](int rc) {
    sgp_ctx_destroy(primary);   // destroys m as well
    return rc;
  };
  bool distinct = true;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devices[i] == devices[j]) distinct = false;
  const char* tr = getenv("SGP_MULTI_TRANSPORT");
  std::string want = tr ? tr : "auto";
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
  const char* pw = getenv("SGP_MULTI_PANEL");
  if (pw && atol(pw) >= TILE) m->W = atol(pw) / TILE * TILE;
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
    for (hipEvent_t* e : {&k.ev_upd, &k.ev_fact, &k.ev_recv[0], &k.ev_recv[1], &k.ev_done, &k.ev_fork, &k.ev_pool[0],
                          &k.ev_pool[1], &k.ev_pool[2]})
      if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return fail(-2);
    for (auto& st : k.s_pool)
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail(-2);
    if (hipEventCreate(&k.ev_t0) != hipSuccess || hipEventCreate(&k.ev_t1) != hipSuccess) return fail(-2);
    if (hipMalloc(&k.d_info, sizeof(int)) != hipSuccess) return fail(-2);
  }
  if (m->transport == TR_P2P || m->transport == TR_RCCL) {
    for (int i = 0; i < ndev; ++i)
      for (int j = 0; j < ndev; ++j) {
        if (i == j) continue;
        int can = 0;
        hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
        if (can) {
          hipSetDevice(devices[i]);
          hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
          if (e != hipSuccess) (void)hipGetLastError();   // already enabled is fine
        }
      }
  }
  if (m->transport == TR_RCCL) {
    std::vector<ncclComm_p> comms(ndev, nullptr);
    int rc = m->rccl.CommInitAll(comms.data(), ndev, devices);
    if (rc != 0) {
      set_error(std::string("ncclCommInitAll failed: ") +
                (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "?"));
      return fail(-4);
    }
    for (int i = 0; i < ndev; ++i) m->r[i].comm = comms[i];
  }
  hipSetDevice(devices[0]);
  *out = primary;
  return 0;
}

extern "C" const char* sgp_ctx_transport(sgp_ctx* ctx) {
  if (!ctx || !ctx->multi) return "single";
  switch (ctx->multi->transport) {
    case TR_RCCL: return "rccl";
    case TR_P2P: return "p2p";
    default: return "loopback";
  }
}

namespace {

// move factored panel J from its owner to every other rank's receive buffer (J % 2)
int broadcast_panel(sgp_multi* m, const Geometry& g, long J) {
  const int o = g.owner(J);
  const size_t count = (size_t)g.ldp(J) * g.width(J);
  const int P = (int)m->r.size();
  if (P == 1 && m->transport != TR_RCCL) return 0;
  Rank& root = m->r[o];
  double* src = root.store + root.off[J / g.P];
  if (m->transport == TR_RCCL) {
    for (int i = 0; i < P; ++i) {   // order the communicator streams behind the data / the buffer's readers
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      if (i == o) {
        M_HIP(hipStreamWaitEvent(k.s_comm, k.ev_fact, 0));
      } else {
        M_HIP(hipStreamWaitEvent(k.s_comm, k.ev_upd, 0));
        if (k.factored_once) M_HIP(hipStreamWaitEvent(k.s_comm, k.ev_fact, 0));
      }
    }
    int rc = m->rccl.GroupStart();
    for (int i = 0; i < P && rc == 0; ++i) {
      Rank& k = m->r[i];
      hipSetDevice(k.dev);
      void* recv = (i == o) ? (void*)src : (void*)k.buf[J % 2];
      rc = m->rccl.Broadcast((i == o) ? (const void*)src : (const void*)k.buf[J % 2], recv, count, NCCL_DOUBLE, o,
                             k.comm, k.s_comm);
    }
    int rc2 = m->rccl.GroupEnd();
    if (rc || rc2) {
      set_error(std::string("ncclBroadcast failed: ") +
                (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc ? rc : rc2) : "?"));
      return -4;
    }
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipEventRecord(k.ev_recv[J % 2], k.s_comm));
    }
    return 0;
  }
  for (int i = 0; i < P; ++i) {
    if (i == o) continue;
    Rank& k = m->r[i];
    M_HIP(hipSetDevice(k.dev));
    M_HIP(hipStreamWaitEvent(k.s_comm, k.ev_upd, 0));                         // readers of this buffer two panels ago
    if (k.factored_once) M_HIP(hipStreamWaitEvent(k.s_comm, k.ev_fact, 0));   // ... on the panel stream as well
    M_HIP(hipStreamWaitEvent(k.s_comm, root.ev_fact, 0));                     // the panel is factored
    if (k.dev == root.dev)
      M_HIP(hipMemcpyAsync(k.buf[J % 2], src, sizeof(double) * count, hipMemcpyDeviceToDevice, k.s_comm));
    else
      M_HIP(hipMemcpyPeerAsync(k.buf[J % 2], k.dev, src, root.dev, sizeof(double) * count, k.s_comm));
    M_HIP(hipEventRecord(k.ev_recv[J % 2], k.s_comm));
  }
  return 0;
}

// panel J as seen by rank i (pointer to its row J0, leading dimension ldp(J))
const double* panel_on(sgp_multi* m, const Geometry& g, long J, int i) {
  Rank& k = m->r[i];
  return (g.owner(J) == i) ? k.store + k.off[J / g.P] : k.buf[J % 2];
}

int wait_panel(sgp_multi* m, const Geometry& g, long J, int i, hipStream_t s) {
  Rank& k = m->r[i];
  if (g.owner(J) == i) return hipStreamWaitEvent(s, k.ev_fact, 0) == hipSuccess ? 0 : -2;
  return hipStreamWaitEvent(s, k.ev_recv[J % 2], 0) == hipSuccess ? 0 : -2;
}

int update_panel(sgp_multi* m, const Geometry& g, long J, long Jp, int i, hipStream_t s) {
  Rank& k = m->r[i];
  const double* Pj = panel_on(m, g, J, i);
  double* C = k.store + k.off[Jp / g.P];              // row Jp0 of panel Jp
  const long c0 = g.col0(Jp);
  // sgp_dev_panel_update indexes C by global row: hand it the (virtual) address of global row 0
  return sgp_dev_panel_update(k.ctx, Pj, g.ldp(J), g.col0(J), g.width(J), C - c0, g.ldp(Jp), c0, g.width(Jp),
                              g.m_tot, (void*)s);
}

}  // namespace

// logpdf(fx, y) over the ranks of ctx->multi (called from sgp_logpdf with the primary context held)
int sgp_multi_logpdf(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                     const double* noise, const double* y, double* out) {
  sgp_multi* m = ctx->multi;
  const int P = (int)m->r.size();
  long N = 0;
  for (int i = 0; i < spec->n_row_blocks; ++i) N += spec->row_len[i];
  M_CHECK_ARG(N >= 1, "sgp_logpdf (multi): empty data");
  M_CHECK_ARG(noise_kind == SGP_NOISE_SCALAR || noise_kind == SGP_NOISE_DIAG,
              "sgp_logpdf (multi): noise kind must be SCALAR or DIAG");
  int64_t n_pad, m_tot;
  sgp_geometry(N, 1, &n_pad, &m_tot);
  Geometry g;
  g.N = N;
  g.n_pad = n_pad;
  g.m_tot = m_tot;
  g.W = std::min<long>(m->W, n_pad);
  g.npan = (n_pad + g.W - 1) / g.W;
  g.P = P;
  const double s2 = noise_kind == SGP_NOISE_SCALAR ? noise[0] : 0.0;
  std::vector<sgp_dspec*> ds(P, nullptr);
  auto cleanup = [&]() {
    for (int i = 0; i < P; ++i) {
      hipSetDevice(m->r[i].dev);
      hipStreamSynchronize(m->r[i].s_upd);
      hipStreamSynchronize(m->r[i].s_panel);
      hipStreamSynchronize(m->r[i].s_comm);
      for (auto st : m->r[i].s_pool) hipStreamSynchronize(st);
      if (ds[i]) sgp_dspec_destroy(ds[i]);
    }
    hipSetDevice(ctx->device);
  };
  auto body = [&]() -> int {
    // ---- per-rank storage and inputs
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      k.off.clear();
      size_t tot = 0;
      for (long J = i; J < g.npan; J += P) {
        k.off.push_back(tot);
        tot += (size_t)g.ldp(J) * g.width(J);
      }
      M_RC(grow(&k.store, &k.store_cap, std::max<size_t>(tot, 1)));
      if (P > 1 || m->transport == TR_RCCL) {
        size_t bc = (size_t)m_tot * g.W;
        if (bc > k.buf_cap) {
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
      M_RC(grow(&k.d_small, &k.small_cap, (size_t)3 * N + 8));
      k.factored_once = false;
      M_RC(sgp_dspec_create(k.ctx, spec, &ds[i]));
      double* dY = k.d_small;
      double* dM = k.d_small + N;
      double* dNz = k.d_small + 2 * N;
      double* dSc = k.d_small + 3 * N;
      M_HIP(hipMemcpyAsync(dY, y, sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
      if (mean) M_HIP(hipMemcpyAsync(dM, mean, sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
      if (noise_kind == SGP_NOISE_DIAG)
        M_HIP(hipMemcpyAsync(dNz, noise, sizeof(double) * N, hipMemcpyHostToDevice, k.s_upd));
      M_HIP(hipMemsetAsync(dSc, 0, sizeof(double) * 8, k.s_upd));
      M_HIP(hipMemsetAsync(k.d_info, 0, sizeof(int), k.s_upd));
      // ---- assembly of the owned panels: no communication
      for (long J = i; J < g.npan; J += P) {
        double* base = k.store + k.off[J / P];
        M_RC(sgp_dev_assemble_cols(k.ctx, ds[i], N, g.col0(J), g.width(J), base - g.col0(J), g.ldp(J), m_tot,
                                   mean ? dM : nullptr, noise_kind, &s2, noise_kind == SGP_NOISE_DIAG ? dNz : nullptr,
                                   dY, N, 1, (void*)k.s_upd));
      }
      M_HIP(hipEventRecord(k.ev_upd, k.s_upd));
    }
    auto factor = [&](long J) -> int {
      const int o = g.owner(J);
      Rank& k = m->r[o];
      M_HIP(hipSetDevice(k.dev));
      double* base = k.store + k.off[J / P];
      M_RC(sgp_dev_panel_factor(k.ctx, base, g.ldp(J), g.ldp(J), g.width(J), g.col0(J), k.d_small + 3 * N, k.d_info,
                                (void*)k.s_panel));
      M_HIP(hipEventRecord(k.ev_fact, k.s_panel));
      k.factored_once = true;
      return 0;
    };
    // ---- panel 0
    {
      Rank& k = m->r[g.owner(0)];
      M_HIP(hipSetDevice(k.dev));
      M_HIP(hipStreamWaitEvent(k.s_panel, k.ev_upd, 0));
      M_RC(factor(0));
      M_RC(broadcast_panel(m, g, 0));
    }
    // ---- right-looking sweep with one-panel look-ahead
    for (long J = 0; J < g.npan; ++J) {
      const long nxt = J + 1;
      for (int i = 0; i < P; ++i) {   // (a) update streams need panel J
        M_HIP(hipSetDevice(m->r[i].dev));
        M_RC(wait_panel(m, g, J, i, m->r[i].s_upd));
      }
      if (nxt < g.npan) {             // (b) look-ahead on the owner of the next panel
        const int o1 = g.owner(nxt);
        Rank& k = m->r[o1];
        M_HIP(hipSetDevice(k.dev));
        M_HIP(hipStreamWaitEvent(k.s_panel, k.ev_upd, 0));      // step J - 1's updates of panel nxt
        M_RC(wait_panel(m, g, J, o1, k.s_panel));
        M_RC(update_panel(m, g, J, nxt, o1, k.s_panel));
        M_RC(factor(nxt));
        M_RC(broadcast_panel(m, g, nxt));
      }
      for (int i = 0; i < P; ++i) {   // (c) the rest of every rank's trailing panels, fanned over the stream pool
        Rank& k = m->r[i];
        M_HIP(hipSetDevice(k.dev));
        M_HIP(hipEventRecord(k.ev_fork, k.s_upd));
        for (auto st : k.s_pool) M_HIP(hipStreamWaitEvent(st, k.ev_fork, 0));
        for (long Jp = i; Jp < g.npan; Jp += P)
          if (Jp > nxt) M_RC(update_panel(m, g, J, Jp, i, k.s_pool[(Jp / P) % Rank::NPOOL]));
        for (int q = 0; q < Rank::NPOOL; ++q) {
          M_HIP(hipEventRecord(k.ev_pool[q], k.s_pool[q]));
          M_HIP(hipStreamWaitEvent(k.s_upd, k.ev_pool[q], 0));
        }
        M_HIP(hipEventRecord(k.ev_upd, k.s_upd));
      }
    }
    // ---- scalars: |L^-1 (y - m)|^2 from the bordered row of the owned panels, logdet
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      if (k.factored_once) M_HIP(hipStreamWaitEvent(k.s_upd, k.ev_fact, 0));
      for (long J = i; J < g.npan; J += P) {
        long nc = std::min(g.width(J), std::max<long>(0, N - g.col0(J)));
        if (nc > 0)
          M_RC(sgp_dev_rowsumsq(k.ctx, k.store + k.off[J / P] + (n_pad - g.col0(J)), g.ldp(J), nc, 1,
                                k.d_small + 3 * N + 1, (void*)k.s_upd));
      }
    }
    double red[2] = {0.0, 0.0};
    if (m->transport == TR_RCCL) {
      int rc = m->rccl.GroupStart();
      for (int i = 0; i < P && rc == 0; ++i) {
        Rank& k = m->r[i];
        hipSetDevice(k.dev);
        double* sc = k.d_small + 3 * N;
        rc = m->rccl.AllReduce(sc, sc + 4, 2, NCCL_DOUBLE, NCCL_SUM, k.comm, k.s_upd);
      }
      int rc2 = m->rccl.GroupEnd();
      if (rc || rc2) {
        set_error("ncclAllReduce failed");
        return -4;
      }
      Rank& k0 = m->r[0];
      M_HIP(hipSetDevice(k0.dev));
      M_HIP(hipMemcpyAsync(red, k0.d_small + 3 * N + 4, sizeof(double) * 2, hipMemcpyDeviceToHost, k0.s_upd));
    }
    int info = 0;
    for (int i = 0; i < P; ++i) {
      Rank& k = m->r[i];
      M_HIP(hipSetDevice(k.dev));
      double sc[2];
      int inf = 0;
      M_HIP(hipMemcpyAsync(sc, k.d_small + 3 * N, sizeof(double) * 2, hipMemcpyDeviceToHost, k.s_upd));
      M_HIP(hipMemcpyAsync(&inf, k.d_info, sizeof(int), hipMemcpyDeviceToHost, k.s_upd));
      M_HIP(hipStreamSynchronize(k.s_upd));
      M_HIP(hipStreamSynchronize(k.s_panel));
      M_HIP(hipStreamSynchronize(k.s_comm));
      if (m->transport != TR_RCCL) {   // fixed rank order: deterministic
        red[0] += sc[0];
        red[1] += sc[1];
      }
      if (inf > 0 && (info == 0 || inf < info)) info = inf;
    }
    if (info > 0) {
      set_error("matrix is not positive definite; Cholesky factorization failed at leading minor " +
                std::to_string(info));
      return info;
    }
    out[0] = -0.5 * ((double)N * 1.8378770664093453 + red[0] + red[1]);
    return 0;
  };
  int rc = body();
  cleanup();
  return rc;
}
