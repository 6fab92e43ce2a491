// Diagnosis harness of the dataflow factorisation (chol_df.hip): launches the kernel on a small SPD matrix and watches
// its task counter / abort word / progress counters from the host through a second stream while it runs; a watchdog raises
// the abort word so the kernel always ends.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stheno.jl_amd/csrc
//   tools/df_probe.hip stheno.jl_amd/csrc/libsthenomi.so -o /tmp/df_probe     usage: df_probe [n_pad=256] [border=128] [wgs=512]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "common.h"
extern "C" const char* sgp_last_error(void);

int main(int argc, char** argv) {
  long n_pad = argc > 1 ? atol(argv[1]) : 256, border = argc > 2 ? atol(argv[2]) : 128;
  int wgs = argc > 3 ? atoi(argv[3]) : 512;
  long m_tot = n_pad + border, ld = m_tot;
  std::vector<double> h((size_t)ld * n_pad, 0.0);
  for (long c = 0; c < n_pad; ++c)
    for (long r = c; r < m_tot; ++r) h[r + c * ld] = (r == c) ? 4.0 + 0.001 * r : (r < n_pad ? std::exp(-0.01 * (r - c) * (r - c)) : 0.01 * ((r * 7 + c * 3) % 11));
  double *dA, *dinv, *dslots;
  int *dstate, *dinfo;
  hipMalloc(&dA, sizeof(double) * h.size());
  hipMalloc(&dinv, sizeof(double) * (n_pad / 128) * 2048);
  hipMalloc(&dslots, sizeof(double) * (n_pad / 128));
  hipMalloc(&dstate, sizeof(int) * (sgp::SGP_DF_STATE_WORDS + m_tot / 128));
  hipMalloc(&dinfo, sizeof(int));
  hipMemcpy(dA, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemset(dinfo, 0, sizeof(int));
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t done;
  hipEventCreate(&done);
  auto t0 = std::chrono::steady_clock::now();
  int rc = sgp::launch_chol_dataflow(dA, ld, n_pad, m_tot, dstate, dinv, dslots, dinfo, wgs, 5.0, s);
  printf("launch rc=%d (%s)\n", rc, rc ? sgp_last_error() : "ok");
  hipEventRecord(done, s);
  int nst = (int)sgp::SGP_DF_STATE_WORDS + (int)(m_tot / 128);
  std::vector<int> st(nst);
  for (int it = 0; it < 200; ++it) {
    bool fin = hipEventQuery(done) == hipSuccess;
    hipMemcpyAsync(st.data(), dstate, sizeof(int) * nst, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%8.1f ms head=%d abort=%d dbg=%d,%d prog=", ms, st[0], st[1], st[2], st[3]);
    for (int i = (int)sgp::SGP_DF_STATE_WORDS; i < nst && i < (int)sgp::SGP_DF_STATE_WORDS + 24; ++i) printf("%d ", st[i]);
    printf("%s\n", fin ? " [done]" : "");
    fflush(stdout);
    if (fin) break;
    if (ms > 1500.0) {
      int one = 1;
      hipMemcpyAsync(dstate + 1, &one, sizeof(int), hipMemcpyHostToDevice, s2);
      hipStreamSynchronize(s2);
      printf("watchdog: abort word raised\n");
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(it < 20 ? 5 : 200));
  }
  int info = 0;
  hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
  printf("info=%d\n", info);
  return 0;
}
