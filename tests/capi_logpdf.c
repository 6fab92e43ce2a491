/* A consumer of libsthenomi.so with NO Python in the loop -- what a Julia `ccall` host does, written in C99:
 * builds the sgp_cov_spec of BASELINE configuration c1 (single GP, SEKernel, N = 2048, D = 2, lengthscale sqrt(D),
 * sigma^2 = 0.1) by hand, then
 *   sgp_ctx_create -> sgp_logpdf -> sgp_posterior_create -> sgp_posterior_predict -> sgp_posterior_destroy -> sgp_ctx_destroy
 * and compares with the committed CPU golden of c1 (tests/golden/baseline_configs.json: logpdf, posterior mean / var at
 * 64 points) to 1e-10 / 1e-8; then a covariance that is NOT positive definite (sigma^2 = -5) must come back as rc > 0
 * (LAPACK's info: the first failing leading minor) with a message in sgp_last_error().
 *
 * tests/test_gpu_capi_consumer.py (-m gpu) compiles this file with gcc against include/sthenomi.h ALONE (not the bench
 * header), writes the inputs and the golden values into one binary file (the RNG stream that defines c1 is NumPy's), and
 * runs it.  File layout (little-endian): int64 N, D, NS; double sigma2, golden_logpdf; X[D*N] (already divided by the
 * lengthscale, ColVecs column-major); y[N]; XS[D*NS]; golden_mean[NS]; golden_var[NS].
 * usage: capi_logpdf <libsthenomi.so> <case.bin>                                                        */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/sthenomi.h"

#define LOAD(name) \
  *(void**)(&p_##name) = dlsym(h, #name); \
  if (!p_##name) { printf("missing symbol %s\n", #name); return 2; }

static int (*p_sgp_abi_version)(void);
static int (*p_sgp_ctx_create)(int, sgp_ctx**);
static int (*p_sgp_ctx_destroy)(sgp_ctx*);
static const char* (*p_sgp_last_error)(void);
static int (*p_sgp_logpdf)(sgp_ctx*, const sgp_cov_spec*, const double*, int, const double*, const double*, int64_t,
                           int64_t, double*);
static int (*p_sgp_posterior_create)(sgp_ctx*, const sgp_cov_spec*, const double*, int, const double*, const double*,
                                     double*, sgp_post**);
static int (*p_sgp_posterior_predict)(sgp_post*, const sgp_cov_spec*, const sgp_cov_spec*, const double*, double*, double*,
                                      double*, int64_t);
static int (*p_sgp_posterior_destroy)(sgp_post*);

static double* read_doubles(FILE* f, size_t n) {
  double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
  if (!p || fread(p, sizeof(double), n, f) != n) { printf("short read\n"); exit(2); }
  return p;
}

/* cov(f, a, b) of ONE process with ONE SE term: one row block, one column block, inputs 0 (rows) and 1 (columns) */
static void one_term_spec(sgp_cov_spec* s, sgp_input* in, sgp_term* t, int32_t* ptr, int64_t* rl, int64_t* cl,
                          const double* xa, int64_t na, const double* xb, int64_t nb, int64_t D, int symmetric) {
  in[0].dim = D; in[0].n = na; in[0].ld = D; in[0].x = xa;
  in[1].dim = D; in[1].n = nb; in[1].ld = D; in[1].x = xb;
  t->kind = SGP_SE; t->row_input = 0; t->col_input = symmetric ? 0 : 1; t->reserved = 0;
  t->coef = 1.0; t->param = 0.0; t->row_scale = NULL; t->col_scale = NULL;
  ptr[0] = 0; ptr[1] = 1;
  rl[0] = na; cl[0] = symmetric ? na : nb;
  s->n_row_blocks = 1; s->n_col_blocks = 1; s->row_len = rl; s->col_len = cl;
  s->n_inputs = symmetric ? 1 : 2; s->inputs = in; s->term_ptr = ptr; s->terms = t;
  s->symmetric = symmetric; s->reserved = 0;
}

int main(int argc, char** argv) {
  void* h;
  FILE* f;
  int64_t hdr[3], N, D, NS, i;
  double sc[2], sigma2, want_lp, lp = 0.0, bad = -5.0, e_m = 0.0, e_v = 0.0, scale_m = 1.0, scale_v = 1.0;
  double *X, *y, *XS, *gm, *gv, *mean, *var, *alpha;
  sgp_ctx* ctx = NULL;
  sgp_post* post = NULL;
  sgp_cov_spec sxx, ssx, sss;
  sgp_input in_xx[2], in_sx[2], in_ss[2];
  sgp_term t_xx, t_sx, t_ss;
  int32_t p_xx[2], p_sx[2], p_ss[2];
  int64_t rl_xx[1], cl_xx[1], rl_sx[1], cl_sx[1], rl_ss[1], cl_ss[1];
  int rc;
  if (argc < 3) { printf("usage: capi_logpdf lib case.bin\n"); return 2; }
  h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 2; }
  LOAD(sgp_abi_version) LOAD(sgp_ctx_create) LOAD(sgp_ctx_destroy) LOAD(sgp_last_error) LOAD(sgp_logpdf)
  LOAD(sgp_posterior_create) LOAD(sgp_posterior_predict) LOAD(sgp_posterior_destroy)
  if (p_sgp_abi_version() != SGP_ABI_VERSION) { printf("ABI mismatch\n"); return 2; }
  f = fopen(argv[2], "rb");
  if (!f || fread(hdr, sizeof(int64_t), 3, f) != 3 || fread(sc, sizeof(double), 2, f) != 2) { printf("bad case file\n"); return 2; }
  N = hdr[0]; D = hdr[1]; NS = hdr[2]; sigma2 = sc[0]; want_lp = sc[1];
  X = read_doubles(f, (size_t)(D * N)); y = read_doubles(f, (size_t)N); XS = read_doubles(f, (size_t)(D * NS));
  gm = read_doubles(f, (size_t)NS); gv = read_doubles(f, (size_t)NS);
  fclose(f);
  mean = (double*)malloc(sizeof(double) * (size_t)NS); var = (double*)malloc(sizeof(double) * (size_t)NS);
  alpha = (double*)malloc(sizeof(double) * (size_t)N);

  rc = p_sgp_ctx_create(0, &ctx);
  if (rc) { printf("sgp_ctx_create rc=%d: %s\n", rc, p_sgp_last_error()); return 1; }
  one_term_spec(&sxx, in_xx, &t_xx, p_xx, rl_xx, cl_xx, X, N, X, N, D, 1);
  rc = p_sgp_logpdf(ctx, &sxx, NULL, SGP_NOISE_SCALAR, &sigma2, y, N, 1, &lp);
  if (rc) { printf("sgp_logpdf rc=%d: %s\n", rc, p_sgp_last_error()); return 1; }
  printf("logpdf %.15g golden %.15g rel %.3e\n", lp, want_lp, fabs(lp - want_lp) / fabs(want_lp));
  if (!(fabs(lp - want_lp) <= 1e-10 * fabs(want_lp))) { printf("FAIL logpdf\n"); return 1; }

  rc = p_sgp_posterior_create(ctx, &sxx, NULL, SGP_NOISE_SCALAR, &sigma2, y, alpha, &post);
  if (rc) { printf("sgp_posterior_create rc=%d: %s\n", rc, p_sgp_last_error()); return 1; }
  one_term_spec(&ssx, in_sx, &t_sx, p_sx, rl_sx, cl_sx, XS, NS, X, N, D, 0);     /* cov(f, x*, x)  */
  one_term_spec(&sss, in_ss, &t_ss, p_ss, rl_ss, cl_ss, XS, NS, XS, NS, D, 1);   /* cov(f, x*)     */
  rc = p_sgp_posterior_predict(post, &ssx, &sss, NULL, mean, var, NULL, 0);
  if (rc) { printf("sgp_posterior_predict rc=%d: %s\n", rc, p_sgp_last_error()); return 1; }
  for (i = 0; i < NS; ++i) {
    if (fabs(gm[i]) > scale_m) scale_m = fabs(gm[i]);
    if (fabs(gv[i]) > scale_v) scale_v = fabs(gv[i]);
  }
  for (i = 0; i < NS; ++i) {
    if (fabs(mean[i] - gm[i]) > e_m) e_m = fabs(mean[i] - gm[i]);
    if (fabs(var[i] - gv[i]) > e_v) e_v = fabs(var[i] - gv[i]);
  }
  printf("posterior at %ld points: max |mean - golden| %.3e, max |var - golden| %.3e\n", (long)NS, e_m, e_v);
  if (!(e_m <= 1e-8 * scale_m) || !(e_v <= 1e-8 * scale_v)) { printf("FAIL posterior\n"); return 1; }
  rc = p_sgp_posterior_destroy(post);
  if (rc) { printf("sgp_posterior_destroy rc=%d\n", rc); return 1; }

  /* not positive definite: rc > 0 = LAPACK's info, the error text says so */
  rc = p_sgp_logpdf(ctx, &sxx, NULL, SGP_NOISE_SCALAR, &bad, y, N, 1, &lp);
  printf("sigma2 = -5: rc %d (%s)\n", rc, p_sgp_last_error());
  if (rc <= 0) { printf("FAIL posdef rc\n"); return 1; }
  /* a bad argument: rc < 0, the context stays usable */
  rc = p_sgp_logpdf(ctx, NULL, NULL, SGP_NOISE_SCALAR, &sigma2, y, N, 1, &lp);
  if (rc >= 0) { printf("FAIL NULL spec accepted\n"); return 1; }
  rc = p_sgp_logpdf(ctx, &sxx, NULL, SGP_NOISE_SCALAR, &sigma2, y, N, 1, &lp);
  if (rc || !(fabs(lp - want_lp) <= 1e-10 * fabs(want_lp))) { printf("FAIL logpdf after errors\n"); return 1; }
  rc = p_sgp_ctx_destroy(ctx);
  if (rc) { printf("sgp_ctx_destroy rc=%d\n", rc); return 1; }
  printf("OK\n");
  return 0;
}
