/* Compiled by tests/test_capi_symbols.py with `gcc -std=c99 -pedantic -Wall -Werror`: proves that
 * include/sthenomi.h is valid plain C (what a Julia `ccall` / cgo / any FFI consumer needs) and
 * prints the size and the offset of every field of every struct that crosses the boundary, which
 * the test compares with the ctypes mirror in stheno.jl_amd/lib.py.  With a library path as
 * argv[1] it also dlopens it and resolves every entry point through the C prototypes. */
#include <stddef.h>
#include <stdio.h>
#include <dlfcn.h>

#include "../include/sthenomi.h"
#include "../include/sthenomi_bench.h"   /* the bench hooks: their own header, their own library (argv[2]) */

#define OFF(T, f) printf("offset " #T "." #f " %zu\n", offsetof(T, f))

/* sizeof(&f) is unevaluated (nothing to link) but makes the compiler check that f is declared */
typedef struct { const char* name; size_t fnptr_size; } entry;
#define E(f) {#f, sizeof(&f)}
/* include/sthenomi.h: what libsthenomi.so exports */
static const entry table[] = {
  E(sgp_abi_version), E(sgp_ctx_create), E(sgp_ctx_create_multi), E(sgp_ctx_ndev), E(sgp_ctx_transport),
  E(sgp_ctx_factor_schedule), E(sgp_ctx_factor_work), E(sgp_cov_spec_suggest_order), E(sgp_ctx_multi_stats),
  E(sgp_ctx_multi_owners), E(sgp_ctx_multi_profile), E(sgp_ctx_multi_profile_get), E(sgp_ctx_destroy),
  E(sgp_ctx_trim), E(sgp_ctx_stage_timing), E(sgp_ctx_stage_ms), E(sgp_last_error), E(sgp_kernelmatrix),
  E(sgp_kernelmatrix_diag), E(sgp_logpdf), E(sgp_logpdf_batch), E(sgp_logpdf_f32), E(sgp_kernelmatrix_f32),
  E(sgp_rand_f32), E(sgp_posterior_mean_var_f32), E(sgp_logpdf_grad), E(sgp_logpdf_grad_x), E(sgp_logpdf_grad_xs),
  E(sgp_rand), E(sgp_posterior_create), E(sgp_posterior_predict), E(sgp_posterior_predict_explicit),
  E(sgp_posterior_destroy), E(sgp_elbo), E(sgp_elbo_grad), E(sgp_elbo_grad_x), E(sgp_elbo_grad_xs),
  E(sgp_kernelmatrix_diag_grad_xs), E(sgp_kernelmatrix_diag_grad), E(sgp_kernelmatrix_diag_grad_x),
  E(sgp_sparse_posterior_create), E(sgp_sparse_posterior_predict), E(sgp_sparse_posterior_destroy),
  E(sgp_dspec_create), E(sgp_dspec_destroy), E(sgp_geometry), E(sgp_dev_logpdf), E(sgp_dev_assemble_cols),
  E(sgp_dev_panel_factor), E(sgp_dev_panel_update), E(sgp_dev_panel_update_batch), E(sgp_dev_rowsumsq),
  E(sgp_dev_assemble_cross_rows), E(sgp_dev_rows_dot), E(sgp_dev_rows_gram), E(sgp_elbo_part_len),
  E(sgp_dev_elbo_partial), E(sgp_dev_elbo_finish),
};
/* include/sthenomi_bench.h: what libsthenomi_bench.so exports (round 6: not the product library) */
static const entry bench_table[] = {
  E(sgp_bench_df_fallbacks), E(sgp_bench_multi_fault), E(sgp_bench_multi_broken), E(sgp_bench_multi_stall),
  E(sgp_bench_multi_profile_pieces), E(sgp_bench_mfma_f64), E(sgp_bench_hbm), E(sgp_bench_potrf),
  E(sgp_bench_potrf_contended), E(sgp_bench_cumask), E(sgp_bench_gemm_stamps), E(sgp_bench_gemm),
};
#undef E

int main(int argc, char** argv) {
  size_t i, n = sizeof(table) / sizeof(table[0]), nb = sizeof(bench_table) / sizeof(bench_table[0]);
  printf("abi %d\n", SGP_ABI_VERSION);
  printf("sizeof sgp_input %zu\n", sizeof(sgp_input));
  OFF(sgp_input, dim); OFF(sgp_input, n); OFF(sgp_input, ld); OFF(sgp_input, x);
  printf("sizeof sgp_term %zu\n", sizeof(sgp_term));
  OFF(sgp_term, kind); OFF(sgp_term, row_input); OFF(sgp_term, col_input); OFF(sgp_term, reserved);
  OFF(sgp_term, coef); OFF(sgp_term, param); OFF(sgp_term, row_scale); OFF(sgp_term, col_scale);
  printf("sizeof sgp_cov_spec %zu\n", sizeof(sgp_cov_spec));
  OFF(sgp_cov_spec, n_row_blocks); OFF(sgp_cov_spec, n_col_blocks); OFF(sgp_cov_spec, row_len);
  OFF(sgp_cov_spec, col_len); OFF(sgp_cov_spec, n_inputs); OFF(sgp_cov_spec, inputs);
  OFF(sgp_cov_spec, term_ptr); OFF(sgp_cov_spec, terms); OFF(sgp_cov_spec, symmetric);
  OFF(sgp_cov_spec, reserved);
  printf("sizeof sgp_panel_src %zu\n", sizeof(sgp_panel_src));
  OFF(sgp_panel_src, base); OFF(sgp_panel_src, ld); OFF(sgp_panel_src, row0); OFF(sgp_panel_src, w);
  printf("sizeof sgp_panel_dst %zu\n", sizeof(sgp_panel_dst));
  OFF(sgp_panel_dst, base); OFF(sgp_panel_dst, ld); OFF(sgp_panel_dst, c0); OFF(sgp_panel_dst, w);
  OFF(sgp_panel_dst, src_first); OFF(sgp_panel_dst, src_count);
  printf("enum SGP_SE %d SGP_CONST %d SGP_NOISE_DENSE %d\n", SGP_SE, SGP_CONST, SGP_NOISE_DENSE);
  if (argc > 1) {
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    for (i = 0; i < n; ++i) {
      if (!dlsym(h, table[i].name)) { printf("missing %s\n", table[i].name); return 3; }
    }
    for (i = 0; i < nb; ++i) {   /* the product library must NOT carry the bench hooks */
      if (dlsym(h, bench_table[i].name)) { printf("product library exports %s\n", bench_table[i].name); return 4; }
    }
    {
      int (*ver)(void);
      *(void**)(&ver) = dlsym(h, "sgp_abi_version");
      printf("loaded abi %d symbols %zu\n", ver(), n);
    }
  }
  if (argc > 2) {
    void* hb = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!hb) { printf("dlopen (bench) failed: %s\n", dlerror()); return 2; }
    for (i = 0; i < nb; ++i) {
      if (!dlsym(hb, bench_table[i].name)) { printf("missing %s\n", bench_table[i].name); return 3; }
    }
    printf("loaded bench symbols %zu\n", nb);
  }
  return 0;
}
