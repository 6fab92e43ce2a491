// Internal entry points of the single-GPU driver (capi.hip) used by the multi-GPU driver (multi.hip): the same
// routines the C-ABI entry points are built from, without locking (the caller serialises: one host thread per
// multi-GPU call) and on an explicit stream.  Not part of the ABI.
#pragma once
#include "ctx.h"

namespace sgp {

// K(rows, cols) of `ds` restricted to the global tile window, element (r, c) at Kv[r + c * ld]
int drv_assemble(const sgp_dspec* ds, double* Kv, long ld, long tile_r_lo, long tile_r_hi, long tile_c_lo,
                 long tile_c_hi, int lower_only, int noise_kind, double sigma2, const double* d_noise_diag,
                 hipStream_t s);
// Factor a PACKED column panel (w columns over m rows, element [0] = (row g0, column g0) of the matrix); keeps the panel's
// inverse 16x16 diagonal blocks (INVD_STRIDE doubles per 128-block) in d_invstore (may be NULL) -- what later solves against
// the factor need -- and adds the logdet contribution of the factored columns to *d_logdet.
// df != 0: one launch of the dataflow kernel on the panel (the hybrid schedule, round 6), with the optional extension px: only
// the first px->n_fact columns are factored (the others updated with them), external source panels applied first.
// lean != 0: the two-workgroups-per-CU instantiation of that kernel (ranks sharing one GPU).
// d_nz / nz_words: the factor's tile pattern (structural zeros), read at the panel's offset g0 / 128; NULL: dense.
int drv_panel_factor(sgp_ctx* ctx, double* P, long ld, long m, long w, long g0, double* d_logdet, int* d_info,
                     double* d_invstore, hipStream_t s, int df = 0, const sgp::sz_word* d_nz = nullptr, int nz_words = 0,
                     const sgp::DfPanel* px = nullptr, int lean = 0);
// R <- R L^-T for nrows (multiple of 128) rows against an n x n lower factor with its kept inverse blocks
int drv_row_trsm(sgp_ctx* ctx, double* R, long ldr, long nrows, const double* L, long ldl, const double* d_invall,
                 long n, hipStream_t s);
// back substitution over the 128-blocks k_first, k_first - 128, ..., k_last of alpha = L^-T z:
// element (r, c) of the factor at Lv[r + c * ld] (virtual base), inverse blocks of 128-block b at
// wall_v + b * INVD_STRIDE; d_z / d_alpha are indexed globally, rows up to n_end are read.
int drv_back_substitute(const double* Lv, long ld, const double* wall_v, long k_first, long k_last, long n_end,
                        double* d_z, double* d_alpha, hipStream_t s);
int drv_diag_of_spec(sgp_ctx* ctx, const sgp_dspec* ds, double* d_out, hipStream_t s);
int drv_dspec_create(sgp_ctx* ctx, const sgp_cov_spec* sp, sgp_dspec** out);   // caller holds the context
void drv_dspec_free(sgp_dspec* ds);
long drv_invd_stride();
// structural zeros (common.h; capi.hip: sz_pattern / sz_upload): rank 0's context computes the tile pattern of the factor
// (host), every rank uploads it; *words = 0 / *d_nz = nullptr: dense
int drv_sz_pattern(sgp_ctx* ctx, const sgp_dspec* ds, int noise_kind, long n_pad, long m_tot, int* words);
int drv_sz_upload(sgp_ctx* ctx, const sgp_ctx* from, int words, hipStream_t s, const sgp::sz_word** d_nz);
// executed tile products per tile column of the factor, from the host spec (empty: structurally dense) -- own_table.h
int drv_sz_col_work(const sgp_ctx* ctx, const sgp_cov_spec* sp, int noise_kind, long n_pad, long m_tot,
                    std::vector<double>& col_work);
double drv_sz_live_fraction(const sgp_ctx* ctx, int words, long c0, long w, long m_tot, long kt0, long kt1);
// dst[i] = src[i * stride], i < n
int drv_copy_strided(const double* src, long stride, long n, double* dst, hipStream_t s);
// C[r + c * ldc] += a * S[r + c * lds], nr x nc
int drv_axpy_block(double* C, long ldc, const double* S, long lds, long nr, long nc, double a, hipStream_t s);
// dst[i + c * ld] = mean[i] (i < N) else 0
int drv_fill_mean_cols(double* dst, long ld, long nrows, long ncols, long N, const double* mean, hipStream_t s);
// sparse-ELBO partial sums of a slice of the data / the final factorisation (see sgp_dev_elbo_partial / _finish);
// keep != 0: the M x M factors stay in the caller's buffers (sparse posterior)
long drv_vfe_part_len(long m_pad);
// takes the context itself (CtxScope inside): d_part receives the partial sums of the slice; dLz / d_wz (optional) keep
// the factor of K(z,z) + Sigma_z and its inverse diagonal blocks
int drv_vfe_partial(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                    const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind, const double* z_noise,
                    const double* y, double* dLz, double* d_wz, double* d_part, long part_len);
// ELBO + gradient on `ctx` (takes the context itself); shard == nullptr: the whole call, with the dataflow fallback
int drv_elbo_grad(sgp_ctx* ctx, const ElboGradArgs& a, const ElboGradShard* shard);
// h6: the six ELBO terms (see vfe_pipeline); d_wg (optional) keeps the inverse diagonal blocks of chol(A A' + I)
int drv_vfe_finish(sgp_ctx* ctx, long M, double* d_part, double* d_wg, double* h6);

}  // namespace sgp
