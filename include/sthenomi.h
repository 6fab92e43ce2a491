/*
 * sthenomi.h -- C ABI of libsthenomi.so: MI355X (gfx950) dense Gaussian-process inference
 * behind the Stheno.jl / AbstractGPs.jl operator surface.
 *
 * The reference (Stheno.jl v0.8.2, /root/reference) has no FFI seam: the seam is Julia
 * multiple dispatch on FiniteGP{<:Union{GPPP,SthenoAbstractGP}}.  Each entry point below
 * names the reference method(s) it replaces (file:line relative to /root/reference; [EXT]
 * marks AbstractGPs.jl / KernelFunctions.jl arithmetic the reference delegates to, see
 * SURVEY.md section 8a).  The Julia-side binding is shown in INTEGRATION.md and
 * julia/SthenoMI355X.jl.
 *
 * Conventions
 *   - all reals are IEEE fp64, all matrices column-major, all sizes int64_t;
 *   - inputs are KernelFunctions.ColVecs layout: a D x n column-major matrix, each point D
 *     contiguous doubles (docs/src/input_types.md:48-55); 1-D inputs are D = 1;
 *   - the caller owns every host buffer; nothing host-side is retained after return;
 *   - return value: 0 ok; >0 LAPACK potrf convention (order of the first non-positive
 *     leading minor -> the shim throws PosDefException(info), as `cholesky` does in the
 *     reference path); <0 bad argument / HIP failure, text via sgp_last_error();
 *   - a ctx serialises its own calls (one in-flight operation per ctx).
 *
 * A covariance "spec" is the flattened form of a Stheno GP tree evaluated at BlockData
 * inputs: for each (row block I, column block J) a list of terms
 *     K[I,J] = sum_t coef_t * diag(rs_t) * k_{kind_t}(Xr_t, Xc_t) * diag(cs_t)
 * which is what src/gp/derived_gp.jl:31-44 + src/affine_transformations/{cross,addition,
 * product,compose}.jl evaluate recursively.  A block pair with zero terms is an exact-zero
 * block (src/gp/atomic_gp.jl:36-38).
 */
#ifndef STHENOMI_H
#define STHENOMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGP_ABI_VERSION 1

/* kernel kinds: KernelFunctions.jl SimpleKernels [EXT] */
enum {
  SGP_SE = 0,       /* SEKernel / SqExponentialKernel: exp(-d^2/2)                       */
  SGP_MATERN12 = 1, /* Matern12Kernel / ExponentialKernel: exp(-d)                       */
  SGP_MATERN32 = 2, /* Matern32Kernel: (1+sqrt3 d) exp(-sqrt3 d)                         */
  SGP_MATERN52 = 3, /* Matern52Kernel: (1+sqrt5 d+5d^2/3) exp(-sqrt5 d)                  */
  SGP_WHITE = 4,    /* WhiteKernel: 1[x == y]                                            */
  SGP_CONST = 5     /* ConstantKernel(c): param                                          */
};

/* noise kinds for FiniteGP(f, x, Sigma_y)  (AbstractGPs FiniteGP [EXT], App. A.2) */
enum {
  SGP_NOISE_SCALAR = 0, /* f(x, s2)  -> s2 * I   (noise[0] = s2; f(x) passes 1e-18)      */
  SGP_NOISE_DIAG = 1,   /* f(x, v)   -> Diagonal(v), noise = v[N]                        */
  SGP_NOISE_DENSE = 2   /* f(x, S)   -> dense N x N column-major, noise = S (ld = N)     */
};

/* one transformed input collection: a D x n ColVecs matrix (host pointer) */
typedef struct {
  int64_t dim;      /* D                                                                 */
  int64_t n;        /* number of points                                                  */
  int64_t ld;       /* leading dimension (>= D)                                          */
  const double* x;  /* host, column-major D x n                                          */
} sgp_input;

/* one term of one block pair */
typedef struct {
  int32_t kind;       /* SGP_SE ...                                                      */
  int32_t row_input;  /* index into sgp_cov_spec.inputs; must have n == row block length */
  int32_t col_input;  /* index into sgp_cov_spec.inputs; must have n == col block length */
  int32_t reserved;
  double coef;        /* product of scalar scales (may be negative)                      */
  double param;       /* SGP_CONST: c                                                    */
  const double* row_scale; /* host, length = row block length, or NULL (== ones)         */
  const double* col_scale; /* host, length = col block length, or NULL                   */
} sgp_term;

/* flattened covariance cov(f, x, x') over BlockData x (rows) and x' (cols).
 * Replaces the recursion entered at src/gaussian_process_probabilistic_programme.jl:51-64
 * (cov(f::GPPP, x[, x'])) -> src/affine_transformations/cross.jl:59-86. */
typedef struct {
  int32_t n_row_blocks, n_col_blocks;
  const int64_t* row_len;   /* [n_row_blocks]                                            */
  const int64_t* col_len;   /* [n_col_blocks]                                            */
  int32_t n_inputs;
  const sgp_input* inputs;
  const int32_t* term_ptr;  /* CSR over block pairs, row-major (I * n_col_blocks + J)    */
  const sgp_term* terms;
  int32_t symmetric;        /* 1: x' === x (cov(f, x)); lets the library build one
                               triangle and guarantees an exactly symmetric result       */
  int32_t reserved;
} sgp_cov_spec;

typedef struct sgp_ctx sgp_ctx;
typedef struct sgp_post sgp_post;
typedef struct sgp_sparse_post sgp_sparse_post;

/* ---- context ---------------------------------------------------------------------- */
int sgp_abi_version(void);
/* device: HIP ordinal.  Fails (<0) when no gfx950 device is present: there is no CPU path. */
int sgp_ctx_create(int device, sgp_ctx** out);
/* One context over several GPUs of the node (BASELINE.json north_star).  On such a context
 *   sgp_logpdf            shards the N x N covariance in column panels, block-cyclic over devices[0..ndev):
 *                         right-looking blocked Cholesky, the chain factorisation -> transport -> look-ahead update
 *                         pipelined in sub-panels, ONE batched update launch per rank and step; panels travel by RCCL
 *                         (ncclCommInitAll inside, one grouped ncclBroadcast per panel over xGMI) or by peer copies as
 *                         scatter + all-gather (every receiver's ndev - 1 ingress links carry a slab each;
 *                         SGP_MULTI_TRANSPORT=rccl|p2p|auto, SGP_MULTI_BCAST=direct for one copy owner -> receiver);
 *                         logdet and |L^-1 (Y - m)|^2 by ncclAllReduce (or summed on the host in rank order);
 *   sgp_posterior_create  keeps that sharded factor, sgp_posterior_predict solves K(x*, x) L^-T against it (left-looking,
 *                         one n* x W reduction per panel) -- any number of predictions per factor;
 *   sgp_rand              multiplies every rank's own panels of L with Z, one reduction;
 *   sgp_elbo              shards the DATA POINTS (sgp_dev_elbo_partial per rank, ONE reduction of M^2 + M + 2 doubles).
 *   sgp_logpdf_grad       (round 4) the kept sharded factor, L^-T through the posterior's row sweep, C^-1 as a sum over
 *                         ranks with a reduce-scatter by column slabs, every rank contracting its slabs with the kernel
 *                         derivatives: gradients w.r.t. the kernel terms, scalar / diagonal noise, y and the mean.
 *   sgp_logpdf_grad_x / _xs  (round 6) the input-point and function-scale gradients on the same sharded result: every rank
 *                         contracts its column slabs of G row-side, per-rank sums added in rank order.
 *   sgp_kernelmatrix      (round 6) the columns in ndev tile-aligned chunks, one per rank, copied straight into the caller's
 *                         matrix; a symmetric spec stays EXACTLY symmetric (upper part by transposition) and bit-equal to the
 *                         one-GPU matrix; sgp_kernelmatrix_diag: every block's points in ndev slices.
 * Dense Sigma_y shards as well (round 4: the owner of a panel adds its column slab at assembly).
 * (a gradient with a dense Sigma_y too, round 6: G = (alpha alpha' - C^-1) / 2 comes back column slab by column slab.)
 *   sgp_elbo_grad / _x / _xs  (round 6) shard the data points like sgp_elbo: every rank runs the pipeline on its slice, the
 *                         sums over data points (A A', A delta, four scalars: M^2 + M + 4 doubles) meet in ONE reduction
 *                         between the two factorisations, the M x M stage runs replicated on identical numbers; per-point
 *                         results come back slice by slice, sums over data points are added in rank order, the K(z,z) side
 *                         is rank 0's.  Fewer than 128 data points per rank: not sharded.
 * One host thread, one `ccall`: the Julia side is unchanged.  Still on devices[0]: the M x M factors of a sparse posterior
 * (0.02 TFLOP: replicated work by design) and the O(N) diagonal entry points' gradients.  Failure (round 6): a HIP / RCCL error on any rank's enqueue thread fails the call with rc < 0 and the root cause
 * in sgp_last_error(), within seconds (every cross-thread wait is bounded: SGP_MULTI_SPIN_TIMEOUT_S, ncclCommInitAll:
 * SGP_MULTI_INIT_TIMEOUT_S); a peer-copy context stays usable, an RCCL context whose communicators had to be aborted refuses
 * further sharded calls and says so.  A device listed several times gives that many ranks on one GPU
 * (test configuration).  SGP_MULTI_PANEL=<cols> sets the panel width (default 1024).  sgp_ctx_ndev -> number of ranks
 * (1 for an ordinary context); sgp_ctx_transport -> "single" | "rccl" | "p2p" | "p2p-staged" (peer access missing:
 * refused unless SGP_MULTI_ALLOW_STAGED=1) | "loopback". */
int sgp_ctx_create_multi(const int* devices, int ndev, sgp_ctx** out);
int sgp_ctx_ndev(sgp_ctx* ctx);
const char* sgp_ctx_transport(sgp_ctx* ctx);
/* Which schedule the blocked Cholesky of an N-point covariance runs on this context (bench / diagnosis): "dataflow-fat" |
 * "dataflow" (one launch of persistent workgroups, chol_df.hip) | "hybrid" (from 24576 columns on: every 2048-column panel by
 * one launch of that kernel, trailing updates as lock-step launches) | "launches-one-panel" | "launches-lookahead" |
 * "launches-serial" | "launches-serial-deep" (capi.hip: chol_bordered).  Every schedule gives the same bits. */
const char* sgp_ctx_factor_schedule(sgp_ctx* ctx, int64_t N);
/* Work of the last factorisation sgp_logpdf / sgp_rand / sgp_posterior_create ran on this context, in tile products
 * (128 x 128 x 128) of the trailing contractions: *executed, and *dense = what the dense schedule executes.  They differ
 * when the model has independent components: the covariance of a Stheno programme then has EXACT zero blocks (no term
 * connects the two processes, src/gp/... cross.jl yields zeros), so has its Cholesky factor, and the factorisation skips
 * every tile product with a structurally zero operand (tile-level symbolic factorisation with fill-in; the skipped
 * products are exact zeros: the factor keeps its bits; the reference's LAPACK path multiplies them out).
 * SGP_STRUCT_ZEROS=0 switches the skipping off.  Dense noise, a single block or a pattern without zeros: executed == dense. */
int sgp_ctx_factor_work(sgp_ctx* ctx, double* executed, double* dense);
/* WHICH tiles of the factor are structurally zero depends on the order of the blocks in the caller's BlockData, as for
 * any sparse direct solver: f3 = f1 + f2 observed as (f1, f2, f3) keeps the (f2, f1) block of the factor zero, (f3, f1, f2)
 * fills it in and the factorisation silently takes the dense time.  The library never permutes (the factor's layout is
 * part of what `posterior` keeps and `rand` multiplies a draw with); this entry point tells ANY host a good order: greedy
 * minimum fill on the block graph of a symmetric spec (two blocks are adjacent when their pair has terms), fill weighted
 * by block lengths, ties: the shorter block first, then the caller's order.  perm_out[k] = index of the block to put at
 * position k (n_row_blocks entries); *changes (may be NULL) = 1 when the suggested order would skip more than the given
 * one.  logpdf, posterior moments and the ELBO do not depend on the order of the observations beyond rounding; the caller
 * applies the permutation to its blocks, y, mean and noise (the Python mirror: stheno.jl_amd/ordering.py permute_blocks;
 * julia/SthenoMI355X.jl: logpdf).  Host-only: no context, no GPU work.  No reference analogue (Stheno builds the dense
 * matrix and LAPACK factors it whatever the order). */
int sgp_cov_spec_suggest_order(const sgp_cov_spec* spec, int32_t* perm_out, int32_t* changes);
/* Ownership and event-binding figures of sgp_ctx_multi_stats (below) when cap >= 11 + 4 * ranks: two more doubles after the
 * enqueue time -- how the panels of the last sharded factorisation were dealt out (0 cyclic, 1 the balanced table built from
 * the symbolic tile pattern of a structured model: one panel per rank and round, heaviest panel to the least loaded rank,
 * csrc/own_table.h; 2 an explicit list, SGP_MULTI_OWNERS=r0,r1,...; SGP_MULTI_OWNERS=cyclic|balanced chooses) and how many
 * cross-thread event waits bound to a later record than the schedule named (cumulative; diagnostic). */
/* Figures of the last sharded factorisation of a multi-GPU context: out[0] ranks, [1] wall ms (enqueue to
 * completion), [2] transport (0 loopback, 1 peer copies, 2 RCCL), [3] ranks the RCCL communicator reports (-1: none),
 * [4] panel width (of the first part; the tail may be narrower), [5] panels, [6] 1 = scatter + all-gather peer copies,
 * [7] panels per update group; then per rank 4 doubles:
 * algorithmic flops of its trailing updates, ms from the start of its first to the end of its last update, panels it
 * factored, bytes it received.  cap >= 8 + 4 * ranks; with cap >= 9 + 4 * ranks one more: the host time (ms) the enqueue
 * thread spent issuing that factorisation (everything is asynchronous: it must stay below the wall time).  *n_out =
 * doubles written. */
int sgp_ctx_multi_stats(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out);
/* Which rank owns which column panel of the last sharded factorisation: out[J] = rank of panel J (*n_out = panels; out may
 * be NULL to ask for the count).  Diagnosis / tools/multi_projection.py. */
int sgp_ctx_multi_owners(sgp_ctx* ctx, int32_t* out, int64_t cap, int64_t* n_out);
/* Profile mode (enable != 0): the following sharded factorisations run SERIALISED, every group of launches alone on
 * the hardware and timed on the host -- per panel J {factor_ms, lookahead_update_ms, panel bytes, then for rank
 * 0 .. P - 1 the three update classes of step J: near_a_ms, near_b_ms, far_ms} (3 + 3 P doubles; csrc/multi.hip explains
 * the classes).  With all ranks on one GPU these are the times each GPU of a node would see for its own share
 * (tools/multi_projection.py).  _get: out may be NULL to query *n_out. */
int sgp_ctx_multi_profile(sgp_ctx* ctx, int enable);
int sgp_ctx_multi_profile_get(sgp_ctx* ctx, double* out, int64_t cap, int64_t* n_out);
int sgp_ctx_destroy(sgp_ctx* ctx);
/* A ctx keeps the device workspaces of finished calls (the m_tot x n_pad factor buffer of
 * sgp_logpdf etc.) in a grow-only cache so that repeated calls of one shape do not pay
 * hipMalloc / hipFree of N^2 doubles each time (hyper-parameter optimisation loops call
 * logpdf hundreds of times: examples/getting_started/script.jl:154-213).  sgp_ctx_trim
 * returns every cached, unused block to the driver.  SGP_POOL=0 disables the cache. */
int sgp_ctx_trim(sgp_ctx* ctx);
/* Optional per-stage device timing (HIP events on the ctx stream) of pipelines a caller cannot time
 * from outside; enable != 0 switches it on and clears the accumulators, sgp_ctx_stage_ms copies the
 * 16 accumulated stage times (ms).  Stage ids of sgp_elbo / sgp_sparse_posterior_create for more
 * than 65536 data points (row-chunked pipeline): 0 K(z,z) assembly + factorisation, 1 K(x,z)
 * assembly (the HBM-bound stage: 8 M N bytes written), 2 Lambda scaling + row solve against Lz
 * (M^2 N flops), 3 A delta / |A|^2 reductions + transpose, 4 Gram product A A' (M^2 N flops) +
 * reduction, 5 factorisation of A A' + I and scalars. */
int sgp_ctx_stage_timing(sgp_ctx* ctx, int enable);
int sgp_ctx_stage_ms(sgp_ctx* ctx, double* out16);
const char* sgp_last_error(void); /* thread-local */

/* ---- covariance assembly (K1-K3, S2-S7) --------------------------------------------
 * sgp_kernelmatrix      : cov(f, x) / cov(f, x, x') -> dense N x M host matrix.
 *                         [EXT kernelmatrix; src/gp/atomic_gp.jl:30-33, cross.jl:59-72]
 * sgp_kernelmatrix_diag : var(f, x) / var(f, x, x') -> N-vector (blocks pairwise equal
 *                         length). [EXT kernelmatrix_diag; src/gp/util.jl:5-7, cross.jl:64-77] */
int sgp_kernelmatrix(sgp_ctx* ctx, const sgp_cov_spec* spec, double* K, int64_t ldk);
int sgp_kernelmatrix_diag(sgp_ctx* ctx, const sgp_cov_spec* spec, double* out);

/* ---- logpdf(fx, y) / logpdf(fx, Y) (A2; AbstractGPs logpdf [EXT], App. A.3) ----------
 * spec must be symmetric; mean = mean(fx) (N, host; NULL == zeros); Y is N x ncols.
 * out[s] = -(N log 2pi + logdet C + |L^-1 (Y[:,s]-m)|^2)/2,  C = K + Sigma_y = L L'. */
int sgp_logpdf(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
               const double* noise, const double* Y, int64_t ldy, int64_t ncols, double* out);
/* logpdf of nspec INDEPENDENT models in one call (round 6): the loop a host runs around logpdf when it restarts an
 * optimiser from several initial hyper-parameters, cross-validates, or evaluates a population of candidates
 * (/root/reference/examples/getting_started/script.jl:154-213 is one such chain; its restarts are B of them).  Member b:
 * specs[b], means[b] (means or means[b] may be NULL == zeros), noises[b] (SCALAR: one value, DIAG: N values; one kind for the
 * batch), ys[b] (one vector); out[b] = logpdf(f_b(x_b, noise_b), y_b) -- bit-equal to the member's own sgp_logpdf call.
  * Members of one PADDED size (the same number of 128-column tiles: equal N, or the folds of a cross-validation, which differ by
 * a point or two) up to SGP_BATCH_MAX_N (12288) padded columns are assembled side by side and factored by ONE launch
 * of the dataflow kernel as a single task pool: at sizes where one factorisation is bound by its diagonal chain (N <= 8192)
 * the B chains hide each other and the aggregate rate is a multiple of the single call's (docs/05).  Anything else
 * (different padded sizes, dense noise, larger members, a multi-GPU context) runs member by member.
 * A member whose matrix is not positive definite gets out[b] = NaN and infos[b] = the failing leading minor (LAPACK's info;
 * 0 for the others); with infos == NULL the call returns the first such info (> 0) instead of 0. */
int sgp_logpdf_batch(sgp_ctx* ctx, int nspec, const sgp_cov_spec* const* specs, const double* const* means,
                     int noise_kind, const double* const* noises, const double* const* ys, double* out, int* infos);

/* ---- fp32 instantiation (the reference is type-stable in Float32:
 * /root/reference/test/gp/util.jl:76-88).  Same spec (the fp64 inputs are rounded to fp32 once on the
 * device), covariance assembly, blocked Cholesky (v_mfma_f32_32x32x2_f32 updates) and forward substitution
 * in single precision; the scalar result is returned as a double holding the fp32-accurate value.
 * y is one vector (N); noise SCALAR or DIAG.  sgp_kernelmatrix_f32: cov(f, x) / cov(f, x, x') as fp32.
 * Limits of the fp32 kernels: input dimension <= 16 and (terms per block pair) x (dimension rounded up to a power of
 * two) <= 64 (rc < 0 otherwise; the host mirror sends such models down the fp64 path and rounds the result). */
int sgp_logpdf_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                   const double* noise, const double* y, double* out);
int sgp_kernelmatrix_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, float* K, int64_t ldk);
/* rand(rng, fx, S) on the fp32 factor: out (N x S floats) = mean .+ L Z, Z (N x S doubles) the caller's draw rounded
 * to fp32 (AbstractGPs rand [EXT], App. A.4; `rand(rng, fx) isa Vector{Float32}`: test/gp/util.jl:76-88). */
int sgp_rand_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                 const double* noise, const double* Z, int64_t ldz, int64_t S, float* out, int64_t ldo);
/* posterior(fx, y) followed by mean_and_var at x* in single precision and ONE factorisation: K(x*, x) rides through
 * the fp32 Cholesky as bordered rows below the observation row.  cross / prior_ss / mean_s as in
 * sgp_posterior_predict; mean_out / var_out: Ns floats (either may be NULL). */
int sgp_posterior_mean_var_f32(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                               const double* noise, const double* y, const sgp_cov_spec* cross,
                               const sgp_cov_spec* prior_ss, const double* mean_s, float* mean_out,
                               float* var_out);

/* ---- logpdf and its reverse-mode gradient -------------------------------
 * What Zygote derives through `logpdf(f(x, s2), y)` on the reference path for hyper-parameter
 * learning (examples/getting_started/script.jl:154-213; AD glue: SURVEY.md section 2 #11).
 * With alpha = C^-1 (y - m) and G = (alpha alpha' - C^-1)/2 = d logpdf / d C:
 *   grad_y[N]    = -alpha          grad_mean[N] = +alpha
 *   grad_noise   = tr G (SCALAR, 1 value), diag G (DIAG, N values) or G itself (DENSE: N x N, column-major, ld = N)
 *   grad_coef[t]    = sum_{i,j in block pair of term t} G_ij rs_i k_t(x_i, x_j) cs_j      = d/d coef_t
 *   grad_inscale[t] = sum G_ij coef_t rs_i cs_j d k_t(g x_i, g x_j)/dg at g = 1  (both inputs of the
 *                     term scaled by g: the derivative w.r.t. an inverse lengthscale / stretch)
 * one entry per element of spec->terms, in that order (mirror-image block pairs (I,J), (J,I) each
 * report their own term; the host adds what belongs to one parameter).  Any output may be NULL.
 * noise_kind must be SCALAR or DIAG. */
int sgp_logpdf_grad(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                    const double* noise, const double* y, double* logpdf_out, double* grad_y,
                    double* grad_mean, double* grad_noise, double* grad_coef, double* grad_inscale);
/* Same, plus the gradient w.r.t. the input points: grad_inputs[k] (may be NULL) receives a packed
 * dim_k x n_k column-major array for spec->inputs[k] -- d logpdf / d (the points the terms read, i.e.
 * after whatever Stretch / Select / Periodic transformation the host applied; chaining back to the
 * user's x is the host's job).  Stationary kernels only depend on x - x', so this is
 * sum_j 2 G_ij coef rs_i cs_j kappa'(d2_ij) 2 (x_i - x'_j) over every term that reads input k.
 * Any input dimension (term and input gradients walk it in chunks of 16 beyond 64 / 16).  Matern-1/2 is not differentiable at coincident points: those pairs
 * contribute 0.  Row / column scale vectors (function-scaled processes) are held fixed here: see
 * sgp_logpdf_grad_xs. */
int sgp_logpdf_grad_x(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                      const double* noise, const double* y, double* logpdf_out, double* grad_y,
                      double* grad_mean, double* grad_noise, double* grad_coef, double* grad_inscale,
                      double* const* grad_inputs);
/* Same, plus the gradient w.r.t. the ROW SCALE vectors of function-scaled processes (sigma(x) * f,
 * /root/reference/src/affine_transformations/product.jl:25-48; on the reference path Zygote differentiates
 * through sigma.(x)): K_ij = coef rs_i k_ij cs_j, and grad_rowscale[t] (one entry per term, NULL to skip; ignored
 * for terms without a row scale) receives, for the row_len[I] points of the term's row block,
 *   2 sum_j G_ij coef k_ij cs_j.
 * The column scale of a term is the row scale of its mirror term in block pair (J, I), so the gradient of ONE
 * scale vector is the sum of the grad_rowscale arrays of every term that carries it as row scale (the host
 * mirror does this by array identity).  grad_inputs may be NULL. */
int sgp_logpdf_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
                       const double* noise, const double* y, double* logpdf_out, double* grad_y,
                       double* grad_mean, double* grad_noise, double* grad_coef, double* grad_inscale,
                       double* const* grad_inputs, double* const* grad_rowscale);

/* ---- rand(rng, fx, S) (A3; App. A.4): out = mean .+ L * Z, Z = randn(rng, N, S) drawn by
 * the caller's RNG (column-major fill order), so the integer RNG stream stays the caller's. */
int sgp_rand(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean, int noise_kind,
             const double* noise, const double* Z, int64_t ldz, int64_t S, double* out,
             int64_t ldo);

/* ---- posterior(fx, y) (A4; App. A.5) -------------------------------------------------
 * Keeps L and L^-1 (y - m) in HBM.  alpha_out (N, may be NULL) receives C^-1 (y - m),
 * the `α` field of AbstractGPs.PosteriorGP.data [EXT]. */
int sgp_posterior_create(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* mean,
                         int noise_kind, const double* noise, const double* y,
                         double* alpha_out, sgp_post** out);
/* cross : cov(f, x*, x) spec (rows = x* blocks, cols = training blocks)  [gppp.jl:60-64]
 * prior_ss: symmetric spec at x* (used for var / cov of the prior)       [gppp.jl:51-58]
 * mean_s : prior mean at x* (Ns; NULL == zeros)
 * mean_out (Ns) / var_out (Ns) / cov_out (Ns x Ns, ld = ldcov): any may be NULL.
 * mean* = m* + K*x alpha; var* = diag K** - colsumsq(L^-1 Kx*); cov* = K** - V'V. */
int sgp_posterior_predict(sgp_post* post, const sgp_cov_spec* cross, const sgp_cov_spec* prior_ss,
                          const double* mean_s, double* mean_out, double* var_out,
                          double* cov_out, int64_t ldcov);
/* The same against a cross-covariance GIVEN as a matrix (round 5): conditioning on top of a process whose covariance is not a
 * sum of kernel terms -- the approximate (VFE) posterior, which the reference returns as an ordinary AbstractGP
 * (src/gp/sparse_finite_gp.jl:60-62) that can be observed and conditioned again.  `post` was created by sgp_posterior_create on
 * a zero-term spec with dense noise = cov(f, x) + Sigma_y; cross = cov(f, x*, x), ns x N column-major (ld ldc); prior_var =
 * var(f, x*) (ns; for var_out), prior_cov = cov(f, x*) (ns x ns, ld ldp; for cov_out), mean_s = mean(f, x*) or NULL.  Host
 * buffers; single-GPU contexts. */
int sgp_posterior_predict_explicit(sgp_post* post, const double* cross, int64_t ldc, int64_t ns, const double* prior_var,
                                   const double* prior_cov, int64_t ldp, const double* mean_s, double* mean_out,
                                   double* var_out, double* cov_out, int64_t ldcov);
int sgp_posterior_destroy(sgp_post* post);

/* ---- elbo and its reverse-mode gradient -----------------------------------
 * What Zygote derives through `elbo(VFE(f(z)), f(x, s2), y)` on the reference path
 * (AbstractGPs.elbo [EXT], App. A.6; src/gp/sparse_finite_gp.jl:37-62).  Arguments as sgp_elbo
 * (Sigma_y scalar or diagonal, as in AbstractGPs.elbo; Sigma_z scalar, diagonal or dense).  Outputs (any but elbo_out
 * may be NULL):
 *   grad_y[N], grad_mean[N] = -grad_y, grad_noise[1 | N], grad_var_x[N] = -1/(2 sy),
 *   grad_z_noise[1 | M | M x M, ld M] (tr / diag of d elbo / d (Kzz + Sigma_z); dense Sigma_z: that cotangent itself),
 *   grad_coef_zz / grad_inscale_zz: one entry per term of zz (every block pair reports its own term),
 *   grad_coef_xz / grad_inscale_xz: one entry per term of xz;  meaning as in sgp_logpdf_grad.
 * The dependence on the prior variances var_x is returned as grad_var_x; chain it through
 * sgp_kernelmatrix_diag_grad for the parameters of the diagonal spec. */
int sgp_elbo_grad(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                  const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                  const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                  double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                  double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                  double* grad_inscale_xz);
/* sum_i w[i] d var_i / d theta for var = sgp_kernelmatrix_diag(spec): per term of the diagonal block
 * pairs (I, I), grad_coef[t] = sum_i w_i rs_i cs_i k_t(x_i, x'_i), grad_inscale[t] likewise with
 * coef_t dk_t/dg; entries of other terms are set to 0.  Both outputs have spec->term_ptr[last] entries. */
int sgp_kernelmatrix_diag_grad(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w, double* grad_coef,
                               double* grad_inscale);
/* Same, plus d (sum_i w_i var_i) / d (input points) per spec input (see sgp_logpdf_grad_x): non-zero
 * only for diagonal terms that read two different inputs (e.g. var of f(a x) + f(b x)). */
int sgp_kernelmatrix_diag_grad_x(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w, double* grad_coef,
                                 double* grad_inscale, double* const* grad_inputs);
/* sgp_elbo_grad plus the gradient w.r.t. the input points (see sgp_logpdf_grad_x):
 * grad_inputs_zz[k] for zz->inputs[k] (inducing points, through K(z,z)) and grad_inputs_xz[k] for
 * xz->inputs[k] (data points on the row side, inducing points on the column side, through K(x,z)).
 * The inducing points appear in both tables: add the matching arrays.  The dependence of var(f, x)
 * on x goes through grad_var_x and sgp_kernelmatrix_diag_grad_x. */
int sgp_elbo_grad_x(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                    const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                    const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                    double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                    double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                    double* grad_inscale_xz, double* const* grad_inputs_zz, double* const* grad_inputs_xz);

/* sgp_elbo_grad_x plus the gradient w.r.t. the scale vectors of function-scaled processes sigma(x) * f
 * (/root/reference/src/affine_transformations/product.jl:25-48; see sgp_logpdf_grad_xs): one array per term, NULL to skip,
 * ignored for terms without that scale.
 *   grad_rowscale_zz[t] (row_len of the term's block, inducing points): 2 sum_j Gzz_ij coef k_ij cs_j -- K(z,z) is
 *       symmetric, the column scale of a term is the row scale of its mirror term, "row side x 2" covers both roles;
 *   grad_rowscale_xz[t] (data points): sum_j Gxz_ij coef k_ij cs_j;   grad_colscale_xz[t] (inducing points, col_len):
 *       sum_i Gxz_ij coef rs_i k_ij.
 * The dependence of var(f, x) on sigma(x) goes through grad_var_x and sgp_kernelmatrix_diag_grad_xs.  grad_inputs_* may
 * be NULL here. */
int sgp_elbo_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
                     const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
                     const double* z_noise, const double* y, double* elbo_out, double* grad_y,
                     double* grad_mean, double* grad_noise, double* grad_var_x, double* grad_z_noise,
                     double* grad_coef_zz, double* grad_inscale_zz, double* grad_coef_xz,
                     double* grad_inscale_xz, double* const* grad_inputs_zz, double* const* grad_inputs_xz,
                     double* const* grad_rowscale_zz, double* const* grad_rowscale_xz,
                     double* const* grad_colscale_xz);
/* sgp_kernelmatrix_diag_grad_x plus, per diagonal term t, grad_rowscale[t][i] = w_i coef cs_i k_t(x_i, x'_i) and
 * grad_colscale[t][i] = w_i coef rs_i k_t(x_i, x'_i) (var_i = sum_t coef rs_i cs_i k_t); grad_inputs may be NULL. */
int sgp_kernelmatrix_diag_grad_xs(sgp_ctx* ctx, const sgp_cov_spec* spec, const double* w, double* grad_coef,
                                  double* grad_inscale, double* const* grad_inputs,
                                  double* const* grad_rowscale, double* const* grad_colscale);

/* ---- elbo(VFE(fz), fx, y) (A5; App. A.6; src/gp/sparse_finite_gp.jl:52-58) -----------
 * zz: symmetric spec at the inducing inputs z (M);  xz: cross spec rows = x (N), cols = z;
 * var_x: prior var(f, x) (N) -- obtain with sgp_kernelmatrix_diag;  mean_x (N; NULL==0);
 * noise_x: diagonal of Sigma_y (noise_kind SCALAR or DIAG only, as the reference requires);
 * z_noise_kind/z_noise: Sigma_z of fz (jitter).  out[0] = elbo. */
int sgp_elbo(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz, const double* var_x,
             const double* mean_x, int noise_kind, const double* noise_x, int z_noise_kind,
             const double* z_noise, const double* y, double* out);

/* posterior(VFE(fz), fx, y) (src/gp/sparse_finite_gp.jl:60-62) */
int sgp_sparse_posterior_create(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz,
                                const double* mean_x, int noise_kind, const double* noise_x,
                                int z_noise_kind, const double* z_noise, const double* y,
                                sgp_sparse_post** out);
/* cross: rows = x*, cols = z.  mean* = m* + K*z alpha; var* = k** - |B|^2 + |Le^-1 B|^2 */
int sgp_sparse_posterior_predict(sgp_sparse_post* post, const sgp_cov_spec* cross,
                                 const sgp_cov_spec* prior_ss, const double* mean_s,
                                 double* mean_out, double* var_out, double* cov_out,
                                 int64_t ldcov);
int sgp_sparse_posterior_destroy(sgp_sparse_post* post);

/* =====================================================================================
 * Device-resident variants (bench / multi-GPU host orchestration).  Pointers prefixed d_
 * are HBM addresses valid on the ctx device; `stream` is a hipStream_t passed as void*
 * (NULL == the ctx's own stream).  These never copy K to the host.
 * ===================================================================================== */
typedef struct sgp_dspec sgp_dspec; /* a cov spec whose inputs / scales live in HBM */
int sgp_dspec_create(sgp_ctx* ctx, const sgp_cov_spec* spec, sgp_dspec** out);
int sgp_dspec_destroy(sgp_dspec* ds);

/* Storage geometry of the bordered factor matrix for N points and S right-hand sides:
 * n_pad = N rounded up to 128, m_tot = n_pad + (S rounded up to 128): the matrix is
 * m_tot x n_pad column-major with ld = m_tot; rows n_pad.. hold (Y - m)' so that the
 * forward substitution L^-1 (Y - m) falls out of the factorisation itself. */
int sgp_geometry(int64_t N, int64_t S, int64_t* n_pad, int64_t* m_tot);

/* whole logpdf with everything resident: d_A is caller-provided scratch of
 * m_tot * n_pad doubles (left holding L and the solved rows); d_mean may be NULL;
 * d_noise: SCALAR -> host value *noise_host is used, DIAG -> d_noise[N];
 * d_Y: N x ncols (ld = ldy); out_host[ncols].  timings (may be NULL) receives
 * {assemble_ms, cholesky_ms, finalize_ms, trailing_update_ms_sum, n_trailing_update_launches,
 *  trailing_update_algorithmic_flops, trailing_update_busy_ms (union of the launch intervals: the
 *  look-ahead runs launches of two streams concurrently), 0} (8 doubles). */
int sgp_dev_logpdf(sgp_ctx* ctx, const sgp_dspec* ds, double* d_A, const double* d_mean,
                   int noise_kind, const double* noise_host, const double* d_noise,
                   const double* d_Y, int64_t ldy, int64_t ncols, double* out_host,
                   double* timings);

/* building blocks used by the multi-GPU host loop (one process per GPU; the host moves
 * panels between ranks with torch.distributed / RCCL) */
/* assemble columns [c0, c0+nc) (nc multiple of 128, c0 global) of K + Sigma_y, rows >= c0
 * only, into d_dst (ld = ldd, column 0 of d_dst == global column c0, row index global);
 * also writes the bordered rows from d_Y/d_mean when ncols > 0. */
int sgp_dev_assemble_cols(sgp_ctx* ctx, const sgp_dspec* ds, int64_t N, int64_t c0, int64_t nc,
                          double* d_dst, int64_t ldd, int64_t m_tot, const double* d_mean,
                          int noise_kind, const double* noise_host, const double* d_noise,
                          const double* d_Y, int64_t ldy, int64_t ncols, void* stream);
/* factor a column panel in place: d_P is m x w (ld), its top w x w block is the diagonal
 * block; on return it holds L11 and L21 = A21 L11^-T.  w multiple of 128.
 * d_logdet[0] += 2 sum log diag;  d_info: device int, set to g0 + j + 1 on the first bad pivot. */
int sgp_dev_panel_factor(sgp_ctx* ctx, double* d_P, int64_t ld, int64_t m, int64_t w,
                         int64_t g0, double* d_logdet, int* d_info, void* stream);
/* trailing update C -= P[r,:] P[c,:]' for the nc columns of d_C (ld = ldc) whose global
 * column indices are c0..c0+nc and rows c0..m_tot: d_P (ld = ldp) is the factored panel
 * with row index global offset p_row0 (d_P[0] is global row p_row0), width w. */
int sgp_dev_panel_update(sgp_ctx* ctx, const double* d_P, int64_t ldp, int64_t p_row0, int64_t w,
                         double* d_C, int64_t ldc, int64_t c0, int64_t nc, int64_t m_tot,
                         void* stream);
/* Batched form of sgp_dev_panel_update (round 4): ONE launch updates `ndst` owned packed panels, destination d with the
 * source panels srcs[src_first .. src_first + src_count) applied in that order (a tile's contraction runs over the sources
 * back to back, accumulators in registers: bit-identical to src_count calls of sgp_dev_panel_update, at the arithmetic
 * intensity of one K = sum of the widths update).  A source is a factored panel, packed: element (global row r, local
 * column k) at base[(r - row0) + k * ld]; a destination is an owned panel, packed: element (global row r, global column c)
 * at base[(r - c0) + (c - c0) * ld], rows c0 .. m_tot.  Widths, c0 multiples of 128; row0 <= c0; nsrc <= 8, ndst any
 * (split into launches of 16 destinations). */
typedef struct sgp_panel_src {
  const double* base;
  int64_t ld, row0, w;
} sgp_panel_src;
typedef struct sgp_panel_dst {
  double* base;
  int64_t ld, c0, w;
  int32_t src_first, src_count;
} sgp_panel_dst;
int sgp_dev_panel_update_batch(sgp_ctx* ctx, const sgp_panel_src* srcs, int nsrc, const sgp_panel_dst* dsts, int ndst,
                               int64_t m_tot, void* stream);
/* sum of squares of bordered row s over columns [0, nc) of a local panel set, accumulated
 * into d_out[s] (atomic-free: one block per s). */
int sgp_dev_rowsumsq(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nc, int64_t nrows,
                     double* d_out, void* stream);

/* ---- posterior on the sharded factor (AbstractGPs.posterior / mean / var / cov of the
 * PosteriorGP [EXT], Appendix A.5; the reference's call sites: /root/reference/test/gp/util.jl) ------------
 * The test points ride through the sharded factorisation as extra bordered rows: the caller sizes its
 * panels for m_tot = n_pad + 128 + pad128(n*) rows, fills rows row0 = n_pad + 128 .. with K(x*, x)
 * (sgp_dev_assemble_cross_rows, after sgp_dev_assemble_cols on the same columns), factors as usual, and
 * reads V' = K(x*, x) L^-T off the same rows:  mean* - m* = V' z  and  sum_c V'^2  per test point
 * (sgp_dev_rows_dot, z' = the observation row), V' V (sgp_dev_rows_gram): sums over columns, i.e. one
 * all-reduce across ranks.  d_dst: as in sgp_dev_assemble_cols (address of global row 0 of column c0). */
int sgp_dev_assemble_cross_rows(sgp_ctx* ctx, const sgp_dspec* cross, int64_t c0, int64_t nc,
                                double* d_dst, int64_t ldd, int64_t row0, void* stream);
/* d_sumsq[r] += sum_{c < nc} R[r,c]^2, d_dot[r] += sum_c R[r,c] z[c] for r < nrows; R = d_rows (ld),
 * z[c] = d_zrow[c * ld] */
int sgp_dev_rows_dot(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nrows, int64_t nc,
                     const double* d_zrow, double* d_sumsq, double* d_dot, void* stream);
/* d_G (nrows_pad x nrows_pad, ld ldg) += R R' over nc columns (nrows_pad multiple of 128, nc of 16) */
int sgp_dev_rows_gram(sgp_ctx* ctx, const double* d_rows, int64_t ld, int64_t nrows_pad, int64_t nc,
                      double* d_G, int64_t ldg, void* stream);

/* ---- N-sharded ELBO (reference entry /root/reference/src/gp/sparse_finite_gp.jl:52-58) --
 * The bound is a sum over data points up to one M x M factorisation: every rank turns ITS slice of the
 * data (rows of xz, var_x, mean_x, noise_x, y: host pointers, as in sgp_elbo) into a "part" -- a
 * contiguous device array of sgp_elbo_part_len(M) doubles holding  sum_n a_n a_n' (lower 128-tiles,
 * ld = m_pad + 128), A delta and four scalars -- the parts are summed across ranks with ONE all-reduce
 * (M^2 + M + 2 meaningful doubles; RCCL over xGMI), and any rank finishes: A A' + I, its Cholesky,
 * the bound.  K(z,z) + Sigma_z is factored redundantly by every rank (M^3 / 3 flops, no traffic).
 * d_part: HBM address on the ctx device, caller-owned (e.g. a torch tensor handed to all_reduce). */
int sgp_elbo_part_len(int64_t M, int64_t* len);
int sgp_dev_elbo_partial(sgp_ctx* ctx, const sgp_cov_spec* zz, const sgp_cov_spec* xz_slice,
                         const double* var_x_slice, const double* mean_x_slice, int noise_kind,
                         const double* noise_x_slice, int z_noise_kind, const double* z_noise,
                         const double* y_slice, double* d_part, int64_t part_len);
/* N_total: number of data points over all ranks; d_part is overwritten (factor of A A' + I) */
int sgp_dev_elbo_finish(sgp_ctx* ctx, int64_t M, int64_t N_total, double* d_part, double* out);

/* (The micro-benchmark and diagnosis hooks sgp_bench_* are declared in include/sthenomi_bench.h: they are what
 * bench.py, tools/ and two GPU tests use to pin the roofline peaks and the MFMA lane maps on the box -- not part of the
 * operator surface a host binds.) */

#ifdef __cplusplus
}
#endif
#endif /* STHENOMI_H */
