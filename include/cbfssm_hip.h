/*
 * cbfssm_hip.h -- C ABI of the MI355X (gfx950) implementation of the CBF-SSM ELBO hot path.
 *
 * The reference (silvanmelchior/CBF-SSM) has no FFI / plugin interface: the path is Python graph-building code over
 * TensorFlow ops.  Each entry point below therefore replaces a *set of TensorFlow op call sites*; the cited
 * file:line ranges are relative to the reference repository root.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer to contiguous row-major float64 unless marked "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates, never synchronises;
 *   - return value: 0 = OK, <0 = argument / launch error (text in cbfssm_last_error(), thread local);
 *   - numerical failure of the Cholesky (leading minor not positive definite; TensorFlow raises
 *     InvalidArgumentError there) is reported asynchronously in pack[scal + CBFSSM_SCAL_INFO] (k>0 = minor k);
 *   - chains: N = B*S particle chains, chain index c = b*S + s; trajectories are kept TIME-MAJOR on the device:
 *     x[t][c][d], y2[t][c][d] (the reference's (B,T,S,d) tensors are stride-permuted views of these).
 */
#ifndef CBFSSM_HIP_H
#define CBFSSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBFSSM_MAX_DOUT 16   /* GP output dimension (dim_x) handled by one 16-row MFMA block */
#define CBFSSM_MAX_DIN 64    /* GP input dimension dim_x + dim_u */
#define CBFSSM_MAX_M 320     /* inducing points */

/* indices into the scalar block of a GP pack */
enum {
    CBFSSM_SCAL_SIGMA2 = 0,  /* kernel variance */
    CBFSSM_SCAL_LOGDET = 1,  /* log det (K_mm + jitter I) */
    CBFSSM_SCAL_KLZ = 2,     /* prior_kl() */
    CBFSSM_SCAL_INFO = 3,    /* 0 = OK, k>0 = leading minor k not positive definite */
    CBFSSM_SCAL_COND = 4,    /* infinity-norm condition number of K_mm + jitter I: |K|_inf |K^-1|_inf */
    CBFSSM_SCAL_JITTER = 5,  /* the jitter this pack was prepared with */
    CBFSSM_SCAL_COUNT = 16
};

/* how GPModel.predict is evaluated by the pass / predict kernels (cbfssm_pack_layout.gp_form) */
enum {
    CBFSSM_GP_FORM_DENSE = 0,   /* A2 = K^-1 k as one dense product, fvar_0 = sigma^2 - k.A2                      */
    CBFSSM_GP_FORM_TRI = 1      /* the reference's own order (gp_tf.py:137-145): A = L^-1 k, fvar_0 = sigma^2 - |A|^2,
                                   A2 = L^-T A, as two triangular products -- same multiply-adds, no cancellation
                                   of k.(K^-1 k) against sigma^2 on an ill-conditioned K_mm                        */
};

/* Offsets (in doubles) of the sections of one GP pack; filled by cbfssm_gp_pack_layout (host struct). */
typedef struct {
    int64_t total;    /* doubles to allocate */
    int64_t Bp;       /* [NBLK][KS][64]  K^-1 as MFMA A-operand image, zero padded to Mp = 16*NBLK           */
    int64_t Zp;       /* [NBLK][DK][64]  Z/lengthscale as MFMA A-operand image                               */
    int64_t cz;       /* [Mp]            -0.5|z~|^2 + log sigma^2 (padding rows: -1e30)                      */
    int64_t muA;      /* [NBLK][4][64]   zeta_mean as A-operand image of the epilogue product                */
    int64_t s2A;      /* [NBLK][4][64]   zeta_var  as A-operand image                                        */
    int64_t invl;     /* [Dp]            1/lengthscale, zero padded                                          */
    int64_t scal;     /* [CBFSSM_SCAL_COUNT]                                                                 */
    int64_t Kmm;      /* [M][M]          kernel matrix, no jitter          (gp_tf.py:129)                    */
    int64_t L;        /* [M][M]          lower Cholesky factor             (gp_tf.py:130)                    */
    int64_t Kinv;     /* [M][M]          (K_mm + jitter I)^-1                                                */
    int64_t Linvt;    /* [M][M]          L^-T (upper triangular), by-product of the factorisation            */
    int64_t Zs;       /* [M][D]          Z / lengthscale                                                     */
    int64_t muB;      /* [NBLK][4][64]   zeta_mean as A-operand image A[row m][k = d]      (adjoint kernels)        */
    int64_t s2B;      /* [NBLK][4][64]   zeta_var, same image                                                        */
    int64_t ZT;       /* [NBLK][JB][4][64] (Z/lengthscale)^T as A-operand image A[row j][k = m]; row D = ones        */
    int64_t rev_slab; /* doubles of one adjoint partial slab (0: no adjoint kernel for this tile height)            */
    int64_t work;     /* [M][M|1]        factorisation workspace when M is too large for LDS                        */
    int64_t Wp;       /* [NBLK][KS][64]  W = L^-1 (lower triangular) as MFMA A-operand image      (gp_tf.py:137)         */
    int64_t WTp;      /* [NBLK][KS][64]  W^T = L^-T as MFMA A-operand image                        (gp_tf.py:145)         */
    int32_t M, D, Do, NBLK, DK, Mp, Dp, KS, JB, rev_stash;   /* rev_stash: 1 = adjoint runs in stash mode (M > 112) */
    int32_t gp_form;  /* CBFSSM_GP_FORM_*: set by the caller after cbfssm_gp_pack_layout (which fills in DENSE); read by
                         cbfssm_gp_predict_f64 and the pass kernels.  Both forms read the same pack.               */
    int32_t reserved;
} cbfssm_pack_layout;

/* Problem description shared by the pass kernels (host struct, passed by pointer). */
typedef struct {
    int32_t B, S, T;            /* sequences, particles per sequence, time steps                             */
    int32_t dim_x, dim_u, dim_y;
    int32_t M;                  /* inducing points                                                           */
    int32_t recog_len;          /* config['recog_len']      cbfssm.py:120,190                                */
    int32_t condition;          /* feed of model.condition  cbfssm.py:227                                    */
    int32_t half;               /* 1: CBFSSMHALF forward pass (cbfssmhalf.py:117-172), 0: CBFSSM                  */
    int32_t group0, ngroups;    /* chain-group split: this call handles the 16-chain groups [group0, group0+ngroups) of
                                   ceil(B*S/16); ngroups = 0 means all.  Chains never interact, so a pass may be issued
                                   in pieces on different streams; partial/slab buffers keep their full-size layout.  */
    double k_factor;            /* config['k_factor']       cbfssm.py:191,214                                */
} cbfssm_problem;

const char* cbfssm_last_error(void);
int cbfssm_version(void);

/* Layout of a GP pack for (M, D = dim_x + dim_u, Do).  Host only. */
int cbfssm_gp_pack_layout(int M, int D, int Do, cbfssm_pack_layout* out);

/*
 * K_mm and its Cholesky factor.
 * Replaces: RBF.K(zeta_pos) (gp_tf.py:33-49,129) and cast_cholesky/_jitter_cholesky (gp_tf.py:52-65,130).
 *   Z (M,D), lengthscales (D), variance (1)  ->  Kmm (M,M) without jitter, L (M,M) lower (upper part zero),
 *   info (1 double: 0 or the failing leading minor).  work: >= M*D + M*(M+1) doubles.
 */
int cbfssm_kmm_chol_f64(int M, int D, const double* Z, const double* lengthscales, const double* variance,
                        double jitter, double* Kmm, double* L, double* info, double* work, void* stream);

/*
 * Everything that is loop invariant for one GPModel, once per ELBO evaluation.
 * Replaces: GPModel.__init__ tail (gp_tf.py:129-130), the operand preparation of GPModel.predict
 * (gp_tf.py:134-159: X/lengthscales, K^-1 instead of two triangular solves) and GPModel.prior_kl (gp_tf.py:163-172).
 *   inputs are the *constrained* values: Z (M,D), lengthscales (D), variance (1), zeta_mean (M,Do), zeta_var (M,Do)
 *   pack: layout.total doubles.
 */
int cbfssm_gp_prepare_f64(const cbfssm_pack_layout* layout, const double* Z, const double* lengthscales,
                          const double* variance, const double* zeta_mean, const double* zeta_var,
                          double jitter, double* pack, void* stream);

/* The same for two GPModels in one launch (gp_f and gp_b of CBFSSM._setup_vars, cbfssm.py:30-48): one workgroup each. */
int cbfssm_gp_prepare2_f64(const cbfssm_pack_layout* layout0, const double* Z0, const double* lengthscales0,
                           const double* variance0, const double* zeta_mean0, const double* zeta_var0, double* pack0,
                           const cbfssm_pack_layout* layout1, const double* Z1, const double* lengthscales1,
                           const double* variance1, const double* zeta_mean1, const double* zeta_var1, double* pack1,
                           double jitter, void* stream);

/*
 * GPModel.predict(Xnew) (gp_tf.py:132-161): X (npts, D) -> fmean (npts, Do), fvar (npts, Do).
 */
int cbfssm_gp_predict_f64(const cbfssm_pack_layout* layout, const double* pack, const double* X, int64_t npts,
                          double* fmean, double* fvar, void* stream);

/*
 * The remaining pieces of gp_tf.py as stand-alone calls (none of them is on a time loop):
 *   cbfssm_cholesky_f64      cast_cholesky / _jitter_cholesky of a GIVEN matrix (gp_tf.py:52-65): mat (M,M) symmetric ->
 *                            L (M,M) lower with mat + jitter I = L L^T, info (1 double: 0 or the failing leading minor);
 *                            work: >= M*(M+1) doubles.  The blocked MFMA factorisation of cbfssm_gp_prepare_f64.
 *   cbfssm_rbf_k_f64         RBF.K(X, X2) (gp_tf.py:33-49): X (n,D), X2 (m,D) -> out (n,m).
 *   cbfssm_gp_predict_fullq_f64   conditional() with a full-matrix q_sqrt (gp_tf.py:68-100, the q_sqrt.ndims == 3 branch):
 *                            the pack is prepared with f as zeta_mean and zeta_var = 0; q_sqrt (Do,M,M) lower triangular
 *                            per output dimension; fvar += sum_j (q_sqrt[d]^T A2)_j^2.  work: cbfssm_gp_predict_fullq_work_elems
 *                            doubles (the A2 tiles).  (The q_sqrt.ndims == 2 branch IS cbfssm_gp_predict_f64 with
 *                            zeta_var = q_sqrt^2; q_sqrt = None is zeta_var = 0.)
 */
int cbfssm_cholesky_f64(int M, const double* mat, double jitter, double* L, double* info, double* work, void* stream);
int cbfssm_rbf_k_f64(int n, int m, int D, const double* X, const double* X2, const double* lengthscales,
                     const double* variance, double* out, void* stream);
int64_t cbfssm_gp_predict_fullq_work_elems(const cbfssm_pack_layout* layout, int64_t npts);
int cbfssm_gp_predict_fullq_f64(const cbfssm_pack_layout* layout, const double* pack, const double* q_sqrt, const double* X,
                                int64_t npts, double* fmean, double* fvar, double* work, void* stream);

/*
 * Both backward (recognition) runs, CBFSSM._backward/_backward_run/_backward_body (cbfssm.py:84-158).
 *   u (B,T,dim_u), y (B,T,dim_y), hid_b (2,T,N), eps_b (2,T,N), var_x (dim_x)
 *   -> y2 (T,N,dim_x-dim_y)  [every t written by exactly one run, cbfssm.py:123-128,151]
 *      h_all (2,T,N,dim_x-dim_y) or NULL: every step's output of both runs (kept for the adjoint)
 *      fmv_b (2,T,N,dim_x-dim_y,2) or NULL: every step's (fmean, fvar) after residual and process noise (kept for
 *      the adjoint, which then needs no recomputation of the predictive products)
 *      a2s_b (cbfssm_saved_a2_elems(p, L, 1) doubles) or NULL: every step's A2 = K_mm^-1 K_mn tiles -- and, for tile
 *      heights up to seven row blocks (M <= 112), its kernel tile K_mn next to it -- kept for the adjoint (which
 *      otherwise recomputes them: one M x M x 16 product, resp. the kernel tile's MFMAs and exponentials, per step less)
 *      ent_part (n_ent_part doubles): per-workgroup partial sums of 0.5*sum(log(2 pi e) + log fvar) over the
 *      written steps (cbfssm.py:154-156); their sum is `entropy` (cbfssm.py:99).
 *   cbfssm_backward_pass_partials(problem) gives n_ent_part.
 */
int64_t cbfssm_backward_pass_partials(const cbfssm_problem* p);
int cbfssm_backward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_b, const double* pack_b,
                             const double* var_x, const double* u, const double* y, const double* hid_b,
                             const double* eps_b, double* y2, double* h_all, double* fmv_b, double* a2s_b,
                             double* ent_part, void* stream);

/*
 * Forward (filter) pass, CBFSSM._forward/_forward_body (cbfssm.py:160-237).
 *   y2 (T,N,dim_x-dim_y) from the backward pass, eps_f (T-1,N), var_x (dim_x), var_y (dim_x)
 *   -> x (T,N,dim_x)   [x[0] = y_tilde[0], cbfssm.py:168]
 *      fmv_f (T-1,N,dim_x,2) or NULL: every step's (fmean, fvar), kept for the adjoint
 *      a2s_f (cbfssm_saved_a2_elems(p, L, 0) doubles) or NULL: every step's A2 (and, M <= 112, kernel) tiles, kept
 *      for the adjoint
 *      kl_part (n_kl_part doubles): per-workgroup partial sums of kl_reg (cbfssm.py:232-235); sum = kl_x.
 */
int64_t cbfssm_forward_pass_partials(const cbfssm_problem* p);

/* Doubles in the optional saved-tile buffer of the backward (backward != 0) or forward pass: one record per step and
 * 16-chain group, T*2 resp. T-1 steps; a record is the M_pad x 16 tile A2 and, when M <= 112, the kernel tile K_mn of
 * the same shape behind it.  The buffer is opaque to the caller: the passes write it, their adjoints read it (no
 * reference counterpart: TF keeps its forward activations for tf.gradients the same way, base_model.py:34-36). */
int64_t cbfssm_saved_a2_elems(const cbfssm_problem* p, const cbfssm_pack_layout* layout, int backward);
int cbfssm_forward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const double* pack_f,
                            const double* var_x, const double* var_y, const double* u, const double* y,
                            const double* y2, const double* eps_f, double* x, double* fmv_f, double* a2s_f,
                            double* kl_part, void* stream);

/*
 * CBFSSMHALF (cbfssm/model/cbfssmhalf.py:97-172): the forward pass with x_0 = recognition-model output x0 (B,dim_x),
 * Kalman update on the first dim_y state dims only (var_y has dim_y entries), no backward runs.  problem->half must be 1.
 * The adjoint also returns gx0 (N,dim_x) = d loss / d x_0 per particle chain (sum over the S particles of a sequence is
 * the gradient of the recognition output).  t range / stash arguments as in cbfssm_forward_pass_bwd_ex_f64.
 */
int cbfssm_half_forward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const double* pack_f,
                                 const double* var_x, const double* var_y, const double* u, const double* y,
                                 const double* x0, const double* eps_f, double* x, double* fmv_f, double* a2s_f,
                                 double* kl_part, void* stream);
int cbfssm_half_forward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const double* pack_f,
                                     const double* var_x, const double* var_y, const double* u, const double* y,
                                     const double* eps_f, const double* x, const double* fmv_f, const double* a2s_f, double cL,
                                     double* gx0, double* gpart, int t_hi, int t_lo, double* gx_carry, double* stash_a,
                                     double* stash_k, int64_t stash_ld, void* stream);

/*
 * Log-likelihood and predictive moments, CBFSSM._build_loss (cbfssm.py:245-251) and _build_prediction
 * (cbfssm.py:264-269).
 *   x (T,N,dim_x), y (B,T,dim_y), var_y (dim_x)
 *   -> ll_part (cbfssm_loglik_partials(problem) doubles, laid out [block][dim_y]: per-workgroup, per-dimension partial
 *      sums of the log-likelihood in a fixed order; their total is `loglik`, their per-dimension totals drive the
 *      gradient with respect to var_y), pred_mean (B,T,dim_y), pred_var (B,T,dim_y),
 *      int_mean (B,T,dim_x) / int_var (B,T,dim_x) or NULL.
 */
int64_t cbfssm_loglik_partials(const cbfssm_problem* p);
int cbfssm_loglik_moments_f64(const cbfssm_problem* p, const double* var_y, const double* y, const double* x,
                              double* ll_part, double* pred_mean, double* pred_var, double* int_mean,
                              double* int_var, void* stream);

/*
 * ELBO combination, CBFSSM._build_loss (cbfssm.py:257-262): deterministic sums of the partial buffers.
 *   out[0..7] = loglik, kl_x, entropy, kl_z_f, kl_z_b, elbo, loss, info (max of the two packs' info).
 */
int cbfssm_elbo_combine_f64(const cbfssm_problem* p, double lambda0, double lambda1,
                            const double* ll_part, int64_t n_ll, const double* kl_part, int64_t n_kl,
                            const double* ent_part, int64_t n_ent, const double* scal_f, const double* scal_b,
                            double* out, void* stream);

/*
 * ---- adjoint (reverse-mode) entry points: what tf.train.AdamOptimizer.minimize differentiates (cbfssm.py:273-275).
 *
 * Slab layout of the parameter adjoints of one GP (layout->rev_slab doubles; C = MFMA accumulator layout,
 * element [blk][r][lane] -> row 16*blk_row + (lane>>4) + 4r, col 16*blk_col + (lane&15)):
 *   [0, NBLK*256)                       d loss / d zeta_mean   (Mp x 16)
 *   [NBLK*256, 2*NBLK*256)              d loss / d zeta_var    (Mp x 16)
 *   then NBLK*NBLK*256                  d loss / d K^-1        (Mp x Mp), data terms only
 *   then NBLK*JB*256                    d loss / d (Z/ls)      (Mp x 16*JB): column D holds the row sums of Ebar,
 *                                       the caller subtracts (Z/ls) o rowsum
 *   then 128 scalars: [0,16) d/d var_x (by state dim), [16,32) d/d var_y, [32,32+16*JB) sum_n xbar~ x~ by input
 *                     row (lengthscale adjoint of the inputs), [96] d/d sigma^2 (direct), [97] d/d log sigma^2
 */
int64_t cbfssm_rev_workgroups(const cbfssm_problem* p, int backward_runs);

/*
 * Adjoint of cbfssm_forward_pass_f64 (reverse of the tf.while_loop in CBFSSM._forward, cbfssm.py:176-237) including
 * the log-likelihood's pull on x (cbfssm.py:245-251).
 *   x, y2, eps_f, fmv_f (and a2s_f, or NULL to recompute A2) as saved by the forward evaluation; cL = loss_factors[0]/S.
 *   -> gy2 (T,N,dim_x-dim_y): d loss / d y2 ; gpart: cbfssm_rev_workgroups(p,0) slabs of layout_f->rev_slab doubles.
 */
int cbfssm_forward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const double* pack_f,
                                const double* var_x, const double* var_y, const double* u, const double* y,
                                const double* y2, const double* eps_f, const double* x, const double* fmv_f,
                                const double* a2s_f, double cL, double* gy2, double* gpart, void* stream);

/*
 * Adjoint of cbfssm_backward_pass_f64 (both runs; reverse of the tf.while_loops in CBFSSM._backward_run,
 * cbfssm.py:107-158).  h_all, fmv_b (and a2s_b, or NULL) as saved by the forward evaluation, gy2 from cbfssm_forward_pass_bwd_f64,
 * cE = loss_factors[1]/S.  -> gpart: cbfssm_rev_workgroups(p,1) slabs of layout_b->rev_slab doubles.
 */
int cbfssm_backward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_b, const double* pack_b,
                                 const double* var_x, const double* u, const double* y, const double* hid_b,
                                 const double* eps_b, const double* h_all, const double* fmv_b, const double* a2s_b,
                                 const double* gy2, double cE, double* gpart, void* stream);

/*
 * General forms of the two adjoint passes: a time range per launch, and "stash mode" for tile heights whose
 * K^-1-adjoint accumulator does not fit the VGPR file (layout->rev_stash == 1, M > 112; the slab then has no
 * d/dK^-1 block).  In stash mode every step writes the two MFMA operand images of d loss / d K^-1 += A2bar K^T to
 * stash_a / stash_k: [slot][NBLK][4][64] doubles each, slot = workgroup * steps_per_workgroup + step, stash_ld = 16 x
 * (slots the buffer can hold) -- "columns" below are 16 per slot -- and cbfssm_stash_contract_f64 contracts them after
 * the launch.
 *   forward:  steps t = t_hi .. t_lo (descending; full pass: T-2 .. 0).  A launch that does not start at T-2 reads the
 *             adjoint of x_{t_hi+1} from gx_carry (N,dim_x); a launch that does not end at 0 writes it there.
 *             columns used: groups * (t_hi - t_lo + 1) * 16.
 *   backward: resample-to-resample segments [seg0, seg1) of both runs (cbfssm_bwd_segments(p) in total), split over
 *             nchunk independent workgroup sets (grid.z).  gpart: groups * 2 * nchunk slabs.
 *             columns used: groups * 2 * nchunk * ceil((seg1-seg0)/nchunk) * 2*recog_len * 16.
 */
int cbfssm_bwd_segments(const cbfssm_problem* p);
int cbfssm_forward_pass_bwd_ex_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const double* pack_f,
                                   const double* var_x, const double* var_y, const double* u, const double* y,
                                   const double* y2, const double* eps_f, const double* x, const double* fmv_f,
                                   const double* a2s_f, double cL, double* gy2, double* gpart, int t_hi, int t_lo,
                                   double* gx_carry, double* stash_a, double* stash_k, int64_t stash_ld, void* stream);
int cbfssm_backward_pass_bwd_ex_f64(const cbfssm_problem* p, const cbfssm_pack_layout* layout_b, const double* pack_b,
                                    const double* var_x, const double* u, const double* y, const double* hid_b,
                                    const double* eps_b, const double* h_all, const double* fmv_b, const double* a2s_b,
                                    const double* gy2, double cE, double* gpart, int seg0, int seg1, int nchunk,
                                    double* stash_a, double* stash_k, int64_t stash_ld, void* stream);

/* out[i] = sum over the nwg slabs, in a fixed order (bitwise reproducible).  gpart must have room for
 * nwg + CBFSSM_REDUCE_SPLIT slabs: the tail is scratch for the first of the two reduction stages. */
#define CBFSSM_REDUCE_SPLIT 32
int cbfssm_reduce_partials_f64(double* gpart, int64_t slab, int64_t nwg, double* out, void* stream);

/*
 * Stash mode (layout->rev_stash, M > 112): the `*_bwd_ex_f64` launches write, per (workgroup, step) slot, the two MFMA
 * operand images of  d loss / d K^-1 += A2bar K^T  (stash_a: A2bar^T, stash_k: K^T; [slot][NBLK][4][64] doubles each, i.e.
 * 16 NBLK x stash_ld doubles per buffer with stash_ld = 16 x slots).  This contracts `nslots` slots and ADDS the SYMMETRIC
 * PART of the result, (B + B^T) / 2 with B = sum A2bar K^T, to ginv_image, an MFMA C-layout image [NBLK][NBLK][4][64] (the
 * layout of the in-register variant's slab section; row = 16 rb + (lane >> 4) + 4 r, column = 16 cb + (lane & 15)).  The
 * symmetric part is all the train tail uses (d loss / d K_mm = -K^-1 (.) K^-1 contracted with a symmetric dK_mm/dtheta), and
 * it halves the accumulator: the kernel holds the lower-triangular blocks of A2bar K^T + K A2bar^T only (cbfssm_contract.hip).
 * CBFSSM_CONTRACT_FULL=1 (measurement switch) adds B itself.  work: cbfssm_stash_contract_work_elems doubles.
 */
int64_t cbfssm_stash_contract_work_elems(const cbfssm_pack_layout* layout, int64_t nslots);
int cbfssm_stash_contract_f64(const cbfssm_pack_layout* layout, const double* stash_a, const double* stash_k, int64_t nslots,
                              double* work, double* ginv_image, void* stream);

/*
 * float32-ARITHMETIC variant of the forward evaluation and of the adjoint (BASELINE.json configs[4]; the reference's model dtype argument,
 * cbfssm.py:12: `CBFSSM(config, dtype=tf.float32)`).  As in the reference's float32 mode the Cholesky of K_mm is computed
 * in float64 and cast (gp_tf.py:57-65): the operands are the float64 pack of cbfssm_gp_prepare_f64 re-packed as float32
 * MFMA images; kernel tile, exp, the K^-1 K contraction (v_mfma_f32_16x16x4_f32) and the step epilogues are float32.
 * Storage stays float64: u, y, noise, trajectories and partial sums are the buffers of the float64 entry points.
 *   cbfssm_pack_f32_elems   floats of a float32 pack for this layout (host)
 *   cbfssm_gp_pack_f32      float64 pack -> float32 pack (after every cbfssm_gp_prepare*_f64)
 *   cbfssm_gp_predict_f32   GPModel.predict (gp_tf.py:132-161)
 *   cbfssm_backward_pass_f32 / cbfssm_forward_pass_f32   CBFSSM._backward / _forward (cbfssm.py:84-237); the partial-sum
 *                           buffers have cbfssm_backward_pass_partials / cbfssm_forward_pass_partials entries; feed
 *                           cbfssm_loglik_moments_f64 and cbfssm_elbo_combine_f64 as usual.
 *   cbfssm_gp_pack_bf16     the same pack with the K^-1 operand rounded to bfloat16; the _f32 passes given such a pack
 *                           also round the kernel tile to bfloat16, i.e. they evaluate the K^-1 K contraction with
 *                           bf16 operands and float32 accumulation (on the float32 MFMA: bf16 x bf16 products are exact in
 *                           float32).  A precision probe for the fp32-vs-bf16 tolerance sweep, not a throughput path.
 *   fmv_* / h_all (may be NULL): what the float32 adjoint below reads of the forward evaluation -- every step's
 *                           (fmean, fvar) and, for the backward runs, every step's output -- in the float64 buffers of
 *                           cbfssm_backward_pass_f64 / cbfssm_forward_pass_f64 (same layout; float32 values).
 *
 * float32 ADJOINT (what `minimize` differentiates when the model dtype is float32, cbfssm.py:12,273-275): the reverse
 * sweeps on v_mfma_f32_16x16x4_f32 with float32 accumulation over the time steps; the kernel tile and A2 = K^-1 k are
 * read back when the passes kept them (a2s_*), recomputed otherwise.  The partial slabs leave as float64 in the layout of the float64 adjoint for NON-stash tile
 * heights -- [mubar | s2bar | G (NBLK x NBLK C-layout images) | Zbar | small] = cbfssm_rev32_slab_elems doubles per
 * workgroup, cbfssm_rev_workgroups workgroups -- so cbfssm_reduce_partials_f64 and cbfssm_train_tail_f64 (the K_mm ->
 * Cholesky -> K^-1 adjoint stays float64, as the reference keeps the Cholesky in float64, gp_tf.py:57-65) take them, with ONE
 * difference: the matrix section holds G = sum (K^-1 A2bar) A2^T = K^-1 (d loss / d K^-1) K^-1, the data part of the K_mm
 * adjoint itself (a float32 accumulator of d loss / d K^-1 would have its rounding multiplied by K^-1 from both sides in the
 * tail).  cbfssm_train_tail_g_f64 takes the section as it is: g_mode = 1 for tile heights up to 10 row blocks (the full
 * matrix G), g_mode = 2 from 13 row blocks (M > 160: two row blocks per wave) -- there the kernel accumulates only the
 * lower-triangular 16 x 16 blocks of S = C A2^T + A2 C^T, C = K^-1 A2bar (diagonal blocks: C A2^T; blocks above the diagonal
 * are never written), the symmetric part (S + S^T) / 2 of G being all that d loss / d K_mm uses: one pass over the time loop at
 * every tile height.  For tile heights above 112 rows hand that section to the tail as gB_dense_* with gB_ld = 0.
 */
int64_t cbfssm_pack_f32_elems(const cbfssm_pack_layout* layout);
int cbfssm_gp_pack_f32(const cbfssm_pack_layout* layout, const double* pack, float* pack32, void* stream);
int cbfssm_gp_pack_bf16(const cbfssm_pack_layout* layout, const double* pack, float* pack32, void* stream);
int cbfssm_gp_predict_f32(const cbfssm_pack_layout* layout, const float* pack32, const double* X, int64_t npts,
                          double* fmean, double* fvar, void* stream);
int cbfssm_backward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_b, const float* pack32_b,
                             const double* var_x, const double* u, const double* y, const double* hid_b,
                             const double* eps_b, double* y2, double* h_all, double* fmv_b, float* a2s_b, double* ent_part,
                             void* stream);
int cbfssm_forward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const float* pack32_f,
                            const double* var_x, const double* var_y, const double* u, const double* y,
                            const double* y2, const double* eps_f, double* x, double* fmv_f, float* a2s_f, double* kl_part,
                            void* stream);
/* a2s_*: NULL, or cbfssm_saved_a2_f32_elems floats -- every step's [A2 | kernel tile] accumulator registers, kept for the float32
 * adjoint, which then reads them back instead of recomputing the kernel tile and the K^-1 K product (2 F instead of 3 F per GP
 * evaluation; C3: 3.4 GB for both GPs).  Opaque to the host. */
int64_t cbfssm_saved_a2_f32_elems(const cbfssm_problem* p, const cbfssm_pack_layout* layout, int backward_runs);
/* The forward pass of the forward-only variants (problem->half = 1: x_0 from the recognition model, Kalman update on the observed
 * dims only; cbfssmhalf.py:117-172, prssm.py:96-118) in float32 arithmetic, and its adjoint (-> gx0 (N, dim_x): d loss / d x_0
 * per chain): the float32 counterparts of cbfssm_half_forward_pass_f64 / _bwd_f64. */
int cbfssm_half_forward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const float* pack32_f,
                                 const double* var_x, const double* var_y, const double* u, const double* y,
                                 const double* x0, const double* eps_f, double* x, double* fmv_f, float* a2s_f,
                                 double* kl_part, void* stream);
int cbfssm_half_forward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const float* pack32_f,
                                     const double* var_x, const double* var_y, const double* u, const double* y,
                                     const double* eps_f, const double* x, const double* fmv_f, const float* a2s_f, double cL,
                                     double* gx0, double* gpart, void* stream);
int64_t cbfssm_rev32_slab_elems(const cbfssm_pack_layout* layout);
int cbfssm_forward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_f, const float* pack32_f,
                                const double* var_x, const double* var_y, const double* u, const double* y,
                                const double* y2, const double* eps_f, const double* x, const double* fmv_f, const float* a2s_f,
                                double cL, double* gy2, double* gpart, void* stream);
int cbfssm_backward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* layout_b, const float* pack32_b,
                                 const double* var_x, const double* u, const double* y, const double* hid_b,
                                 const double* eps_b, const double* h_all, const double* fmv_b, const float* a2s_b,
                                 const double* gy2, double cE, double* gpart, void* stream);

/*
 * ---- once-per-step tail of a train step ------------------------------------------------------------------------------
 * The twelve trainable tensors of CBFSSM._setup_vars (cbfssm.py:30-58) as ONE flat float64 vector, in this order:
 *   f.zeta_pos (M,D) f.zeta_mean (M,dim_x) f.zeta_var_unc (M,dim_x) f.variance_unc (1) f.lengthscales_unc (D)
 *   b.zeta_pos (M,D) b.zeta_mean (M,dob)   b.zeta_var_unc (M,dob)   b.variance_unc (1) b.lengthscales_unc (D)
 *   var_x_unc (dim_x) var_y_unc (dim_x)                      D = dim_x + dim_u, dob = dim_x - dim_y
 */
typedef struct cbfssm_param_layout {
    int64_t off[12];
    int64_t total;
    int32_t M, D, dim_x, dim_y;
} cbfssm_param_layout;

int cbfssm_param_layout_init(int M, int dim_x, int dim_u, int dim_y, cbfssm_param_layout* out);

/* Positivity transform softplus(x) + 1e-10 of every *_unc tensor (tf_transform.py:19-21); the other tensors are copied.
 * cflat has the layout of pflat. */
int cbfssm_constrain_f64(const cbfssm_param_layout* pl, const double* pflat, double* cflat, void* stream);

/* Scratch doubles cbfssm_train_tail_f64 needs. */
int64_t cbfssm_train_tail_work_elems(const cbfssm_pack_layout* layout_f, const cbfssm_pack_layout* layout_b);

/*
 * What tf.gradients (base_model.py:34-36) does after the time loops: the adjoint of GPModel.__init__ / prior_kl
 * (K_mm -> Cholesky -> K^-1, gp_tf.py:33-65,129-130,163-172) for gp_f and gp_b from the reduced adjoint slabs, and the
 * chain through the positivity transforms.  red = [slab_f | slab_b | loglik, kl_x, entropy, dloss/dvar_y[dim_y]] as
 * produced by cbfssm_reduce_partials_f64 (summed over the ranks for a multi-GPU step); gB_dense_*: the K^-1 adjoints
 * of stash mode (M > 112), NULL otherwise -- dense [.][gB_ld] matrices, or with gB_ld = 0 the C-layout images of
 * cbfssm_stash_contract_f64.  Writes d loss / d (flat parameter vector) to gflat.
 */
int cbfssm_train_tail_f64(const cbfssm_param_layout* pl, const cbfssm_pack_layout* layout_f, const double* pack_f,
                          const cbfssm_pack_layout* layout_b, const double* pack_b, const double* red,
                          const double* gB_dense_f, const double* gB_dense_b, int64_t gB_ld, const double* pflat,
                          const double* cflat, double* work, double* gflat, void* stream);
/* The same with the matrix section in another form (what the float32 adjoint leaves, cbfssm_*_pass_bwd_f32):
 * g_mode 0: d loss / d K^-1 (= cbfssm_train_tail_f64); 1: G = K^-1 (d loss / d K^-1) K^-1, the data part of d loss / d K_mm
 * itself, as a full matrix; 2: the lower-triangular 16 x 16 blocks of G + G^T (diagonal blocks: of G), blocks above the
 * diagonal ignored.  Only the symmetric part of G enters (dK_mm/dtheta is symmetric, gp_tf.py:33-49). */
int cbfssm_train_tail_g_f64(const cbfssm_param_layout* pl, const cbfssm_pack_layout* layout_f, const double* pack_f,
                            const cbfssm_pack_layout* layout_b, const double* pack_b, const double* red,
                            const double* gB_dense_f, const double* gB_dense_b, int64_t gB_ld, int g_mode, const double* pflat,
                            const double* cflat, double* work, double* gflat, void* stream);

/*
 * The same tail for the forward-only variants -- CBFSSMHALF (cbfssmhalf.py:20-47,174-199) and the PR-SSM baseline
 * (prssm.py:28-47,81-82,96): ONE GP (gp_f), var_y with dim_y entries.  Flat vectors (pflat unconstrained, cflat constrained,
 * gflat the gradient) in the order zeta_pos [M][D] | zeta_mean [M][Do] | zeta_var [M][Do] | variance [1] | lengthscales [D, or
 * 1 with shared_ls: PR-SSM's one lengthscale for all input dimensions, prssm.py:40] | var_x [Do] | var_y [dim_y].
 * red = [slab | loglik, kl_x, entropy (0), dloss/dvar_y[dim_y]] (cbfssm_reduce_partials_f64 + cbfssm_data_tail_f64).
 * pack_kl: NULL, or (PR-SSM) the pack of the same parameters prepared with jitter 0 -- the prior-KL terms then use the
 * jitter-free K_mm^-1 as the reference factorises the prior without jitter.  g_mode: the form of the matrix section, as in
 * cbfssm_train_tail_g_f64 (0 after the float64 adjoint, 1 / 2 after cbfssm_half_forward_pass_bwd_f32).
 * work: cbfssm_train_tail_half_work_elems doubles.
 */
int64_t cbfssm_train_tail_half_work_elems(const cbfssm_pack_layout* layout);
int cbfssm_train_tail_half_f64(const cbfssm_pack_layout* layout, const double* pack, const double* pack_kl, int shared_ls,
                               const double* red, const double* gB_dense, int64_t gB_ld, int g_mode, int dim_y,
                               const double* pflat, const double* cflat, double* work, double* gflat, void* stream);

/*
 * Recognition model of the forward-only variants: x_0 = dense(GRUCell(16)(the first recog_len steps of [u, y], reversed))
 * (cbfssm/model/cbfssmhalf.py:82-93, cbfssm/model/prssm.py:132-141; TF-1.8 GRUCell gate layout).  params: the six tensors
 * behind each other -- gate kernel [dim_u + dim_y + 16][32], gate bias [32], candidate kernel [dim_u + dim_y + 16][16],
 * candidate bias [16], dense kernel [16][dim_x], dense bias [dim_x] = cbfssm_gru_recog_param_elems doubles.  One wave per
 * sequence; act (cbfssm_gru_recog_act_elems doubles, or NULL when no gradient follows) keeps every step's h, r, u, c.
 * The backward call takes d loss / d x_0 per sequence (the adjoint pass's gx0 summed over the particles) and writes one
 * gradient slab of cbfssm_gru_recog_param_elems doubles per sequence (gpart: room for B + CBFSSM_REDUCE_SPLIT slabs);
 * cbfssm_reduce_partials_f64(gpart, elems, B, out) sums them in a fixed order.
 */
int64_t cbfssm_gru_recog_param_elems(int dim_u, int dim_y, int dim_x);
int64_t cbfssm_gru_recog_act_elems(int B, int recog_len);
int cbfssm_gru_recog_f64(int B, int T, int dim_u, int dim_y, int dim_x, int recog_len, const double* u, const double* y,
                         const double* params, double* x0, double* act, void* stream);
int cbfssm_gru_recog_bwd_f64(int B, int T, int dim_u, int dim_y, int dim_x, int recog_len, const double* u, const double* y,
                             const double* params, const double* act, const double* gx0, double* gpart, void* stream);

/* The rank-local data terms of the flat reduce buffer: tail[0..2] = loglik, kl_x, entropy (from the ELBO combination's
 * out[0..2]); tail[3 + d] = d loss / d var_y[d] through the log-likelihood (cbfssm.py:245-251), d < dim_y, from the
 * per-dimension totals of ll_part (cbfssm_loglik_moments_f64).  cL = loss_factors[0] / S. */
int cbfssm_data_tail_f64(const cbfssm_problem* p, const double* var_y, const double* ll_part, const double* out8, double cL,
                         double* tail, void* stream);

/* tf.train.AdamOptimizer(learning_rate).minimize (cbfssm.py:273-275; TF 1.8 rule): t_dev (one double on the device)
 * is incremented, then lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), p -= lr_t m / (sqrt(v) + eps). */
int cbfssm_adam_step_f64(int64_t n, double* pflat, const double* gflat, double* m, double* v, double* t_dev, double lr,
                         double beta1, double beta2, double eps, void* stream);

/* ---- noise.  The reference draws its standard normals inside the graph (tf.random_normal: cbfssm.py:134,149,209;
 * cbfssmhalf.py:142; prssm.py:126), one per (b, s) chain and step; the passes above take them as arrays (hid_b, eps_b, eps_f).
 * cbfssm_normal_f64 fills such an array on the device: Philox4x32-10 (counter = element pair index, key = seed) + Box-Muller in
 * float64.  out[i] is a pure function of (seed, offset + i) -- a caller that advances `offset` by the elements it has drawn
 * gets one reproducible stream per seed, however the draws are split over calls, streams or devices.
 * cbfssm_philox4x32_10_u32: the generator's four 32-bit words for the counters first_counter .. first_counter + ncounters - 1
 * (out: 4 * ncounters words) -- the known-answer vectors of the Philox paper are checked through it. */
int cbfssm_normal_f64(uint64_t seed, uint64_t offset, int64_t n, double* out, void* stream);
int cbfssm_philox4x32_10_u32(uint64_t seed, uint64_t first_counter, int64_t ncounters, uint32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
