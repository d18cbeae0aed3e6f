// C ABI of the CBF-SSM ELBO hot path for MI355X (see include/cbfssm_hip.h) and the once-per-evaluation kernels:
// K_mm build + Cholesky + inverse + operand packing + prior KL, log-likelihood/moments, ELBO combination.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <cstdlib>
#include "../../include/cbfssm_hip.h"
#include "cbfssm_inst.hpp"
#include "cbfssm_adjoint_inst.hpp"

CBF_FOR_EACH_NBLK(CBF_DECLARE)
CBF_FOR_EACH_REV_NBLK(CBF_REV_DECLARE)

namespace cbfssm {

static thread_local char g_err[512] = "";
static double* g_dbg = nullptr;   // diagnostic builds only (cbfssm_debug_set_buffer)

int fail(int code, const char* fmt, ...)   // also used by cbfssm_tail.hip
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-int(e) - 1000, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// GP preparation: one workgroup per GPModel.
//   K_mm (gp_tf.py:33-49,129), L = chol(K_mm + jitter I) (gp_tf.py:52-65,130), K^-1, operand images, prior_kl
//   (gp_tf.py:163-172).
// The factorisation is a right-looking Cholesky of the bordered matrix [[K, I], [I, 0]]: after M steps the
// (1,1) block is L, the (2,1) block L^-T and the (2,2) block -K^-1 -- one pass, one barrier per column.
// ---------------------------------------------------------------------------------------------------------------------
struct PrepArgs {
    int M, D, Do, NBLK, DK;
    const double* Z;
    const double* ls;
    const double* var;
    const double* zmean;   // may be null (kmm_chol only)
    const double* zvar;
    double jitter;
    double* Kmm;           // M*M
    double* Lout;          // M*M  lower Cholesky factor
    double* Gout;          // M*M  L^-T (upper) or null
    double* Kinv;          // M*M  or null
    double* Zs;            // M*D
    double* gmat;          // global fallback for the working matrix when it does not fit LDS (M*(M+1)) or null
    // pack sections (null for kmm_chol only)
    double* Bp;
    double* Zp;
    double* cz;
    double* muA;
    double* s2A;
    double* invl;
    double* scal;
    double* muB;
    double* s2B;
    double* ZT;
    int JB;
    double* Wp;            // triangular operand images (two-triangular GP form)
    double* WTp;
    double* info_out;      // kmm_chol: 1 double
    const double* Kin;     // cbfssm_cholesky_f64: factorise this M x M matrix instead of building K(Z, Z)
    double refine_cond;    // condition number above which G = L^-T gets its Newton step (refine_threshold())
};

struct PrepArgs2 {
    PrepArgs g[2];
};

#define PREP_NT 1024
#define PREP_NB 32
#define PREP_LDS_MAX_M 140
// Infinity-norm condition number of K_mm + jitter I above which G = L^-T gets its Newton step in doubled precision.  The
// left residual of W = L^-1 matters where the kernels multiply by W -- the two-triangular GP form, which the Python surface
// runs above 3e7 and keeps down to a quarter of that (hysteresis) -- so the default is that lower edge; below it the
// dense form's K^-1 = G G^T is used, whose error is the cancellation in sigma^2 - k.(K^-1 k), not the factor's.
// CBFSSM_REFINE_COND overrides (0: always).  Cost when it runs: 0.07 ms at M = 100, 0.5 ms at M = 200, 1.9 ms at M = 300.
static double refine_threshold()
{
    static const double v = [] {
        const char* e = getenv("CBFSSM_REFINE_COND");
        return e ? atof(e) : 7.5e6;
    }();
    return v;
}   // (M | 1) * M doubles must fit the 160 KiB LDS next to the small arrays

// One workgroup per GPModel (blockIdx.x).  The working matrix W (row stride LD, odd) starts as
//   lower triangle + diagonal: K_mm + jitter I ;  strict upper triangle: 0 (the off-diagonal of the bordering identity)
// and is swept column by column (right-looking Cholesky of [[K, I],[I, *]] restricted to the blocks that are needed):
// afterwards column j of the lower part is L[:,j] * sqrt(piv_j) and row i of the strict upper part is
// L^-T[i,:] * sqrt(piv) (unit diagonal implied).  One barrier per column.
template <bool LDSW>
__global__ __launch_bounds__(PREP_NT) void prepare_kernel(PrepArgs2 aa)
{
    const PrepArgs& a = aa.g[blockIdx.x];
    extern __shared__ double smem[];
    __shared__ double Xs[CBFSSM_MAX_M];
    __shared__ double piv[CBFSSM_MAX_M];
    __shared__ double red[PREP_NT / 64];
    __shared__ int s_info;
    const int tid = threadIdx.x;
    const int M = a.M, D = a.D;
    const int LD = M | 1;
    // (a compile-time choice keeps the address space known: ds_read/ds_write instead of flat accesses)
    double* Wm;
    if constexpr (LDSW) Wm = smem; else Wm = a.gmat;
    const int tx = tid & 31, ty = tid >> 5;   // 32 x 32 thread tile

    if (tid == 0) s_info = 0;
    // X / lengthscales and row norms (gp_tf.py:34-35)
    if (!a.Kin) {
        for (int i = tid; i < M * D; i += PREP_NT) a.Zs[i] = a.Z[i] / a.ls[i % D];
        __syncthreads();
        for (int m = tid; m < M; m += PREP_NT) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += a.Zs[m * D + j] * a.Zs[m * D + j];
            Xs[m] = s;
        }
    }
    __syncthreads();
    const double var = a.Kin ? 0.0 : a.var[0];
    constexpr int NWAVE = PREP_NT / 64;
    const int NBT = (M + 15) / 16;
    const int wv = tid >> 6, l = tid & 63, g = l >> 4, nl = l & 15;
    // X X^T in 16 x 16 tiles on the f64 MFMA units (gp_tf.py:36: the matmul of the -2 X X^T + |x|^2 + |x'|^2 expansion)
    if (a.Kin) {
        // cast_cholesky of a given matrix (gp_tf.py:52-65): the working matrix is its lower triangle + jitter on the diagonal
        for (int idx = tid; idx < M * M; idx += PREP_NT) {
            const int i = idx / M, k = idx - i * M;
            const double kv = a.Kin[idx];
            Wm[i * LD + k] = (k < i) ? kv : ((k == i) ? kv + a.jitter : 0.0);
        }
    }
    for (int t = wv; t < (a.Kin ? 0 : NBT * NBT); t += NWAVE) {
        const int ib = t / NBT, kb = t - ib * NBT;
        const int ia = 16 * ib + nl, ka = 16 * kb + nl;
        d4 dot = {0, 0, 0, 0};
        for (int s = 0; 4 * s < D; ++s) {
            const int j = 4 * s + g;
            const double av = (ia < M && j < D) ? a.Zs[ia * D + j] : 0.0;
            const double bv = (ka < M && j < D) ? a.Zs[ka * D + j] : 0.0;
            dot = CBF_MFMA(av, bv, dot);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ib + g + 4 * r, k = 16 * kb + nl;
            if (i < M && k < M) {
                const double d2 = -2.0 * dot[r] + Xs[i] + Xs[k];              // gp_tf.py:37-38, no clamp
                const double kv = var * exp(-0.5 * d2);                       // gp_tf.py:49
                a.Kmm[i * M + k] = kv;
                Wm[i * LD + k] = (k < i) ? kv : ((k == i) ? kv + a.jitter : 0.0);   // gp_tf.py:53
            }
        }
    }
    __syncthreads();

    // Bordered elimination of [[K + jitter I, I], [I, .]] in the one M x M working matrix: the lower triangle ends up as
    // L diag(piv)^1/2 (column j scaled), the strict upper triangle as the rows of L^-T (same scaling).  Step j subtracts
    // c_ij W[k][j] from W[i][k], k > j, with c_ij = W[i][j] / p_j (rows i != j; lower rows only up to k <= i) and
    // c_jj = 1 / p_j (the border row born at step j).  Blocked right-looking form: panels of PREP_NB columns are
    // eliminated in LDS (all M rows x PREP_NB columns), then ONE rank-PREP_NB update W -= C P^T of everything to the
    // right of the panel runs on the f64 MFMA units -- M / PREP_NB passes over the matrix instead of M.
    constexpr int NB = PREP_NB, PL = PREP_NB + 1;
    double* P = nullptr;
    if constexpr (!LDSW) P = smem;                       // [16 ceil(M/16)][PL] panel copy (W itself lives in global memory)
#ifdef CBF_PREP_STAMPS
    long long tk0 = clock64(), tk_pan = 0, tk_upd = 0;
#endif
    for (int j0 = 0; j0 < M; j0 += NB) {
#ifdef CBF_PREP_STAMPS
        long long tp0 = clock64();
#endif
        const int nb = min(NB, M - j0), j1 = j0 + nb;
        auto pan = [&](int i, int jj) -> double& {
            if constexpr (LDSW) return Wm[i * LD + j0 + jj];
            else return P[i * PL + jj];
        };
        if constexpr (!LDSW) {
            for (int idx = tid; idx < 16 * NBT * NB; idx += PREP_NT) {
                const int i = idx / NB, jj = idx - i * NB;
                P[i * PL + jj] = (i < M && jj < nb) ? Wm[i * LD + j0 + jj] : 0.0;
            }
            __syncthreads();
        }
        // ---- panel: the unblocked steps restricted to the panel's columns
        for (int jj = 0; jj < nb; ++jj) {
            const int j = j0 + jj;
            const double p = pan(j, jj);
            if (tid == 0) {
                piv[j] = p;
                if (!(p > 0.0) && s_info == 0) s_info = j + 1;
            }
            const double rp = 1.0 / p;
            {
                const int kk = jj + 1 + tx, k = j0 + kk;     // 32 x 32 thread tile: tx -> panel column, ty -> row
                if (kk < nb) {
                    const double wkj = rp * pan(k, jj);
                    for (int i = ty; i < M; i += 32) {
                        if (i > j && k > i) continue;        // a lower row is updated up to its diagonal only
                        const double c = (i == j) ? 1.0 : pan(i, jj);
                        pan(i, kk) -= c * wkj;
                    }
                }
            }
            __syncthreads();
        }
        if constexpr (!LDSW) {
            for (int idx = tid; idx < M * nb; idx += PREP_NT) {
                const int i = idx / nb, jj = idx - i * nb;
                Wm[i * LD + j0 + jj] = P[i * PL + jj];
            }
        }
#ifdef CBF_PREP_STAMPS
        __syncthreads();
        tk_pan += clock64() - tp0;
        tp0 = clock64();
#endif
        if (j1 >= M) break;
        // ---- rank-nb update of the columns k >= j1 (nb == NB here; 16 | j1): W[i][k] -= sum_jj C[i][jj] W[k][j0+jj]
        int cnt = 0;
        for (int ib = 0; ib < NBT; ++ib) {
            const int i0 = 16 * ib;
            const int kb_lo = j1 >> 4, kb_hi = (i0 < j1) ? NBT - 1 : ib;
            for (int kb = kb_lo; kb <= kb_hi; ++kb, ++cnt) {
                if ((cnt % NWAVE) != wv) continue;
                const int k0 = 16 * kb;
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + g + 4 * r, k = k0 + nl;
                    acc[r] = (i < M && k < M) ? Wm[i * LD + k] : 0.0;
                }
                const int ia = i0 + nl;                  // A-operand row of this lane
                const int kbr = k0 + nl;                 // B-operand column (= row of W) of this lane
#pragma unroll
                for (int s = 0; s < NB / 4; ++s) {
                    const int jj = 4 * s + g;
                    double av = 0.0;
                    if (ia < M) {
                        const double rp = 1.0 / piv[j0 + jj];
                        if (ia < j0 || ia >= j1) av = pan(ia, jj) * rp;
                        else if (jj > ia - j0) av = pan(ia, jj) * rp;
                        else if (jj == ia - j0) av = rp;
                    }
                    const double bv = (kbr < M) ? pan(kbr, jj) : 0.0;
                    acc = CBF_MFMA(-av, bv, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + g + 4 * r, k = k0 + nl;
                    if (i < M && k < M && (i < j1 || k <= i)) Wm[i * LD + k] = acc[r];
                }
            }
        }
        __syncthreads();
#ifdef CBF_PREP_STAMPS
        tk_upd += clock64() - tp0;
#endif
    }
#ifdef CBF_PREP_STAMPS
    long long tk1 = clock64();
#endif
    // log det(K + jitter I) = sum_j log piv_j (fixed-order block sum)
    double ld_part = 0.0;
    for (int j = tid; j < M; j += PREP_NT) ld_part += log(piv[j]);
    const double logdet = block_sum(ld_part, red, tid, PREP_NT);
    // outputs: L = W_lower diag(piv)^-1/2, L^-T = W_upper diag(piv)^-1/2 (diagonal 1/sqrt(piv))
    for (int i = ty; i < M; i += 32) {
        for (int k = tx; k < M; k += 32) {
            const double sp = sqrt(piv[k]);
            const double w = Wm[i * LD + k];
            a.Lout[i * M + k] = (k < i) ? w / sp : ((k == i) ? sp : 0.0);
            const double gv = (k > i) ? w / sp : ((k == i) ? 1.0 / sp : 0.0);
            if (a.Gout) a.Gout[i * M + k] = gv;
        }
    }
    __syncthreads();
    if (a.info_out && tid == 0) a.info_out[0] = double(s_info);
    if (!a.Kinv) return;
    // scale both halves in place (own element only): lower part L (the values just written to Lout), upper part and
    // diagonal G = L^-T; then K^-1 = G G^T
    for (int i = ty; i < M; i += 32)
        for (int k = tx; k < M; k += 32) {
            const double sp = sqrt(piv[k]);
            const double w = Wm[i * LD + k];
            Wm[i * LD + k] = (k > i) ? w / sp : ((k == i) ? 1.0 / sp : w / sp);    // diagonal now holds G[i][i]
        }
    __syncthreads();
    // K^-1[i][k] = sum_{q >= max(i,k)} G[i][q] G[k][q]: 16 x 16 tiles of the lower triangle on the f64 MFMA units
    auto kinv_product = [&]() {
        int cnt = 0;
        for (int ib = 0; ib < NBT; ++ib) {
            for (int kb = 0; kb <= ib; ++kb, ++cnt) {
                if ((cnt % NWAVE) != wv) continue;
                const int ia = 16 * ib + nl, kr = 16 * kb + nl;
                d4 acc = {0, 0, 0, 0};
                for (int qb = ib; qb < NBT; ++qb) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int q = 16 * qb + 4 * s + g;
                        const double av = (ia < M && q < M && q >= ia) ? Wm[ia * LD + q] : 0.0;
                        const double bv = (kr < M && q < M && q >= kr) ? Wm[kr * LD + q] : 0.0;
                        acc = CBF_MFMA(av, bv, acc);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ib + g + 4 * r, k = 16 * kb + nl;
                    if (i < M && k < M) {
                        a.Kinv[i * M + k] = acc[r];
                        if (ib != kb) a.Kinv[k * M + i] = acc[r];      // (a diagonal tile holds both of its halves)
                    }
                }
            }
        }
    };
    kinv_product();
    __syncthreads();
    // infinity-norm condition number of K_mm + jitter I (row sums of |K| and |K^-1|): what the caller's choice between
    // the dense and the two-triangular GP form goes by, and what decides about the refinement of G below
    __shared__ double cn[2][PREP_NT / 64];
    __shared__ double s_cond;
    {
        double kn = 0.0, kin = 0.0;
        for (int i = wv; i < M; i += NWAVE) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = l; k < M; k += 64) {
                const double kv = a.Kin ? a.Kin[i * M + k] : a.Kmm[i * M + k];
                s1 += fabs(kv) + (k == i ? a.jitter : 0.0);
                s2 += fabs(a.Kinv[i * M + k]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            kn = fmax(kn, s1);
            kin = fmax(kin, s2);
        }
        if (l == 0) { cn[0][wv] = kn; cn[1][wv] = kin; }
        __syncthreads();
        if (tid == 0) {
            double c0 = 0.0, c1 = 0.0;
            for (int i = 0; i < NWAVE; ++i) { c0 = fmax(c0, cn[0][i]); c1 = fmax(c1, cn[1][i]); }
            s_cond = c0 * c1;
        }
        __syncthreads();
    }
    // ---- refinement of G = L^-T on ill-conditioned K_mm.  The elimination gives W = L^-1 = G^T column by column with a
    // small RIGHT residual I - L W (each column is a backward-stable substitution), but the kernels multiply by W from
    // the left (A = W k, gp_tf.py:137; A2 = W^T A, gp_tf.py:145) where the reference back-substitutes, and W L - I is of
    // order cond(L) eps: at cond(K_mm) 2e9 that put the predictive mean of a 250-step recurrence 5e-5 from the oracle, ten
    // times what two LAPACK codings differ by.  One Newton step W <- W + W (I - L W) with the residual accumulated in
    // doubled precision (compensated dot products: exact products by FMA, two-sum accumulation) squares the residual
    // (1e-11 -> 1e-22): W becomes the correctly rounded inverse of the L written above, and W k is then as accurate
    // as the substitution.  Runs above CBF_REFINE_COND only (a uniform branch on the measured condition number).
    if (s_cond > a.refine_cond && s_info == 0 && a.Bp) {
        double* Rg = a.Kinv;        // scratch until K^-1 is rebuilt: lower + diagonal = R, strict upper = the correction of G
        double* cdiag = a.Bp;       // (diagonal of the correction: the image section is written further down)
        {
#pragma clang fp contract(off)
            // four rows of one column per thread: four independent compensated sums (one alone is a chain of six dependent
            // float64 additions per term), the W operand shared
            const int NG4 = (M + 3) >> 2;
            for (int idx = tid; idx < NG4 * M; idx += PREP_NT) {
                const int ig = idx / M, j = idx - ig * M;
                const int i0 = 4 * ig;
                if (j > i0 + 3 || j >= M) continue;
                const int ihi = min(i0 + 3, M - 1);
                double s[4], c[4], dg[4];
                int row[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] = (i0 + r == j) ? 1.0 : 0.0;
                    c[r] = 0.0;
                    row[r] = min(i0 + r, M - 1) * LD;                 // (rows beyond M: a copy of the last row, dropped below)
                    dg[r] = sqrt(piv[min(i0 + r, M - 1)]);            // L[i][i]
                }
                auto term = [&](int r, double lv, double gv) {
                    const double p = -(lv * gv);
                    const double e = __builtin_fma(-lv, gv, -p);       // exact: lv gv = -(p + e)
                    const double t = s[r] + p;
                    const double z = t - s[r];
                    c[r] += ((s[r] - (t - z)) + (p - z)) + e;          // two-sum error of s + p, plus the product's
                    s[r] = t;
                };
                // k < i0: strictly below the diagonal for all four rows -- no predicates in the loop
                int k = j;
                for (; k < i0; ++k) {
                    const double gv = Wm[j * LD + k];                 // G[j][k] = W[k][j]
#pragma unroll
                    for (int r = 0; r < 4; ++r) term(r, Wm[row[r] + k], gv);
                }
                // the diagonal block: L[i][k] for k <= i, L[i][i] = sqrt(piv_i), zero above
                for (; k <= ihi; ++k) {
                    const double gv = Wm[j * LD + k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + r;
                        if (k <= i) term(r, (k == i) ? dg[r] : Wm[row[r] + k], gv);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    if (i < M && j <= i) Rg[i * M + j] = s[r] + c[r];
                }
            }
        }
        __syncthreads();
        // correction of G: C[j][k] = sum_{i=j..k} R[i][j] G[i][k]   (W + W R, transposed), 16 x 16 tiles of the upper
        // triangle on the f64 MFMA units: A[row j][k = i] = R[i][j] (zero for i < j), B[k = i][col k] = G[i][k] (zero for k < i)
        // (R sits in the lower triangle of the scratch incl. its diagonal, the correction goes to the strict upper triangle and
        //  `cdiag`: the tiles are written as they are finished)
        {
            int cnt = 0;
            for (int jb = 0; jb < NBT; ++jb) {
                for (int kb = jb; kb < NBT; ++kb, ++cnt) {
                    if ((cnt % NWAVE) != wv) continue;
                    d4 acc = {0, 0, 0, 0};
                    const int ja = 16 * jb + nl, kc = 16 * kb + nl;
                    for (int ib = jb; ib <= kb; ++ib) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            const int i = 16 * ib + 4 * s4 + g;
                            const double av = (ja < M && i < M && i >= ja) ? Rg[i * M + ja] : 0.0;
                            const double bv = (kc < M && i < M && kc >= i) ? Wm[i * LD + kc] : 0.0;
                            acc = CBF_MFMA(av, bv, acc);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * jb + g + 4 * r, k = 16 * kb + nl;
                        if (j < M && k < M && k >= j) {
                            if (k == j) cdiag[j] = acc[r]; else Rg[j * M + k] = acc[r];
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < M * M; idx += PREP_NT) {
            const int j = idx / M, k = idx - j * M;
            if (k < j) continue;
            const double gn = Wm[j * LD + k] + ((k == j) ? cdiag[j] : Rg[j * M + k]);
            Wm[j * LD + k] = gn;
            if (a.Gout) a.Gout[j * M + k] = gn;
        }
        __syncthreads();
        kinv_product();
        __syncthreads();
    }
#ifdef CBF_PREP_STAMPS
    long long tk2 = clock64();
#endif
    if (!a.Bp) return;
    const double* Cm = a.Kinv;

    // ---- operand images for the time-loop kernels
    const int Do = a.Do, NBLK = a.NBLK, DK = a.DK, Mp = 16 * NBLK, KS = Mp / 4;
    for (int i = tid; i < NBLK * KS * 64; i += PREP_NT) {
        const int l = i & 63, s = (i >> 6) % KS, rb = (i >> 6) / KS;
        const int row = 16 * rb + (l & 15), col = 4 * s + (l >> 4);
        a.Bp[i] = (row < M && col < M) ? Cm[row * M + col] : 0.0;
    }
    // W = L^-1 = G^T (G = L^-T, upper triangular, in a.Gout) and W^T = G as A-operand images of the two triangular
    // products A = W K, A2 = W^T A (gp_tf.py:137,145); the kernels skip the zero blocks
    for (int i = tid; i < NBLK * KS * 64; i += PREP_NT) {
        const int l = i & 63, s = (i >> 6) % KS, rb = (i >> 6) / KS;
        const int row = 16 * rb + (l & 15), col = 4 * s + (l >> 4);
        const bool in = row < M && col < M;
        a.Wp[i] = (in && col <= row) ? a.Gout[col * M + row] : 0.0;
        a.WTp[i] = (in && row <= col) ? a.Gout[row * M + col] : 0.0;
    }
    for (int i = tid; i < NBLK * DK * 64; i += PREP_NT) {
        const int l = i & 63, s = (i >> 6) % DK, rb = (i >> 6) / DK;
        const int row = 16 * rb + (l & 15), col = 4 * s + (l >> 4);
        a.Zp[i] = (row < M && col < D) ? a.Zs[row * D + col] : 0.0;
    }
    const double logvar = log(var);
    for (int m = tid; m < Mp; m += PREP_NT) a.cz[m] = (m < M) ? (-0.5 * Xs[m] + logvar) : -1e30;
    for (int i = tid; i < NBLK * 4 * 64; i += PREP_NT) {
        const int l = i & 63, r = (i >> 6) & 3, rb = i >> 8;
        const int m = 16 * rb + 4 * r + (l >> 4), d = l & 15;
        const bool ok = (m < M) && (d < Do);
        a.muA[i] = ok ? a.zmean[m * Do + d] : 0.0;
        a.s2A[i] = ok ? a.zvar[m * Do + d] : 0.0;
    }
    for (int j = tid; j < 4 * DK; j += PREP_NT) a.invl[j] = (j < D) ? 1.0 / a.ls[j] : 0.0;
    // adjoint-kernel images
    for (int i = tid; i < NBLK * 4 * 64; i += PREP_NT) {
        const int l = i & 63, s = (i >> 6) & 3, rb = i >> 8;
        const int m = 16 * rb + (l & 15), d = 4 * s + (l >> 4);
        const bool ok = (m < M) && (d < Do);
        a.muB[i] = ok ? a.zmean[m * Do + d] : 0.0;
        a.s2B[i] = ok ? a.zvar[m * Do + d] : 0.0;
    }
    for (int i = tid; i < NBLK * a.JB * 4 * 64; i += PREP_NT) {
        const int l = i & 63, s = (i >> 6) & 3, jb = (i >> 8) % a.JB, rb = (i >> 8) / a.JB;
        const int j = 16 * jb + (l & 15), m = 16 * rb + 4 * s + (l >> 4);
        double v = 0.0;
        if (m < M) v = (j < D) ? a.Zs[m * D + j] : ((j == D) ? 1.0 : 0.0);
        a.ZT[i] = v;
    }

    // ---- prior KL: 0.5 sum_d [ tr(K^-1 S_d) + mu_d^T K^-1 mu_d - M + log det K - log det S_d ]   (gp_tf.py:163-172)
    // K^-1 mu_d for all d at once from the operand images just written: rows of row block rb = sum over the k-steps of
    // (K^-1 A-operand image) x (zeta_mean B-operand image), one wave per row block
    __syncthreads();
    double acc = 0.0;
    for (int rb = wv; rb < NBLK; rb += NWAVE) {
        d4 km = {0, 0, 0, 0};
        for (int s = 0; s < KS; ++s) km = CBF_MFMA(a.Bp[(rb * KS + s) * 64 + l], a.muA[s * 64 + l], km);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * rb + g + 4 * r, d = nl;           // C-layout: row (lane >> 4) + 4 r, column lane & 15
            if (m < M && d < Do) {
                const double s2 = a.zvar[m * Do + d];
                acc += Cm[m * M + m] * s2 + a.zmean[m * Do + d] * km[r] - log(s2);
            }
        }
    }
    const double tot = block_sum(acc, red, tid, PREP_NT);
    if (tid == 0) {
        a.scal[CBFSSM_SCAL_SIGMA2] = var;
        a.scal[CBFSSM_SCAL_LOGDET] = logdet;
        a.scal[CBFSSM_SCAL_KLZ] = 0.5 * (tot + double(Do) * (logdet - double(M)));
        a.scal[CBFSSM_SCAL_INFO] = double(s_info);
        for (int i = 4; i < CBFSSM_SCAL_COUNT; ++i) a.scal[i] = 0.0;
        a.scal[CBFSSM_SCAL_COND] = s_cond;
        a.scal[CBFSSM_SCAL_JITTER] = a.jitter;
#ifdef CBF_PREP_STAMPS
        a.scal[8] = double(tk_pan); a.scal[9] = double(tk_upd); a.scal[10] = double(tk2 - tk1); a.scal[11] = double(clock64() - tk2);
        // (scal[8..11]: cycles in panels, rank updates, outputs + K^-1 = G G^T, operand images + KL; Kmm build = rest)
#endif
    }
}

static int launch_prepare(PrepArgs2& aa, int n, int maxM, hipStream_t st)
{
    size_t lds = 0;
    if (maxM <= PREP_LDS_MAX_M) lds = size_t(maxM | 1) * maxM * sizeof(double);
    if (lds > 0) {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prepare_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            if (e != hipSuccess) return fail(-int(e) - 1000, "prepare: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(prepare_kernel<true>, dim3(n), dim3(PREP_NT), lds, st, aa);
    } else {
        const size_t pl = size_t(16 * ((maxM + 15) / 16)) * (PREP_NB + 1) * sizeof(double);   // panel copy
        if (pl > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prepare_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(pl));
            if (e != hipSuccess) return fail(-int(e) - 1000, "prepare: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(prepare_kernel<false>, dim3(n), dim3(PREP_NT), pl, st, aa);
    }
    return check_launch("gp_prepare");
}

// ---------------------------------------------------------------------------------------------------------------------
// Log-likelihood and moments over particles (cbfssm.py:245-251,264-269).  One thread per (b, t, d).
// ---------------------------------------------------------------------------------------------------------------------
struct LlArgs {
    int B, S, T, dim_x, dim_y;
    const double* var_y;
    const double* y;
    const double* x;     // (T,N,dim_x)
    double* ll_part;     // (B*T*dim_y)
    double* pred_mean;   // (B,T,dim_y)
    double* pred_var;
    double* int_mean;    // (B,T,dim_x) or null
    double* int_var;
};

__global__ void loglik_moments_kernel(LlArgs a)
{
    // per-block, per-dimension partial sums of the log-likelihood (fixed order): ll_part[block][dim_y]
    __shared__ double sh[256];
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = int64_t(a.B) * a.T * a.dim_x;
    double ll = 0.0;
    if (idx < total) {
        // thread <-> (t, b, d), d fastest, then b: x is (T, N, dim_x), so a wave walks ONE contiguous span of x as it
        // steps through the particles (S dim_x doubles per (t, b)); the (B, T, .) outputs take the scattered writes,
        // which are 1/(2S) of the bytes.  (HBM-bound kernel: 8 S dim_x bytes read per 24 bytes written.)
        const int d = int(idx % a.dim_x);
        const int64_t tb = idx / a.dim_x;
        const int b = int(tb % a.B), t = int(tb / a.B);
        const int64_t bt = int64_t(b) * a.T + t;
        const int64_t N = int64_t(a.B) * a.S;
        const double* xp = a.x + (int64_t(t) * N + int64_t(b) * a.S) * a.dim_x + d;
        // population moments over the particles (tf.nn.moments, cbfssm.py:267) in one pass over HBM: sums of the
        // differences to the first particle (a shift inside the sample range keeps the cancellation harmless)
        const double x0 = xp[0];
        double s1 = 0.0, s2 = 0.0;
        // (19 loads in flight per lane -- all of S = 20 -- before the first is consumed: 36.7 us at C3 against 38.7 with eight)
        constexpr int CH = 19;
        int s = 1;
        for (; s + CH <= a.S; s += CH) {
            double v[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = xp[int64_t(s + j) * a.dim_x];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const double dv = v[j] - x0;
                s1 += dv;
                s2 += dv * dv;
            }
        }
#pragma unroll 8
        for (; s < a.S; ++s) {
            const double dv = xp[int64_t(s) * a.dim_x] - x0;
            s1 += dv;
            s2 += dv * dv;
        }
        const double m1 = s1 / a.S;
        const double mean = x0 + m1;
        const double ss = fmax(s2 - s1 * m1, 0.0);                             // sum_s (x_s - mean)^2
        const double var = ss / a.S;
        if (a.int_mean) { a.int_mean[bt * a.dim_x + d] = mean; a.int_var[bt * a.dim_x + d] = var; }   // cbfssm.py:269
        if (d < a.dim_y) {
            const double vy = a.var_y[d];
            const int64_t o = bt * a.dim_y + d;
            a.pred_mean[o] = mean;
            a.pred_var[o] = var + vy;                                          // cbfssm.py:268
            // sum_s log N(y | x_s, vy) = -0.5 [ sum_s (y - x_s)^2 / vy + S (log 2 pi + log vy) ]   (cbfssm.py:247-251)
            const double yo = a.y[o];
            const double dm = yo - mean;
            const double sq = ss + a.S * dm * dm;
            ll = -0.5 * (sq / vy + a.S * (1.8378770664093454836 + log(vy)));
        }
    }
    sh[threadIdx.x] = ll;
    __syncthreads();
    if (int(threadIdx.x) < a.dim_y) {
        const int d = threadIdx.x;
        const int base = int((int64_t(blockIdx.x) * 256) % a.dim_x);
        double s = 0.0;
        for (int k = (d - base + a.dim_x) % a.dim_x; k < 256; k += a.dim_x) s += sh[k];
        a.ll_part[int64_t(blockIdx.x) * a.dim_y + d] = s;
    }
}

// (Measured and NOT kept, round 4: a wave per (t, b) span -- fully coalesced 512-byte wave loads of the S dim_x contiguous
// doubles, parked in an LDS strip, lanes d < dim_x reducing over the particles, the next span's loads in registers meanwhile:
// 42.5 us against 38.0 us at C3, 327 against 205 us at C5, 37 against 10 us at C2 on a buffer rotation larger than the
// Infinity Cache.  The 112-byte stride of the form above costs partial lines at L1, not HBM efficiency; what the span form
// loses is bytes in flight -- 14 reducing lanes per wave behind an LDS round trip.  Also measured: a persistent grid of 4 .. 16
// workgroups per CU walking the blocks, ten non-temporal loads in flight per thread: 46.6 .. 50.0 us against 37.8 -- the lanes of
// a wave touch every 128-byte line in two consecutive iterations, and loads that bypass the cache fetch those lines twice.
// Two adjacent dims per thread (16-byte loads, half the threads): 44.3 us.  One wave per (t, b) with the particles spread over the
// lanes (every wave load one contiguous run, the per-lane (count, mean, M2) triples merged in a shuffle tree): 52.4 us at C3,
// 493 against 204 us at C5.  Every variant with fewer loads in flight per CU than this one's 20 per lane loses.)
struct CombineArgs {
    const double* ll; int64_t n_ll;
    const double* kl; int64_t n_kl;
    const double* ent; int64_t n_ent;
    const double* scal_f;
    const double* scal_b;
    double lambda0, lambda1, inv_s;
    double* out;
};

__global__ __launch_bounds__(1024) void combine_kernel(CombineArgs a)
{
    __shared__ double red[16];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = tid; i < a.n_ll; i += 1024) s0 += a.ll[i];
    for (int64_t i = tid; i < a.n_kl; i += 1024) s1 += a.kl[i];
    for (int64_t i = tid; i < a.n_ent; i += 1024) s2 += a.ent[i];
    const double loglik = block_sum(s0, red, tid, 1024);
    const double kl_x = block_sum(s1, red, tid, 1024);
    const double entropy = block_sum(s2, red, tid, 1024);
    if (tid == 0) {
        const double klf = a.scal_f[CBFSSM_SCAL_KLZ], klb = a.scal_b ? a.scal_b[CBFSSM_SCAL_KLZ] : 0.0;
        // cbfssm.py:257-262
        const double elbo = loglik * a.lambda0 * a.inv_s - kl_x * a.lambda0 * a.inv_s + entropy * a.lambda1 * a.inv_s
                            - klf - klb;
        a.out[0] = loglik; a.out[1] = kl_x; a.out[2] = entropy; a.out[3] = klf; a.out[4] = klb;
        a.out[5] = elbo; a.out[6] = -elbo;
        a.out[7] = fmax(a.scal_f[CBFSSM_SCAL_INFO], a.scal_b ? a.scal_b[CBFSSM_SCAL_INFO] : 0.0);
    }
}

// Sum the per-workgroup slabs in a fixed order: out[i] = sum_wg gpart[wg][i].  Two stages (RSPLIT partial sums per
// element, then their sum) so that a few thousand slabs still fill the chip; the order is fixed => reproducible.
#define RSPLIT 32
__global__ void reduce_partials_stage1(const double* gpart, int64_t slab, int nwg, double* tmp)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= slab) return;
    const int part = blockIdx.y;
    const int per = (nwg + RSPLIT - 1) / RSPLIT;
    const int k0 = part * per, k1 = min(nwg, k0 + per);
    double s = 0.0;
    for (int k = k0; k < k1; ++k) s += gpart[int64_t(k) * slab + i];
    tmp[int64_t(part) * slab + i] = s;
}

__global__ void reduce_partials_stage2(const double* tmp, int64_t slab, double* out)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= slab) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < RSPLIT; ++k) s += tmp[int64_t(k) * slab + i];
    out[i] = s;
}

__global__ void reduce_partials_kernel(const double* gpart, int64_t slab, int nwg, double* out)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= slab) return;
    double s = 0.0;
    for (int k = 0; k < nwg; ++k) s += gpart[int64_t(k) * slab + i];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
static const int kNblk[] = {1, 2, 4, 7, 10, 13, 16, 20};

static int dispatch_predict(int NBLK, int DK, const PredictArgs& a, hipStream_t st)
{
    switch (NBLK) {
#define X(NB) case NB: return launch_predict_nb##NB(DK, a, st);
        CBF_FOR_EACH_NBLK(X)
#undef X
    }
    return -2;
}

static int dispatch_pass(int NBLK, int DK, int mode, const PassArgs& a, dim3 grid, int nc, hipStream_t st)
{
    switch (NBLK) {
#define X(NB) case NB: return launch_pass_nb##NB(DK, mode, a, grid, nc, st);
        CBF_FOR_EACH_NBLK(X)
#undef X
    }
    return -2;
}

static int64_t rev_slab(int NBLK, int DK)
{
    switch (NBLK) {
#define X(NB) case NB: return rev_slab_nb##NB(DK);
        CBF_FOR_EACH_REV_NBLK(X)
#undef X
    }
    return 0;
}

static int dispatch_rev(int NBLK, int DK, int mode, const RevArgs& a, dim3 grid, hipStream_t st)
{
    switch (NBLK) {
#define X(NB) case NB: return launch_rev_nb##NB(DK, mode, a, grid, st);
        CBF_FOR_EACH_REV_NBLK(X)
#undef X
    }
    return -3;
}

static PackPtrs pack_ptrs(const cbfssm_pack_layout* L, const double* pack)
{
    PackPtrs p;
    p.Bp = pack + L->Bp; p.Zp = pack + L->Zp; p.cz = pack + L->cz; p.muA = pack + L->muA; p.s2A = pack + L->s2A;
    p.invl = pack + L->invl; p.scal = pack + L->scal;
    p.KSr = (L->M + 3) / 4;
    p.Wp = pack + L->Wp; p.WTp = pack + L->WTp;
    return p;
}

static int check_problem(const cbfssm_problem* p, const cbfssm_pack_layout* L, int Do)
{
    if (!p || !L) return fail(-1, "null problem/layout");
    if (p->B < 1 || p->S < 1 || p->T < 1) return fail(-1, "B, S, T must be >= 1");
    if (p->dim_x < 1 || p->dim_x > CBFSSM_MAX_DOUT) return fail(-1, "dim_x must be in [1,%d]", CBFSSM_MAX_DOUT);
    if (p->dim_y < 0 || p->dim_y > p->dim_x || p->dim_u < 0) return fail(-1, "bad dim_y/dim_u");
    if (p->recog_len < 1) return fail(-1, "recog_len must be >= 1");
    if (L->D != p->dim_x + p->dim_u) return fail(-1, "pack D=%d != dim_x+dim_u=%d", L->D, p->dim_x + p->dim_u);
    if (L->Do != Do) return fail(-1, "pack Do=%d, expected %d", L->Do, Do);
    if (L->M != p->M) return fail(-1, "pack M=%d != problem M=%d", L->M, p->M);
    if (int64_t(p->B) * p->S > (int64_t(1) << 30)) return fail(-1, "too many chains");
    return 0;
}

// column blocks (16 chains each) per workgroup of the forward-evaluation pass kernels: two when there are enough
// chains to still cover the chip (every K^-1 operand then feeds two MFMAs and every barrier covers twice the work)
static int pass_nc(const cbfssm_problem* p, int mode)
{
    if (p->ngroups > 0) {
        // chain-group split (units of 16 chains): the two-block kernels need an even range; only the skewed backward
        // runs are worth it there
        const bool even = ((p->group0 | p->ngroups) & 1) == 0;
        const int64_t nn = int64_t(p->ngroups) * 16;
        // (measured at C3, 256 + 64 groups: the skewed kernel on the even main piece is no faster than the one-block
        //  kernel there -- train 13.84 vs 13.73 ms, eval 4.08 vs 3.89 -- so it is opt-in: CBFSSM_NC_BWD=2)
        const char* e2 = getenv("CBFSSM_NC_BWD");
        return (e2 && atoi(e2) == 2 && even && mode == MODE_BWD && !p->half && p->M <= 112 && nn >= 32 * 128) ? 2 : 1;
    }
    const int64_t n = int64_t(p->B) * p->S;
    const char* e = getenv(mode == MODE_FWD ? "CBFSSM_NC_FWD" : "CBFSSM_NC_BWD");
    // tile heights 13..16 (two row blocks per wave): two column blocks share every streamed K^-1 operand load; their
    // tiles fit the LDS up to M = 256.  Measured at C4 against the compiler-scheduled one-block kernel: backward pass
    // 8.25 -> 6.96 ms, forward pass 6.60 -> 5.53 ms (CBFSSM_NC_FWD / CBFSSM_NC_BWD = 3).
    const bool shared_ok = p->M > 192 && p->M <= 256 && !p->half;
    if (e && !p->half) {
        const int v = atoi(e);
        if (p->M > 112) return (v == 3 && shared_ok) ? 3 : 1;
        return (v == 2 || v == 3) ? v : 1;
    }
    if (p->M > 112) return 1;   // (the skewed kernel's two tiles do not fit; the shared-operand variant 3 equals the
                                //  hand-scheduled one-block kernel at C4: 6.96 vs 6.95 ms, so it stays opt-in)
    // measured at C3 (round 1): the skewed two-group kernel 3 % faster on the many-workgroup backward runs, 2 % slower on
    // the forward pass (320 -> 160 workgroups).  Round 2, with the compile-time trim of the all-padding k-steps: the
    // one-group kernel is the faster one on the backward runs too (2.26 vs 2.34 ms), so the skewed kernel is opt-in
    // (CBFSSM_NC_BWD=2).
    (void)n;
    return 1;   // (also required by half mode: only the one-group kernel knows x0)
}

// chain-group range of a call: (first group, number of groups, total) in units of 16*nc chains
static int group_range(const cbfssm_problem* p, int nc, int* g0, int* ng, int* gt)
{
    const int64_t n = int64_t(p->B) * p->S;
    const int total16 = int((n + 15) / 16);
    const int cols = (nc == 1) ? 1 : 2;
    *gt = int((n + 16 * cols - 1) / (16 * cols));
    if (p->ngroups <= 0) { *g0 = 0; *ng = *gt; return 0; }
    if (p->group0 < 0 || p->group0 + p->ngroups > total16) return fail(-1, "bad chain-group range [%d, +%d) of %d", p->group0, p->ngroups, total16);
    if (nc != 1) {
        if ((p->group0 | p->ngroups) & 1) return fail(-1, "internal: an odd chain-group range needs the one-group kernels");
        *g0 = p->group0 / 2; *ng = p->ngroups / 2;
        return 0;
    }
    *g0 = p->group0; *ng = p->ngroups;
    return 0;
}

static void bwd_segments(const cbfssm_problem* p, int* nseg0, int* nseg1)
{
    const int P = 2 * p->recog_len;
    *nseg0 = p->T / P + 1;
    *nseg1 = (p->T + p->recog_len) / P + 1;
}

}  // namespace cbfssm

using namespace cbfssm;

extern "C" {

const char* cbfssm_last_error(void) { return g_err; }
#ifdef CBF_REV_STAMPS
void cbfssm_debug_set_buffer(double* p) { g_dbg = p; }
#endif
int cbfssm_version(void) { return 1; }

int cbfssm_gp_pack_layout(int M, int D, int Do, cbfssm_pack_layout* out)
{
    if (!out) return fail(-1, "null layout");
    if (M < 1 || M > CBFSSM_MAX_M) return fail(-1, "M=%d outside [1,%d]", M, CBFSSM_MAX_M);
    if (D < 1 || D > 24) return fail(-1, "D=%d outside [1,24]", D);
    if (Do < 1 || Do > CBFSSM_MAX_DOUT) return fail(-1, "Do=%d outside [1,%d]", Do, CBFSSM_MAX_DOUT);
    int nblk = 0;
    for (int v : kNblk) if (16 * v >= M) { nblk = v; break; }
    const int dk = (D <= 8) ? 2 : ((D <= 16) ? 4 : 6);
    memset(out, 0, sizeof(*out));
    out->M = M; out->D = D; out->Do = Do; out->NBLK = nblk; out->DK = dk; out->Mp = 16 * nblk; out->Dp = 4 * dk;
    out->KS = out->Mp / 4;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += (n + 63) / 64 * 64; return r; };
    out->Bp = take(int64_t(nblk) * out->KS * 64);
    out->Zp = take(int64_t(nblk) * dk * 64);
    out->cz = take(out->Mp);
    out->muA = take(int64_t(nblk) * 256);
    out->s2A = take(int64_t(nblk) * 256);
    out->invl = take(out->Dp);
    out->scal = take(CBFSSM_SCAL_COUNT);
    out->Kmm = take(int64_t(M) * M);
    out->L = take(int64_t(M) * M);
    out->Kinv = take(int64_t(M) * M);
    out->Linvt = take(int64_t(M) * M);
    out->Zs = take(int64_t(M) * D);
    out->JB = (4 * dk + 1 + 15) / 16;
    out->muB = take(int64_t(nblk) * 256);
    out->s2B = take(int64_t(nblk) * 256);
    out->ZT = take(int64_t(nblk) * out->JB * 256);
    out->rev_slab = rev_slab(nblk, dk);
    out->rev_stash = (nblk > 7) ? 1 : 0;
    out->work = take(M > PREP_LDS_MAX_M ? int64_t(M) * (M | 1) : 0);
    out->Wp = take(int64_t(nblk) * out->KS * 64);
    out->WTp = take(int64_t(nblk) * out->KS * 64);
    out->gp_form = CBFSSM_GP_FORM_DENSE;
    out->total = o;
    return 0;
}

int cbfssm_kmm_chol_f64(int M, int D, const double* Z, const double* lengthscales, const double* variance,
                        double jitter, double* Kmm, double* L, double* info, double* work, void* stream)
{
    if (M < 1 || M > CBFSSM_MAX_M || D < 1 || D > CBFSSM_MAX_DIN) return fail(-1, "bad M/D");
    if (!Z || !lengthscales || !variance || !Kmm || !L || !info || !work) return fail(-1, "null pointer");
    PrepArgs2 aa;
    memset(&aa, 0, sizeof(aa));
    PrepArgs& a = aa.g[0];
    a.M = M; a.D = D; a.Z = Z; a.ls = lengthscales; a.var = variance; a.jitter = jitter;
    a.Kmm = Kmm; a.Lout = L; a.Zs = work; a.gmat = (M > PREP_LDS_MAX_M) ? work + int64_t(M) * D : nullptr;
    a.info_out = info;
    return launch_prepare(aa, 1, M, (hipStream_t)stream);
}

static int fill_prep(PrepArgs& a, const cbfssm_pack_layout* L, const double* Z, const double* lengthscales,
                     const double* variance, const double* zeta_mean, const double* zeta_var, double jitter,
                     double* pack)
{
    if (!L || !Z || !lengthscales || !variance || !zeta_mean || !zeta_var || !pack) return fail(-1, "null pointer");
    a.M = L->M; a.D = L->D; a.Do = L->Do; a.NBLK = L->NBLK; a.DK = L->DK;
    a.Z = Z; a.ls = lengthscales; a.var = variance; a.zmean = zeta_mean; a.zvar = zeta_var; a.jitter = jitter;
    a.Kmm = pack + L->Kmm; a.Lout = pack + L->L; a.Gout = pack + L->Linvt; a.Kinv = pack + L->Kinv;
    a.Zs = pack + L->Zs;
    a.gmat = (L->M > PREP_LDS_MAX_M) ? pack + L->work : nullptr;
    a.Bp = pack + L->Bp; a.Zp = pack + L->Zp; a.cz = pack + L->cz; a.muA = pack + L->muA; a.s2A = pack + L->s2A;
    a.invl = pack + L->invl; a.scal = pack + L->scal;
    a.muB = pack + L->muB; a.s2B = pack + L->s2B; a.ZT = pack + L->ZT; a.JB = L->JB;
    a.Wp = pack + L->Wp; a.WTp = pack + L->WTp;
    a.refine_cond = refine_threshold();
    return 0;
}

int cbfssm_gp_prepare_f64(const cbfssm_pack_layout* L, const double* Z, const double* lengthscales,
                          const double* variance, const double* zeta_mean, const double* zeta_var, double jitter,
                          double* pack, void* stream)
{
    PrepArgs2 aa;
    memset(&aa, 0, sizeof(aa));
    int rc = fill_prep(aa.g[0], L, Z, lengthscales, variance, zeta_mean, zeta_var, jitter, pack);
    if (rc) return rc;
    return launch_prepare(aa, 1, L->M, (hipStream_t)stream);
}

int cbfssm_gp_prepare2_f64(const cbfssm_pack_layout* L0, const double* Z0, const double* ls0, const double* var0,
                           const double* zmean0, const double* zvar0, double* pack0,
                           const cbfssm_pack_layout* L1, const double* Z1, const double* ls1, const double* var1,
                           const double* zmean1, const double* zvar1, double* pack1, double jitter, void* stream)
{
    PrepArgs2 aa;
    memset(&aa, 0, sizeof(aa));
    int rc = fill_prep(aa.g[0], L0, Z0, ls0, var0, zmean0, zvar0, jitter, pack0);
    if (rc) return rc;
    rc = fill_prep(aa.g[1], L1, Z1, ls1, var1, zmean1, zvar1, jitter, pack1);
    if (rc) return rc;
    return launch_prepare(aa, 2, L0->M > L1->M ? L0->M : L1->M, (hipStream_t)stream);
}

int cbfssm_gp_predict_f64(const cbfssm_pack_layout* L, const double* pack, const double* X, int64_t npts,
                          double* fmean, double* fvar, void* stream)
{
    if (!L || !pack || !X || !fmean || !fvar) return fail(-1, "null pointer");
    if (npts < 0 || npts > (int64_t(1) << 34)) return fail(-1, "bad npts");
    if (npts == 0) return 0;
    PredictArgs a;
    a.pk = pack_ptrs(L, pack); a.X = X; a.npts = npts; a.D = L->D; a.Do = L->Do; a.fmean = fmean; a.fvar = fvar;
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    a.a2o = nullptr;
    int rc = dispatch_predict(L->NBLK, L->DK, a, (hipStream_t)stream);
    if (rc) return fail(rc, "gp_predict launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    return 0;
}

int cbfssm_cholesky_f64(int M, const double* mat, double jitter, double* L, double* info, double* work, void* stream)
{
    if (M < 1 || M > CBFSSM_MAX_M) return fail(-1, "bad M");
    if (!mat || !L || !info || !work) return fail(-1, "null pointer");
    PrepArgs2 aa;
    memset(&aa, 0, sizeof(aa));
    PrepArgs& a = aa.g[0];
    a.M = M; a.D = 1; a.jitter = jitter; a.Kin = mat;
    a.Lout = L; a.gmat = (M > PREP_LDS_MAX_M) ? work : nullptr;
    a.info_out = info;
    return launch_prepare(aa, 1, M, (hipStream_t)stream);
}

namespace cbfssm {
__global__ void rbf_k_kernel(int n, int m, int D, const double* X, const double* X2, const double* ls, const double* var,
                             double* out)
{
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= int64_t(n) * m) return;
    const int i = int(idx / m), k = int(idx - int64_t(i) * m);
    // the reference's expansion (gp_tf.py:33-43): -2 x.x' + |x|^2 + |x'|^2 on lengthscale-scaled inputs, no clamp
    double dot = 0.0, xs = 0.0, x2s = 0.0;
    for (int j = 0; j < D; ++j) {
        const double a = X[int64_t(i) * D + j] / ls[j], b = X2[int64_t(k) * D + j] / ls[j];
        dot += a * b; xs += a * a; x2s += b * b;
    }
    out[idx] = var[0] * exp(-0.5 * (-2.0 * dot + xs + x2s));
}

// fvar[p][d] += sum_j ( sum_m q_sqrt[d][m][j] A2[m][p] )^2   (conditional(), full-matrix q_sqrt, gp_tf.py:89-94)
__global__ __launch_bounds__(256) void fullq_kernel(const double* a2t, const double* q, int M, int NBLK, int Do,
                                                    int64_t npts, double* fvar)
{
    extern __shared__ double sh[];                    // [16 NBLK][16] A2 of this 16-point group, dense
    __shared__ double red[256];
    const int g = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const double* t = a2t + int64_t(g) * NBLK * 256;
    for (int i = tid; i < NBLK * 256; i += 256) {
        const int rb = i >> 8, r = (i >> 6) & 3, l = i & 63;
        sh[(16 * rb + (l >> 4) + 4 * r) * 16 + (l & 15)] = t[i];
    }
    __syncthreads();
    const int n = tid & 15, tj = tid >> 4;
    double acc = 0.0;
    for (int j = tj; j < M; j += 16) {
        double v = 0.0;
        for (int m = 0; m < M; ++m) v += q[(int64_t(d) * M + m) * M + j] * sh[m * 16 + n];
        acc += v * v;
    }
    red[tid] = acc;
    __syncthreads();
    if (tj == 0) {
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += red[k * 16 + n];
        const int64_t p = int64_t(g) * 16 + n;
        if (p < npts) fvar[p * Do + d] += s;
    }
}
}  // namespace cbfssm

int cbfssm_rbf_k_f64(int n, int m, int D, const double* X, const double* X2, const double* lengthscales,
                     const double* variance, double* out, void* stream)
{
    if (n < 1 || m < 1 || D < 1 || D > CBFSSM_MAX_DIN) return fail(-1, "bad n/m/D");
    if (!X || !X2 || !lengthscales || !variance || !out) return fail(-1, "null pointer");
    const int64_t tot = int64_t(n) * m;
    hipLaunchKernelGGL(rbf_k_kernel, dim3(unsigned((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, m, D, X, X2,
                       lengthscales, variance, out);
    return check_launch("rbf_k");
}

int64_t cbfssm_gp_predict_fullq_work_elems(const cbfssm_pack_layout* L, int64_t npts)
{
    if (!L || npts < 0) return -1;
    return (npts + 15) / 16 * L->NBLK * 256;
}

int cbfssm_gp_predict_fullq_f64(const cbfssm_pack_layout* L, const double* pack, const double* q_sqrt, const double* X,
                                int64_t npts, double* fmean, double* fvar, double* work, void* stream)
{
    if (!L || !pack || !q_sqrt || !X || !fmean || !fvar || !work) return fail(-1, "null pointer");
    if (npts < 0 || npts > (int64_t(1) << 34)) return fail(-1, "bad npts");
    if (npts == 0) return 0;
    PredictArgs a;
    a.pk = pack_ptrs(L, pack); a.X = X; a.npts = npts; a.D = L->D; a.Do = L->Do; a.fmean = fmean; a.fvar = fvar;
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    a.a2o = work;
    int rc = dispatch_predict(L->NBLK, L->DK, a, (hipStream_t)stream);
    if (rc) return fail(rc, "gp_predict launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    const size_t lds = size_t(16) * L->NBLK * 16 * sizeof(double);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fullq_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(-int(e) - 1000, "fullq: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(fullq_kernel, dim3(unsigned((npts + 15) / 16), unsigned(L->Do)), dim3(256), lds, (hipStream_t)stream,
                       (const double*)work, q_sqrt, L->M, L->NBLK, L->Do, npts, fvar);
    return check_launch("gp_predict_fullq");
}

int64_t cbfssm_backward_pass_partials(const cbfssm_problem* p)
{
    if (!p || p->recog_len < 1) return -1;
    int n0, n1;
    bwd_segments(p, &n0, &n1);
    const int64_t groups = (int64_t(p->B) * p->S + 15) / 16;     // upper bound over both tilings (1 or 2 groups per WG)
    return groups * (n0 + n1);
}

static int save_k(const cbfssm_pack_layout* L);

int cbfssm_backward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_b,
                             const double* var_x, const double* u, const double* y, const double* hid_b,
                             const double* eps_b, double* y2, double* h_all, double* fmv_b, double* a2s_b,
                             double* ent_part, void* stream)
{
    int rc = check_problem(p, L, p ? p->dim_x - p->dim_y : 0);
    if (rc) return rc;
    if (p->dim_x - p->dim_y < 1) return fail(-1, "backward pass needs dim_x > dim_y (use CBFSSMHALF otherwise)");
    if (!pack_b || !var_x || !u || !y || !hid_b || !eps_b || !y2 || !ent_part) return fail(-1, "null pointer");
    PassArgs a;
    memset(&a, 0, sizeof(a));
    a.pk = pack_ptrs(L, pack_b);
    a.N = p->B * p->S; a.S = p->S; a.T = p->T; a.B = p->B;
    a.dim_x = p->dim_x; a.dim_u = p->dim_u; a.dim_y = p->dim_y; a.Do = p->dim_x - p->dim_y; a.D = L->D;
    a.recog_len = p->recog_len; a.condition = p->condition; a.k_factor = p->k_factor;
    a.var_x = var_x; a.u = u; a.y = y; a.eps = eps_b; a.hid = hid_b; a.y2_out = y2; a.h_all = h_all;
    a.part_out = ent_part;
    a.dbg = g_dbg;
    a.fmv = fmv_b;
    a.a2s = a2s_b;
    a.ksave = save_k(L);
    int n0, n1;
    bwd_segments(p, &n0, &n1);
    a.nseg0 = n0;
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    const int nc = a.tri ? 1 : pass_nc(p, MODE_BWD);
    int g0, ng, gt;
    rc = group_range(p, nc, &g0, &ng, &gt);
    if (rc) return rc;
    a.group0 = g0; a.gtotal = gt;
    dim3 grid(unsigned(ng), unsigned(n0 + n1));
    rc = dispatch_pass(L->NBLK, L->DK, MODE_BWD, a, grid, nc, (hipStream_t)stream);
    if (rc) return fail(rc, "backward_pass launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    return 0;
}

// Tile heights whose adjoint holds its accumulators in registers (M <= 112) also keep the kernel tile K = k(Z, x_t) of
// every step next to its A2 tile: the adjoint then reads it instead of rebuilding it (6 MFMAs and four exponentials per
// lane and step, a whole phase of its step), for as many bytes again.  CBFSSM_NO_SAVE_K=1 switches it off (A/B runs).
static int save_k(const cbfssm_pack_layout* L) { return (L->NBLK <= 7 && !getenv("CBFSSM_NO_SAVE_K")) ? 1 : 0; }

int64_t cbfssm_saved_a2_elems(const cbfssm_problem* p, const cbfssm_pack_layout* L, int backward)
{
    if (!p || !L) return -1;
    const int64_t g16 = (int64_t(p->B) * p->S + 15) / 16;
    const int64_t slots = backward ? 2 * int64_t(p->T) : (p->T > 1 ? p->T - 1 : 0);
    return slots * g16 * saved_tile_stride(L->NBLK, save_k(L));
}

int64_t cbfssm_forward_pass_partials(const cbfssm_problem* p)
{
    if (!p) return -1;
    return (int64_t(p->B) * p->S + 15) / 16;
}

static int forward_pass_impl(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                             const double* var_x, const double* var_y, const double* u, const double* y,
                             const double* y2, const double* x0, const double* eps_f, double* x, double* fmv_f,
                             double* a2s_f, double* kl_part, void* stream)
{
    int rc = check_problem(p, L, p ? p->dim_x : 0);
    if (rc) return rc;
    if (!pack_f || !var_x || !var_y || !u || !y || !x || !kl_part) return fail(-1, "null pointer");
    if (p->half ? !x0 : (p->dim_x > p->dim_y && !y2)) return fail(-1, p->half ? "x0 is null" : "y2 is null");
    if (p->T > 1 && !eps_f) return fail(-1, "eps_f is null");
    PassArgs a;
    memset(&a, 0, sizeof(a));
    a.pk = pack_ptrs(L, pack_f);
    a.N = p->B * p->S; a.S = p->S; a.T = p->T; a.B = p->B;
    a.dim_x = p->dim_x; a.dim_u = p->dim_u; a.dim_y = p->dim_y; a.Do = p->dim_x; a.D = L->D;
    a.recog_len = p->recog_len; a.condition = p->condition; a.k_factor = p->k_factor;
    a.var_x = var_x; a.var_y = var_y; a.u = u; a.y = y; a.eps = eps_f; a.y2_in = y2; a.x_out = x;
    a.part_out = kl_part;
    a.dbg = g_dbg;
    a.half = p->half; a.x0 = x0;
    a.fmv = fmv_f;
    a.a2s = a2s_f;
    a.ksave = save_k(L);
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    const int nc = a.tri ? 1 : pass_nc(p, MODE_FWD);
    int g0, ng, gt;
    rc = group_range(p, nc, &g0, &ng, &gt);
    if (rc) return rc;
    a.group0 = g0; a.gtotal = gt;
    dim3 grid(unsigned(ng), 1);
    rc = dispatch_pass(L->NBLK, L->DK, MODE_FWD, a, grid, nc, (hipStream_t)stream);
    if (rc) return fail(rc, "forward_pass launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    return 0;
}

int cbfssm_forward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                            const double* var_x, const double* var_y, const double* u, const double* y,
                            const double* y2, const double* eps_f, double* x, double* fmv_f, double* a2s_f,
                            double* kl_part, void* stream)
{
    if (p && p->half) return fail(-1, "problem->half is set: use cbfssm_half_forward_pass_f64");
    return forward_pass_impl(p, L, pack_f, var_x, var_y, u, y, y2, nullptr, eps_f, x, fmv_f, a2s_f, kl_part, stream);
}

int cbfssm_half_forward_pass_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                                 const double* var_x, const double* var_y, const double* u, const double* y,
                                 const double* x0, const double* eps_f, double* x, double* fmv_f, double* a2s_f,
                                 double* kl_part, void* stream)
{
    if (!p || !p->half) return fail(-1, "problem->half must be 1");
    return forward_pass_impl(p, L, pack_f, var_x, var_y, u, y, nullptr, x0, eps_f, x, fmv_f, a2s_f, kl_part, stream);
}

int cbfssm_loglik_moments_f64(const cbfssm_problem* p, const double* var_y, const double* y, const double* x,
                              double* ll_part, double* pred_mean, double* pred_var, double* int_mean,
                              double* int_var, void* stream)
{
    if (!p || !var_y || !y || !x || !ll_part || !pred_mean || !pred_var) return fail(-1, "null pointer");
    if ((int_mean == nullptr) != (int_var == nullptr)) return fail(-1, "int_mean/int_var must both be given");
    LlArgs a;
    a.B = p->B; a.S = p->S; a.T = p->T; a.dim_x = p->dim_x; a.dim_y = p->dim_y;
    a.var_y = var_y; a.y = y; a.x = x; a.ll_part = ll_part; a.pred_mean = pred_mean; a.pred_var = pred_var;
    a.int_mean = int_mean; a.int_var = int_var;
    const int64_t total = int64_t(p->B) * p->T * p->dim_x;
    const int64_t nblk = (total + 255) / 256;
    hipLaunchKernelGGL(loglik_moments_kernel, dim3(unsigned(nblk)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("loglik_moments");
}

int64_t cbfssm_loglik_partials(const cbfssm_problem* p)
{
    if (!p) return -1;
    return (int64_t(p->B) * p->T * p->dim_x + 255) / 256 * p->dim_y;
}

int cbfssm_elbo_combine_f64(const cbfssm_problem* p, double lambda0, double lambda1, const double* ll_part,
                            int64_t n_ll, const double* kl_part, int64_t n_kl, const double* ent_part, int64_t n_ent,
                            const double* scal_f, const double* scal_b, double* out, void* stream)
{
    if (!p || !scal_f || !out) return fail(-1, "null pointer");   // scal_b may be null (CBFSSMHALF has no gp_b)
    CombineArgs a;
    a.ll = ll_part; a.n_ll = ll_part ? n_ll : 0; a.kl = kl_part; a.n_kl = kl_part ? n_kl : 0;
    a.ent = ent_part; a.n_ent = ent_part ? n_ent : 0;
    a.scal_f = scal_f; a.scal_b = scal_b; a.lambda0 = lambda0; a.lambda1 = lambda1; a.inv_s = 1.0 / p->S;
    a.out = out;
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("elbo_combine");
}

static int rev_chunks(const cbfssm_problem* p)
{
    // enough workgroups for ~5 rounds over 256 CUs (one adjoint workgroup per CU), never more chunks than segments
    const int64_t groups = (int64_t(p->B) * p->S + 15) / 16;
    const int P = 2 * p->recog_len;
    const int nseg = p->T / P + 1;
    int64_t c = (1280 + 2 * groups - 1) / (2 * groups);
    if (c > nseg) c = nseg;
    if (c < 1) c = 1;
    if (c > 64) c = 64;
    return int(c);
}

static int bwd_total_segments(const cbfssm_problem* p)
{
    // segment k of run r starts at max(0, 2R*k - r*R); run 1 has the most: floor((T-1+R)/2R) + 1
    return (p->T - 1 + p->recog_len) / (2 * p->recog_len) + 1;
}

int64_t cbfssm_rev_workgroups(const cbfssm_problem* p, int backward_runs)
{
    if (!p || p->recog_len < 1) return -1;
    return (int64_t(p->B) * p->S + 15) / 16 * (backward_runs ? 2 * rev_chunks(p) : 1);
}

int cbfssm_bwd_segments(const cbfssm_problem* p)
{
    if (!p || p->recog_len < 1) return -1;
    return bwd_total_segments(p);
}

static int fill_rev(RevArgs& a, const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack, int Do)
{
    memset(&a, 0, sizeof(a));
    a.pk = pack_ptrs(L, pack);
    a.rk.muB = pack + L->muB; a.rk.s2B = pack + L->s2B; a.rk.ZT = pack + L->ZT;
    a.N = p->B * p->S; a.S = p->S; a.T = p->T; a.B = p->B;
    a.dim_x = p->dim_x; a.dim_u = p->dim_u; a.dim_y = p->dim_y; a.Do = Do; a.D = L->D;
    a.recog_len = p->recog_len; a.condition = p->condition; a.k_factor = p->k_factor;
    a.slab = L->rev_slab;
    a.KSr = (L->M + 3) / 4;
    if (L->rev_slab <= 0) return fail(-3, "no adjoint kernel for M=%d (tile height %d)", L->M, L->NBLK);
    return 0;
}

static int set_stash(RevArgs& a, const cbfssm_pack_layout* L, double* stash_a, double* stash_k, int64_t stash_ld,
                     int64_t nwg, int chunk_steps)
{
    if (L->rev_stash) {
        if (!stash_a || !stash_k) return fail(-1, "this tile height (M=%d) needs stash buffers", L->M);
        if (stash_ld < nwg * chunk_steps * 16) return fail(-1, "stash_ld too small: need %lld", (long long)(nwg * chunk_steps * 16));
        a.stash_a = stash_a; a.stash_k = stash_k; a.stash_ld = stash_ld; a.chunk_steps = chunk_steps;
    }
    return 0;
}

static int forward_pass_bwd_impl(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                                 const double* var_x, const double* var_y, const double* u, const double* y,
                                 const double* y2, const double* eps_f, const double* x, const double* fmv_f,
                                 const double* a2s_f, double cL, double* gy2,
                                 double* gx0, double* gpart, int t_hi, int t_lo, double* gx_carry, double* stash_a,
                                 double* stash_k, int64_t stash_ld, void* stream)
{
    int rc = check_problem(p, L, p ? p->dim_x : 0);
    if (rc) return rc;
    if (!pack_f || !var_x || !var_y || !u || !y || !x || !gpart) return fail(-1, "null pointer");
    if (p->T > 1 && !fmv_f) return fail(-1, "fmv_f (saved fmean/fvar of the forward pass) is null");
    if (p->half ? !gx0 : (p->dim_x > p->dim_y && (!y2 || !gy2))) return fail(-1, "y2/gy2/gx0 is null");
    if (p->T > 1 && !eps_f) return fail(-1, "eps_f is null");
    if (t_hi > p->T - 2 || t_lo < 0) return fail(-1, "bad step range [%d, %d]", t_lo, t_hi);
    if ((t_hi < p->T - 2 || t_lo > 0) && t_hi >= t_lo && !gx_carry) return fail(-1, "partial range needs gx_carry");
    RevArgs a;
    rc = fill_rev(a, p, L, pack_f, p->dim_x);
    if (rc) return rc;
    a.cL = cL; a.var_x = var_x; a.var_y = var_y; a.u = u; a.y = y; a.eps = eps_f; a.x = x; a.y2 = y2; a.gy2 = gy2;
    a.gpart = gpart; a.t_hi = t_hi; a.t_lo = t_lo; a.gx_carry = gx_carry;
    a.half = p->half; a.gx0 = gx0;
    a.fmv = fmv_f;
    a.a2s = a2s_f;
    a.ksave = a2s_f ? save_k(L) : 0;
    const int64_t groups = (a.N + 15) / 16;
    const int steps = t_hi >= t_lo ? t_hi - t_lo + 1 : 0;
    rc = set_stash(a, L, stash_a, stash_k, stash_ld, groups, steps);
    if (rc) return rc;
    int g0, ng, gt;
    rc = group_range(p, 1, &g0, &ng, &gt);
    if (rc) return rc;
    a.group0 = g0; a.gtotal = gt;
    dim3 grid(unsigned(ng), 1);
    rc = dispatch_rev(L->NBLK, L->DK, MODE_FWD, a, grid, (hipStream_t)stream);
    if (rc) return fail(rc, "forward_pass_bwd launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    return 0;
}

int cbfssm_forward_pass_bwd_ex_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                                   const double* var_x, const double* var_y, const double* u, const double* y,
                                   const double* y2, const double* eps_f, const double* x, const double* fmv_f,
                                   const double* a2s_f, double cL, double* gy2, double* gpart, int t_hi, int t_lo,
                                   double* gx_carry, double* stash_a, double* stash_k, int64_t stash_ld, void* stream)
{
    if (p && p->half) return fail(-1, "problem->half is set: use cbfssm_half_forward_pass_bwd_f64");
    return forward_pass_bwd_impl(p, L, pack_f, var_x, var_y, u, y, y2, eps_f, x, fmv_f, a2s_f, cL, gy2, nullptr, gpart,
                                 t_hi, t_lo, gx_carry, stash_a, stash_k, stash_ld, stream);
}

int cbfssm_half_forward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                                     const double* var_x, const double* var_y, const double* u, const double* y,
                                     const double* eps_f, const double* x, const double* fmv_f, const double* a2s_f, double cL,
                                     double* gx0, double* gpart, int t_hi, int t_lo, double* gx_carry, double* stash_a,
                                     double* stash_k, int64_t stash_ld, void* stream)
{
    if (!p || !p->half) return fail(-1, "problem->half must be 1");
    return forward_pass_bwd_impl(p, L, pack_f, var_x, var_y, u, y, nullptr, eps_f, x, fmv_f, a2s_f, cL, nullptr, gx0, gpart,
                                 t_hi, t_lo, gx_carry, stash_a, stash_k, stash_ld, stream);
}

int cbfssm_forward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_f,
                                const double* var_x, const double* var_y, const double* u, const double* y,
                                const double* y2, const double* eps_f, const double* x, const double* fmv_f,
                                const double* a2s_f, double cL, double* gy2, double* gpart, void* stream)
{
    if (L && L->rev_stash) return fail(-3, "M=%d runs in stash mode: use cbfssm_forward_pass_bwd_ex_f64", L->M);
    return cbfssm_forward_pass_bwd_ex_f64(p, L, pack_f, var_x, var_y, u, y, y2, eps_f, x, fmv_f, a2s_f, cL, gy2, gpart,
                                          p ? p->T - 2 : -1, 0, nullptr, nullptr, nullptr, 0, stream);
}

int cbfssm_backward_pass_bwd_ex_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_b,
                                    const double* var_x, const double* u, const double* y, const double* hid_b,
                                    const double* eps_b, const double* h_all, const double* fmv_b, const double* a2s_b,
                                    const double* gy2, double cE, double* gpart, int seg0, int seg1, int nchunk, double* stash_a,
                                    double* stash_k, int64_t stash_ld, void* stream)
{
    int rc = check_problem(p, L, p ? p->dim_x - p->dim_y : 0);
    if (rc) return rc;
    if (!pack_b || !var_x || !u || !y || !hid_b || !eps_b || !h_all || !fmv_b || !gy2 || !gpart)
        return fail(-1, "null pointer");
    if (seg0 < 0 || seg1 <= seg0 || seg1 > bwd_total_segments(p) || nchunk < 1 || nchunk > seg1 - seg0)
        return fail(-1, "bad segment range [%d, %d) / chunks %d", seg0, seg1, nchunk);
    RevArgs a;
    rc = fill_rev(a, p, L, pack_b, p->dim_x - p->dim_y);
    if (rc) return rc;
    a.cE = cE; a.var_x = var_x; a.u = u; a.y = y; a.eps = eps_b; a.hid = hid_b; a.h_all = h_all;
    a.gy2 = const_cast<double*>(gy2); a.gpart = gpart;
    a.fmv = fmv_b;
    a.a2s = a2s_b;
    a.ksave = a2s_b ? save_k(L) : 0;
    a.seg0 = seg0; a.seg1 = seg1; a.nchunk = nchunk;
    const int64_t groups = (a.N + 15) / 16;
    const int per = (seg1 - seg0 + nchunk - 1) / nchunk;
    rc = set_stash(a, L, stash_a, stash_k, stash_ld, groups * 2 * nchunk, per * 2 * p->recog_len);
    if (rc) return rc;
    int g0, ng, gt;
    rc = group_range(p, 1, &g0, &ng, &gt);
    if (rc) return rc;
    a.group0 = g0; a.gtotal = gt;
    dim3 grid(unsigned(ng), 2, unsigned(nchunk));
    rc = dispatch_rev(L->NBLK, L->DK, MODE_BWD, a, grid, (hipStream_t)stream);
    if (rc) return fail(rc, "backward_pass_bwd launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    return 0;
}

int cbfssm_backward_pass_bwd_f64(const cbfssm_problem* p, const cbfssm_pack_layout* L, const double* pack_b,
                                 const double* var_x, const double* u, const double* y, const double* hid_b,
                                 const double* eps_b, const double* h_all, const double* fmv_b, const double* a2s_b,
                                 const double* gy2, double cE, double* gpart, void* stream)
{
    if (L && L->rev_stash) return fail(-3, "M=%d runs in stash mode: use cbfssm_backward_pass_bwd_ex_f64", L->M);
    if (!p || p->recog_len < 1) return fail(-1, "null problem");
    return cbfssm_backward_pass_bwd_ex_f64(p, L, pack_b, var_x, u, y, hid_b, eps_b, h_all, fmv_b, a2s_b, gy2, cE, gpart, 0,
                                           bwd_total_segments(p), rev_chunks(p), nullptr, nullptr, 0, stream);
}

int cbfssm_reduce_partials_f64(double* gpart, int64_t slab, int64_t nwg, double* out, void* stream)
{
    if (!gpart || !out || slab < 1 || nwg < 1 || nwg > (1 << 30)) return fail(-1, "bad reduce arguments");
    const unsigned gx = unsigned((slab + 255) / 256);
    if (nwg >= 4 * RSPLIT) {
        // the caller's buffer holds nwg slabs; the stage-1 partial sums go to the tail it reserves behind them
        double* tmp = gpart + nwg * slab;
        hipLaunchKernelGGL(reduce_partials_stage1, dim3(gx, RSPLIT), dim3(256), 0, (hipStream_t)stream, gpart, slab,
                           int(nwg), tmp);
        hipLaunchKernelGGL(reduce_partials_stage2, dim3(gx), dim3(256), 0, (hipStream_t)stream, tmp, slab, out);
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, gpart, slab, int(nwg),
                           out);
    }
    return check_launch("reduce_partials");
}

}  // extern "C"
