// Once-per-step tail of a train step: positivity transforms, the O(M^3) adjoint of K_mm -> K^-1 with the prior-KL
// gradient, the chain through the transforms, and the TF-1.8 Adam update.  These are ~170 tiny tensor ops when written
// in a tensor library (half of a C1 train step); here they are five launches.
//
// Reference call sites: tf_transform.py:19-21 (softplus + 1e-10), gp_tf.py:33-49 (RBF), :129-130 (K_mm, Cholesky),
// :163-172 (prior KL), base_model.py:34-36 (tf.gradients through all of it), cbfssm.py:273-275 (AdamOptimizer).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/cbfssm_hip.h"

namespace cbfssm {

int fail(int rc, const char* fmt, ...);   // cbfssm_api.hip

struct GpTail {
    const double* slab;       // reduced adjoint slab of this GP (Slab<> layout, cbfssm_adjoint.hpp)
    const double* gB_dense;   // stash mode: K^-1 adjoint, dense [.][gB_ld] or (gB_ld = 0) a C-layout image; else NULL (image
                              // inside the slab)
    int64_t gB_ld;
    const double* Kinv;       // [M][M]
    const double* Kmm;        // [M][M]
    const double* Zs;         // [M][D]   z / lengthscale
    double* T;                // scratch [M][max(M,48)]: Kinv G, then WZ [M][D] and K^-1 mu [M][Do]
    double* G2;               // scratch [M][M]
    double* dv;               // scratch [M]: per-row terms of sum(Kbar o K_mm) in the form that does not cancel (tail_gemm<1>)
    const double* scal;       // the pack's scalar block (jitter)
    const double* zmean;      // [M][Do]
    const double* zvar;       // [M][Do]  constrained
    const double* ls;         // [D]      constrained
    const double* var;        // [1]      constrained
    const double* zvar_unc;
    const double* var_unc;
    const double* ls_unc;
    double* g_z;              // [M][D]
    double* g_mu;             // [M][Do]
    double* g_s2;             // [M][Do]  w.r.t. the unconstrained variable
    double* g_var;            // [1]
    double* g_ls;             // [D]
    int M, D, Do, NBLK, JB, stash;
    const double* Kkl;        // PR-SSM: the prior-KL terms use THIS inverse, of K_mm without jitter (prssm.py:81-82); NULL: Kinv
    double* T2;               // PR-SSM scratch [M][M]: Kkl B_KL
    int shared_ls;            // PR-SSM: one lengthscale for all input dimensions (prssm.py:40): ls / ls_unc / g_ls have one entry
    int gmode;                // 0: the image holds d loss / d K^-1 (float64 adjoint).  1, 2: it holds G = the K_mm adjoint's data
                              // part itself, K^-1 (d loss / d K^-1) K^-1 (float32 adjoint, cbfssm_rev32.hip) -- 1: the full
                              // matrix, 2: only its lower-triangular 16 x 16 blocks, as S = C A2^T + A2 C^T (diagonal blocks:
                              // C A2^T); either way only the symmetric part of G enters
};

struct TailArgs {
    GpTail gp[2];
    const double* vx_unc;
    const double* vy_unc;
    double* g_vx;
    double* g_vy;
    const double* tail;       // [3 + dim_y]: loglik, kl_x, entropy, d loss/d var_y from the log-likelihood
    int dim_x, dim_y;
    int ngp;                  // 2: CBFSSM (gp_f, gp_b); 1: the forward-only variants (CBFSSMHALF, PRSSM: gp_f only)
    int n_vy;                 // entries of var_y: dim_x (CBFSSM, cbfssm.py:60-63) or dim_y (the forward-only variants)
};

__device__ __forceinline__ double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// element (row, col) of an MFMA C-layout image [nrb][ncb][4][64]: row = 16 rb + (lane >> 4) + 4 r, col = 16 cb + (lane & 15)
__device__ __forceinline__ double c_image(const double* img, int ncb, int row, int col)
{
    const int rb = row >> 4, rr = row & 15, cb = col >> 4, nl = col & 15;
    return img[((rb * ncb + cb) * 4 + (rr >> 2)) * 64 + 16 * (rr & 3) + nl];
}

__device__ __forceinline__ double block_sum_t(double v, double* red, int tid, int nt)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (nt >> 6); ++i) s += red[i];   // fixed order: deterministic
    return s;
}

// the data part's image, element (k, j)
__device__ __forceinline__ double img_elem(const GpTail& p, int k, int j)
{
    if (p.gB_dense) return (p.gB_ld > 0) ? p.gB_dense[int64_t(k) * p.gB_ld + j] : c_image(p.gB_dense, p.NBLK, k, j);
    return c_image(p.slab + 2 * p.NBLK * 256, p.NBLK, k, j);
}
// gmode 1, 2: the symmetric part of G, element (k, j)
__device__ __forceinline__ double gsym_elem(const GpTail& p, int k, int j)
{
    if (p.gmode == 2) {
        const int bk = k >> 4, bj = j >> 4;
        return 0.5 * ((bk >= bj ? img_elem(p, k, j) : 0.0) + (bj >= bk ? img_elem(p, j, k) : 0.0));
    }
    return 0.5 * (img_elem(p, k, j) + img_elem(p, j, k));
}

// G = Kinvbar (data) + Kinvbar (prior KL):  0.5 (diag(sum_d zvar) + zmean zmean^T)          (gp_tf.py:163-172)
// (gmode != 0: the prior-KL part only -- the data part arrives K^-1-applied and is added in stage 1)
__device__ __forceinline__ double kl_elem(const GpTail& p, int k, int j)
{
    double dot = 0.0;
    for (int d = 0; d < p.Do; ++d) dot = fma(p.zmean[k * p.Do + d], p.zmean[j * p.Do + d], dot);
    if (k == j)
        for (int d = 0; d < p.Do; ++d) dot += p.zvar[k * p.Do + d];
    return 0.5 * dot;
}
__device__ __forceinline__ double g_elem(const GpTail& p, int k, int j)
{
    const double v = p.gmode ? 0.0 : img_elem(p, k, j);
    if (p.Kkl) return v;                                  // (PR-SSM: the prior-KL part goes through Kkl, tail_gemm<3>)
    double dot = 0.0;
    for (int d = 0; d < p.Do; ++d) dot = fma(p.zmean[k * p.Do + d], p.zmean[j * p.Do + d], dot);
    if (k == j)
        for (int d = 0; d < p.Do; ++d) dot += p.zvar[k * p.Do + d];
    return v + 0.5 * dot;
}

// STAGE 0: T = Kinv G.   STAGE 1: G2 = (-T Kinv + 0.5 Do Kinv) o Kmm   (K^-1 = (K_mm + jitter I)^-1, log det of the KL)
// STAGE 2 (column tiles 0,1: Ws Z~ with Ws = -0.5 (G2 + G2^T); column tile 2: K^-1 zeta_mean), results behind each other
// in the T scratch: WZ [M][D], KM [M][Do]
template <int STAGE>
__global__ __launch_bounds__(256) void tail_gemm(TailArgs a)
{
    const GpTail& p = a.gp[blockIdx.z];
    const int M = p.M;
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    if (int(blockIdx.y) * 16 >= M) return;
    if constexpr (STAGE == 2) {
        const bool km = (blockIdx.x == 2);
        const int jc = km ? tx : j;                       // output column inside WZ / KM
        const int ncol = km ? p.Do : p.D;
        if (!km && int(blockIdx.x) * 16 >= p.D) return;
        double acc = 0.0;
        for (int k0 = 0; k0 < M; k0 += 16) {
            double av = 0.0;
            if (i < M && k0 + tx < M) {
                const int k = k0 + tx;
                av = km ? (p.Kkl ? p.Kkl : p.Kinv)[int64_t(i) * M + k] : -0.5 * (p.G2[int64_t(i) * M + k] + p.G2[int64_t(k) * M + i]);
            }
            As[ty][tx] = av;
            double bv = 0.0;
            if (k0 + ty < M && jc < ncol) bv = km ? p.zmean[(k0 + ty) * p.Do + jc] : p.Zs[(k0 + ty) * p.D + jc];
            Bs[ty][tx] = bv;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = fma(As[ty][kk], Bs[kk][tx], acc);
            __syncthreads();
        }
        if (i < M && jc < ncol) {
            if (km) p.T[int64_t(M) * p.D + i * p.Do + jc] = acc;
            else p.T[i * p.D + jc] = acc;
        }
        return;
    }
    if (int(blockIdx.x) * 16 >= M) return;
    if constexpr (STAGE == 3) {
        if (!p.Kkl) return;                               // PR-SSM only: T2 = Kkl B_KL
    }
    const double* A = (STAGE == 0) ? p.Kinv : (STAGE == 3 ? p.Kkl : p.T);
    double acc = 0.0;
    for (int k0 = 0; k0 < M; k0 += 16) {
        As[ty][tx] = (i < M && k0 + tx < M) ? A[int64_t(i) * M + k0 + tx] : 0.0;
        double b = 0.0;
        if (k0 + ty < M && j < M)
            b = (STAGE == 0) ? g_elem(p, k0 + ty, j) : (STAGE == 3 ? kl_elem(p, k0 + ty, j) : p.Kinv[int64_t(k0 + ty) * M + j]);
        Bs[ty][tx] = b;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = fma(As[ty][kk], Bs[kk][tx], acc);
        __syncthreads();
    }
    double acc2 = 0.0;                                    // PR-SSM, stage 1: (Kkl B_KL) Kkl
    if (STAGE == 1 && p.Kkl) {
        for (int k0 = 0; k0 < M; k0 += 16) {
            As[ty][tx] = (i < M && k0 + tx < M) ? p.T2[int64_t(i) * M + k0 + tx] : 0.0;
            Bs[ty][tx] = (k0 + ty < M && j < M) ? p.Kkl[int64_t(k0 + ty) * M + j] : 0.0;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc2 = fma(As[ty][kk], Bs[kk][tx], acc2);
            __syncthreads();
        }
    }
    if (i < M && j < M) {
        if (STAGE == 0) p.T[int64_t(i) * M + j] = acc;
        else if (STAGE == 3) p.T2[int64_t(i) * M + j] = acc;
        else if (p.Kkl) {
            // K_mm adjoint with the jitter-free prior (prssm.py:81-82,96): -K^-1 B K^-1 - Kkl B_KL Kkl + 0.5 Do Kkl, and
            // tr(Kbar K_mm) = -tr(K^-1 B) + jitter tr(K^-1 B K^-1) - tr(Kkl B_KL) + 0.5 Do M   (Kkl K_mm = I)
            const double jit = p.scal[CBFSSM_SCAL_JITTER];
            double tdiag = (i == j) ? p.T[int64_t(i) * M + i] : 0.0;
            if (p.gmode) {                                // (float32 adjoint: the data part arrives as G = K^-1 B K^-1, see below)
                const double gs = gsym_elem(p, i, j);
                if (i == j) {
                    double d = 0.0;
                    for (int k = 0; k < M; ++k) d = fma(gsym_elem(p, i, k), p.Kmm[int64_t(k) * M + i], d);
                    tdiag += d + jit * gs;
                }
                acc += gs;
            }
            p.G2[int64_t(i) * M + j] = (-acc - acc2 + 0.5 * p.Do * p.Kkl[int64_t(i) * M + j]) * p.Kmm[int64_t(i) * M + j];
            if (i == j) p.dv[i] = -tdiag + jit * acc - p.T2[int64_t(i) * M + i] + 0.5 * p.Do;
        } else {
            const double kinv = p.Kinv[int64_t(i) * M + j];
            double tdiag = (i == j) ? p.T[int64_t(i) * M + i] : 0.0;
            if (p.gmode) {
                // K^-1 (B + B_KL) K^-1 = G + K^-1 B_KL K^-1, and the diagonal of K^-1 B = G (K_mm + jitter I); the jitter
                // terms of the trace below cancel between the two
                const double gs = gsym_elem(p, i, j);
                if (i == j) {
                    double d = 0.0;
                    for (int k = 0; k < M; ++k) d = fma(gsym_elem(p, i, k), p.Kmm[int64_t(k) * M + i], d);
                    tdiag += d;
                }
                acc += gs;
                if (i == j) tdiag += p.scal[CBFSSM_SCAL_JITTER] * gs;
            }
            p.G2[int64_t(i) * M + j] = (-acc + 0.5 * p.Do * kinv) * p.Kmm[int64_t(i) * M + j];
            // d loss / d sigma^2 needs sum_ij Kbar_ij K_mm,ij = tr(Kbar K_mm).  Summing the entries of G2 cancels twice
            // (entries of K^-1 G K^-1 are of order cond^2 |G|): with K^-1 K_mm = I - jitter K^-1 the same trace is
            //     -tr(T) + jitter tr(T K^-1) + 0.5 Do (M - jitter tr K^-1),
            // whose terms are of order cond |G| only (measured on a trained-like K_mm with cond 3e7: 3.5e-2 -> see DESIGN).
            if (i == j) {
                const double jit = p.scal[CBFSSM_SCAL_JITTER];
                p.dv[i] = -tdiag + jit * acc + 0.5 * p.Do * (1.0 - jit * kinv);
            }
        }
    }
}

// One workgroup per GP: everything that is O(M^2 D) or smaller, and the chain through the positivity transforms.
__global__ __launch_bounds__(256) void tail_finish(TailArgs a)
{
    const GpTail& p = a.gp[blockIdx.x];
    const int M = p.M, D = p.D, Do = p.Do, NBLK = p.NBLK, JB = p.JB;
    const int tid = threadIdx.x, NT = 256;
    __shared__ double wsrow[320];
    __shared__ double red[8];
    __shared__ double glsj[32];
    const double* gMu = p.slab;
    const double* gS2 = p.slab + NBLK * 256;
    const double* gZ = p.slab + 2 * NBLK * 256 + (p.stash ? 0 : NBLK * NBLK * 256);
    const double* small = gZ + NBLK * JB * 256;
    const double* G2 = p.G2;

    // K = var exp(-0.5 d2(z~)): Kbar o K = G2;  Wd = -0.5 G2, Ws = Wd + Wd^T                           (gp_tf.py:33-49)
    const double* WZ = p.T;                              // Ws z~   [M][D]     (tail_gemm<2>)
    const double* KM = p.T + int64_t(M) * D;             // K^-1 mu [M][Do]
    const int wv = tid >> 6, l = tid & 63;
    double tot = 0.0;
    for (int i = wv; i < M; i += NT / 64) {              // one wave per row: lanes over the columns
        double s = 0.0;
        for (int j = l; j < M; j += 64) s += G2[int64_t(i) * M + j] + G2[int64_t(j) * M + i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (l == 0) wsrow[i] = -0.5 * s;
    }
    for (int i = tid; i < M; i += NT) tot += p.dv[i];    // tr(Kbar K_mm), per-row terms from tail_gemm<1>
    tot = block_sum_t(tot, red, tid, NT);
    __syncthreads();

    // Z~bar = Ebar x~^T - z~ o rowsum(Ebar) + 2 (rowsum(Ws) o z~ - Ws z~);   zbar = Z~bar / ls
    // lengthscales: -(colsum(Z~bar o z~) + inputs' part) / ls; thread j < D walks its column (fixed order)
    for (int idx = tid; idx < M * D; idx += NT) {
        const int i = idx / D, j = idx - i * D;
        const double zs = p.Zs[idx];
        const double gzt = c_image(gZ, JB, i, j) - zs * c_image(gZ, JB, i, D) + 2.0 * (wsrow[i] * zs - WZ[idx]);
        p.g_z[idx] = gzt / p.ls[p.shared_ls ? 0 : j];
    }
    __syncthreads();
    for (int j = wv; j < D; j += NT / 64) {              // one wave per column: lanes over the rows (fixed order)
        double s = 0.0;
        for (int i = l; i < M; i += 64) s += p.g_z[i * D + j] * p.Zs[i * D + j];      // = Z~bar o z~ / ls
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (l == 0) {
            if (p.shared_ls) glsj[j] = -(s + small[32 + j] / p.ls[0]);
            else p.g_ls[j] = -(s + small[32 + j] / p.ls[j]) * sigmoid(p.ls_unc[j]);
        }
    }
    if (p.shared_ls) {                                   // one lengthscale: its adjoint is the sum over the input dimensions
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += glsj[j];
            p.g_ls[0] = s * sigmoid(p.ls_unc[0]);
        }
    }
    if (tid == 0) {
        const double var = p.var[0];
        p.g_var[0] = (tot / var + small[96] + small[97] / var) * sigmoid(p.var_unc[0]);
    }
    // inducing mean / variance: data part from the slab + prior KL                                    (gp_tf.py:163-172)
    for (int idx = tid; idx < M * Do; idx += NT) {
        const int i = idx / Do, d = idx - i * Do;
        p.g_mu[idx] = c_image(gMu, 1, i, d) + KM[idx];
        const double gs2 = c_image(gS2, 1, i, d) + 0.5 * ((p.Kkl ? p.Kkl : p.Kinv)[int64_t(i) * M + i] - 1.0 / p.zvar[idx]);
        p.g_s2[idx] = gs2 * sigmoid(p.zvar_unc[idx]);
    }
    // process / observation noise (workgroup of gp_f): per-dimension sums of both slabs + the log-likelihood's pull
    if (blockIdx.x == 0 && tid < a.dim_x) {
        double gvx = small[tid];
        if (a.ngp > 1) {
            const GpTail& pb = a.gp[1];
            const double* small_b = pb.slab + 2 * pb.NBLK * 256 + (pb.stash ? 0 : pb.NBLK * pb.NBLK * 256) + pb.NBLK * pb.JB * 256;
            if (tid < pb.Do) gvx += small_b[tid];
        }
        a.g_vx[tid] = gvx * sigmoid(a.vx_unc[tid]);
        if (tid < a.n_vy) {
            double gvy = small[16 + tid];
            if (tid < a.dim_y) gvy += a.tail[3 + tid];
            a.g_vy[tid] = gvy * sigmoid(a.vy_unc[tid]);
        }
    }
}

struct ConstrainArgs {
    const double* p;
    double* c;
    int64_t off[13];          // segment starts (PARAM order) + total
    int unc[12];
};

__global__ __launch_bounds__(256) void constrain_kernel(ConstrainArgs a)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= a.off[12]) return;
    int seg = 0;
#pragma unroll
    for (int s = 1; s < 12; ++s) seg += (i >= a.off[s]);
    const double x = a.p[i];
    // softplus(x) + 1e-10                                                                           (tf_transform.py:19-21)
    a.c[i] = a.unc[seg] ? (fmax(x, 0.0) + log1p(exp(-fabs(x))) + 1e-10) : x;
}

// Data scalars of a rank and the log-likelihood's pull on var_y (cbfssm.py:245-251) into the tail of the flat reduce
// buffer: tail = [loglik, kl_x, entropy, d loss/d var_y[0..dim_y)].  One workgroup per dimension (it was one wave per
// dimension in ONE workgroup: 55 dependent strided loads per lane, 36 us at C3), fixed order.
__global__ __launch_bounds__(256) void data_tail_kernel(const double* ll_part, int64_t nblk, int dim_y, const double* var_y,
                                                        const double* out8, double cL, double bts, double* tail)
{
    __shared__ double red[4];
    const int tid = threadIdx.x, d = blockIdx.x;
    double s = 0.0;
    for (int64_t k = tid; k < nblk; k += 256) s += ll_part[k * dim_y + d];
    const double tot = block_sum_t(s, red, tid, 256);
    if (tid == 0) {
        const double vy = var_y[d];
        // ll_d = -0.5 [ sq_d / vy + B T S (log 2 pi + log vy) ]  =>  sq_d, and d(-cL ll)/d vy
        const double sq = (-2.0 * tot - bts * (1.8378770664093454836 + log(vy))) * vy;
        tail[3 + d] = -cL * 0.5 * (sq / (vy * vy) - bts / vy);
    }
    if (d == 0 && tid < 3) tail[tid] = out8[tid];
}

__global__ void adam_tick_kernel(double* t) { t[0] += 1.0; }

// tf.train.AdamOptimizer (TF 1.8): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);  p -= lr_t m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_kernel(int64_t n, double* p, const double* g, double* m, double* v, const double* t,
                                                   double lr, double b1, double b2, double eps)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const double tt = t[0];
    const double lr_t = lr * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt));
    const double gi = g[i];
    const double mi = b1 * m[i] + (1.0 - b1) * gi;
    const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrt(vi) + eps);
}

}  // namespace cbfssm

using namespace cbfssm;

extern "C" {

int cbfssm_param_layout_init(int M, int dim_x, int dim_u, int dim_y, cbfssm_param_layout* out)
{
    if (!out) return fail(-1, "null layout");
    if (M < 1 || dim_x < 1 || dim_u < 0 || dim_y < 1 || dim_y > dim_x) return fail(-1, "bad dimensions");
    const int D = dim_x + dim_u;
    const int Do[2] = {dim_x, dim_x - dim_y};
    int64_t o = 0;
    int k = 0;
    for (int g = 0; g < 2; ++g) {
        out->off[k++] = o; o += int64_t(M) * D;        // zeta_pos
        out->off[k++] = o; o += int64_t(M) * Do[g];    // zeta_mean
        out->off[k++] = o; o += int64_t(M) * Do[g];    // zeta_var_unc
        out->off[k++] = o; o += 1;                     // variance_unc
        out->off[k++] = o; o += D;                     // lengthscales_unc
    }
    out->off[k++] = o; o += dim_x;                     // var_x_unc
    out->off[k++] = o; o += dim_x;                     // var_y_unc
    out->total = o;
    out->M = M; out->D = D; out->dim_x = dim_x; out->dim_y = dim_y;
    return 0;
}

int cbfssm_constrain_f64(const cbfssm_param_layout* pl, const double* pflat, double* cflat, void* stream)
{
    if (!pl || !pflat || !cflat) return fail(-1, "null pointer");
    ConstrainArgs a;
    a.p = pflat; a.c = cflat;
    for (int i = 0; i < 12; ++i) a.off[i] = pl->off[i];
    a.off[12] = pl->total;
    static const int unc[12] = {0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1};
    memcpy(a.unc, unc, sizeof(unc));
    const unsigned nb = unsigned((pl->total + 255) / 256);
    hipLaunchKernelGGL(constrain_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "constrain launch failed");
}

int64_t cbfssm_train_tail_work_elems(const cbfssm_pack_layout* Lf, const cbfssm_pack_layout* Lb)
{
    if (!Lf || !Lb) return -1;
    auto one = [](const cbfssm_pack_layout* L) {
        return int64_t(L->M) * (L->M > 48 ? L->M : 48) + int64_t(L->M) * L->M + (int64_t(L->M) + 63) / 64 * 64;
    };
    return one(Lf) + one(Lb);
}

int cbfssm_train_tail_f64(const cbfssm_param_layout* pl, const cbfssm_pack_layout* Lf, const double* pack_f,
                          const cbfssm_pack_layout* Lb, const double* pack_b, const double* red, const double* gB_dense_f,
                          const double* gB_dense_b, int64_t gB_ld, const double* pflat, const double* cflat, double* work,
                          double* gflat, void* stream)
{
    return cbfssm_train_tail_g_f64(pl, Lf, pack_f, Lb, pack_b, red, gB_dense_f, gB_dense_b, gB_ld, 0, pflat, cflat, work, gflat,
                                   stream);
}

int cbfssm_train_tail_g_f64(const cbfssm_param_layout* pl, const cbfssm_pack_layout* Lf, const double* pack_f,
                            const cbfssm_pack_layout* Lb, const double* pack_b, const double* red, const double* gB_dense_f,
                            const double* gB_dense_b, int64_t gB_ld, int g_mode, const double* pflat, const double* cflat,
                            double* work, double* gflat, void* stream)
{
    if (g_mode < 0 || g_mode > 2) return fail(-1, "g_mode must be 0, 1 or 2");
    if (!pl || !Lf || !Lb || !pack_f || !pack_b || !red || !pflat || !cflat || !work || !gflat)
        return fail(-1, "null pointer");
    if (Lf->M != pl->M || Lb->M != pl->M || Lf->D != pl->D || Lb->D != pl->D) return fail(-1, "layouts disagree");
    if (Lf->Do != pl->dim_x || Lb->Do != pl->dim_x - pl->dim_y) return fail(-1, "output dims disagree");
    if (Lf->rev_slab <= 0 || Lb->rev_slab <= 0) return fail(-3, "no adjoint slab for M=%d", Lf->M);
    if (Lf->M > 320 || pl->D > 32) return fail(-3, "tail kernel limits: M <= 320, D <= 32");
    if ((Lf->rev_stash != 0) != (gB_dense_f != nullptr) || (Lb->rev_stash != 0) != (gB_dense_b != nullptr))
        return fail(-1, "dense K^-1 adjoints are required exactly in stash mode");
    TailArgs a;
    memset(&a, 0, sizeof(a));
    const cbfssm_pack_layout* L[2] = {Lf, Lb};
    const double* pack[2] = {pack_f, pack_b};
    const double* gBd[2] = {gB_dense_f, gB_dense_b};
    const double* slab = red;
    double* w = work;
    for (int g = 0; g < 2; ++g) {
        GpTail& p = a.gp[g];
        const int M = L[g]->M;
        p.slab = slab; slab += L[g]->rev_slab;
        p.gB_dense = gBd[g]; p.gB_ld = gB_ld;
        p.Kinv = pack[g] + L[g]->Kinv; p.Kmm = pack[g] + L[g]->Kmm; p.Zs = pack[g] + L[g]->Zs;
        p.T = w; w += int64_t(M) * (M > 48 ? M : 48);
        p.G2 = w; w += int64_t(M) * M;
        p.dv = w; w += (int64_t(M) + 63) / 64 * 64;
        p.scal = pack[g] + L[g]->scal;
        const int64_t* off = pl->off + 5 * g;
        p.zmean = cflat + off[1]; p.zvar = cflat + off[2]; p.var = cflat + off[3]; p.ls = cflat + off[4];
        p.zvar_unc = pflat + off[2]; p.var_unc = pflat + off[3]; p.ls_unc = pflat + off[4];
        p.g_z = gflat + off[0]; p.g_mu = gflat + off[1]; p.g_s2 = gflat + off[2]; p.g_var = gflat + off[3];
        p.g_ls = gflat + off[4];
        p.M = M; p.D = L[g]->D; p.Do = L[g]->Do; p.NBLK = L[g]->NBLK; p.JB = L[g]->JB; p.stash = L[g]->rev_stash;
        p.gmode = g_mode;
        p.Kkl = nullptr; p.T2 = nullptr; p.shared_ls = 0;
    }
    a.ngp = 2; a.n_vy = pl->dim_x;
    a.tail = slab;
    a.vx_unc = pflat + pl->off[10]; a.vy_unc = pflat + pl->off[11];
    a.g_vx = gflat + pl->off[10]; a.g_vy = gflat + pl->off[11];
    a.dim_x = pl->dim_x; a.dim_y = pl->dim_y;
    const unsigned nb = unsigned((pl->M + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tail_gemm<0>, dim3(nb, nb, 2), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_gemm<1>, dim3(nb, nb, 2), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_gemm<2>, dim3(3, nb, 2), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_finish, dim3(2), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "train tail launch failed");
}

int64_t cbfssm_train_tail_half_work_elems(const cbfssm_pack_layout* L)
{
    if (!L) return -1;
    return int64_t(L->M) * (L->M > 48 ? L->M : 48) + 2 * int64_t(L->M) * L->M + (int64_t(L->M) + 63) / 64 * 64;
}

int cbfssm_train_tail_half_f64(const cbfssm_pack_layout* L, const double* pack, const double* pack_kl, int shared_ls,
                               const double* red, const double* gB_dense, int64_t gB_ld, int g_mode, int dim_y, const double* pflat,
                               const double* cflat, double* work, double* gflat, void* stream)
{
    if (g_mode < 0 || g_mode > 2) return fail(-1, "g_mode must be 0, 1 or 2");
    if (!L || !pack || !red || !pflat || !cflat || !work || !gflat) return fail(-1, "null pointer");
    if (L->rev_slab <= 0) return fail(-3, "no adjoint slab for M=%d", L->M);
    if (L->M > 320 || L->D > 32) return fail(-3, "tail kernel limits: M <= 320, D <= 32");
    if ((L->rev_stash != 0) != (gB_dense != nullptr)) return fail(-1, "the dense K^-1 adjoint is required exactly in stash mode");
    if (dim_y < 1 || dim_y > L->Do) return fail(-1, "bad dim_y");
    TailArgs a;
    memset(&a, 0, sizeof(a));
    GpTail& p = a.gp[0];
    const int M = L->M, D = L->D, Do = L->Do;
    p.slab = red;
    p.gB_dense = gB_dense; p.gB_ld = gB_ld;
    p.Kinv = pack + L->Kinv; p.Kmm = pack + L->Kmm; p.Zs = pack + L->Zs;
    double* w = work;
    p.T = w; w += int64_t(M) * (M > 48 ? M : 48);
    p.G2 = w; w += int64_t(M) * M;
    p.T2 = w; w += int64_t(M) * M;
    p.dv = w;
    p.scal = pack + L->scal;
    p.Kkl = pack_kl ? pack_kl + L->Kinv : nullptr;
    p.shared_ls = shared_ls ? 1 : 0;
    // flat layout: zeta_pos [M][D] | zeta_mean [M][Do] | zeta_var(_unc) [M][Do] | variance(_unc) [1] | lengthscales(_unc)
    // [D or 1] | var_x(_unc) [Do] | var_y(_unc) [dim_y]
    const int64_t o1 = int64_t(M) * D, o2 = o1 + int64_t(M) * Do, o3 = o2 + int64_t(M) * Do, o4 = o3 + 1,
                  o5 = o4 + (shared_ls ? 1 : D), o6 = o5 + Do;
    p.zmean = cflat + o1; p.zvar = cflat + o2; p.var = cflat + o3; p.ls = cflat + o4;
    p.zvar_unc = pflat + o2; p.var_unc = pflat + o3; p.ls_unc = pflat + o4;
    p.g_z = gflat; p.g_mu = gflat + o1; p.g_s2 = gflat + o2; p.g_var = gflat + o3; p.g_ls = gflat + o4;
    p.M = M; p.D = D; p.Do = Do; p.NBLK = L->NBLK; p.JB = L->JB; p.stash = L->rev_stash; p.gmode = g_mode;
    a.tail = red + L->rev_slab;
    a.vx_unc = pflat + o5; a.vy_unc = pflat + o6;
    a.g_vx = gflat + o5; a.g_vy = gflat + o6;
    a.dim_x = Do; a.dim_y = dim_y; a.ngp = 1; a.n_vy = dim_y;
    const unsigned nb = unsigned((M + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tail_gemm<0>, dim3(nb, nb, 1), dim3(16, 16), 0, st, a);
    if (pack_kl) hipLaunchKernelGGL(tail_gemm<3>, dim3(nb, nb, 1), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_gemm<1>, dim3(nb, nb, 1), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_gemm<2>, dim3(3, nb, 1), dim3(16, 16), 0, st, a);
    hipLaunchKernelGGL(tail_finish, dim3(1), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "train tail (forward-only variant) launch failed");
}

int cbfssm_data_tail_f64(const cbfssm_problem* p, const double* var_y, const double* ll_part, const double* out8, double cL,
                         double* tail, void* stream)
{
    if (!p || !var_y || !ll_part || !out8 || !tail) return fail(-1, "null pointer");
    const int64_t nblk = (int64_t(p->B) * p->T * p->dim_x + 255) / 256;
    hipLaunchKernelGGL(data_tail_kernel, dim3(unsigned(p->dim_y > 0 ? p->dim_y : 1)), dim3(256), 0, (hipStream_t)stream, ll_part, nblk, p->dim_y, var_y, out8, cL,
                       double(p->B) * p->T * p->S, tail);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "data tail launch failed");
}

int cbfssm_adam_step_f64(int64_t n, double* pflat, const double* gflat, double* m, double* v, double* t_dev, double lr,
                         double beta1, double beta2, double eps, void* stream)
{
    if (n < 1 || !pflat || !gflat || !m || !v || !t_dev) return fail(-1, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, t_dev);
    hipLaunchKernelGGL(adam_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, n, pflat, gflat, m, v,
                       (const double*)t_dev, lr, beta1, beta2, eps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "adam launch failed");
}

}  // extern "C"
