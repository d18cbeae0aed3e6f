// Recognition model of the forward-only variants: x_0 = dense(GRU(16)(reversed first recog_len steps of [u, y]))
// (cbfssm/model/cbfssmhalf.py:82-93, cbfssm/model/prssm.py:132-141: tf.contrib.rnn.GRUCell(16) + tf.layers.dense, TF 1.8
// gate layout: [r | u] = sigmoid([x, h] W_g + b_g), c = tanh([x, r o h] W_c + b_c), h' = u o h + (1 - u) o c).
//
// A few hundred FLOPs per sequence and step -- but written with a tensor library it is ~15 launches per GRU step forwards
// and ~35 backwards: 800 launches for recog_len = 16, most of a CBFSSMHALF train step at the small-scale shapes (3.5 ms
// of a 4.1 ms step at the Actuator shape even inside a HIP graph).  Here: ONE wave per sequence walks the recog_len steps,
// forwards (keeping h, r, u, c of every step: 64 doubles) and backwards; the weight gradients leave as one slab per
// sequence and are summed in a fixed order by cbfssm_reduce_partials_f64 (no atomics: reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cbfssm_hip.h"

namespace cbfssm {

int fail(int rc, const char* fmt, ...);   // cbfssm_api.hip

constexpr int GRU_H = 16;                 // cbfssmhalf.py:84
constexpr int GRU_MAXIN = 32;             // dim_u + dim_y

struct GruArgs {
    int B, T, dim_u, dim_y, dim_x, R;
    const double* u;
    const double* y;
    const double* Wg;     // [n_in + 16][32]
    const double* bg;     // [32]
    const double* Wc;     // [n_in + 16][16]
    const double* bc;     // [16]
    const double* Wd;     // [16][dim_x]
    const double* bd;     // [dim_x]
    double* x0;           // [B][dim_x]
    double* act;          // [B][R][64]: h (before the step), r, u, c     (+ [B][16]: h after the last step, behind it)
    const double* gx0;    // [B][dim_x]   d loss / d x_0
    double* gpart;        // [B][P]       per-sequence gradient slabs, the six tensors behind each other
    int64_t P;
};

__device__ __forceinline__ double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }

__device__ __forceinline__ double gru_input(const GruArgs& a, int b, int step, int i)
{
    const int t = a.R - 1 - step;                                           // the reversed window (cbfssmhalf.py:86)
    return (i < a.dim_u) ? a.u[(int64_t(b) * a.T + t) * a.dim_u + i] : a.y[(int64_t(b) * a.T + t) * a.dim_y + (i - a.dim_u)];
}

__global__ __launch_bounds__(64) void gru_forward_kernel(GruArgs a)
{
    __shared__ double xs[GRU_MAXIN], hs[GRU_H], rh[GRU_H];
    const int l = threadIdx.x, b = blockIdx.x;
    const int n_in = a.dim_u + a.dim_y;
    double h = 0.0;                                                          // lane j < 16 carries h_j
    if (l < GRU_H) hs[l] = 0.0;
    for (int step = 0; step < a.R; ++step) {
        if (l < n_in) xs[l] = gru_input(a, b, step, l);
        __syncthreads();
        double r = 0.0, z = 0.0;
        if (l < 2 * GRU_H) {                                                 // gate column l: r (l < 16) or u (l >= 16)
            double s = a.bg[l];
            for (int i = 0; i < n_in; ++i) s = fma(xs[i], a.Wg[i * 32 + l], s);
            for (int k = 0; k < GRU_H; ++k) s = fma(hs[k], a.Wg[(n_in + k) * 32 + l], s);
            s = sigm(s);
            if (l < GRU_H) r = s; else z = s;
        }
        z = __shfl(z, (l & 15) + 16);                                        // lane j < 16: r_j and u_j
        if (l < GRU_H) rh[l] = r * h;
        __syncthreads();
        if (l < GRU_H) {
            double s = a.bc[l];
            for (int i = 0; i < n_in; ++i) s = fma(xs[i], a.Wc[i * 16 + l], s);
            for (int k = 0; k < GRU_H; ++k) s = fma(rh[k], a.Wc[(n_in + k) * 16 + l], s);
            const double c = tanh(s);
            if (a.act) {
                double* o = a.act + (int64_t(b) * a.R + step) * 64;
                o[l] = h; o[16 + l] = r; o[32 + l] = z; o[48 + l] = c;
            }
            h = z * h + (1.0 - z) * c;
        }
        __syncthreads();                                                     // xs, hs, rh are rewritten below
        if (l < GRU_H) hs[l] = h;
    }
    __syncthreads();
    if (a.act && l < GRU_H) a.act[int64_t(a.B) * a.R * 64 + int64_t(b) * GRU_H + l] = h;
    if (l < a.dim_x) {
        double s = a.bd[l];
        for (int k = 0; k < GRU_H; ++k) s = fma(hs[k], a.Wd[k * a.dim_x + l], s);
        a.x0[int64_t(b) * a.dim_x + l] = s;
    }
}

// reverse mode through the dense layer and the recog_len GRU steps of one sequence
__global__ __launch_bounds__(64) void gru_backward_kernel(GruArgs a)
{
    __shared__ double xs[GRU_MAXIN], hp[GRU_H], rs[GRU_H], dcp[GRU_H], dgp[2 * GRU_H], dhs[GRU_H], gxs[GRU_H];
    const int l = threadIdx.x, b = blockIdx.x;
    const int n_in = a.dim_u + a.dim_y, nrow = n_in + GRU_H;
    double* slab = a.gpart + int64_t(b) * a.P;
    double* gWg = slab;
    double* gbg = gWg + int64_t(nrow) * 32;
    double* gWc = gbg + 32;
    double* gbc = gWc + int64_t(nrow) * 16;
    double* gWd = gbc + 16;
    double* gbd = gWd + int64_t(GRU_H) * a.dim_x;
    // dense layer: x_0 = h_R W_d + b_d
    if (l < GRU_H) gxs[l] = (l < a.dim_x) ? a.gx0[int64_t(b) * a.dim_x + l] : 0.0;
    const double hR = (l < GRU_H) ? a.act[int64_t(a.B) * a.R * 64 + int64_t(b) * GRU_H + l] : 0.0;
    __syncthreads();
    double dh = 0.0;
    if (l < GRU_H) {
        for (int d = 0; d < a.dim_x; ++d) {
            gWd[l * a.dim_x + d] = hR * gxs[d];
            dh = fma(a.Wd[l * a.dim_x + d], gxs[d], dh);
        }
    }
    if (l < a.dim_x) gbd[l] = gxs[l];
    // accumulators of this lane: W_g column l & 31, rows (l >> 5) + 2 k;  W_c column l & 15, rows (l >> 4) + 4 k
    constexpr int NG = (GRU_MAXIN + GRU_H + 1) / 2, NC = (GRU_MAXIN + GRU_H + 3) / 4;
    double accg[NG], accc[NC], accbg = 0.0, accbc = 0.0;
#pragma unroll
    for (int k = 0; k < NG; ++k) accg[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) accc[k] = 0.0;
    for (int step = a.R - 1; step >= 0; --step) {
        const double* o = a.act + (int64_t(b) * a.R + step) * 64;
        double h = 0.0, r = 0.0, z = 0.0, c = 0.0;
        if (l < GRU_H) { h = o[l]; r = o[16 + l]; z = o[32 + l]; c = o[48 + l]; hp[l] = h; rs[l] = r; }
        if (l < n_in) xs[l] = gru_input(a, b, step, l);
        // h' = u h + (1 - u) c
        double dz = 0.0, dhn = 0.0;
        if (l < GRU_H) {
            dz = dh * (h - c);
            dcp[l] = dh * (1.0 - z) * (1.0 - c * c);                          // through tanh
            dhn = dh * z;
        }
        __syncthreads();
        // candidate: c = tanh([x, r o h] W_c + b_c)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int i = (l >> 4) + 4 * k;
            if (i < nrow) accc[k] = fma((i < n_in) ? xs[i] : rs[i - n_in] * hp[i - n_in], dcp[l & 15], accc[k]);
        }
        if (l < GRU_H) accbc += dcp[l];
        double dr = 0.0;
        if (l < GRU_H) {
            double drh = 0.0;                                                 // d loss / d (r o h)_l
            for (int j = 0; j < GRU_H; ++j) drh = fma(a.Wc[(n_in + l) * 16 + j], dcp[j], drh);
            dr = drh * h;
            dhn = fma(drh, r, dhn);
            dgp[l] = dr * r * (1.0 - r);                                      // through the sigmoids
            dgp[16 + l] = dz * z * (1.0 - z);
        }
        __syncthreads();
        // gates: [r | u] = sigmoid([x, h] W_g + b_g)
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = (l >> 5) + 2 * k;
            if (i < nrow) accg[k] = fma((i < n_in) ? xs[i] : hp[i - n_in], dgp[l & 31], accg[k]);
        }
        if (l < 2 * GRU_H) accbg += dgp[l];
        if (l < GRU_H) {
            for (int j = 0; j < 2 * GRU_H; ++j) dhn = fma(a.Wg[(n_in + l) * 32 + j], dgp[j], dhn);
            dh = dhn;
        }
        __syncthreads();                                                      // the shared vectors are rewritten above
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = (l >> 5) + 2 * k;
        if (i < nrow) gWg[i * 32 + (l & 31)] = accg[k];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int i = (l >> 4) + 4 * k;
        if (i < nrow) gWc[i * 16 + (l & 15)] = accc[k];
    }
    if (l < 2 * GRU_H) gbg[l] = accbg;
    if (l < GRU_H) gbc[l] = accbc;
}

static int fill_gru(GruArgs& a, int B, int T, int dim_u, int dim_y, int dim_x, int recog_len, const double* u, const double* y,
                    const double* params)
{
    if (B < 1 || T < 1 || dim_u < 0 || dim_y < 1 || dim_x < 1 || recog_len < 1) return fail(-1, "bad dimensions");
    if (dim_u + dim_y > GRU_MAXIN || dim_x > GRU_H) return fail(-3, "recognition kernel limits: dim_u + dim_y <= 32, dim_x <= 16");
    if (recog_len > T) return fail(-1, "recog_len exceeds the sequence length");
    if (!u && dim_u > 0) return fail(-1, "null pointer");
    if (!y || !params) return fail(-1, "null pointer");
    const int nrow = dim_u + dim_y + GRU_H;
    a.B = B; a.T = T; a.dim_u = dim_u; a.dim_y = dim_y; a.dim_x = dim_x; a.R = recog_len;
    a.u = u; a.y = y;
    a.Wg = params; a.bg = a.Wg + int64_t(nrow) * 32; a.Wc = a.bg + 32; a.bc = a.Wc + int64_t(nrow) * 16;
    a.Wd = a.bc + 16; a.bd = a.Wd + int64_t(GRU_H) * dim_x;
    a.P = int64_t(nrow) * 48 + 48 + int64_t(GRU_H) * dim_x + dim_x;
    a.x0 = nullptr; a.act = nullptr; a.gx0 = nullptr; a.gpart = nullptr;
    return 0;
}

}  // namespace cbfssm

using namespace cbfssm;

extern "C" {

int64_t cbfssm_gru_recog_param_elems(int dim_u, int dim_y, int dim_x)
{
    if (dim_u < 0 || dim_y < 1 || dim_x < 1) return -1;
    return int64_t(dim_u + dim_y + GRU_H) * 48 + 48 + int64_t(GRU_H) * dim_x + dim_x;
}

int64_t cbfssm_gru_recog_act_elems(int B, int recog_len)
{
    if (B < 1 || recog_len < 1) return -1;
    return int64_t(B) * recog_len * 64 + int64_t(B) * GRU_H;
}

int cbfssm_gru_recog_f64(int B, int T, int dim_u, int dim_y, int dim_x, int recog_len, const double* u, const double* y,
                         const double* params, double* x0, double* act, void* stream)
{
    GruArgs a;
    int rc = fill_gru(a, B, T, dim_u, dim_y, dim_x, recog_len, u, y, params);
    if (rc) return rc;
    if (!x0) return fail(-1, "null pointer");
    a.x0 = x0; a.act = act;
    hipLaunchKernelGGL(gru_forward_kernel, dim3(unsigned(B)), dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "gru forward launch failed");
}

int cbfssm_gru_recog_bwd_f64(int B, int T, int dim_u, int dim_y, int dim_x, int recog_len, const double* u, const double* y,
                             const double* params, const double* act, const double* gx0, double* gpart, void* stream)
{
    GruArgs a;
    int rc = fill_gru(a, B, T, dim_u, dim_y, dim_x, recog_len, u, y, params);
    if (rc) return rc;
    if (!act || !gx0 || !gpart) return fail(-1, "null pointer");
    a.act = const_cast<double*>(act); a.gx0 = gx0; a.gpart = gpart;
    hipLaunchKernelGGL(gru_backward_kernel, dim3(unsigned(B)), dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "gru backward launch failed");
}

}  // extern "C"
