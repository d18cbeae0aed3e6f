// float32-arithmetic variant of the forward evaluation (BASELINE.json configs[4]: reduced-precision ELBO sweep; the
// reference's `CBFSSM(config, dtype)` argument, cbfssm/model/cbfssm.py:12).  As in the reference's float32 mode the
// Cholesky of K_mm is computed in float64 and cast (gp_tf.py:57-65): the operands come from the float64 pack of
// cbfssm_gp_prepare_f64, re-packed as float32 MFMA images by cbfssm_gp_pack_f32; everything inside the time loops --
// kernel tile, exp, the K^-1 K contraction on v_mfma_f32_16x16x4_f32, the step epilogues -- is float32.  Storage in HBM
// (inputs, noise, trajectories) stays float64 so that the buffers of the float64 path are shared; the per-pass sums of
// the KL / entropy terms are kept in float64 (a float32 sum over 10^7 terms would measure the summation, not the model).
//
// v_mfma_f32_16x16x4_f32 layouts: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// C/D[row = 4 (lane >> 4) + reg][col = lane & 15]  -- the C layout differs from the f64 instruction's
// (row = (lane >> 4) + 4 reg), so the "accumulator is the next product's B operand" chain of the f64 kernels holds here
// with a PERMUTED k order: register r of row block b is the B operand of a k-step whose four k indices are the rows
// 16 b + r, + 4, + 8, + 12.  The float32 operand images are packed in that order (pack32_kernel).
#include "cbfssm_f32.hpp"

namespace cbfssm {
namespace f32 {

__global__ void pack32_kernel(const double* pack, cbfssm_pack_layout L, float* out, Off32 o, int bf16)
{
    const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = int64_t(gridDim.x) * blockDim.x;
    const int NBLK = L.NBLK, KS = L.KS, DK = L.DK, Mp = L.Mp, Dp = L.Dp, JB = L.JB;
    // K^-1 (and W = L^-1, W^T for the two-triangular form): target (rb, s, g, nl) holds Kinv[16 rb + nl][16 (s >> 2) + 4 g +
    // (s & 3)]; the f64 image holds Kinv[16 rb + nl][4 s' + g'] at (rb, s', g', nl)  ->  s' = 4 (s >> 2) + g, g' = s & 3
    for (int64_t i = tid; i < int64_t(NBLK) * KS * 64; i += nt) {
        const int l = int(i & 63), s = int((i >> 6) % KS), rb = int((i >> 6) / KS);
        const int g = l >> 4, nl = l & 15;
        const int sp = 4 * (s >> 2) + g, gp = s & 3;
        const int64_t src = (int64_t(rb) * KS + sp) * 64 + gp * 16 + nl;
        const float kv = float(pack[L.Bp + src]);
        out[o.Bp + i] = bf16 ? round_bf16(kv) : kv;
        out[o.Wp + i] = float(pack[L.Wp + src]);
        out[o.WTp + i] = float(pack[L.WTp + src]);
        out[o.BpN + i] = float(pack[L.Bp + i]);                 // natural k order (adjoint: B operands from LDS tiles)
        out[o.WpN + i] = float(pack[L.Wp + i]);
        out[o.WTpN + i] = float(pack[L.WTp + i]);
    }
    for (int64_t i = tid; i < int64_t(NBLK) * DK * 64; i += nt) out[o.Zp + i] = float(pack[L.Zp + i]);
    for (int64_t i = tid; i < Mp; i += nt) out[o.cz + i] = float(pack[L.cz + i]);
    // zeta_mean / zeta_var: target (rb, r, g, nl) holds z[16 rb + 4 g + r][nl]; the f64 image holds z[16 rb + 4 r' + g'][nl]
    for (int64_t i = tid; i < int64_t(NBLK) * 256; i += nt) {
        const int l = int(i & 63), r = int((i >> 6) & 3), rb = int(i >> 8);
        const int g = l >> 4, nl = l & 15;
        const int64_t src = (int64_t(rb) * 4 + g) * 64 + r * 16 + nl;
        out[o.mu + i] = float(pack[L.muA + src]);
        out[o.s2 + i] = float(pack[L.s2A + src]);
        out[o.muB + i] = float(pack[L.muB + i]);
        out[o.s2B + i] = float(pack[L.s2B + i]);
    }
    // (Z~)^T: target ((rb, jb), r, g, nl) holds ZT[j = 16 jb + nl][m = 16 rb + 4 g + r]; the f64 image holds m = 16 rb + 4 s' + g'
    for (int64_t i = tid; i < int64_t(NBLK) * JB * 256; i += nt) {
        const int l = int(i & 63), r = int((i >> 6) & 3);
        const int64_t blk = i >> 8;
        const int g = l >> 4, nl = l & 15;
        out[o.ZTq + i] = float(pack[L.ZT + (blk * 4 + g) * 64 + r * 16 + nl]);
    }
    for (int64_t i = tid; i < Dp; i += nt) out[o.invl + i] = float(pack[L.invl + i]);
    if (tid == 0) {
        out[o.scal] = float(pack[L.scal + CBFSSM_SCAL_SIGMA2]);
        out[o.scal + 1] = bf16 ? 1.0f : 0.0f;
    }
}

struct Args32 {
    Pack32 pk;
    int N, S, T, B;
    int dim_x, dim_u, dim_y;
    int Do, D;
    int recog_len, condition;
    float k_factor;
    const double* var_x;
    const double* var_y;
    const double* u;
    const double* y;
    const double* eps;
    const double* hid;
    const double* y2_in;
    double* y2_out;
    double* x_out;
    double* part_out;
    double* fmv;          // optional: (fmean, fvar) of every step, kept for the adjoint (PassArgs::fmv layout)
    float* a2s;           // optional: every step's [A2 | kernel tile] accumulator registers, [slot][group][2][NBLK][4][64] floats
                          // (slot = t, or run T + t for the backward runs): the float32 adjoint reads them instead of
                          // recomputing both (cbfssm_rev32.hip, KSV)
    double* h_all;        // optional (backward runs): every step's output of both runs
    int tri;              // 1: two-triangular GP form (layout->gp_form == CBFSSM_GP_FORM_TRI)
    int half;             // forward-only variants (CBFSSMHALF / PRSSM, cbfssmhalf.py:117-172): x_0 from x0, the Kalman update on
    const double* x0;     // the observed dims only (d < dim_y), var_y with dim_y entries; x0: (B, dim_x) recognition-model output
    int group0;           // chain-group split: this launch covers the 16-chain groups [group0, group0 + gridDim.x)
    int nseg0;
    // predict
    const double* X;
    int64_t npts;
    double* fmean;
    double* fvar;
};

template <int NBLK, int RB, int DK>
struct Tile32 {
    static constexpr int W = (NBLK + RB - 1) / RB;
    static constexpr int NT = 64 * W;
    static constexpr int MP = 16 * NBLK;
    static constexpr int KS = MP / 4;
    static constexpr int QPW = (4 + W - 1) / W;
    static constexpr int LDS_FLOATS = DK * 64 + 2 * MP * 16 + W * 512;     // xq, K tile, A tile (two-triangular form), partials

    // M <= 112 (one row block per wave): the K^-1 rows of the wave stay in VGPRs for the whole pass (28 registers in
    // float32) -- no operand stream in the time loop
    static constexpr bool BREG = (NBLK <= 7 && RB == 1);
    float Zreg[RB][DK];
    float czr[RB][4];
    float Breg[BREG ? KS : 1];
    const float* Bp;
    const float* Wp;
    const float* WTp;
    const float* mu;
    const float* s2;
    float sigma2;
    bool bf16;

    __device__ __forceinline__ void load(const Pack32& pk, int w, int l)
    {
        Bp = pk.Bp; Wp = pk.Wp; WTp = pk.WTp; mu = pk.mu; s2 = pk.s2; sigma2 = pk.scal[0];
        bf16 = pk.scal[1] != 0.0f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            const bool ok = rb < NBLK;
            const int rbc = ok ? rb : 0;
#pragma unroll
            for (int s = 0; s < DK; ++s) Zreg[i][s] = ok ? pk.Zp[(rbc * DK + s) * 64 + l] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) czr[i][r] = ok ? pk.cz[16 * rbc + 4 * (l >> 4) + r] : -1e30f;   // C row = 4 g + r
        }
        if constexpr (BREG) {
            const int rbc = min(w, NBLK - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) Breg[s] = pk.Bp[(rbc * KS + s) * 64 + l];
        }
    }

    // kernel tile (gp_tf.py:33-49,134) -> LDS, then A2 = K^-1 K and the predictive products (gp_tf.py:137-159).
    // TRI: the reference's own order (gp_tf.py:137-145) as two triangular products that skip the zero blocks --
    // A = W K with W = L^-1 (row block rb needs the k-blocks 0..rb), fvar_0 = sigma^2 - colsum(A o A), A2 = W^T A (k-blocks
    // rb..NBLK-1; the rows of A travel through a second LDS tile, one more workgroup barrier).  In float32 this is the form
    // that keeps fvar_0 meaningful on an ill-conditioned K_mm: a sum of squares instead of sigma^2 - k.(K^-1 k), whose two
    // terms agree to cond eps_32.
    template <bool TRI>
    __device__ __forceinline__ void gp(const float* xq, float* Kt, float* At, float* part, int w, int l, float* rec = nullptr)
    {
        float bx[DK], xx = 0.0f;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            bx[s] = xq[64 * s + l];
            xx = fmaf(bx[s], bx[s], xx);
        }
        xx += __shfl_xor(xx, 16);
        xx += __shfl_xor(xx, 32);
        float kreg[RB][4];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                f4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = czr[i][r] - 0.5f * xx;
#pragma unroll
                for (int s = 0; s < DK; ++s) e = CBF_MFMA32(Zreg[i][s], bx[s], e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kreg[i][r] = expf(e[r]);
                    if (!TRI && bf16) kreg[i][r] = round_bf16(kreg[i][r]);  // (the kernel tile is the contraction's other operand)
                    Kt[(4 * rb + r) * 64 + l] = kreg[i][r];         // B operand of the (permuted) k-step 4 rb + r
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) kreg[i][r] = 0.0f;
            }
        }
        __syncthreads();
        f4 acc[RB][2];
#pragma unroll
        for (int i = 0; i < RB; ++i) { acc[i][0] = f4{0, 0, 0, 0}; acc[i][1] = f4{0, 0, 0, 0}; }
        float q = 0.0f;
        if constexpr (!TRI && BREG) {
            if (w < NBLK) {
#pragma unroll
                for (int s = 0; s < KS; ++s) acc[0][s & 1] = CBF_MFMA32(Breg[s], Kt[64 * s + l], acc[0][s & 1]);
            }
        } else if constexpr (!TRI) {
#pragma unroll 1
            for (int s0 = 0; s0 < KS; s0 += 4) {
                float b[4], aop[RB][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = Kt[64 * (s0 + j) + l];
#pragma unroll
                    for (int i = 0; i < RB; ++i) aop[i][j] = Bp[(min(w * RB + i, NBLK - 1) * KS + s0 + j) * 64 + l];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (w * RB + i < NBLK) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
            }
        } else {
            // A = W K: k-block kb contributes to row block rb when kb <= rb (W is lower triangular)
#pragma unroll 1
            for (int kb = 0; kb < NBLK; ++kb) {
                if (kb > min(w * RB + RB - 1, NBLK - 1)) break;
                float b[4], aop[RB][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = Kt[64 * (4 * kb + j) + l];
#pragma unroll
                    for (int i = 0; i < RB; ++i) aop[i][j] = Wp[(min(w * RB + i, NBLK - 1) * KS + 4 * kb + j) * 64 + l];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (w * RB + i < NBLK && kb <= w * RB + i) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int rb = w * RB + i;
                if (rb < NBLK) {
                    const f4 av = acc[i][0] + acc[i][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        q = fmaf(av[r], av[r], q);                  // colsum(A o A), this wave's rows   (gp_tf.py:140)
                        At[(4 * rb + r) * 64 + l] = av[r];
                    }
                }
                acc[i][0] = f4{0, 0, 0, 0}; acc[i][1] = f4{0, 0, 0, 0};
            }
            __syncthreads();
            // A2 = W^T A: k-blocks kb >= rb
#pragma unroll 1
            for (int kb = w * RB; kb < NBLK; ++kb) {
                float b[4], aop[RB][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = At[64 * (4 * kb + j) + l];
#pragma unroll
                    for (int i = 0; i < RB; ++i) aop[i][j] = WTp[(min(w * RB + i, NBLK - 1) * KS + 4 * kb + j) * 64 + l];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (w * RB + i < NBLK && kb >= w * RB + i) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
            }
        }
        f4 P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                const f4 a2 = acc[i][0] + acc[i][1];
                if (rec) {                                  // kept for the adjoint: the registers as they are (row 4 g + r)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rec[(rb * 4 + r) * 64 + l] = a2[r];
                        rec[NBLK * 256 + (rb * 4 + r) * 64 + l] = kreg[i][r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    P1 = CBF_MFMA32(mu[(rb * 4 + r) * 64 + l], a2[r], P1);
                    P2 = CBF_MFMA32(s2[(rb * 4 + r) * 64 + l], a2[r] * a2[r], P2);
                    if constexpr (!TRI) q = fmaf(kreg[i][r], a2[r], q);
                }
            }
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            part[((w * 2 + 0) * 4 + r) * 64 + l] = P1[r];
            part[((w * 2 + 1) * 4 + r) * 64 + l] = P2[r] - q;
        }
    }

    // GP output of register q at this lane: row d = 4 (l >> 4) + q, chain l & 15
    __device__ __forceinline__ void gather(const float* part, int q, int l, float& fm, float& fv) const
    {
        float s1 = 0.0f, sv = 0.0f;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
            s1 += part[((ww * 2 + 0) * 4 + q) * 64 + l];
            sv += part[((ww * 2 + 1) * 4 + q) * 64 + l];
        }
        fm = s1;
        fv = sigma2 + sv;
    }
};

template <int NBLK, int RB, int DK, bool TRI>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void predict32_kernel(Args32 a)
{
    typedef Tile32<NBLK, RB, DK> TT;
    extern __shared__ float lds32[];
    float* xq = lds32;
    float* Kt = xq + DK * 64;
    float* At = Kt + TT::MP * 16;
    float* part = At + TT::MP * 16;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    TT tile;
    tile.load(a.pk, w, l);
    const int64_t p0 = int64_t(blockIdx.x) * 16;
    for (int i = tid; i < DK * 64; i += TT::NT) {
        const int j = i >> 4, n = i & 15;
        const int64_t p = p0 + n;
        float v = 0.0f;
        if (j < a.D && p < a.npts) v = float(a.X[p * a.D + j]) * a.pk.invl[j];
        xq[i] = v;
    }
    __syncthreads();
    tile.template gp<TRI>(xq, Kt, At, part, w, l);
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < TT::QPW; ++qi) {
        const int q = w + qi * TT::W;
        if (q < 4) {
            float fm, fv;
            tile.gather(part, q, l, fm, fv);
            const int d = 4 * (l >> 4) + q;
            const int64_t p = p0 + (l & 15);
            if (d < a.Do && p < a.npts) {
                a.fmean[p * a.Do + d] = double(fm);
                a.fvar[p * a.Do + d] = double(fv);
            }
        }
    }
}

// Persistent pass kernel in float32 arithmetic: MODE_FWD = CBFSSM._forward_body loop (cbfssm.py:176-237), MODE_BWD = one
// resample-to-resample segment of one _backward_body run (cbfssm.py:107-158).  Same structure as pass_kernel
// (cbfssm_kernels.hpp): one workgroup = 16 chains, T looped inside, three workgroup barriers per step.
template <int NBLK, int RB, int DK, int MODE, bool TRI>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void pass32_kernel(Args32 a)
{
    typedef Tile32<NBLK, RB, DK> TT;
    constexpr int W = TT::W, NT = TT::NT, QPW = TT::QPW;
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    extern __shared__ float lds32[];
    __shared__ double red[16];
    float* xq = lds32;
    float* Kt = xq + DK * 64;
    float* At = Kt + TT::MP * 16;
    float* part = At + TT::MP * 16;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, nl = l & 15;
    const int N = a.N, S = a.S, T = a.T, Do = a.Do;
    const int naux = a.D - Do;
    const int gx = blockIdx.x + a.group0;
    const int c0 = gx * 16;
    const int G16 = (N + 15) >> 4;

    int t_first, nsteps, dir, run = 0;
    const int R = a.recog_len, P = 2 * R;
    if (MODE == MODE_FWD) {
        t_first = 0; nsteps = T - 1; dir = 1;
    } else {
        int k;
        if (int(blockIdx.y) < a.nseg0) { run = 0; k = blockIdx.y + 1; }
        else { run = 1; k = blockIdx.y - a.nseg0 + 1; }
        const int o = run * R;
        const int hi = min(P * k - 1 - o, T - 1);
        const int lo = (k > 1) ? (P * (k - 1) - o) : 0;
        t_first = hi; nsteps = hi - lo + 1; dir = -1;
        if (nsteps <= 0) {
            if (tid == 0) a.part_out[blockIdx.y * G16 + gx] = 0.0;
            return;
        }
    }

    TT tile;
    tile.load(a.pk, w, l);

    // epilogue lanes: task q = w + qi W < 4 handles state row d = 4 g + q of chain nl
    float vx[QPW], vy[QPW], il[QPW], hcur[QPW];
    double lin[QPW];
    LogProd lp[QPW];
    bool act[QPW];
    const int cg = c0 + nl;
    const int c = min(cg, N - 1);
    const bool cval = cg < N;
    const int bq = c / S;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * g + q;
        act[qi] = (q < 4) && (d < Do);
        const int dc = act[qi] ? d : 0;
        vx[qi] = float(a.var_x[dc]);
        vy[qi] = (MODE == MODE_FWD) ? float(a.var_y[(a.half && dc >= a.dim_y) ? 0 : dc]) : 0.0f;   // half: var_y has dim_y entries
        il[qi] = a.pk.invl[dc];
        lin[qi] = 0.0;
        lp[qi].init();
        hcur[qi] = 0.0f;
    }

    const double* auxp[AUXR];
    int auxs[AUXR];
    float auxl[AUXR];
#pragma unroll
    for (int k2 = 0; k2 < AUXR; ++k2) {
        const int i = tid + k2 * NT, ja = i >> 4, n = i & 15;
        auxp[k2] = a.eps; auxs[k2] = 0; auxl[k2] = 0.0f;       // (no row: a valid dummy address -- loads only happen when a next step exists --, factor 0)
        if (i < 16 * naux) {
            const int b = min(c0 + n, N - 1) / S;
            if (ja < a.dim_u) { auxp[k2] = a.u + int64_t(b) * T * a.dim_u + ja; auxs[k2] = a.dim_u; }
            else { auxp[k2] = a.y + int64_t(b) * T * a.dim_y + (ja - a.dim_u); auxs[k2] = a.dim_y; }
            auxl[k2] = a.pk.invl[Do + ja];
        }
    }
    // raw float64 value: converted and scaled at the LDS write (see pass_kernel in cbfssm_kernels.hpp)
    auto aux_load = [&](int k2, int t) -> double { return auxp[k2][int64_t(t) * auxs[k2]]; };

    for (int i = tid; i < DK * 64; i += NT) xq[i] = 0.0f;
    __syncthreads();
    const bool resample0 = (MODE == MODE_BWD) && (((t_first + 1 + run * R) % P) == 0);
    const int tm0 = (MODE == MODE_BWD) ? (t_first % P) : 0;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * g + q;
        if (act[qi]) {
            float v;
            if (MODE == MODE_FWD) {
                // x_0 = y_tilde[:, 0] = [y_0, y2_0]  (cbfssm.py:97,168); half: the recognition model's output (cbfssmhalf.py:106)
                const double v64 = a.half ? a.x0[int64_t(bq) * a.dim_x + d]
                                          : ((d < a.dim_y) ? a.y[(int64_t(bq) * T) * a.dim_y + d]
                                                           : a.y2_in[int64_t(c) * (a.dim_x - a.dim_y) + (d - a.dim_y)]);
                if (cval) a.x_out[int64_t(c) * a.dim_x + d] = v64;                      // x_0 = y_tilde[:, 0]  (cbfssm.py:168)
                v = float(v64);
            } else {
                v = resample0 ? float(a.hid[(int64_t(run) * T + t_first) * N + c]) : 0.0f;   // cbfssm.py:106,133-136
            }
            hcur[qi] = v;
            xq[16 * d + nl] = v * il[qi];
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < AUXR; ++k2) {
        const int i = tid + k2 * NT;
        if (i < 16 * naux) xq[16 * Do + i] = float(aux_load(k2, t_first)) * auxl[k2];
    }

    for (int step = 0; step < nsteps; ++step) {
        const int t = t_first + dir * step;
        const int tn = t + dir;
        int tmod = tm0 - step; if (tmod < 0) tmod += P;
        const int tmn = (tmod == 0) ? P - 1 : tmod - 1;
        const bool has_next = (step + 1 < nsteps);
        __syncthreads();                                  // xq complete

        float eps_t, ytil[QPW], hidn = 0.0f;
        double auxr[AUXR];
        bool resample_n = false;
        if (MODE == MODE_BWD) resample_n = has_next && (tmn + 1 + run * R == P);            // cbfssm.py:124,127
        if (MODE == MODE_FWD) {
            eps_t = float(a.eps[int64_t(t) * N + c]);                                       // cbfssm.py:209
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * g + (w + qi * W);
                ytil[qi] = 0.0f;
                if (act[qi]) {
                    if (d < a.dim_y) ytil[qi] = float(a.y[(int64_t(bq) * T + (t + 1)) * a.dim_y + d]);
                    else if (!a.half) ytil[qi] = float(a.y2_in[(int64_t(t + 1) * N + c) * (a.dim_x - a.dim_y) + (d - a.dim_y)]);
                }
            }
        } else {
            eps_t = float(a.eps[(int64_t(run) * T + t) * N + c]);                           // cbfssm.py:149
            if (resample_n) hidn = float(a.hid[(int64_t(run) * T + tn) * N + c]);
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) ytil[qi] = 0.0f;
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) auxr[k2] = has_next ? aux_load(k2, tn) : 0.0;

        float* rec = nullptr;
        if (a.a2s) {
            const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
            rec = a.a2s + (slot * G16 + gx) * (2 * NBLK * 256);
        }
        tile.template gp<TRI>(xq, Kt, At, part, w, l, rec);    // (one workgroup barrier inside, two in the two-triangular form)
        __syncthreads();                                  // part complete; xq and Kt free

#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = w + qi * W;
            if (q < 4) {
                float fm, fv;
                tile.gather(part, q, l, fm, fv);
                const int d = 4 * g + q;
                float outv = 0.0f;
                if (act[qi]) {
                    const float fmean = fm + hcur[qi];                                     // cbfssm.py:145,205
                    const float fvar = fv + vx[qi];                                        // cbfssm.py:146,206
                    if (a.fmv && cval) {
                        const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                        double* o = a.fmv + ((slot * N + c) * Do + d) * 2;
                        o[0] = double(fmean); o[1] = double(fvar);
                    }
                    if (MODE == MODE_FWD) {
                        const float vyt = vy[qi] + (a.k_factor - 1.0f) * fvar;             // cbfssm.py:212-214
                        const float sm = vyt + fvar;
                        const float rs = rcp32(sm);
                        const float kk = fvar * rs;
                        const float ydiff = ytil[qi] - fmean;
                        const float mu = fmean + kk * ydiff;
                        const float sig = kk * vyt;                 // = (1-k)^2 fvar + k^2 v with 1 - k = v / s   (:219-220)
                        // (half: the hidden dims get no Kalman update -- mu = fmean, sig = fvar, no KL term)
                        const bool do_cond = (a.condition || (t < R - 1)) && !(a.half && d >= a.dim_y);   // :227
                        outv = do_cond ? (mu + eps_t * (sig * rsqrt32(sig))) : (fmean + eps_t * (fvar * rsqrt32(fvar)));
                        if (do_cond && cval) {
                            lin[qi] += double(kk * (ydiff * ydiff * rs - 1.0f));           // (sig + (mu-fmean)^2)/fvar - 1  (:232)
                            lp[qi].mul(double(vyt * rs));                                  // sig / fvar
                        }
                        if (cval) a.x_out[(int64_t(t + 1) * N + c) * a.dim_x + d] = double(outv);
                    } else {
                        outv = fmean + eps_t * (fvar * rsqrt32(fvar));                     // cbfssm.py:150
                        const bool write = (run == 0) ? (tmod < R) : (tmod >= R);          // :125,128
                        if (cval && write) {
                            a.y2_out[(int64_t(t) * N + c) * Do + d] = double(outv);        // :151
                            lp[qi].mul(double(fvar));                                      // :154-156
                            lin[qi] += 1.0;
                        }
                        if (cval && a.h_all) a.h_all[((int64_t(run) * T + t) * N + c) * Do + d] = double(outv);
                    }
                }
                const float hn = (MODE == MODE_BWD && resample_n) ? hidn : outv;           // cbfssm.py:133-136,158
                hcur[qi] = act[qi] ? hn : 0.0f;
                if (has_next && act[qi]) xq[16 * d + nl] = hn * il[qi];
            }
        }
        if (has_next) {
#pragma unroll
            for (int k2 = 0; k2 < AUXR; ++k2) {
                const int i = tid + k2 * NT;
                if (i < 16 * naux) xq[16 * Do + i] = float(auxr[k2]) * auxl[k2];
            }
        }
    }

    double v = 0.0;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        if (act[qi] && cval) {
            if (MODE == MODE_FWD) v += 0.5 * (lin[qi] - lp[qi].log());
            else v += 0.5 * (lin[qi] * 2.8378770664093453391 + lp[qi].log());              // log(2 pi e)
        }
    }
    const double tot = block_sum(v, red, tid, NT);
    if (tid == 0) a.part_out[blockIdx.y * G16 + gx] = tot;
}

template <int NBLK, int DK, bool TRI>
static int launch32(int mode, const Args32& a, dim3 grid, hipStream_t st)
{
    constexpr int RB = (NBLK >= 13) ? 2 : 1;
    typedef Tile32<NBLK, RB, DK> TT;
    const size_t lds = size_t(TT::LDS_FLOATS) * sizeof(float);
    hipError_t e = hipSuccess;
    auto setlds = [&](const void* k) {
        if (lds > 48 * 1024) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    };
    if (mode == 2) {
        auto k = predict32_kernel<NBLK, RB, DK, TRI>;
        setlds(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return -int(e) - 1000;
        hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
    } else if (mode == MODE_FWD) {
        auto k = pass32_kernel<NBLK, RB, DK, MODE_FWD, TRI>;
        setlds(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return -int(e) - 1000;
        hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
    } else {
        auto k = pass32_kernel<NBLK, RB, DK, MODE_BWD, TRI>;
        setlds(reinterpret_cast<const void*>(k));
        if (e != hipSuccess) return -int(e) - 1000;
        hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
    }
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK>
static int launch32_n(int DK, int mode, const Args32& a, dim3 grid, hipStream_t st)
{
    if (a.tri) {
        switch (DK) {
            case 2: return launch32<NBLK, 2, true>(mode, a, grid, st);
            case 4: return launch32<NBLK, 4, true>(mode, a, grid, st);
            case 6: return launch32<NBLK, 6, true>(mode, a, grid, st);
        }
        return -2;
    }
    switch (DK) {
        case 2: return launch32<NBLK, 2, false>(mode, a, grid, st);
        case 4: return launch32<NBLK, 4, false>(mode, a, grid, st);
        case 6: return launch32<NBLK, 6, false>(mode, a, grid, st);
    }
    return -2;
}

static int dispatch32(int NBLK, int DK, int mode, const Args32& a, dim3 grid, hipStream_t st)
{
    switch (NBLK) {
        case 1: return launch32_n<1>(DK, mode, a, grid, st);
        case 2: return launch32_n<2>(DK, mode, a, grid, st);
        case 4: return launch32_n<4>(DK, mode, a, grid, st);
        case 7: return launch32_n<7>(DK, mode, a, grid, st);
        case 10: return launch32_n<10>(DK, mode, a, grid, st);
        case 13: return launch32_n<13>(DK, mode, a, grid, st);
        case 16: return launch32_n<16>(DK, mode, a, grid, st);
        case 20: return launch32_n<20>(DK, mode, a, grid, st);
    }
    return -2;
}

static int fill32(Args32& a, const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32, int Do)
{
    if (!p || !L || !pack32) return fail(-1, "null pointer");
    if (p->B < 1 || p->S < 1 || p->T < 1 || p->recog_len < 1) return fail(-1, "B, S, T, recog_len must be >= 1");
    if (p->ngroups > 0 && (p->group0 < 0 || p->group0 + p->ngroups > (p->B * p->S + 15) / 16)) return fail(-1, "bad chain-group range");
    if (L->D != p->dim_x + p->dim_u || L->Do != Do || L->M != p->M) return fail(-1, "pack does not match the problem");
    memset(&a, 0, sizeof(a));
    a.pk = pack32_ptrs(L, pack32);
    a.N = p->B * p->S; a.S = p->S; a.T = p->T; a.B = p->B;
    a.dim_x = p->dim_x; a.dim_u = p->dim_u; a.dim_y = p->dim_y; a.Do = Do; a.D = L->D;
    a.recog_len = p->recog_len; a.condition = p->condition; a.k_factor = float(p->k_factor);
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    a.group0 = p->ngroups > 0 ? p->group0 : 0;
    return 0;
}

}  // namespace f32
}  // namespace cbfssm

using namespace cbfssm;
using namespace cbfssm::f32;

extern "C" {

int64_t cbfssm_pack_f32_elems(const cbfssm_pack_layout* L)
{
    if (!L) return -1;
    return pack32_offsets(L).total;
}

static int pack32_launch(const cbfssm_pack_layout* L, const double* pack, float* pack32, int bf16, void* stream)
{
    if (!L || !pack || !pack32) return fail(-1, "null pointer");
    const Off32 o = pack32_offsets(L);
    hipLaunchKernelGGL(pack32_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, pack, *L, pack32, o, bf16);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-int(e) - 1000, "gp_pack_f32: %s", hipGetErrorString(e));
}

int cbfssm_gp_pack_f32(const cbfssm_pack_layout* L, const double* pack, float* pack32, void* stream)
{
    return pack32_launch(L, pack, pack32, 0, stream);
}

int cbfssm_gp_pack_bf16(const cbfssm_pack_layout* L, const double* pack, float* pack32, void* stream)
{
    return pack32_launch(L, pack, pack32, 1, stream);
}

int cbfssm_gp_predict_f32(const cbfssm_pack_layout* L, const float* pack32, const double* X, int64_t npts, double* fmean,
                          double* fvar, void* stream)
{
    if (!L || !pack32 || !X || !fmean || !fvar) return fail(-1, "null pointer");
    if (npts <= 0) return npts == 0 ? 0 : fail(-1, "bad npts");
    Args32 a;
    memset(&a, 0, sizeof(a));
    a.pk = pack32_ptrs(L, pack32);
    a.X = X; a.npts = npts; a.D = L->D; a.Do = L->Do; a.fmean = fmean; a.fvar = fvar;
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    int rc = dispatch32(L->NBLK, L->DK, 2, a, dim3(unsigned((npts + 15) / 16)), (hipStream_t)stream);
    return rc ? fail(rc, "gp_predict_f32 launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc) : 0;
}

int64_t cbfssm_saved_a2_f32_elems(const cbfssm_problem* p, const cbfssm_pack_layout* L, int backward_runs)
{
    if (!p || !L) return -1;
    const int64_t groups = (int64_t(p->B) * p->S + 15) / 16;
    const int64_t slots = backward_runs ? 2 * int64_t(p->T) : (p->T > 1 ? int64_t(p->T) - 1 : 0);
    return slots * groups * 2 * L->NBLK * 256;            // [A2 | kernel tile] accumulator registers of every step and group
}

int cbfssm_backward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_b,
                             const double* var_x, const double* u, const double* y, const double* hid_b,
                             const double* eps_b, double* y2, double* h_all, double* fmv_b, float* a2s_b, double* ent_part,
                             void* stream)
{
    if (p && p->half) return fail(-1, "the forward-only variants have no backward runs");
    Args32 a;
    int rc = fill32(a, p, L, pack32_b, p ? p->dim_x - p->dim_y : 0);
    if (rc) return rc;
    if (!var_x || !u || !y || !hid_b || !eps_b || !y2 || !ent_part) return fail(-1, "null pointer");
    a.var_x = var_x; a.u = u; a.y = y; a.eps = eps_b; a.hid = hid_b; a.y2_out = y2; a.part_out = ent_part;
    a.h_all = h_all; a.fmv = fmv_b; a.a2s = a2s_b;
    const int P = 2 * p->recog_len;
    const int n0 = p->T / P + 1, n1 = (p->T + p->recog_len) / P + 1;       // as cbfssm_backward_pass_f64 counts them
    a.nseg0 = n0;
    dim3 grid(unsigned(p->ngroups > 0 ? p->ngroups : (a.N + 15) / 16), unsigned(n0 + n1));
    rc = dispatch32(L->NBLK, L->DK, MODE_BWD, a, grid, (hipStream_t)stream);
    return rc ? fail(rc, "backward_pass_f32 launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc) : 0;
}

static int forward32_impl(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f, const double* var_x,
                          const double* var_y, const double* u, const double* y, const double* y2, const double* x0,
                          const double* eps_f, double* x, double* fmv_f, float* a2s_f, double* kl_part, void* stream)
{
    Args32 a;
    int rc = fill32(a, p, L, pack32_f, p ? p->dim_x : 0);
    if (rc) return rc;
    if (!var_x || !var_y || !u || !y || !x || !kl_part || (p->T > 1 && !eps_f)) return fail(-1, "null pointer");
    if (p->half ? !x0 : (p->dim_x > p->dim_y && !y2)) return fail(-1, p->half ? "x0 is null" : "y2 is null");
    a.var_x = var_x; a.var_y = var_y; a.u = u; a.y = y; a.eps = eps_f; a.y2_in = y2; a.x_out = x; a.part_out = kl_part;
    a.fmv = fmv_f; a.a2s = a2s_f; a.half = p->half; a.x0 = x0;
    dim3 grid(unsigned(p->ngroups > 0 ? p->ngroups : (a.N + 15) / 16), 1);
    rc = dispatch32(L->NBLK, L->DK, MODE_FWD, a, grid, (hipStream_t)stream);
    return rc ? fail(rc, "forward_pass_f32 launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc) : 0;
}

int cbfssm_half_forward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f,
                                 const double* var_x, const double* var_y, const double* u, const double* y,
                                 const double* x0, const double* eps_f, double* x, double* fmv_f, float* a2s_f,
                                 double* kl_part, void* stream)
{
    if (!p || !p->half) return fail(-1, "problem->half must be 1");
    return forward32_impl(p, L, pack32_f, var_x, var_y, u, y, nullptr, x0, eps_f, x, fmv_f, a2s_f, kl_part, stream);
}

int cbfssm_forward_pass_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f,
                            const double* var_x, const double* var_y, const double* u, const double* y,
                            const double* y2, const double* eps_f, double* x, double* fmv_f, float* a2s_f, double* kl_part,
                            void* stream)
{
    if (p && p->half) return fail(-1, "problem->half is set: use cbfssm_half_forward_pass_f32");
    return forward32_impl(p, L, pack32_f, var_x, var_y, u, y, y2, nullptr, eps_f, x, fmv_f, a2s_f, kl_part, stream);
}

}  // extern "C"
