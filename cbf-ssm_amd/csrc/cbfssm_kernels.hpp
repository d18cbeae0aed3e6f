// CBF-SSM ELBO hot path for MI355X (gfx950 / CDNA4): device code of the time-loop kernels.
//
// One workgroup owns 16 particle chains (one MFMA column block) for a whole pass and loops over time inside the
// kernel; chains never interact (cbfssm/model/cbfssm.py:114-158,185-237 act row-wise on the (B*S, .) state matrix),
// so there is no grid-level synchronisation.  A GP conditional (cbfssm/model/gp_tf.py:132-161) for 16 points is
//
//   phase 1   E  = Z~ X~^T - .5|z~|^2 - .5|x~|^2 + log s2      v_mfma_f64_16x16x4, rows = inducing points
//             K  = exp(E)                                       (gp_tf.py:33-49,134) written to LDS
//   phase 2   A2 = K_mm^-1 K                                    v_mfma_f64_16x16x4, K^-1 rows live in VGPRs
//             P1 = mu_z^T A2,  P2 = s2_z^T (A2 o A2) - colsum(K o A2)      (gp_tf.py:137-159, contraction form)
//   phase 3   cross-wave sum of P1/P2, then the per-(chain, state-dim) step epilogue of the pass
//
// The f64 MFMA C/D layout (row = (lane>>4) + 4*reg, col = lane&15) equals its B-operand layout for k-step
// 4*block + reg, so the exp'd accumulator of phase 1 is phase 2's B operand and phase 2's accumulator is the B
// operand of the P1/P2 products with no lane movement (checked on hardware by csrc/probe/mfma_f64_probe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cbfssm {

typedef double d4 __attribute__((ext_vector_type(4)));

#define CBF_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

enum { MODE_FWD = 0, MODE_BWD = 1 };

struct PackPtrs {
    const double* Bp;
    const double* Zp;
    const double* cz;
    const double* muA;
    const double* s2A;
    const double* invl;
    const double* scal;
    int KSr;
};

struct PassArgs {
    PackPtrs pk;
    int N, S, T, B;
    int dim_x, dim_u, dim_y;
    int Do;        // GP output dim = number of state rows carried by the chain
    int D;         // GP input dim
    int recog_len, condition;
    double k_factor;
    const double* var_x;   // (dim_x)
    const double* var_y;   // (dim_x)
    const double* u;       // (B,T,dim_u)
    const double* y;       // (B,T,dim_y)
    const double* eps;     // fwd: (T-1,N); bwd: (2,T,N)
    const double* hid;     // bwd: (2,T,N)
    const double* y2_in;   // fwd: (T,N,dim_x-dim_y)
    double* y2_out;        // bwd
    double* h_all;         // bwd, optional (2,T,N,Do)
    double* x_out;         // fwd: (T,N,dim_x)
    double* part_out;      // one partial per workgroup
    int nseg0;             // bwd: number of segment slots of run 0 (blockIdx.y < nseg0 -> run 0)
};

struct PredictArgs {
    PackPtrs pk;
    const double* X;   // (npts, D)
    int64_t npts;
    int D, Do;
    double* fmean;     // (npts, Do)
    double* fvar;
};

template <int NBLK, int RB, int DK, bool BREG>
struct Tile {
    static constexpr int W = (NBLK + RB - 1) / RB;   // waves per workgroup
    static constexpr int NT = 64 * W;
    static constexpr int MP = 16 * NBLK;
    static constexpr int KS = MP / 4;                // k-steps of the K^-1 K product
    static constexpr int QPW = (4 + W - 1) / W;      // state-row groups (4 rows each) per wave in phase 3
    static constexpr int LDS_DOUBLES = DK * 64 + MP * 16 + W * 512 + 64;

    // loop-invariant MFMA A operands of this wave
    double Zreg[RB][DK];
    double czr[RB][4];
    double muA[RB][4];
    double s2A[RB][4];
    double Breg[BREG ? RB : 1][BREG ? KS : 1];
    const double* Bp;
    const double* muAg;
    const double* s2Ag;
    int lane_;
    double sigma2;
    int KSr;       // k-steps of K^-1 that carry data: ceil(M/4) <= KS

    template <bool WITH_EPI = true>
    __device__ __forceinline__ void load_operands(const PackPtrs& pk, int w, int l)
    {
        Bp = pk.Bp;
        muAg = pk.muA;
        s2Ag = pk.s2A;
        lane_ = l;
        sigma2 = pk.scal[0];
        KSr = pk.KSr;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            const bool ok = rb < NBLK;
            const int rbc = ok ? rb : 0;
#pragma unroll
            for (int s = 0; s < DK; ++s) Zreg[i][s] = ok ? pk.Zp[(rbc * DK + s) * 64 + l] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                czr[i][r] = ok ? pk.cz[16 * rbc + 4 * r + (l >> 4)] : -1e30;
                if (WITH_EPI && BREG) {
                    muA[i][r] = ok ? pk.muA[(rbc * 4 + r) * 64 + l] : 0.0;
                    s2A[i][r] = ok ? pk.s2A[(rbc * 4 + r) * 64 + l] : 0.0;
                }
            }
            if (BREG) {
#pragma unroll
                for (int s = 0; s < KS; ++s) Breg[i][s] = ok ? pk.Bp[(rbc * KS + s) * 64 + l] : 0.0;
            }
        }
    }

    // phases 1 and 2 for the 16 points whose scaled inputs sit in xq; leaves P1/P2 partials of this wave in `part`.
    // Contains two workgroup barriers; the caller must barrier before reading `part` and before rewriting xq.
    __device__ __forceinline__ void gp_phases(const double* xq, double* Kt, double* part, int w, int l)
    {
        // ---- phase 1: kernel tile rows of this wave
        double bx[DK];
        double xx = 0.0;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            bx[s] = xq[64 * s + l];
            xx = fma(bx[s], bx[s], xx);
        }
        xx += __shfl_xor(xx, 16);
        xx += __shfl_xor(xx, 32);
        double kreg[RB][4];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                d4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = czr[i][r] - 0.5 * xx;
#pragma unroll
                for (int s = 0; s < DK; ++s) e = CBF_MFMA(Zreg[i][s], bx[s], e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kreg[i][r] = exp(e[r]);
                    Kt[256 * rb + 64 * r + l] = kreg[i][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) kreg[i][r] = 0.0;
            }
        }
        __syncthreads();

        // ---- phase 2: A2 rows of this wave, then the P1/P2 products
        d4 acc[RB][2];
#pragma unroll
        for (int i = 0; i < RB; ++i) { acc[i][0] = d4{0, 0, 0, 0}; acc[i][1] = d4{0, 0, 0, 0}; }
        if constexpr (BREG) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double b = Kt[64 * s + l];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int rb = w * RB + i;
                    if (rb < NBLK) {
                        constexpr int two = (RB == 1) ? 1 : 0;
                        acc[i][(s & 1) * two] = CBF_MFMA(Breg[i][s], b, acc[i][(s & 1) * two]);
                    }
                }
            }
        } else {
            // K^-1 streamed from L2 as a lane-linear A-operand image: 512 contiguous bytes per (row block, k-step)
            static_assert(KS % 4 == 0, "KS must be a multiple of 4");
#pragma unroll 1
            for (int s0 = 0; s0 < KSr; s0 += 4) {
                double b[4], aop[RB][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = Kt[64 * (s0 + j) + l];
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        const int rb = min(w * RB + i, NBLK - 1);
                        aop[i][j] = Bp[(rb * KS + s0 + j) * 64 + l];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        if (w * RB + i < NBLK) {
                            constexpr int two = (RB == 1) ? 1 : 0;
                            acc[i][(j & 1) * two] = CBF_MFMA(aop[i][j], b[j], acc[i][(j & 1) * two]);
                        }
                    }
                }
            }
        }
        d4 P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0};
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                const d4 a2 = acc[i][0] + acc[i][1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double mu_op = BREG ? muA[i][r] : muAg[(rb * 4 + r) * 64 + lane_];
                    const double s2_op = BREG ? s2A[i][r] : s2Ag[(rb * 4 + r) * 64 + lane_];
                    P1 = CBF_MFMA(mu_op, a2[r], P1);
                    P2 = CBF_MFMA(s2_op, a2[r] * a2[r], P2);
                    q = fma(kreg[i][r], a2[r], q);
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

    // phase 3 helper: GP output for state-row group q at this lane's (row = 4q + (l>>4), chain = l&15)
    __device__ __forceinline__ void gather(const double* part, int q, int l, double& fm, double& fv) const
    {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
            s1 += part[((ww * 2 + 0) * 4 + q) * 64 + l];
            s2 += part[((ww * 2 + 1) * 4 + q) * 64 + l];
        }
        fm = s1;
        fv = sigma2 + s2;
    }
};

__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nthreads)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0) {
        for (int i = 0; i < nthreads / 64; ++i) s += red[i];
    }
    return s;   // valid in thread 0
}

// log-sum accumulated as a product with the exponent split off every step (one f64 log at the end of the pass
// instead of one per step: the software f64 log costs ~380 cycles per wave on gfx950, frexp is two instructions)
struct LogProd {
    double mant;
    int ex;
    __device__ __forceinline__ void init() { mant = 1.0; ex = 0; }
    __device__ __forceinline__ void mul(double v)
    {
        mant *= v;
        int e;
        mant = frexp(mant, &e);
        ex += e;
    }
    __device__ __forceinline__ double log() const { return ::log(mant) + double(ex) * 0.6931471805599453094; }
};

// ---------------------------------------------------------------------------------------------------------------------
// GPModel.predict for arbitrary points (gp_tf.py:132-161)
// ---------------------------------------------------------------------------------------------------------------------
template <int NBLK, int RB, int DK, bool BREG>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void predict_kernel(PredictArgs a)
{
    typedef Tile<NBLK, RB, DK, BREG> TT;
    extern __shared__ double lds[];
    double* xq = lds;
    double* Kt = xq + DK * 64;
    double* part = Kt + TT::MP * 16;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    TT tile;
    tile.load_operands(a.pk, w, l);
    const int64_t p0 = int64_t(blockIdx.x) * 16;
    for (int i = tid; i < DK * 64; i += TT::NT) {
        const int j = i >> 4, n = i & 15;
        const int64_t p = p0 + n;
        double v = 0.0;
        if (j < a.D && p < a.npts) v = a.X[p * a.D + j] * a.pk.invl[j];
        xq[i] = v;
    }
    __syncthreads();
    tile.gp_phases(xq, Kt, part, w, l);
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < TT::QPW; ++qi) {
        const int q = w + qi * TT::W;
        if (q < 4) {
            double fm, fv;
            tile.gather(part, q, l, fm, fv);
            const int d = 4 * q + (l >> 4);
            const int64_t p = p0 + (l & 15);
            if (d < a.Do && p < a.npts) {
                a.fmean[p * a.Do + d] = fm;
                a.fvar[p * a.Do + d] = fv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent pass kernel.  MODE_FWD: CBFSSM._forward_body loop (cbfssm.py:176-237);
// MODE_BWD: one resample-to-resample segment of one CBFSSM._backward_body run (cbfssm.py:107-158).
// ---------------------------------------------------------------------------------------------------------------------
template <int NBLK, int RB, int DK, bool BREG, int MODE>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void pass_kernel(PassArgs a)
{
    typedef Tile<NBLK, RB, DK, BREG> TT;
    constexpr int W = TT::W, NT = TT::NT, QPW = TT::QPW;
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    extern __shared__ double lds[];
    double* xq = lds;
    double* Kt = xq + DK * 64;
    double* part = Kt + TT::MP * 16;
    double* red = part + W * 512;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, nl = l & 15;
    const int N = a.N, S = a.S, T = a.T, Do = a.Do;
    const int naux = a.D - Do;                       // rows of the GP input that are not chain state
    const int c0 = blockIdx.x * 16;
    const int c = min(c0 + nl, N - 1);               // clamped chain of this lane (phase 3)
    const bool cvalid = (c0 + nl) < N;
    const int bq = c / S;                            // its sequence

    // ---- time range of this workgroup
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
            if (tid == 0) a.part_out[blockIdx.y * gridDim.x + blockIdx.x] = 0.0;
            return;
        }
    }

    TT tile;
    tile.load_operands(a.pk, w, l);

    // per-lane constants of the state rows this lane finishes in phase 3
    double vx[QPW], vy[QPW], il[QPW];
    double hcur[QPW];
    double lin[QPW];
    LogProd lp[QPW];
    bool act[QPW];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * q + g;
        act[qi] = (q < 4) && (d < Do);
        const int dc = act[qi] ? d : 0;
        vx[qi] = a.var_x[dc];
        vy[qi] = (MODE == MODE_FWD) ? a.var_y[dc] : 0.0;
        il[qi] = a.pk.invl[dc];
        lin[qi] = 0.0;
        lp[qi].init();
        hcur[qi] = 0.0;
    }

    // auxiliary (non-state) input rows: fwd u_t; bwd [u_t, y_t]           (cbfssm.py:137,197)
    auto aux_load = [&](int i, int t) -> double {
        const int ja = i >> 4, n = i & 15;
        if (ja >= naux) return 0.0;
        const int cc = min(c0 + n, N - 1);
        const int b = cc / S;
        double v;
        if (ja < a.dim_u) v = a.u[(int64_t(b) * T + t) * a.dim_u + ja];
        else v = a.y[(int64_t(b) * T + t) * a.dim_y + (ja - a.dim_u)];
        return v * a.pk.invl[Do + ja];
    };

    // ---- initial state and first input
    // xq rows [0,Do) carry the chain state, rows [Do,D) the auxiliary inputs, rows [D,4*DK) stay zero
    for (int i = tid; i < DK * 64; i += NT) xq[i] = 0.0;
    __syncthreads();
    const bool resample0 = (MODE == MODE_BWD) && (((t_first + 1 + run * R) % P) == 0);
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * q + g;
        if (act[qi]) {
            double v;
            if (MODE == MODE_FWD) {
                // x_0 = y_tilde[:, 0] = [y_0, y2_0]                       (cbfssm.py:97,168)
                v = (d < a.dim_y) ? a.y[(int64_t(bq) * T) * a.dim_y + d]
                                  : a.y2_in[int64_t(c) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                if (cvalid) a.x_out[int64_t(c) * a.dim_x + d] = v;
            } else {
                v = resample0 ? a.hid[(int64_t(run) * T + t_first) * N + c] : 0.0;   // cbfssm.py:106,133-136
            }
            hcur[qi] = v;
            xq[64 * q + l] = v * il[qi];
        }
    }
    double auxr[AUXR];
#pragma unroll
    for (int k2 = 0; k2 < AUXR; ++k2) {
        const int i = tid + k2 * NT;
        if (i < 16 * naux) xq[16 * Do + i] = aux_load(i, t_first);
    }

    for (int step = 0; step < nsteps; ++step) {
        const int t = t_first + dir * step;
        const int tn = t + dir;                       // time index of the next GP input
        const bool has_next = (step + 1 < nsteps);
        __syncthreads();                              // xq complete

        // ---- prefetch this step's epilogue inputs and the next step's auxiliary rows
        double eps_t, ytil[QPW], hidn = 0.0;
        bool resample_n = false;
        if (MODE == MODE_FWD) {
            eps_t = a.eps[int64_t(t) * N + c];                                         // cbfssm.py:209
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                ytil[qi] = 0.0;
                if (act[qi]) {
                    ytil[qi] = (d < a.dim_y) ? a.y[(int64_t(bq) * T + (t + 1)) * a.dim_y + d]      // cbfssm.py:196
                                             : a.y2_in[(int64_t(t + 1) * N + c) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                }
            }
        } else {
            eps_t = a.eps[(int64_t(run) * T + t) * N + c];                             // cbfssm.py:149
            resample_n = has_next && (((tn + 1 + run * R) % P) == 0);                  // cbfssm.py:124,127
            if (resample_n) hidn = a.hid[(int64_t(run) * T + tn) * N + c];
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT;
            auxr[k2] = (has_next && i < 16 * naux) ? aux_load(i, tn) : 0.0;
        }

        tile.gp_phases(xq, Kt, part, w, l);
        __syncthreads();                              // part complete; xq and Kt free

        // ---- phase 3
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = w + qi * W;
            if (q < 4) {
                double fm, fv;
                tile.gather(part, q, l, fm, fv);
                const int d = 4 * q + g;
                double outv = 0.0;
                if (act[qi]) {
                    const double fmean = fm + hcur[qi];                                // cbfssm.py:145,205
                    const double fvar = fv + vx[qi];                                   // cbfssm.py:146,206
                    if (MODE == MODE_FWD) {
                        const double vyt = vy[qi] + (a.k_factor - 1.0) * fvar;         // cbfssm.py:212-214
                        const double s = vyt + fvar;                                   // :216
                        const double k = fvar * (1.0 / s);                             // :217
                        const double ydiff = ytil[qi] - fmean;                         // :215
                        const double mu = fmean + k * ydiff;                           // :218
                        const double omk = 1.0 - k;
                        const double sig = omk * omk * fvar + k * k * vyt;             // :219-220
                        const bool do_cond = a.condition || (t < R - 1);               // :227
                        outv = do_cond ? (mu + eps_t * sqrt(sig)) : (fmean + eps_t * sqrt(fvar));   // :221-229
                        if (do_cond && cvalid) {
                            // kl_reg = log fvar - log sig + (sig + (mu - fmean)^2)/fvar - 1        (:232)
                            const double rf = 1.0 / fvar;
                            const double dm = mu - fmean;
                            lin[qi] += (sig + dm * dm) * rf - 1.0;
                            lp[qi].mul(sig * rf);
                        }
                        if (cvalid) a.x_out[(int64_t(t + 1) * N + c) * a.dim_x + d] = outv;         // :229
                    } else {
                        outv = fmean + eps_t * sqrt(fvar);                             // cbfssm.py:150
                        const bool write = (run == 0) ? ((t % P) < R) : ((t % P) >= R);             // :125,128
                        if (cvalid) {
                            if (write) {
                                a.y2_out[(int64_t(t) * N + c) * Do + d] = outv;        // :151
                                lp[qi].mul(fvar);                                      // :154-156
                                lin[qi] += 1.0;
                            }
                            if (a.h_all) a.h_all[((int64_t(run) * T + t) * N + c) * Do + d] = outv;
                        }
                    }
                }
                const double hn = (MODE == MODE_BWD && resample_n) ? hidn : outv;      // cbfssm.py:133-136,158
                hcur[qi] = act[qi] ? hn : 0.0;
                if (has_next && act[qi]) xq[64 * q + l] = hn * il[qi];
            }
        }
        if (has_next) {
#pragma unroll
            for (int k2 = 0; k2 < AUXR; ++k2) {
                const int i = tid + k2 * NT;
                if (i < 16 * naux) xq[16 * Do + i] = auxr[k2];
            }
        }
    }

    // ---- per-workgroup partial of the regulariser
    double v = 0.0;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        if (act[qi] && cvalid) {
            if (MODE == MODE_FWD) v += 0.5 * (lin[qi] - lp[qi].log());
            else v += 0.5 * (lin[qi] * 2.8378770664093453391 + lp[qi].log());          // log(2 pi e)
        }
    }
    const double tot = block_sum(v, red, tid, NT);
    if (tid == 0) a.part_out[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

}  // namespace cbfssm
