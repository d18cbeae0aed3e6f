// float32-arithmetic adjoint of the CBF-SSM time loops: what `minimize` differentiates (cbfssm/model/cbfssm.py:273-275)
// when the model was built with dtype = float32 (cbfssm.py:12, base_model.py:8-10).  As in the reference's float32 mode
// only the Cholesky is float64 (gp_tf.py:57-65): K^-1 and the other operand images come from the float64 pack and are
// cast (cbfssm_gp_pack_f32); the kernel tile, A2 = K^-1 k, the reverse sweep and every accumulation over the time steps
// run on v_mfma_f32_16x16x4_f32 / float32 VALU.  Storage in HBM (trajectories, noise, the y2 adjoint) stays float64 and
// the partial slabs leave the kernel as float64 in the layout of the float64 adjoint (cbfssm_adjoint.hpp: Slab), so the
// fixed-order reduction, the K_mm / Cholesky adjoint and the optimizer step (cbfssm_tail.hip) are the float64 ones --
// again as the reference does, whose float32 graph differentiates through a float64 Cholesky.
//
// Same step structure as rev_kernel (phases B, C, E, F, G, D between four workgroup barriers), written once for every
// tile height, nothing kept by the forward evaluation except (fmean, fvar) and the trajectories: the kernel tile and A2
// are recomputed.  The accumulator of the K_mm adjoint's data part (tiles of 4 VGPRs; see phase F: it holds
// (K^-1 A2bar) A2^T, not A2bar K^T) stays in registers for the whole pass.  One row block per wave (up to 10 row blocks): the
// wave's NBLK tiles of the full matrix.  Two row blocks per wave (13, 16, 20 row blocks; SYMG): only the SYMMETRIC part of
// that matrix is ever used (it is d loss / d K_mm, contracted with symmetric dK_mm/dtheta), so the waves accumulate the
// lower-triangular blocks of  S = C A2^T + A2 C^T,  C = K^-1 A2bar  -- NBLK (NBLK + 1) / 2 tiles, dealt so that wave w owns
// the block rows w and NBLK - 1 - w (NBLK + 1 tiles each: 84 VGPRs at 20 row blocks) -- in ONE pass over the time loop.
// (Round 3 held NBLK / 2 column blocks of the full matrix per pass and ran the time loop twice above 13 row blocks.)
// This is the path of a reduced-precision model, not the headline.
//
// v_mfma_f32_16x16x4_f32: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// C[row = 4 (lane >> 4) + reg][col = lane & 15] -- register r of lane group g is row 4 g + r (the f64 instruction: 4 r + g).
#include "cbfssm_f32.hpp"
#include "cbfssm_adjoint.hpp"

namespace cbfssm {
namespace f32 {

struct Rev32Args {
    Pack32 pk;
    int N, S, T, B;
    int dim_x, dim_u, dim_y;
    int Do, D;
    int recog_len, condition;
    float k_factor;
    float cL, cE;
    const double* var_x;
    const double* var_y;
    const double* u;
    const double* y;
    const double* eps;
    const double* hid;
    const double* x;
    const double* y2;
    const double* h_all;
    const double* fmv;
    const float* a2s;      // optional: every step's [A2 | kernel tile] registers as kept by the float32 passes (Args32::a2s)
    double* gy2;
    double* gpart;
    int64_t slab;
    int KSr;
    int seg0, seg1, nchunk;
    int cb0;               // first Kinvbar column block this launch accumulates
    int first;             // 1: this launch also produces every other adjoint (and the y2 adjoint); 0: Kinvbar columns only
    int tri;               // 1: the products with K^-1 run as two triangular products (layout->gp_form == CBFSSM_GP_FORM_TRI)
    int half;              // forward-only variants (CBFSSMHALF / PRSSM): the forward pass of problem->half = 1
    double* gx0;           // half: (N, dim_x) d loss / d x_0 per chain (summed over the particles by the caller)
    int group0, gtotal;    // chain-group split: this launch covers groups [group0, group0 + gridDim.x) of gtotal
};

// SYMG accumulation.  Wave w of a two-row-blocks-per-wave tile owns the block rows R1 = NBLK - 1 - w and R0 = w < R1 of the
// lower triangle: NBLK + 1 accumulator tiles, tile K = block (R1, K) for K <= R1 and block (R0, K - R1 - 1) above.  The tile
// INDEX is a constant in the source (the accumulators stay in registers); which block a tile is depends on the wave, so the
// LDS addresses of its column operands and the "diagonal block" test are wave-uniform run-time values -- one code path for
// every wave.  (First version: one statically indexed instantiation per wave behind an if-chain on w; the merged register
// webs of ten instantiations cost 300 more spilled registers at 20 row blocks and the pass was slower than the two it
// replaced.)   S[row][col] += C_row A2_col^T + A2_row C_col^T; the diagonal block takes C A2^T only, the tail symmetrises.
// cT / aT: the row's C and A2 rows as A operands [m][k = chain]; the column blocks' rows as B operands [k = chain][m] come
// from the same two LDS tiles (an A-operand image of X is the B-operand image of X^T).
template <int NBLK, int K>
__device__ __forceinline__ void symg_tiles(f4 (&gS)[NBLK + 1], int R0, int R1, const float (&c1)[4], const float (&a1)[4],
                                           const float (&c0)[4], const float (&a0)[4], const float* Ct, const float* A2k, int g, int nl)
{
    constexpr int PD = 17;
    if constexpr (K <= NBLK) {
        if (K <= R1) {
            const float* pa = A2k + (16 * K + nl) * PD + g;
            const float* pc = Ct + (16 * K + nl) * PD + g;
#pragma unroll
            for (int s = 0; s < 4; ++s) gS[K] = CBF_MFMA32(c1[s], pa[4 * s], gS[K]);
            if (K != R1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) gS[K] = CBF_MFMA32(a1[s], pc[4 * s], gS[K]);
            }
        } else if (R1 > R0) {
            const int col = K - R1 - 1;                                   // (<= R0 for every K <= NBLK)
            const float* pa = A2k + (16 * col + nl) * PD + g;
            const float* pc = Ct + (16 * col + nl) * PD + g;
#pragma unroll
            for (int s = 0; s < 4; ++s) gS[K] = CBF_MFMA32(c0[s], pa[4 * s], gS[K]);
            if (col != R0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) gS[K] = CBF_MFMA32(a0[s], pc[4 * s], gS[K]);
            }
        }
        symg_tiles<NBLK, K + 1>(gS, R0, R1, c1, a1, c0, a0, Ct, A2k, g, nl);
    }
}
template <int NBLK, int K>
__device__ __forceinline__ void symg_store(const f4 (&gS)[NBLK + 1], int R0, int R1, double* gBslab, int img)
{
    if constexpr (K <= NBLK) {
        const bool hi = (K <= R1);
        if (hi || R1 > R0) {
            const int row = hi ? R1 : R0, col = hi ? K : K - R1 - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) gBslab[(row * NBLK + col) * 256 + img + r * 16] = double(gS[K][r]);
        }
        symg_store<NBLK, K + 1>(gS, R0, R1, gBslab, img);
    }
}
template <int N, int K>
__device__ __forceinline__ void symg_zero(f4 (&gS)[N])
{
    if constexpr (K < N) {
        gS[K] = f4{0, 0, 0, 0};
        symg_zero<N, K + 1>(gS);
    }
}

// TRI: every product with K^-1 = W^T W (A2 = K^-1 K in phase C, K^-1 A2bar in phase F) runs as two triangular products
// W (.) then W^T (.) with W = L^-1, the way the reference back-substitutes twice (gp_tf.py:137,145) -- the zero blocks are
// skipped, the intermediate rows travel through one more LDS tile and one more workgroup barrier per product.  In float32
// the explicit K^-1 loses cond eps_32 in each of these products; the triangular factors lose sqrt(cond) eps_32.
// KSV: the forward evaluation kept every step's A2 and kernel tile (Rev32Args::a2s): phases B and C -- 6 MFMAs and four
// exponentials, then the whole K^-1 K product per row block -- become two loads issued a step ahead, the barrier between
// them goes, and the Z~ rows / row constants / K^-1 rows of the wave are not held.
template <int NBLK, int RB, int DK, int MODE, int NCB, bool TRI, bool KSV = false>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void rev32_kernel(Rev32Args a)
{
    constexpr int W = (NBLK + RB - 1) / RB, NT = 64 * W, MP = 16 * NBLK, KS = MP / 4;
    constexpr int JB = (4 * DK + 1 + 15) / 16;
    constexpr int NG = 4 * JB;
    constexpr int GPW = (NG + W - 1) / W;
    constexpr int QPW = (4 + W - 1) / W;
    constexpr int PD = 17;
    constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    constexpr bool SYMG = (RB == 2);                    // symmetric K_mm-adjoint accumulator, block rows (w, NBLK - 1 - w) per wave
    static_assert(!SYMG || NCB == 1, "SYMG tiles accumulate the whole triangle in one pass");
    typedef Slab<NBLK, JB, false> SL;

    extern __shared__ float lds32r[];
    __shared__ double red[16];
    float* xq0 = lds32r;                       // [2][4 DK][17]
    float* Kt = xq0 + 2 * 4 * DK * PD;         // [MP][17]
    float* A2t = Kt + MP * PD;                 // [MP][17]
    float* Fm = A2t + MP * PD;                 // [16][17]
    float* Fv = Fm + 16 * PD;
    float* part = Fv + 16 * PD;                // [W][PSL]
    float* A2k = part + W * PSL;               // [MP][17]  A2 rows of every wave, kept from phase E for the accumulation in F
    float* At = A2k + MP * PD;                 // [MP][17]  (TRI: rows of W K / W A2bar between the two triangular products)
    float* Ct = At + (TRI ? MP * PD : 0);      // [MP][17]  (SYMG: C = K^-1 A2bar rows of every wave, from phase F to barrier 6)

    const int tid = threadIdx.x, l = tid & 63, g = l >> 4, nl = l & 15;
    const int w = SYMG ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int N = a.N, S = a.S, T = a.T, Do = a.Do, D = a.D;
    const int naux = D - Do;
    const int dob = a.dim_x - a.dim_y;
    const int gx = blockIdx.x + a.group0;
    const int c0 = gx * 16;
    const int c = min(c0 + nl, N - 1);
    const bool cvalid = (c0 + nl) < N;
    const int bq = c / S;
    const int run = (MODE == MODE_BWD) ? int(blockIdx.y) : 0;
    const int R = a.recog_len, P = 2 * R;
    const int KSr = a.KSr;
    const int64_t wg_linear = (int64_t(blockIdx.z) * gridDim.y + blockIdx.y) * a.gtotal + gx;
    const bool first = a.first != 0;

    bool ok[RB];
    int rbs[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if constexpr (SYMG) {
            const int rb = (i == 0) ? w : NBLK - 1 - w;            // (the middle wave of an odd NBLK owns one row block)
            ok[i] = (i == 0) || (rb > w);
            rbs[i] = ok[i] ? rb : (NBLK - 1);
        } else {
            ok[i] = (w * RB + i) < NBLK;
            rbs[i] = ok[i] ? (w * RB + i) : (NBLK - 1);
        }
    }
    const int rb_lo = rbs[0], rb_hi = ok[RB - 1] ? rbs[RB - 1] : rbs[0];      // (two-triangular products: the k-block ranges)
    // Z~ rows and row constants of the owned row blocks (kernel tile, gp_tf.py:33-49)
    float Zreg[RB][DK], czr[RB][4];
    if constexpr (!KSV) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
#pragma unroll
            for (int s = 0; s < DK; ++s) Zreg[i][s] = a.pk.Zp[(rbs[i] * DK + s) * 64 + l];
#pragma unroll
            for (int r = 0; r < 4; ++r) czr[i][r] = a.pk.cz[16 * rbs[i] + 4 * g + r];
        }
    }
    // M <= 112 (one row block per wave), dense form: the K^-1 rows of the wave in VGPRs for the whole pass
    constexpr bool BREG = (NBLK <= 7 && RB == 1 && !TRI);
    float Breg[BREG ? KS : 1];
    if constexpr (BREG) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Breg[s] = a.pk.BpN[(int64_t(rbs[0]) * KS + s) * 64 + l];
    }

    f4 gMu[RB], gS2[RB], gZ[RB][JB], gB[RB][NCB];
    f4 gS[SYMG ? NBLK + 1 : 1];
    symg_zero<(SYMG ? NBLK + 1 : 1), 0>(gS);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        gMu[i] = f4{0, 0, 0, 0};
        gS2[i] = f4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < JB; ++j) gZ[i][j] = f4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NCB; ++j) gB[i][j] = f4{0, 0, 0, 0};
    }

    // phase D / G lane state.  Phase D lanes: state row d = 4 g + q of chain nl, q = w + qi W < 4
    float vx[QPW], vy[QPW], il[QPW], ivy[QPW], gcar[QPW], gdir[QPW];
    double gvx[QPW], gvy[QPW];                 // (sums over the whole pass: kept in float64 like the ELBO partial sums)
    double gsig = 0.0, glogsig = 0.0;
    bool act[QPW];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * g + q;
        act[qi] = (q < 4) && (d < Do);
        const int dc = act[qi] ? d : 0;
        vx[qi] = float(a.var_x[dc]);
        vy[qi] = (MODE == MODE_FWD) ? float(a.var_y[(a.half && dc >= a.dim_y) ? 0 : dc]) : 0.0f;
        il[qi] = a.pk.invl[dc];
        ivy[qi] = (MODE == MODE_FWD) ? 1.0f / vy[qi] : 0.0f;
        gcar[qi] = 0.0f; gdir[qi] = 0.0f; gvx[qi] = 0.0; gvy[qi] = 0.0;
    }
    double glx[GPW];
#pragma unroll
    for (int k2 = 0; k2 < GPW; ++k2) glx[k2] = 0.0;

    for (int i = tid; i < 2 * 4 * DK * PD; i += NT) xq0[i] = 0.0f;
    for (int i = tid; i < 16 * PD; i += NT) { Fm[i] = 0.0f; Fv[i] = 0.0f; }
    __syncthreads();

    // time range (as rev_kernel): forward-pass adjoint t = T-2 .. 0; backward runs: chunk z of run y covers whole
    // resample-to-resample segments, walked upwards in t
    int t_begin = 0, nsteps = 0;
    if (MODE == MODE_FWD) {
        nsteps = max(0, T - 1);
    } else {
        const int o = run * R;
        const int z = blockIdx.z, nz = a.nchunk;
        const int nsg = a.seg1 - a.seg0;
        const int k0 = a.seg0 + (z * nsg) / nz, k1 = a.seg0 + ((z + 1) * nsg) / nz;
        const int tb = (k0 <= 0) ? 0 : min(T, P * k0 - o);
        const int te = min(T, max(0, P * k1 - o));
        t_begin = tb;
        nsteps = max(0, te - tb);
    }
    if (MODE == MODE_FWD && nsteps > 0) {
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            if (act[qi] && d < a.dim_y) {
                // adjoint of x_{T-1}: only the log-likelihood sees it      (cbfssm.py:245-251)
                const float xv = float(a.x[(int64_t(T - 1) * N + c) * a.dim_x + d]);
                const float yv = float(a.y[(int64_t(bq) * T + (T - 1)) * a.dim_y + d]);
                gcar[qi] = -a.cL * (yv - xv) / vy[qi];
            }
        }
    }
    auto t_of = [&](int step) -> int { return (MODE == MODE_FWD) ? (T - 2 - step) : (t_begin + step); };
    int tmod = (MODE == MODE_BWD) ? (t_begin % P) : 0;

    const double* auxp[AUXR];
    int auxs[AUXR];
    float auxl[AUXR];
#pragma unroll
    for (int k2 = 0; k2 < AUXR; ++k2) {
        const int i = tid + k2 * NT, ja = i >> 4, n = i & 15;
        auxp[k2] = a.eps; auxs[k2] = 0; auxl[k2] = 0.0f;
        if (i < 16 * naux) {
            const int b = min(c0 + n, N - 1) / S;
            if (ja < a.dim_u) { auxp[k2] = a.u + int64_t(b) * T * a.dim_u + ja; auxs[k2] = a.dim_u; }
            else { auxp[k2] = a.y + int64_t(b) * T * a.dim_y + (ja - a.dim_u); auxs[k2] = a.dim_y; }
            auxl[k2] = a.pk.invl[Do + ja];
        }
    }
    auto load_inputs = [&](int t, int tm, float (&hv)[QPW], double (&av)[AUXR]) {
        bool rs = false;
        if (MODE == MODE_BWD) rs = (tm + 1 + run * R == P);                                        // cbfssm.py:124,127
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            hv[qi] = 0.0f;
            if (act[qi]) {
                double v;
                if (MODE == MODE_FWD) v = a.x[(int64_t(t) * N + c) * a.dim_x + d];
                else if (rs) v = a.hid[(int64_t(run) * T + t) * N + c];
                else if (t == T - 1) v = 0.0;                                                      // cbfssm.py:106
                else v = a.h_all[((int64_t(run) * T + (t + 1)) * N + c) * Do + d];                 // h_t = out_{t+1}
                hv[qi] = float(v);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) av[k2] = auxp[k2][int64_t(t) * auxs[k2]];
    };
    auto store_inputs = [&](float* xb, const float (&hv)[QPW], const double (&av)[AUXR]) {
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            if (act[qi]) xb[d * PD + nl] = hv[qi] * il[qi];
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT;
            if (i < 16 * naux) xb[(Do + (i >> 4)) * PD + (i & 15)] = float(av[k2]) * auxl[k2];
        }
    };
    // inputs of phase D of step t: {eps, y~ (fwd) or the y2 adjoint (bwd), fmean, fvar}.  Issued a phase early (at the top of
    // phase F, under its matrix loop) -- loaded where they are used they put two dependent HBM / L2 round trips on the serial
    // chain of every step (the float64 kernel's epilogue_load, cbfssm_adjoint.hpp).
    auto epilogue_load = [&](int t, int tm, float& eps_t, float (&yin)[QPW], float (&fm)[QPW], float (&fv)[QPW]) {
        eps_t = 0.0f;
        if (MODE == MODE_FWD) eps_t = float(a.eps[int64_t(t) * N + c]);
        else eps_t = float(a.eps[(int64_t(run) * T + t) * N + c]);
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            yin[qi] = 0.0f; fm[qi] = 0.0f; fv[qi] = 1.0f;
            if (act[qi]) {
                const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                const double* o = a.fmv + ((slot * N + c) * Do + d) * 2;
                fm[qi] = float(o[0]); fv[qi] = float(o[1]);
                if (MODE == MODE_FWD) {
                    if (d < a.dim_y) yin[qi] = float(a.y[(int64_t(bq) * T + (t + 1)) * a.dim_y + d]);
                    else if (!a.half) yin[qi] = float(a.y2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)]);
                } else {
                    const bool write = (run == 0) ? (tm < R) : (tm >= R);
                    if (write) yin[qi] = float(a.gy2[(int64_t(t) * N + c) * Do + d]);
                }
            }
        }
    };
    // phase D of step t: adjoint of the step epilogue (cbfssm.py:145-156, 205-235) from the carried state adjoint
    auto epilogue_adjoint = [&](int t, int tm, const float eps_t, const float (&yin)[QPW], const float (&fmi)[QPW],
                                const float (&fvi)[QPW]) {
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = w + qi * W;
            if (q < 4) {
                const int d = 4 * g + q;
                float gfm = 0.0f, gfv = 0.0f;
                if (act[qi] && cvalid) {
                    const float fmean = fmi[qi], fvar = fvi[qi];
                    const float gout = gcar[qi];
                    if (MODE == MODE_FWD) {
                        const bool do_cond = (a.condition || (t < R - 1)) && !(a.half && d >= a.dim_y);   // cbfssm.py:227
                        if (do_cond) {
                            const float ytil = yin[qi];
                            const float kf1 = a.k_factor - 1.0f;
                            const float vyt = vy[qi] + kf1 * fvar;
                            const float s = vyt + fvar;
                            const float rs = rcp32(s);
                            const float k = fvar * rs;
                            const float ydiff = ytil - fmean;
                            const float mu = fmean + k * ydiff;
                            const float omk = 1.0f - k;
                            const float sig = omk * omk * fvar + k * k * vyt;
                            const float rf = rcp32(fvar), rsig = rcp32(sig);
                            const float dm = mu - fmean;
                            // x' = mu + eps sqrt(sig);  kl = .5[log fvar - log sig + (sig + dm^2)/fvar - 1]
                            const float gmu = gout + a.cL * dm * rf;
                            const float gsg = gout * eps_t * 0.5f * rsqrt32(sig) + a.cL * 0.5f * (rf - rsig);
                            gfm = -a.cL * dm * rf;
                            gfv = a.cL * 0.5f * (rf - (sig + dm * dm) * rf * rf);
                            gfm += gmu * omk;                                   // mu = fmean + k (ytil - fmean)
                            float gk = gmu * ydiff;
                            const float gyt = gmu * k;
                            gk += gsg * (-2.0f * omk * fvar + 2.0f * k * vyt);  // sig = (1-k)^2 fvar + k^2 vyt
                            gfv += gsg * omk * omk;
                            float gvyt = gsg * k * k;
                            gfv += gk * rs;                                     // k = fvar / s ; s = vyt + fvar
                            const float gs = -gk * k * rs;
                            gvyt += gs;
                            gfv += gs;
                            if (first) gvy[qi] += double(gvyt);
                            gfv += kf1 * gvyt;                                  // vyt = vy + (kf - 1) fvar
                            if (first && d >= a.dim_y && !a.half) a.gy2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)] = double(gyt);
                        } else {
                            gfm = gout;                                         // x' = fmean + eps sqrt(fvar), no KL term
                            gfv = gout * eps_t * 0.5f * rsqrt32(fvar);
                            if (first && d >= a.dim_y && !a.half) a.gy2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)] = 0.0;
                        }
                    } else {
                        // out = fmean + eps sqrt(fvar); entropy term on written steps          (cbfssm.py:150-156)
                        const bool write = (run == 0) ? (tm < R) : (tm >= R);
                        const float gtot = gout + yin[qi];                      // (the y2 adjoint: zero on unwritten steps)
                        gfm = gtot;
                        gfv = gtot * eps_t * 0.5f * rsqrt32(fvar) - (write ? a.cE * 0.5f * rcp32(fvar) : 0.0f);
                    }
                    if (first) { gvx[qi] += double(gfv); gsig += double(gfv); }
                }
                gdir[qi] = gfm;
                Fm[d * PD + nl] = gfm;
                Fv[d * PD + nl] = gfv;
            }
        }
    };

    // KSV: the kept [A2 | kernel tile] registers of this wave's row blocks for the NEXT step, issued behind barrier 5
    const int64_t G16 = (int64_t(N) + 15) >> 4;
    f4 a2n[KSV ? RB : 1], kn[KSV ? RB : 1];
    auto load_saved = [&](int t) {
        if constexpr (KSV) {
            const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
            const float* rp = a.a2s + (slot * G16 + gx) * (2 * NBLK * 256) + l;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                a2n[i] = f4{0, 0, 0, 0}; kn[i] = f4{0, 0, 0, 0};
                if (ok[i]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a2n[i][r] = rp[(rbs[i] * 4 + r) * 64];
                        kn[i][r] = rp[NBLK * 256 + (rbs[i] * 4 + r) * 64];
                    }
                }
            }
        }
    };
    float hcur[QPW];
    if (nsteps > 0) load_saved(t_of(0));
    if (nsteps > 0) {
        double av[AUXR];
        load_inputs(t_of(0), tmod, hcur, av);
        store_inputs(xq0, hcur, av);
        float e0, y0[QPW], m0[QPW], v0[QPW];
        epilogue_load(t_of(0), tmod, e0, y0, m0, v0);
        epilogue_adjoint(t_of(0), tmod, e0, y0, m0, v0);
    }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int t = t_of(step);
        const bool has_next = (step + 1 < nsteps);
        const int tn = has_next ? t_of(step + 1) : t;
        float* xq = xq0 + (step & 1) * (4 * DK * PD);
        float* xqn = xq0 + ((step + 1) & 1) * (4 * DK * PD);
        bool resample_t = false;
        const int tmn = (tmod + 1 == P) ? 0 : tmod + 1;
        if (MODE == MODE_BWD) resample_t = (tmod + 1 + run * R == P);                              // cbfssm.py:124,127

        float hnext[QPW];
        double auxn[AUXR];
        if (has_next) load_inputs(tn, tmn, hnext, auxn);

        // ---- B: kernel tile rows of this wave -> LDS  (KSV: kept by the forward evaluation, loaded a step ahead)
        f4 kreg[RB], a2[RB];
        if constexpr (!KSV) {
        float bx[DK], xx = 0.0f;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            bx[s] = xq[(4 * s + g) * PD + nl];
            xx = fmaf(bx[s], bx[s], xx);
        }
        xx += __shfl_xor(xx, 16);
        xx += __shfl_xor(xx, 32);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            kreg[i] = f4{0, 0, 0, 0};
            if (ok[i]) {
                f4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = czr[i][r] - 0.5f * xx;
#pragma unroll
                for (int s = 0; s < DK; ++s) e = CBF_MFMA32(Zreg[i][s], bx[s], e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kreg[i][r] = expf(e[r]);
                    Kt[(16 * rbs[i] + 4 * g + r) * PD + nl] = kreg[i][r];
                }
            }
        }
        __syncthreads();                                                                           // 1
        }

        // rows of this wave of K^-1 X for the 16-column tile X in LDS ([row m][17]); the operand images stream from L2 in
        // natural k order.  TRI: W X -> At (own rows), barrier, W^T At
        auto kinv_times = [&](const float* X, f4 (&out)[RB]) {
            f4 acc[RB][2];
#pragma unroll
            for (int i = 0; i < RB; ++i) { acc[i][0] = f4{0, 0, 0, 0}; acc[i][1] = f4{0, 0, 0, 0}; }
            if constexpr (BREG) {
                if (ok[0]) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc[0][s & 1] = CBF_MFMA32(Breg[s], X[(4 * s + g) * PD + nl], acc[0][s & 1]);
                }
            } else if constexpr (!TRI) {
#pragma unroll 1
                for (int s0 = 0; s0 < KSr; s0 += 4) {
                    float b[4], aop[RB][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < RB; ++i) aop[i][j] = a.pk.BpN[(int64_t(rbs[i]) * KS + s0 + j) * 64 + l];
                        b[j] = X[(4 * (s0 + j) + g) * PD + nl];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (ok[i]) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
                }
            } else {
#pragma unroll 1
                for (int kb = 0; kb <= rb_hi; ++kb) {                // W is lower triangular: k-blocks kb <= rb
                    float b[4], aop[RB][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < RB; ++i) aop[i][j] = a.pk.WpN[(int64_t(rbs[i]) * KS + 4 * kb + j) * 64 + l];
                        b[j] = X[(4 * (4 * kb + j) + g) * PD + nl];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (ok[i] && kb <= rbs[i]) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
                }
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    if (ok[i]) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) At[(16 * rbs[i] + 4 * g + r) * PD + nl] = acc[i][0][r] + acc[i][1][r];
                    }
                    acc[i][0] = f4{0, 0, 0, 0}; acc[i][1] = f4{0, 0, 0, 0};
                }
                __syncthreads();
#pragma unroll 1
                for (int kb = rb_lo; kb < NBLK; ++kb) {              // W^T is upper triangular: k-blocks kb >= rb
                    float b[4], aop[RB][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < RB; ++i) aop[i][j] = a.pk.WTpN[(int64_t(rbs[i]) * KS + 4 * kb + j) * 64 + l];
                        b[j] = At[(4 * (4 * kb + j) + g) * PD + nl];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (ok[i] && kb >= rbs[i]) acc[i][j & 1] = CBF_MFMA32(aop[i][j], b[j], acc[i][j & 1]);
                }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) out[i] = acc[i][0] + acc[i][1];
        };

        // ---- C: A2 = K^-1 K, rows of this wave
        if constexpr (KSV) {
#pragma unroll
            for (int i = 0; i < RB; ++i) { kreg[i] = kn[i]; a2[i] = a2n[i]; }
        } else {
            kinv_times(Kt, a2);
        }

        // ---- E: A2bar and the parameter adjoints that contract over the 16 chains
        float fvsum = 0.0f, fmB[4], fvB[4], fmT[4], fvT[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fmB[s] = Fm[(4 * s + g) * PD + nl];
            fvB[s] = Fv[(4 * s + g) * PD + nl];
            fvsum += fvB[s];
            fmT[s] = Fm[nl * PD + 4 * s + g];
            fvT[s] = Fv[nl * PD + 4 * s + g];
        }
        fvsum += __shfl_xor(fvsum, 16);
        fvsum += __shfl_xor(fvsum, 32);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (ok[i]) {
                f4 T1 = {0, 0, 0, 0}, T2 = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    T1 = CBF_MFMA32(a.pk.muB[(rbs[i] * 4 + s) * 64 + l], fmB[s], T1);
                    T2 = CBF_MFMA32(a.pk.s2B[(rbs[i] * 4 + s) * 64 + l], fvB[s], T2);
                }
                f4 a2bar;
#pragma unroll
                for (int r = 0; r < 4; ++r) a2bar[r] = T1[r] + 2.0f * a2[i][r] * T2[r] - kreg[i][r] * fvsum;
                // 16 x 16 transposes through this wave's own rows of the A2bar tile: C layout (row 4 g + r, col nl) ->
                // A-operand layout (row nl, k = 4 s + g)
                float a2T[4];
                float* own = A2t + 16 * rbs[i] * PD;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    own[(4 * g + r) * PD + nl] = a2[i][r];
                    A2k[(16 * rbs[i] + 4 * g + r) * PD + nl] = a2[i][r];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < 4; ++s) a2T[s] = own[nl * PD + 4 * s + g];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) own[(4 * g + r) * PD + nl] = a2bar[r];      // stays: A2bar tile of phase F
                if (first) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        gMu[i] = CBF_MFMA32(a2T[s], fmT[s], gMu[i]);                    // mubar[m][d] += A2[m][n] Fm[d][n]
                        gS2[i] = CBF_MFMA32(a2T[s] * a2T[s], fvT[s], gS2[i]);           // s2bar[m][d] += A2[m][n]^2 Fv[d][n]
                    }
                }
            }
        }
        if (has_next) store_inputs(xqn, hnext, auxn);
        __syncthreads();                                                                           // 4

        // ---- F: Kbar = K^-1 A2bar - A2 o colsum(Fv), Ebar = Kbar o K, input adjoint partials, Zbar~
        // (phase D inputs of the next step and this step's observation for phase G: issued here, consumed after barrier 5)
        float eps_n = 0.0f, yin_n[QPW], fm_n[QPW], fv_n[QPW], ycur[QPW];
        if (has_next) epilogue_load(tn, tmn, eps_n, yin_n, fm_n, fv_n);
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            ycur[qi] = 0.0f;
            if (MODE == MODE_FWD && act[qi] && d < a.dim_y && (t >= 1 || a.half))
                ycur[qi] = float(a.y[(int64_t(bq) * T + t) * a.dim_y + d]);
        }
        f4 ebar[RB];
        {
            f4 kb_[RB];
            kinv_times(A2t, kb_);                     // (TRI: the A tile was last read in phase C of every wave, before barrier 4)
            // The K_mm adjoint's data part, accumulated as  G += (K^-1 A2bar) A2^T  (= K^-1 (A2bar K^T) K^-1, K^-1 being
            // symmetric) instead of  d loss / d K^-1 += A2bar K^T: both factors are K^-1-applied already, so nothing
            // multiplies the float32 accumulator by K^-1 from both sides afterwards (that amplified its rounding by
            // cond(K_mm): 1.6e-3 on the gradient at cond 4e4).  The host hands K_mm' G K_mm' to the float64 tail, which
            // expects d loss / d K^-1.  Transposes through this wave's own rows of the K tile (dead after phase E).
            if constexpr (SYMG) {
                // C rows of this wave -> the C tile; the accumulation follows barrier 5 (every wave's rows are there then)
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    if (ok[i])
#pragma unroll
                        for (int r = 0; r < 4; ++r) Ct[(16 * rbs[i] + 4 * g + r) * PD + nl] = kb_[i][r];
            } else {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (ok[i]) {
                    float kbT[4];
                    float* ownk = Kt + 16 * rbs[i] * PD;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ownk[(4 * g + r) * PD + nl] = kb_[i][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int s = 0; s < 4; ++s) kbT[s] = ownk[nl * PD + 4 * s + g];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        if (a.cb0 + cb < NBLK) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const float a2c = A2k[(16 * (a.cb0 + cb) + nl) * PD + 4 * s + g];
                                gB[i][cb] = CBF_MFMA32(kbT[s], a2c, gB[i][cb]);        // G[m'][m] += (K^-1 A2bar)[m'][n] A2[m][n]
                            }
                        }
                    }
                }
            }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ebar[i][r] = (kb_[i][r] - a2[i][r] * fvsum) * kreg[i][r];
        }
        f4 xp[JB];
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) xp[jb] = f4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (ok[i]) {
                // (the accumulator is the B operand: register r = rows 4 g + r, which is the k order of the ZTq image)
#pragma unroll
                for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        xp[jb] = CBF_MFMA32(a.pk.ZTq[((rbs[i] * JB + jb) * 4 + r) * 64 + l], ebar[i][r], xp[jb]);
            }
        }
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[w * PSL + (jb * 4 + r) * 64 + l] = xp[jb][r];
        if (first) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (ok[i]) {
                    float ebT[4];
                    float* ownk = Kt + 16 * rbs[i] * PD;           // the K tile is dead after phase E: own rows as scratch
#pragma unroll
                    for (int r = 0; r < 4; ++r) ownk[(4 * g + r) * PD + nl] = ebar[i][r];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int s = 0; s < 4; ++s) ebT[s] = ownk[nl * PD + 4 * s + g];
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) {
                        const int j = 16 * jb + nl;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            float xT = (j < 4 * DK) ? xq[j * PD + 4 * s + g] : 0.0f;
                            if (j == D) xT = 1.0f;                                      // ones column: row sums of Ebar
                            gZ[i][jb] = CBF_MFMA32(ebT[s], xT, gZ[i][jb]);              // Zbar~[m][j] += Ebar[m][n] x~[j][n]
                        }
                    }
                }
            }
        }
        __syncthreads();                                                                           // 5
        if (has_next) load_saved(tn);                  // (KSV) a2 / kreg were last read in phase F
        // SYMG: S += C A2^T + A2 C^T on this wave's block rows, both tiles complete and untouched until the next step's
        // phase E / F -- in the shadow of phases G / D, which are vector latency on the first waves
        if constexpr (SYMG) {
            const int R0 = w, R1 = NBLK - 1 - w;
            float c1[4], a1[4], c0[4], a0[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c1[s] = Ct[(16 * R1 + nl) * PD + 4 * s + g]; a1[s] = A2k[(16 * R1 + nl) * PD + 4 * s + g];
                c0[s] = Ct[(16 * R0 + nl) * PD + 4 * s + g]; a0[s] = A2k[(16 * R0 + nl) * PD + 4 * s + g];
            }
            symg_tiles<NBLK, 0>(gS, R0, R1, c1, a1, c0, a0, Ct, A2k, g, nl);
        }

        // ---- G: input adjoint, carried to the next reverse step.  Lane (g, nl) of group gi = 4 jb + q holds input row
        // j = 16 jb + 4 g + q of chain nl (register q of the xbar tile)
        float esum = 0.0f;
        {
            const int jbD = D >> 4, gD = (D >> 2) & 3, qD = D & 3;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) esum += part[ww * PSL + (jbD * 4 + qD) * 64 + gD * 16 + nl];
        }
#pragma unroll
        for (int k2 = 0; k2 < GPW; ++k2) {
            const int gi = w + k2 * W;
            if (gi < NG) {
                const int jb = gi >> 2, q = gi & 3;
                const int j = 16 * jb + 4 * g + q;
                float xb = 0.0f;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) xb += part[ww * PSL + (jb * 4 + q) * 64 + l];
                if (j < D && cvalid) {
                    const float xt = xq[j * PD + nl];
                    xb -= xt * esum;
                    if (first) glx[k2] += double(xb * xt);                               // lengthscale adjoint (inputs)
                }
                if (first && j == D && cvalid) glogsig += double(xb);
                if (jb == 0) {
                    // state rows hand their adjoint to the phase-D lanes of the same (d, chain): gi = q < 4 is group k2 = qi
                    // of this wave in both phases
#pragma unroll
                    for (int qi = 0; qi < QPW; ++qi) {
                        if (w + qi * W == q) {
                            float gin = 0.0f;
                            if (act[qi] && cvalid) gin = gdir[qi] + xb * il[qi];
                            if (MODE == MODE_FWD) {
                                const int d = 4 * g + q;
                                if (act[qi] && cvalid) {
                                    if ((t >= 1 || a.half) && d < a.dim_y) {
                                        gin += -a.cL * (ycur[qi] - hcur[qi]) * ivy[qi];  // log-likelihood term of x_t
                                    }
                                    if (first && t == 0) {
                                        if (a.half) a.gx0[int64_t(c) * a.dim_x + d] = double(gin);        // x_0 = recognition model
                                        else if (d >= a.dim_y) a.gy2[int64_t(c) * dob + (d - a.dim_y)] = double(gin);   // x_0 = y_tilde_0
                                    }
                                }
                                gcar[qi] = gin;
                            } else {
                                gcar[qi] = resample_t ? 0.0f : gin;                      // h_t = out_{t+1} unless resampled
                            }
                        }
                    }
                }
            }
        }
        // ---- D of the next step: same lanes as the carried adjoint just produced
        if (has_next) {
            epilogue_adjoint(tn, tmn, eps_n, yin_n, fm_n, fv_n);
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) hcur[qi] = hnext[qi];
        }
        tmod = tmn;
        __syncthreads();                                                                           // 6
    }

    if (MODE == MODE_FWD && nsteps == 0 && first) {
        // T == 1: x_0 = y_tilde_0 only feeds the log-likelihood through its observed dims -> no gradient to y2
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * g + (w + qi * W);
            if (act[qi] && cvalid) {
                if (a.half) {
                    float gv = 0.0f;
                    if (d < a.dim_y) gv = -a.cL * (float(a.y[(int64_t(bq) * T) * a.dim_y + d]) - float(a.x[int64_t(c) * a.dim_x + d])) / vy[qi];
                    a.gx0[int64_t(c) * a.dim_x + d] = double(gv);
                } else if (d >= a.dim_y) {
                    a.gy2[int64_t(c) * dob + (d - a.dim_y)] = 0.0;
                }
            }
        }
    }

    // ---- this workgroup's slab, float64, in the float64 kernels' C-layout images [r'][g'][nl] with row = 4 r' + g':
    // register r of lane group g is row 4 g + r here, i.e. image position r' = g, g' = r
    double* slab = a.gpart + wg_linear * a.slab;
    const int img = g * 64 + nl;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if (ok[i]) {
            const int rb = rbs[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (first) {
                    slab[SL::gMu + rb * 256 + img + r * 16] = double(gMu[i][r]);
                    slab[SL::gS2 + rb * 256 + img + r * 16] = double(gS2[i][r]);
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb) slab[SL::gZ + (rb * JB + jb) * 256 + img + r * 16] = double(gZ[i][jb][r]);
                }
                if constexpr (!SYMG) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
                        if (a.cb0 + cb < NBLK) slab[SL::gB + (rb * NBLK + a.cb0 + cb) * 256 + img + r * 16] = double(gB[i][cb][r]);
                }
            }
        }
    }
    // (SYMG: the lower-triangular blocks; the blocks above the diagonal of the slab section are never written and stay zero)
    if constexpr (SYMG) symg_store<NBLK, 0>(gS, w, NBLK - 1 - w, slab + SL::gB, img);
    if (!first) return;
    for (int i = tid; i < 192; i += NT) slab[SL::small + i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        double v1 = gvx[qi], v2 = gvy[qi];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
        if (q < 4 && nl == 0) {
            slab[SL::small + 4 * g + q] = v1;                 // d = 4 g + q
            slab[SL::small + 16 + 4 * g + q] = v2;
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < GPW; ++k2) {
        const int gi = w + k2 * W;
        double v = glx[k2];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (gi < NG && nl == 0) slab[SL::small + 32 + 16 * (gi >> 2) + 4 * g + (gi & 3)] = v;      // input row j
    }
    const double s1 = block_sum(gsig, red, tid, NT);
    const double s2 = block_sum(glogsig, red, tid, NT);
    if (tid == 0) {
        slab[SL::small + 96] = s1;
        slab[SL::small + 97] = s2;
    }
}

// column blocks of the K_mm-adjoint accumulator one launch holds per row block: all of them (one pass over the time loop at
// every tile height since the two-row-block tiles accumulate the symmetric part only)
constexpr int rev32_ncb(int nblk) { return nblk; }

template <int NBLK, int DK, bool TRI>
static int launch_rev32(int mode, const Rev32Args& a, dim3 grid, hipStream_t st)
{
    constexpr int RB = (NBLK >= 13) ? 2 : 1;
    constexpr int W = (NBLK + RB - 1) / RB;
    constexpr int JB = (4 * DK + 1 + 15) / 16;
    constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    constexpr int NCB = (RB == 2) ? 1 : rev32_ncb(NBLK);        // (RB == 2: the symmetric accumulator, see the kernel)
    const size_t lds = size_t(2 * 4 * DK * 17 + ((TRI ? 4 : 3) + (RB == 2 ? 1 : 0)) * 16 * NBLK * 17 + 2 * 16 * 17 + W * PSL) * sizeof(float);
    hipError_t e = hipSuccess;
    auto go = [&](auto k) {
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e == hipSuccess) hipLaunchKernelGGL(k, grid, dim3(64 * W), lds, st, a);
    };
    if (mode == MODE_FWD) {
        if (a.a2s) go(rev32_kernel<NBLK, RB, DK, MODE_FWD, NCB, TRI, true>);
        else go(rev32_kernel<NBLK, RB, DK, MODE_FWD, NCB, TRI, false>);
    } else {
        if (a.a2s) go(rev32_kernel<NBLK, RB, DK, MODE_BWD, NCB, TRI, true>);
        else go(rev32_kernel<NBLK, RB, DK, MODE_BWD, NCB, TRI, false>);
    }
    if (e != hipSuccess) return -int(e) - 1000;
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK>
static int launch_rev32_n(int DK, int mode, const Rev32Args& a, dim3 grid, hipStream_t st)
{
    if (a.tri) {
        switch (DK) {
            case 2: return launch_rev32<NBLK, 2, true>(mode, a, grid, st);
            case 4: return launch_rev32<NBLK, 4, true>(mode, a, grid, st);
            case 6: return launch_rev32<NBLK, 6, true>(mode, a, grid, st);
        }
        return -2;
    }
    switch (DK) {
        case 2: return launch_rev32<NBLK, 2, false>(mode, a, grid, st);
        case 4: return launch_rev32<NBLK, 4, false>(mode, a, grid, st);
        case 6: return launch_rev32<NBLK, 6, false>(mode, a, grid, st);
    }
    return -2;
}

static int dispatch_rev32(int NBLK, int DK, int mode, const Rev32Args& a, dim3 grid, hipStream_t st)
{
    switch (NBLK) {
        case 1: return launch_rev32_n<1>(DK, mode, a, grid, st);
        case 2: return launch_rev32_n<2>(DK, mode, a, grid, st);
        case 4: return launch_rev32_n<4>(DK, mode, a, grid, st);
        case 7: return launch_rev32_n<7>(DK, mode, a, grid, st);
        case 10: return launch_rev32_n<10>(DK, mode, a, grid, st);
        case 13: return launch_rev32_n<13>(DK, mode, a, grid, st);
        case 16: return launch_rev32_n<16>(DK, mode, a, grid, st);
        case 20: return launch_rev32_n<20>(DK, mode, a, grid, st);
    }
    return -2;
}

static int64_t slab32(const cbfssm_pack_layout* L)
{
    const int JB = L->JB;
    return int64_t(L->NBLK) * 256 * 2 + int64_t(L->NBLK) * L->NBLK * 256 + int64_t(L->NBLK) * JB * 256 + 192;
}

static int fill_rev32(Rev32Args& a, const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32, int Do)
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
    a.slab = slab32(L);
    a.KSr = (L->M + 3) / 4;
    a.tri = (L->gp_form == CBFSSM_GP_FORM_TRI);
    a.gtotal = (a.N + 15) / 16;
    a.group0 = p->ngroups > 0 ? p->group0 : 0;
    return 0;
}

}  // namespace f32
}  // namespace cbfssm

using namespace cbfssm;
using namespace cbfssm::f32;

extern "C" {

int64_t cbfssm_rev32_slab_elems(const cbfssm_pack_layout* L)
{
    if (!L) return -1;
    return slab32(L);
}

static int forward_bwd32_impl(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f, const double* var_x,
                              const double* var_y, const double* u, const double* y, const double* y2, const double* eps_f,
                              const double* x, const double* fmv_f, const float* a2s_f, double cL, double* gy2, double* gx0,
                              double* gpart, void* stream)
{
    Rev32Args a;
    int rc = fill_rev32(a, p, L, pack32_f, p ? p->dim_x : 0);
    if (rc) return rc;
    if (!var_x || !var_y || !u || !y || !x || !gpart || (p->T > 1 && (!fmv_f || !eps_f))) return fail(-1, "null pointer");
    if (p->half ? !gx0 : (p->dim_x > p->dim_y && (!y2 || !gy2))) return fail(-1, "y2/gy2/gx0 is null");
    a.cL = float(cL); a.var_x = var_x; a.var_y = var_y; a.u = u; a.y = y; a.eps = eps_f; a.x = x; a.y2 = y2; a.gy2 = gy2;
    a.gpart = gpart; a.fmv = fmv_f; a.a2s = a2s_f; a.half = p->half; a.gx0 = gx0;
    const int ncb = rev32_ncb(L->NBLK);
    dim3 grid(unsigned(p->ngroups > 0 ? p->ngroups : (a.N + 15) / 16), 1, 1);
    for (int cb0 = 0; cb0 < L->NBLK; cb0 += ncb) {
        a.cb0 = cb0; a.first = (cb0 == 0);
        rc = dispatch_rev32(L->NBLK, L->DK, MODE_FWD, a, grid, (hipStream_t)stream);
        if (rc) return fail(rc, "forward_pass_bwd_f32 launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    }
    return 0;
}

int cbfssm_forward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f,
                                const double* var_x, const double* var_y, const double* u, const double* y,
                                const double* y2, const double* eps_f, const double* x, const double* fmv_f, const float* a2s_f,
                                double cL, double* gy2, double* gpart, void* stream)
{
    if (p && p->half) return fail(-1, "problem->half is set: use cbfssm_half_forward_pass_bwd_f32");
    return forward_bwd32_impl(p, L, pack32_f, var_x, var_y, u, y, y2, eps_f, x, fmv_f, a2s_f, cL, gy2, nullptr, gpart, stream);
}

int cbfssm_half_forward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_f,
                                     const double* var_x, const double* var_y, const double* u, const double* y,
                                     const double* eps_f, const double* x, const double* fmv_f, const float* a2s_f, double cL,
                                     double* gx0, double* gpart, void* stream)
{
    if (!p || !p->half) return fail(-1, "problem->half must be 1");
    return forward_bwd32_impl(p, L, pack32_f, var_x, var_y, u, y, nullptr, eps_f, x, fmv_f, a2s_f, cL, nullptr, gx0, gpart, stream);
}

int cbfssm_backward_pass_bwd_f32(const cbfssm_problem* p, const cbfssm_pack_layout* L, const float* pack32_b,
                                 const double* var_x, const double* u, const double* y, const double* hid_b,
                                 const double* eps_b, const double* h_all, const double* fmv_b, const float* a2s_b,
                                 const double* gy2, double cE, double* gpart, void* stream)
{
    Rev32Args a;
    int rc = fill_rev32(a, p, L, pack32_b, p ? p->dim_x - p->dim_y : 0);
    if (rc) return rc;
    if (!var_x || !u || !y || !hid_b || !eps_b || !h_all || !fmv_b || !gy2 || !gpart) return fail(-1, "null pointer");
    a.cE = float(cE); a.var_x = var_x; a.u = u; a.y = y; a.eps = eps_b; a.hid = hid_b; a.h_all = h_all;
    a.gy2 = const_cast<double*>(gy2); a.gpart = gpart; a.fmv = fmv_b; a.a2s = a2s_b;
    const int nseg = cbfssm_bwd_segments(p);
    const int nchunk = int(cbfssm_rev_workgroups(p, 1) / (2 * ((int64_t(a.N) + 15) / 16)));      // as the float64 adjoint chunks
    a.seg0 = 0; a.seg1 = nseg; a.nchunk = nchunk < 1 ? 1 : (nchunk > nseg ? nseg : nchunk);
    const int ncb = rev32_ncb(L->NBLK);
    dim3 grid(unsigned(p->ngroups > 0 ? p->ngroups : (a.N + 15) / 16), 2, unsigned(a.nchunk));
    for (int cb0 = 0; cb0 < L->NBLK; cb0 += ncb) {
        a.cb0 = cb0; a.first = (cb0 == 0);
        rc = dispatch_rev32(L->NBLK, L->DK, MODE_BWD, a, grid, (hipStream_t)stream);
        if (rc) return fail(rc, "backward_pass_bwd_f32 launch failed (NBLK=%d DK=%d rc=%d)", L->NBLK, L->DK, rc);
    }
    return 0;
}

}  // extern "C"
