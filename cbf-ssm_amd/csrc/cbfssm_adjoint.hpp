// Reverse-mode (adjoint) time-loop kernels of the CBF-SSM ELBO: what tf.train.AdamOptimizer.minimize differentiates
// through the two tf.while_loops (cbfssm/model/cbfssm.py:107-111,176-179,273-275), re-derived by hand.
//
// One workgroup owns 16 chains and walks the recurrence backwards from the saved trajectory (x_t or h_t).  Of the
// forward evaluation it reads the per-step (fmean, fvar) and, when they were kept, the A2 = K^-1 K tiles; the kernel
// tile K = k(Z, x_t) is recomputed (and A2 too when no tiles were kept: phase C).  Per step, between four workgroup
// barriers:
//
//   B   K tile of this wave's rows (MFMA + exp) -> LDS;           (the next step's inputs are loaded meanwhile)
//   E   A2bar = mu Fm + 2 A2 o (s2 Fv) - K o colsum(Fv)                                                  MFMA
//       mubar += A2 Fm^T,  s2bar += (A2 o A2) Fv^T,  Kinvbar += A2bar K^T      (k-dim = the 16 chains)    MFMA
//   F   Kbar = Kinv A2bar - A2 o colsum(Fv);  Ebar = Kbar o K                                            MFMA
//       xbar~ = Z~^T Ebar - x~ o colsum(Ebar),  Zbar~ += Ebar x~^T                                       MFMA
//   G   carry d loss/d x_t (or d loss/d h_t) to the next reverse step, then in the same lanes
//   D   per-(chain, dim) adjoint of the NEXT step's epilogue -> Fm = d loss/d fmean, Fv = d loss/d fvar  (16 x 16 each)
//
// Parameter adjoints accumulate in VGPRs over the whole pass and leave the kernel once, as one partial slab per
// workgroup (summed in a fixed order by reduce_partials_kernel: bitwise reproducible, no atomics).
#pragma once
#include "cbfssm_kernels.hpp"

namespace cbfssm {

struct RevPackPtrs {
    const double* muB;   // [NBLK][4][64]   A[row m][k = d]
    const double* s2B;
    const double* ZT;    // [NBLK][JB][4][64]  A[row j][k = m], row D is all ones (for m < M)
};

struct RevArgs {
    PackPtrs pk;
    RevPackPtrs rk;
    int N, S, T, B;
    int dim_x, dim_u, dim_y;
    int Do, D;
    int recog_len, condition;
    double k_factor;
    double cL;             // lambda0 / S : weight of (kl_x - loglik) in the loss      (cbfssm.py:257-262)
    double cE;             // lambda1 / S : weight of -entropy
    const double* var_x;
    const double* var_y;
    const double* u;
    const double* y;
    const double* eps;     // fwd: (T-1,N); bwd: (2,T,N)
    const double* hid;     // bwd: (2,T,N)
    const double* x;       // fwd: (T,N,dim_x) saved trajectory
    const double* y2;      // fwd: (T,N,dob)
    const double* h_all;   // bwd: (2,T,N,dob) saved outputs of both runs
    double* gy2;           // (T,N,dob): written by the fwd reverse, read by the bwd reverse
    double* gpart;         // [workgroup][slab]
    int64_t slab;          // doubles per workgroup
    int KSr;               // ceil(M / 4)
    // time range of this launch
    int t_hi, t_lo;        // fwd: steps t = t_hi .. t_lo (descending), t_hi <= T-2
    int seg0, seg1;        // bwd: resample-to-resample segments [seg0, seg1) of each run, split over grid.z chunks
    int nchunk;            // bwd: grid.z
    double* gx_carry;      // fwd: (N, dim_x) adjoint of x_{t_lo} handed to the next launch (null: single launch)
    // stash mode (tile heights whose K^-1-adjoint does not fit the VGPR file): the A2bar^T and K^T operand images of
    // every step go to HBM, [slot = workgroup * chunk_steps + step][NBLK][4][64] doubles each (stash_ld = 16 x slots);
    // cbfssm_stash_contract_f64 contracts them after the launch
    double* stash_a;
    double* stash_k;
    int64_t stash_ld;
    int chunk_steps;
    int half;              // CBFSSMHALF forward pass
    double* gx0;           // half: (N, dim_x) d loss / d x_0 per chain (summed over the particles by the caller)
    int group0, gtotal;    // this launch covers chain groups [group0, group0 + gridDim.x) of gtotal
    const double* a2s;     // optional: A2 = K^-1 k tiles of every step as saved by the forward evaluation
                           // (PassArgs::a2s layout); NULL -> phase C recomputes them
    const double* fmv;     // (fmean, fvar) of every step as saved by the forward evaluation (PassArgs::fmv layout)
    int ksave;             // 1: every saved record is [A2 tile][kernel tile] (PassArgs::ksave)
};

// slab layout (doubles), all in MFMA C-layout [r][lane] blocks of 256
template <int NBLK, int JB, bool STASH>
struct Slab {
    static constexpr int gMu = 0;                                         // [NBLK][256]
    static constexpr int gS2 = gMu + NBLK * 256;                          // [NBLK][256]
    static constexpr int gB = gS2 + NBLK * 256;                           // [NBLK][NBLK][256]  (absent in stash mode)
    static constexpr int gZ = gB + (STASH ? 0 : NBLK * NBLK * 256);       // [NBLK][JB][256]
    static constexpr int small = gZ + NBLK * JB * 256;   // [0,16) gvx, [16,32) gvy, [32,32+16*JB) glx, 96 gsig, 97 glogsig
    static constexpr int total = small + 192;            // [100,192): diagnostic stamps (CBF_REV_STAMPS builds only)
};


// LDS budget of one adjoint workgroup (doubles).  Stash-mode tiles re-read the Z~ operand images every step instead of
// holding them in registers; when the K^-1 image does not fit anyway, the LDS left over holds those images (a read
// that misses L1 costs an L2 round trip right in front of the MFMAs that need it).
template <int NBLK, int RB, int DK, bool STASH>
struct RevLds {
    static constexpr int W = (NBLK + RB - 1) / RB;
    static constexpr int JB = (4 * DK + 1 + 15) / 16;
    static constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    static constexpr int BASE_PLAIN = 2 * 4 * DK * 17 + 2 * (16 * NBLK) * 17 + 2 * 16 * 17 + W * PSL + 64;
    // Stash-mode tiles that stream K^-1: the per-wave partial tiles of the input adjoint (`part`, written at the end of
    // phase F, read in phase G) live in the K tile's LDS, which is dead by then -- one more workgroup barrier per step
    // buys W x PSL doubles (28 KB at NBLK = 13) for the operand images below.
    static constexpr bool PALIAS = STASH;
    static constexpr int KTR = PALIAS ? ((16 * NBLK) * 17 > W * PSL ? (16 * NBLK) * 17 : W * PSL) : (16 * NBLK) * 17;
    static constexpr int BASE = PALIAS ? 2 * 4 * DK * 17 + KTR + (16 * NBLK) * 17 + 2 * 16 * 17 + 64 : BASE_PLAIN;
    static constexpr int LIMIT = 163840 / 8;
    static constexpr int ZP = NBLK * DK * 64 + 16 * NBLK;          // Z~ A-operand image + row constants
    static constexpr int ZT = NBLK * JB * 256;                     // (Z~)^T A-operand image
    static constexpr int MU = NBLK * 256;                          // mu_z B-operand image (phase E)
    // filled in this order (measured at NBLK = 13 / D = 21, where only part fits: {(Z~)^T} and {Z~, mu} are within 1 %)
    static constexpr bool ZTLDS = STASH && (BASE + ZT <= LIMIT);
    static constexpr bool ZLDS = STASH && (BASE + ZP + (ZTLDS ? ZT : 0) <= LIMIT);
    static constexpr bool MULDS = ZLDS && (BASE + ZP + (ZTLDS ? ZT : 0) + MU <= LIMIT);
    static constexpr int EXTRA = (ZLDS ? ZP : 0) + (ZTLDS ? ZT : 0) + (MULDS ? MU : 0);   // variants without the K^-1 image
};

// BLDS: the K^-1 A-operand image lives in LDS for the whole pass (one copy per workgroup, trimmed to the ceil(M/4)
// k-steps that carry data); otherwise it streams from L2.  RB: 16-row blocks of inducing points per wave.
// Seven row-block waves load the four SIMDs 2-2-2-1.  The in-register tiles of that height get an EIGHTH wave, which
// lands next to wave 3 and takes the last REV_XCB column blocks of the Kinvbar accumulation for ALL row blocks (its
// operands, A2bar and K, are complete in the LDS tiles between the barriers that end phases E and F); it has no other
// role and leaves before the epilogue (a wave that has ended no longer counts at s_barrier).
// Measured on the C3 kernels (ms, forward-pass / backward-run adjoint; without the wave 5.12 / 5.83):
//   1 block 5.14 / 5.59, 2 blocks 4.89 / 5.60, 3 blocks 4.81 / 5.72, 4 blocks 5.11 / 6.01 (the extra wave's 112 MFMAs
//   no longer fit into phase F next to wave 3's own); with the operand loads of phases E and F issued early (below):
//   2 blocks 4.68 / 5.54, 3 blocks 4.30 / 5.40, 4 blocks 4.66 / -.
#ifndef CBF_REV_XCB
#define CBF_REV_XCB 3
#endif
constexpr int REV_XCB = CBF_REV_XCB;
// ... and it STARTS EARLY.  Its operands of row block rb exist as soon as that row block's wave has written its A2bar rows,
// a quarter into phase E -- not only after the barrier that ends E.  Measured per wave (profiles/r02/
// adjoint_phase_shares_per_wave.log): with the whole of its 84 MFMAs between the barriers of phase F, wave 3's SIMD carries
// 41 + 84 MFMAs there against 82 on the others (phase F 41 % of a step on wave 3, 33-35 % elsewhere, everybody waiting
// for it) and idles half of phase E (wave 3 waits 9-11 % there).  So the row-block waves raise an LDS flag per row block
// (flag_release / flag_wait, cbfssm_kernels.hpp) and the extra wave takes the first REV_XEARLY row blocks before that
// barrier, the rest after it.
#ifndef CBF_REV_XEARLY
#define CBF_REV_XEARLY 3
#endif
constexpr int REV_XEARLY = CBF_REV_XEARLY;
#ifndef CBF_REV_XLATE
#define CBF_REV_XLATE 1
#endif
constexpr int REV_XLATE = CBF_REV_XLATE;      // row blocks the extra wave takes behind barrier 5 (see the kernel)
#ifdef CBF_NO_XW          // diagnostic builds: the seven-wave form
constexpr bool rev_extra_wave(int, bool) { return false; }
#else
// Stash-mode tiles get an extra wave too, with another job: it is the STASH WRITER.  It copies the two operand images of every
// step (K^T right after barrier 1, A2bar^T after barrier 4) from the LDS tiles to HBM, so that the row-block waves issue
// no global stores inside the time loop: vmcnt retires in order and counts stores, and at 256 VGPRs with two row blocks per
// wave these kernels reload spilled registers all through a step -- every such reload was an `s_waitcnt vmcnt(0)` that also
// waited for the step's stash stores to reach HBM (isa_wait_audit.py: `vmcnt(1): waits for 9 [WWWWWWWWS]`).  It lands on
// the SIMD that carries the fewest row blocks.  C4 adjoints 9.80 / 13.78 -> 9.58 / 13.34 ms, train step 38.9 -> 38.0 ms.
// Not at 16 row blocks: eight waves would become nine and the three-wave SIMD would cap everybody at 168 VGPRs.
// (Measured and NOT kept, round 3: the same wave also taking over Zbar~ += Ebar x~^T for all row blocks through an Ebar
// tile in LDS -- the row-block waves shed 32 accumulator VGPRs (spill slots 64 -> 48 / 34 -> 12) and 16 MFMAs per step, the
// kernels did not move: 9.68 / 13.35 ms.  And kernel tiles kept by the forward evaluation for these tile heights, read back
// instead of rebuilt: 10.9 / 15.9 ms loaded at the step top, 11.1 / 15.9 ms prefetched a step ahead -- slower both ways.)
#ifdef CBF_REV_RB13
constexpr bool rev_extra_wave(int nblk, bool stash) { return stash ? (nblk != 16 && nblk != 13) : (nblk == 7); }
#else
constexpr bool rev_extra_wave(int nblk, bool stash) { return stash ? (nblk != 16) : (nblk == 7); }
#endif
#endif

// KD: k-steps of the products whose k index is the GP output dimension (mu Fm, s2 Fv): 4 in general, 2 when the launcher
// knows Do <= 8 (the backward runs of the Sarcos class: dim_x - dim_y = 7) -- the other two would multiply zeros.
// KSV: the forward evaluation kept the kernel tile of every step next to its A2 tile (RevArgs::ksave, non-stash tiles):
// phase B is a copy -- registers that were loaded a step ahead go to the LDS tile -- instead of 6 MFMAs and four
// exponentials per lane, and the Z~ rows / row constants of the wave are not held at all.
template <int NBLK, int RB, int DK, bool BLDS, bool STASH, int MODE, int KD = 4, bool KSV = false>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB + (rev_extra_wave(NBLK, STASH) ? 1 : 0))) void rev_kernel(RevArgs a)
{
    static_assert(!(KSV && STASH), "kernel tiles are kept for the register-resident tile heights only");
    constexpr bool BREG = false;
    typedef Tile<NBLK, RB, DK, BREG> TT;
    constexpr int W = TT::W, NT = TT::NT, MP = TT::MP, KS = TT::KS;
    constexpr int JB = (4 * DK + 1 + 15) / 16;          // 16-row blocks covering the D inputs + the ones row
    constexpr int NG = 4 * JB;                          // 4-row groups of the input-adjoint tile
    constexpr int GPW = (NG + W - 1) / W;               // groups per wave in phase G
    constexpr int PD = 17;                              // padded row length of the LDS tiles
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    constexpr bool XW = rev_extra_wave(NBLK, STASH) && !STASH;
    constexpr bool SW = rev_extra_wave(NBLK, STASH) && STASH;      // stash-writer wave
    constexpr int XCB = XW ? REV_XCB : 0;               // column blocks of Kinvbar accumulated by the extra wave
    constexpr int NCB = STASH ? 1 : NBLK - XCB;         // ... and by the wave that owns the row block
    constexpr bool ALL_OK = (NBLK % RB == 0);           // every wave owns RB real row blocks (no predicate around MFMAs)
    typedef Slab<NBLK, JB, STASH> SL;

    extern __shared__ double lds[];
    double* xq0 = lds;                                  // [2][4*DK][17]: this step's and the next step's inputs
    typedef RevLds<NBLK, RB, DK, STASH> RL;
    constexpr bool PALIAS = RL::PALIAS && !BLDS;        // `part` shares the K tile's LDS (see RevLds)
    constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    double* Kt = xq0 + 2 * 4 * DK * PD;                 // [MP][17]
    double* A2t = Kt + (PALIAS ? RL::KTR : MP * PD);    // [MP][17]
    double* Fm = A2t + MP * PD;                         // [16][17]
    double* Fv = Fm + 16 * PD;                          // [16][17]
    double* part = PALIAS ? Kt : (Fv + 16 * PD);        // [W][max(2,JB)][4][64]
    double* red = PALIAS ? (Fv + 16 * PD) : (part + W * PSL);   // 64
    int* xflag = reinterpret_cast<int*>(red);           // extra wave: "A2bar rows of row block rb are written" flags
                                                        // (red itself is only used by the block sums after the time loop)
    double* Bl = red + 64;                              // BLDS: [NBLK][KSr][64]
    constexpr bool ZLDS = RL::ZLDS && !BLDS, ZTLDS = RL::ZTLDS && !BLDS, MULDS = RL::MULDS && !BLDS;
    double* ZTl = red + 64;                             // ZTLDS: [NBLK][JB][4][64]
    double* Zl = ZTl + (ZTLDS ? NBLK * JB * 256 : 0);   // ZLDS: [NBLK][DK][64] then cz [MP]
    double* czl = Zl + NBLK * DK * 64;
    double* mul = Zl + (ZLDS ? RL::ZP : 0);             // MULDS: [NBLK][4][64]

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, nl = l & 15;
    const int N = a.N, S = a.S, T = a.T, Do = a.Do, D = a.D;
    const int naux = D - Do;
    const int dob = a.dim_x - a.dim_y;
    const int gx = blockIdx.x + a.group0;               // chain group of this workgroup
    const int c0 = gx * 16;
    const int c = min(c0 + nl, N - 1);
    const bool cvalid = (c0 + nl) < N;
    const int bq = c / S;
    const int run = (MODE == MODE_BWD) ? int(blockIdx.y) : 0;
    const int R = a.recog_len, P = 2 * R;
    const int KSr = a.KSr;                              // k-steps of K^-1 that carry data: ceil(M/4)
    const int64_t wg_linear = (int64_t(blockIdx.z) * gridDim.y + blockIdx.y) * a.gtotal + gx;

    // ---- loop-invariant operands (Z~ rows and cz of the owned row blocks stay in VGPRs; the small operand images
    // muA/s2A/muB/s2B/ZT are re-read from L1/L2 where they are used: the VGPRs hold the adjoint accumulators)
    TT tile;
    if constexpr (!KSV) tile.template load_operands<false>(a.pk, w, l);
    bool ok[RB];
    int rbs[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        ok[i] = (w * RB + i) < NBLK;
        rbs[i] = ok[i] ? (w * RB + i) : (NBLK - 1);
    }

    // ---- accumulators of the parameter adjoints (whole launch)
    d4 gMu[RB], gS2[RB], gZ[RB][JB], gB[RB][NCB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        gMu[i] = d4{0, 0, 0, 0};
        gS2[i] = d4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < JB; ++j) gZ[i][j] = d4{0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NCB; ++j) gB[i][j] = d4{0, 0, 0, 0};
    }

    // ---- phase D/G lane state.  Phase D lanes: (d = 4q+g, chain nl), q = w + qi*W < 4.
    constexpr int QPW = TT::QPW;
    double vx[QPW], vy[QPW], il[QPW], ivy[QPW];
    double gcar[QPW];          // adjoint of the chain state arriving from the previously processed step
    double gdir[QPW];          // direct (residual) path adjoint of this step's input state
    double gvx[QPW], gvy[QPW];
    double gsig = 0.0;
    bool act[QPW];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        const int d = 4 * q + g;
        act[qi] = (q < 4) && (d < Do);
        const int dc = act[qi] ? d : 0;
        vx[qi] = a.var_x[dc];
        vy[qi] = (MODE == MODE_FWD) ? a.var_y[(a.half && dc >= a.dim_y) ? 0 : dc] : 0.0;
        il[qi] = a.pk.invl[dc];
        ivy[qi] = (MODE == MODE_FWD) ? 1.0 / vy[qi] : 0.0;
        gcar[qi] = 0.0; gdir[qi] = 0.0; gvx[qi] = 0.0; gvy[qi] = 0.0;
    }
    // Phase G lanes: input row j = 4*gi + g, gi = w + k*W < NG
    double glx[GPW];
    double glogsig = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < GPW; ++k2) glx[k2] = 0.0;


    for (int i = tid; i < 2 * 4 * DK * PD; i += NT) xq0[i] = 0.0;
    double* xq = xq0;
    for (int i = tid; i < 16 * PD; i += NT) { Fm[i] = 0.0; Fv[i] = 0.0; }
    if (tid < 64) { xflag[tid] = 0; xflag[tid + 64] = 0; }
    if (BLDS) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if (ok[i])
                for (int s = 0; s < KSr; ++s) Bl[(rbs[i] * KSr + s) * 64 + l] = a.pk.Bp[(rbs[i] * KS + s) * 64 + l];
    }
    if constexpr (ZLDS) {
        for (int i = tid; i < NBLK * DK * 64; i += NT) Zl[i] = a.pk.Zp[i];
        for (int i = tid; i < MP; i += NT) czl[i] = a.pk.cz[i];
    }
    if constexpr (ZTLDS) {
        for (int i = tid; i < NBLK * JB * 256; i += NT) ZTl[i] = a.rk.ZT[i];
    }
    if constexpr (MULDS) {
        for (int i = tid; i < NBLK * 256; i += NT) mul[i] = a.rk.muB[i];
    }
    const double* bop[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) bop[i] = BLDS ? (Bl + rbs[i] * KSr * 64 + l) : (a.pk.Bp + rbs[i] * KS * 64 + l);
    __syncthreads();

    // ---- time range.  Backward runs: chunk z of run y covers whole resample-to-resample segments (the carried
    // adjoint is zero at a segment start, cbfssm.py:133-136), so chunks are independent workgroups.
    int t_begin = 0, nsteps = 0;
    if (MODE == MODE_FWD) {
        nsteps = a.t_hi - a.t_lo + 1;
        if (nsteps < 0) nsteps = 0;
    } else {
        const int o = run * R;
        const int z = blockIdx.z, nz = a.nchunk;
        const int nsg = a.seg1 - a.seg0;
        const int k0 = a.seg0 + (z * nsg) / nz, k1 = a.seg0 + ((z + 1) * nsg) / nz;   // segment k starts at max(0, P*k - o)
        const int tb = (k0 <= 0) ? 0 : min(T, P * k0 - o);
        const int te = min(T, max(0, P * k1 - o));
        t_begin = tb;
        nsteps = max(0, te - tb);
    }
    if (MODE == MODE_FWD && nsteps > 0) {
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * (w + qi * W) + g;
            if (act[qi]) {
                if (a.t_hi == T - 2) {
                    // adjoint of x_{T-1}: only the log-likelihood sees it      (cbfssm.py:245-251)
                    if (d < a.dim_y) {
                        const double xv = a.x[(int64_t(T - 1) * N + c) * a.dim_x + d];
                        const double yv = a.y[(int64_t(bq) * T + (T - 1)) * a.dim_y + d];
                        gcar[qi] = -a.cL * (yv - xv) / vy[qi];
                    }
                } else {
                    gcar[qi] = a.gx_carry[int64_t(c) * a.dim_x + d];      // from the launch that handled t_hi + 1
                }
            }
        }
    }

    // ---- step pipeline.  The GP input of step s+1 (saved trajectory, no dependence on the reverse sweep) is loaded
    // while step s computes and lands in the other xq buffer; the epilogue adjoint of step s+1 (phase D) follows phase
    // G of step s in the same lanes, so a step has four workgroup barriers and no load latency on its critical path.
    auto t_of = [&](int step) -> int { return (MODE == MODE_FWD) ? (a.t_hi - step) : (t_begin + step); };
    int tmod = (MODE == MODE_BWD) ? (t_begin % P) : 0;          // t mod 2R of the step being processed (backward runs)
    // auxiliary input rows of this thread: base pointer, time stride and 1/lengthscale, fixed for the whole pass
    const double* auxp[AUXR];
    int auxs[AUXR];
    double auxl[AUXR];
#pragma unroll
    for (int k2 = 0; k2 < AUXR; ++k2) {
        const int i = (NT - 1 - tid) + k2 * NT, ja = i >> 4, n = i & 15;   // from the last wave down: waves 0-3 carry D/G
        auxp[k2] = a.pk.invl; auxs[k2] = 0; auxl[k2] = 0.0;            // (no row: a valid dummy address, factor 0)
        if (i >= 0 && i < 16 * naux) {
            const int b = min(c0 + n, N - 1) / S;
            if (ja < a.dim_u) { auxp[k2] = a.u + int64_t(b) * T * a.dim_u + ja; auxs[k2] = a.dim_u; }
            else { auxp[k2] = a.y + int64_t(b) * T * a.dim_y + (ja - a.dim_u); auxs[k2] = a.dim_y; }
            auxl[k2] = a.pk.invl[Do + ja];
        }
    }
    // (tm = t mod 2R is carried along the steps: a runtime modulo per step costs ~30 VALU instructions a wave)
    auto load_inputs = [&](int t, int tm, double (&hv)[QPW], double (&av)[AUXR]) {
        bool rs = false;
        if (MODE == MODE_BWD) rs = (tm + 1 + run * R == P);           // (t + 1 + run R) mod 2R == 0   cbfssm.py:124,127
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * (w + qi * W) + g;
            hv[qi] = 0.0;
            if (act[qi]) {
                double v;
                if (MODE == MODE_FWD) v = a.x[(int64_t(t) * N + c) * a.dim_x + d];
                else if (rs) v = a.hid[(int64_t(run) * T + t) * N + c];
                else if (t == T - 1) v = 0.0;                                                  // cbfssm.py:106
                else v = a.h_all[((int64_t(run) * T + (t + 1)) * N + c) * Do + d];             // h_t = out_{t+1}
                hv[qi] = v;
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) av[k2] = auxp[k2][int64_t(t) * auxs[k2]];   // raw: scaled in store_inputs (a multiply
                                                                                        // here is a vmcnt(0) wait at the step top)
    };
    auto store_inputs = [&](double* xb, const double (&hv)[QPW], const double (&av)[AUXR]) {
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * (w + qi * W) + g;
            if (act[qi]) xb[d * PD + nl] = hv[qi] * il[qi];
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = (NT - 1 - tid) + k2 * NT;
            if (i < 16 * naux) xb[(Do + (i >> 4)) * PD + (i & 15)] = av[k2] * auxl[k2];
        }
    };
    // phase D of step t: adjoint of the step epilogue from the carried state adjoint -> Fm, Fv tiles, gdir
    // inputs of phase D of step t: {eps, y~ (fwd) or the y2 adjoint (bwd), fmean, fvar}
    auto epilogue_load = [&](int t, double& eps_t, double (&ytil)[QPW], double (&fmv_m)[QPW], double (&fmv_v)[QPW]) {
        double (&gy2in)[QPW] = ytil;
        if (MODE == MODE_FWD) {
            eps_t = a.eps[int64_t(t) * N + c];
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                ytil[qi] = 0.0;
                if (act[qi]) {
                    if (d < a.dim_y) ytil[qi] = a.y[(int64_t(bq) * T + (t + 1)) * a.dim_y + d];
                    else if (!a.half) ytil[qi] = a.y2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)];
                }
            }
        } else {
            eps_t = a.eps[(int64_t(run) * T + t) * N + c];
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                // (a branch, not a select: a select on the loaded value is an s_waitcnt right behind the load)
                gy2in[qi] = 0.0;
                if (act[qi]) gy2in[qi] = a.gy2[(int64_t(t) * N + c) * Do + d];
            }
        }
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int d = 4 * (w + qi * W) + g;
            fmv_m[qi] = 0.0; fmv_v[qi] = 1.0;
            if (act[qi]) {
                const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                const double* o = a.fmv + ((slot * N + c) * Do + d) * 2;
                fmv_m[qi] = o[0]; fmv_v[qi] = o[1];
            }
        }
    };
    auto epilogue_adjoint = [&](int t, int tm, const double eps_t, const double (&ytil)[QPW], const double (&fmv_m)[QPW],
                                const double (&fmv_v)[QPW]) {
        const double (&gy2in)[QPW] = ytil;
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = w + qi * W;
            if (q < 4) {
                const int d = 4 * q + g;
                double gfm = 0.0, gfv = 0.0;
                if (act[qi] && cvalid) {
                    const double fmean = fmv_m[qi];        // saved by the forward evaluation (no recompute of P1/P2)
                    const double fvar = fmv_v[qi];
                    const double gout = gcar[qi];
                    if (MODE == MODE_FWD) {
                        const bool do_cond = (a.condition || (t < R - 1)) && !(a.half && d >= a.dim_y);   // cbfssm.py:227
                        if (do_cond) {
                            const double kf1 = a.k_factor - 1.0;
                            const double vyt = vy[qi] + kf1 * fvar;
                            const double s = vyt + fvar;
                            const double rs = fast_rcp(s);
                            const double k = fvar * rs;
                            const double ydiff = ytil[qi] - fmean;
                            const double mu = fmean + k * ydiff;
                            const double omk = 1.0 - k;
                            const double sig = omk * omk * fvar + k * k * vyt;
                            const double rf = fast_rcp(fvar), rsig = fast_rcp(sig);
                            const double dm = mu - fmean;
                            // x' = mu + eps sqrt(sig);  kl = .5[log fvar - log sig + (sig + dm^2)/fvar - 1]
                            const double gmu = gout + a.cL * dm * rf;
                            const double gsg = gout * eps_t * 0.5 * fast_rsqrt(sig) + a.cL * 0.5 * (rf - rsig);
                            gfm = -a.cL * dm * rf;
                            gfv = a.cL * 0.5 * (rf - (sig + dm * dm) * rf * rf);
                            // mu = fmean + k (ytil - fmean)
                            gfm += gmu * omk;
                            double gk = gmu * ydiff;
                            const double gyt = gmu * k;
                            // sig = (1-k)^2 fvar + k^2 vyt
                            gk += gsg * (-2.0 * omk * fvar + 2.0 * k * vyt);
                            gfv += gsg * omk * omk;
                            double gvyt = gsg * k * k;
                            // k = fvar / s ; s = vyt + fvar ; vyt = vy + (kf-1) fvar
                            gfv += gk * rs;
                            const double gs = -gk * k * rs;
                            gvyt += gs;
                            gfv += gs;
                            gvy[qi] += gvyt;
                            gfv += kf1 * gvyt;
                            if (d >= a.dim_y && !a.half) a.gy2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)] = gyt;
                        } else {
                            // x' = fmean + eps sqrt(fvar), no KL term                           (cbfssm.py:224,234)
                            gfm = gout;
                            gfv = gout * eps_t * 0.5 * fast_rsqrt(fvar);
                            if (d >= a.dim_y && !a.half) a.gy2[(int64_t(t + 1) * N + c) * dob + (d - a.dim_y)] = 0.0;
                        }
                    } else {
                        // out = fmean + eps sqrt(fvar); entropy term on written steps          (cbfssm.py:150-156)
                        const bool write = (run == 0) ? (tm < R) : (tm >= R);
                        const double gtot = gout + (write ? gy2in[qi] : 0.0);
                        gfm = gtot;
                        gfv = gtot * eps_t * 0.5 * fast_rsqrt(fvar) - (write ? a.cE * 0.5 * fast_rcp(fvar) : 0.0);
                    }
                    gvx[qi] += gfv;
                    gsig += gfv;
                }
                gdir[qi] = gfm;
                if (d < 16) {
                    Fm[d * PD + nl] = gfm;
                    Fv[d * PD + nl] = gfv;
                }
            }
        }
    };

    if constexpr (XW) {
        if (w == W) {
            d4 xacc[NBLK][XCB];
#pragma unroll
            for (int rb = 0; rb < NBLK; ++rb)
#pragma unroll
                for (int cq = 0; cq < XCB; ++cq) xacc[rb][cq] = d4{0, 0, 0, 0};
            __syncthreads();                                         // (the barrier in front of the step loop)
            for (int step = 0; step < nsteps; ++step) {
                __syncthreads();                                     // 1: kernel tile complete (and intact until barrier 6)
                double kT[XCB][4];
#pragma unroll
                for (int cq = 0; cq < XCB; ++cq)
#pragma unroll
                    for (int s = 0; s < 4; ++s) kT[cq][s] = Kt[(16 * (NCB + cq) + nl) * PD + 4 * s + g];
                // the first row blocks while the row-block waves are still in phase E (their A2bar rows are flagged) ...
#pragma unroll
                for (int rb = 0; rb < NBLK; ++rb) {
                    if (rb < REV_XEARLY) {
                        flag_wait(xflag, rb, step + 1);
                        double abT[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) abT[s] = A2t[(16 * rb + nl) * PD + 4 * s + g];
#pragma unroll
                        for (int cq = 0; cq < XCB; ++cq)
#pragma unroll
                            for (int s = 0; s < 4; ++s) xacc[rb][cq] = CBF_MFMA(abT[s], kT[cq][s], xacc[rb][cq]);
                    }
                }
                __syncthreads();                                     // 4: every A2bar row is written
                // ... some inside phase F, the last REV_XLATE row blocks behind barrier 5 (the A2bar tile stays intact until
                // phase E of the next step): phases G / D are mostly vector latency on the first four waves, this wave's
                // SIMD partner (wave 3) carries little or none of it.  The accumulators are pinned in front of each
                // barrier: left alone, the compiler sinks most of these MFMAs (they touch registers only) below BOTH
                // barriers -- 34 of them ran after barrier 6, where the seven row-block waves had nothing but a copy
                // to do and waited at the next barrier for this wave (phase B: 12 % of a step, measured per wave).
                auto xblock = [&](int rb) {
                    double abT[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) abT[s] = A2t[(16 * rb + nl) * PD + 4 * s + g];
#pragma unroll
                    for (int cq = 0; cq < XCB; ++cq)
#pragma unroll
                        for (int s = 0; s < 4; ++s) xacc[rb][cq] = CBF_MFMA(abT[s], kT[cq][s], xacc[rb][cq]);
                };
                auto xpin = [&](int rb) {
#pragma unroll
                    for (int cq = 0; cq < XCB; ++cq) asm volatile("" : "+v"(xacc[rb][cq]));
                };
#pragma unroll
                for (int rb = 0; rb < NBLK; ++rb)
                    if (rb >= REV_XEARLY && rb < NBLK - REV_XLATE) xblock(rb);
#pragma unroll
                for (int rb = 0; rb < NBLK; ++rb)
                    if (rb >= REV_XEARLY && rb < NBLK - REV_XLATE) xpin(rb);
                __syncthreads();                                     // 5
#pragma unroll
                for (int rb = 0; rb < NBLK; ++rb)
                    if (rb >= REV_XEARLY && rb >= NBLK - REV_XLATE) xblock(rb);
#pragma unroll
                for (int rb = 0; rb < NBLK; ++rb)
                    if (rb >= REV_XEARLY && rb >= NBLK - REV_XLATE) xpin(rb);
                __syncthreads();                                     // 6
            }
            double* slab = a.gpart + wg_linear * a.slab;
#pragma unroll
            for (int rb = 0; rb < NBLK; ++rb)
#pragma unroll
                for (int cq = 0; cq < XCB; ++cq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[SL::gB + (rb * NBLK + NCB + cq) * 256 + r * 64 + l] = xacc[rb][cq][r];
            return;
        }
    }
    if constexpr (SW) {
        if (w == W) {
            __syncthreads();                                         // (the barrier in front of the step loop)
            for (int step = 0; step < nsteps; ++step) {
                const int64_t slot = wg_linear * a.chunk_steps + step;
                double* pk = a.stash_k + slot * NBLK * 256 + l;
                double* pa = a.stash_a + slot * NBLK * 256 + l;
                __syncthreads();                                     // 1: kernel tile complete (intact until phase F)
#pragma unroll 2
                for (int rb = 0; rb < NBLK; ++rb) {
                    double v[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = Kt[(16 * rb + nl) * PD + 4 * s + g];      // B[k = chain][col m]
#pragma unroll
                    for (int s = 0; s < 4; ++s) pk[rb * 256 + s * 64] = v[s];
                }
                __syncthreads();                                     // 4: every A2bar row is written (intact until the
                                                                     //    next step's phase E, behind its barrier 1)
#pragma unroll 2
                for (int rb = 0; rb < NBLK; ++rb) {
                    double v[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = A2t[(16 * rb + nl) * PD + 4 * s + g];     // A[row m][k = chain]
#pragma unroll
                    for (int s = 0; s < 4; ++s) pa[rb * 256 + s * 64] = v[s];
                }
                if constexpr (PALIAS) __syncthreads();               // (the row-block waves: partial tiles over the K tile)
                __syncthreads();                                     // 5
                __syncthreads();                                     // 6
            }
            return;
        }
    }
    CBF_STAMP_DECL;
    double hcur[QPW];
    if (nsteps > 0) {
        double av[AUXR];
        load_inputs(t_of(0), tmod, hcur, av);
        store_inputs(xq, hcur, av);
        double e0, y0[QPW], m0[QPW], v0[QPW];
        epilogue_load(t_of(0), e0, y0, m0, v0);
        epilogue_adjoint(t_of(0), tmod, e0, y0, m0, v0);
    }
    // KSV: the saved [A2 | K] rows of this wave for the NEXT step, issued behind barrier 5 of the current one
    const int64_t G16 = (N + 15) >> 4;
    const int TS = saved_tile_stride(NBLK, a.ksave);
    d4 a2n[KSV ? RB : 1], kn[KSV ? RB : 1];
    auto load_saved = [&](int t) {
        if constexpr (KSV) {
            const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
            const double* ap = a.a2s + (slot * G16 + (c0 >> 4)) * TS + l;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a2n[i][r] = ap[rbs[i] * 256 + r * 64];
                    kn[i][r] = ap[NBLK * 256 + rbs[i] * 256 + r * 64];
                }
            }
        }
    };
    if (nsteps > 0) load_saved(t_of(0));
    // mean / variance B operands of phase E (eight-wave tiles).  Without the kernel-tile build the step has no phase whose
    // arithmetic would hide their latency, and the Z~ rows it no longer holds leave the registers: KSV keeps them for the pass.
    constexpr bool MUPRE = XW && RB == 1;
    double mBv[MUPRE ? 4 : 1], sBv[MUPRE ? 4 : 1];
    auto load_mu = [&]() {
        if constexpr (MUPRE) {
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                mBv[s] = a.rk.muB[rbs[0] * 256 + s * 64 + l];
                sBv[s] = a.rk.s2B[rbs[0] * 256 + s * 64 + l];
            }
        }
    };
    // (two k-steps = 8 registers: the backward runs of the Sarcos class; four would spill in the forward-pass adjoint)
    constexpr bool MURES = KSV && MUPRE && KD == 2;
    if constexpr (MURES) {
        load_mu();
        // landed before the loop: otherwise the compiler has to assume them in flight at the loop header and its first
        // wait inside the loop also waits, in every iteration, for whatever the step has issued by then
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    }
    __syncthreads();
    CBF_STAMP_START();
#ifdef CBF_REV_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    CBF_STAMP_MARK0();
    for (int step = 0; step < nsteps; ++step) {
        CBF_STAMP_MARK(11);
        const int t = t_of(step);
        const bool has_next = (step + 1 < nsteps);
        const int tn = has_next ? t_of(step + 1) : t;
        double* xq = xq0 + (step & 1) * (4 * DK * PD);          // this step's scaled inputs
        double* xqn = xq0 + ((step + 1) & 1) * (4 * DK * PD);   // filled for the next step during this one
        bool resample_t = false;
        const int tmn = (tmod + 1 == P) ? 0 : tmod + 1;             // (t + 1) mod 2R: the backward-run adjoint walks t upwards
        if (MODE == MODE_BWD) resample_t = (tmod + 1 + run * R == P);                         // cbfssm.py:124,127

        // ---- B with kept kernel tiles (KSV): the rows loaded a step ahead go to the LDS tile -- FIRST thing in the step:
        // vmcnt retires in order, so anything issued before this wait (the next step's inputs below) would be waited for too
        d4 kreg[RB];
        CBF_STAMP_MARK0();
        if constexpr (KSV) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                kreg[i] = kn[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) Kt[(16 * rbs[i] + 4 * r + g) * PD + nl] = kreg[i][r];
            }
        }
        CBF_STAMP_MARK(9);
        // next step's inputs: issued now, written to LDS at the end of phase E
        double hnext[QPW], auxn[AUXR];
        if (has_next) load_inputs(tn, tmn, hnext, auxn);
        CBF_STAMP_MARK(10);
        // A2 rows of this wave, if the forward evaluation kept them (consumed after the second barrier)
        d4 a2[RB];
        if constexpr (KSV) {
#pragma unroll
            for (int i = 0; i < RB; ++i) a2[i] = a2n[i];
        } else if (a.a2s) {
            const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
            const double* ap = a.a2s + (slot * G16 + (c0 >> 4)) * TS + l;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                a2[i] = d4{0, 0, 0, 0};
                if (ok[i]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a2[i][r] = ap[rbs[i] * 256 + r * 64];
                }
            }
        }

        // mean / variance B operands of phase E: issued here, under phase B
        // (measured: forward-pass adjoint 4.47 -> 4.32 ms)
        if constexpr (!MURES) load_mu();

        // ---- B: kernel tile (rows of this wave)
        if constexpr (!KSV) {
        double bx[DK], xx = 0.0;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            bx[s] = xq[(4 * s + g) * PD + nl];
            xx = fma(bx[s], bx[s], xx);
        }
        xx += __shfl_xor(xx, 16);
        xx += __shfl_xor(xx, 32);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            kreg[i] = d4{0, 0, 0, 0};
            if (ok[i]) {
                d4 e;
                if constexpr (STASH) {
                    // (stash mode runs two row blocks per wave at the VGPR cap: the Z~ operand and the row constants
                    //  are re-read from their L1-resident images here instead of living in 40 registers for the pass)
                    const double* czs = ZLDS ? czl : a.pk.cz;
                    const double* Zs = ZLDS ? Zl : a.pk.Zp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = czs[16 * rbs[i] + 4 * r + g] - 0.5 * xx;
#pragma unroll
                    for (int s = 0; s < DK; ++s) e = CBF_MFMA(Zs[(rbs[i] * DK + s) * 64 + l], bx[s], e);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = tile.czr[i][r] - 0.5 * xx;
#pragma unroll
                    for (int s = 0; s < DK; ++s) e = CBF_MFMA(tile.Zreg[i][s], bx[s], e);
                }
                kreg[i] = tile_exp4(e);
#pragma unroll
                for (int r = 0; r < 4; ++r) Kt[(16 * rbs[i] + 4 * r + g) * PD + nl] = kreg[i][r];
            }
        }
        }
        CBF_STAMP_MARK(2);
        CBF_STAMP_BARRIER(1);

        // ---- C: A2 rows of this wave (only when the forward evaluation did not keep them)
        CBF_STAMP_MARK0();
        if (!a.a2s) {
            d4 acc[RB][2];
#pragma unroll
            for (int i = 0; i < RB; ++i) { acc[i][0] = d4{0, 0, 0, 0}; acc[i][1] = d4{0, 0, 0, 0}; }
            if constexpr (BLDS) {
                int s = 0;
#pragma unroll 2
                for (; s + 1 < KSr; s += 2) {
                    const double b0 = Kt[(4 * s + g) * PD + nl], b1 = Kt[(4 * s + 4 + g) * PD + nl];
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        if (ok[i]) {
                            acc[i][0] = CBF_MFMA(bop[i][s * 64], b0, acc[i][0]);
                            acc[i][1] = CBF_MFMA(bop[i][(s + 1) * 64], b1, acc[i][1]);
                        }
                    }
                }
                if (s < KSr) {
                    const double b0 = Kt[(4 * s + g) * PD + nl];
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (ok[i]) acc[i][0] = CBF_MFMA(bop[i][s * 64], b0, acc[i][0]);
                }
            } else if constexpr (RB == 2 && NBLK >= 16) {
                // (measured: -6 % on the C5 step at NBLK = 20; at NBLK = 13 the 48 fixed registers cost more in spills than
                // the loop gains)
                // K^-1 streams from L2 through the hand-scheduled loop (cbfssm_kernels.hpp); the image is zero-padded to
                // KS = 4 NBLK k-steps, the tile rows beyond M are finite, a non-existent second row block is dropped
                stream_kinv_rb2<4 * PD * 8>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], bop[0], bop[1],
                                            lds_addr(Kt + g * PD + nl), (KSr + 3) >> 2);
            } else {
                // K^-1 streams from L2: issue the operand loads of four k-steps together, ahead of their MFMAs
#pragma unroll 1
                for (int s0 = 0; s0 < KSr; s0 += 4) {
                    double b[4], aop[RB][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < RB; ++i) aop[i][j] = bop[i][(s0 + j) * 64];
                        b[j] = Kt[(4 * (s0 + j) + g) * PD + nl];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (ok[i]) acc[i][j & 1] = CBF_MFMA(aop[i][j], b[j], acc[i][j & 1]);
                }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) a2[i] = acc[i][0] + acc[i][1];
        }
        CBF_STAMP_MARK(0);
        // (no P1/P2 here: fmean / fvar come from the forward evaluation; phase E follows without a barrier, it needs
        //  only this wave's own A2 rows and the Fm/Fv tiles written two barriers ago)

        // ---- E: A2bar, and the parameter adjoints that contract over the 16 chains
        CBF_STAMP_MARK0();
        double fvsum = 0.0;
        double fmB[4], fvB[4];
#pragma unroll
        for (int s = 0; s < KD; ++s) {                  // (rows d >= 4 KD of the Fm / Fv tiles are zero)
            fmB[s] = Fm[(4 * s + g) * PD + nl];
            fvB[s] = Fv[(4 * s + g) * PD + nl];
            fvsum += fvB[s];
        }
        fvsum += __shfl_xor(fvsum, 16);
        fvsum += __shfl_xor(fvsum, 32);
        double fmT[4], fvT[4];      // the same tiles with the chain index as k: [n = 4s+g][col = nl]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fmT[s] = Fm[nl * PD + 4 * s + g];
            fvT[s] = Fv[nl * PD + 4 * s + g];
        }
        d4 a2bar[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            a2bar[i] = d4{0, 0, 0, 0};
            if (ok[i]) {
                const double* mBp = (MULDS ? mul : a.rk.muB) + rbs[i] * 256 + l;
                const double* sBp = a.rk.s2B + rbs[i] * 256 + l;
                d4 T1 = {0, 0, 0, 0}, T2 = {0, 0, 0, 0};
                if constexpr (MUPRE) {
#pragma unroll
                    for (int s = 0; s < KD; ++s) {
                        T1 = CBF_MFMA(mBv[s], fmB[s], T1);
                        T2 = CBF_MFMA(sBv[s], fvB[s], T2);
                    }
                } else {
                    // the operand rows of this row block: ALL issued, then one wait (pinned) -- left at their MFMAs the
                    // compiler emits load, s_waitcnt vmcnt(0), MFMA eight times over: eight L1 / L2 round trips in a row
                    double mv[KD], sv[KD];
#pragma unroll
                    for (int s = 0; s < KD; ++s) { mv[s] = mBp[s * 64]; sv[s] = sBp[s * 64]; }
#pragma unroll
                    for (int s = 0; s < KD; ++s) asm volatile("" : "+v"(mv[s]), "+v"(sv[s]));
#pragma unroll
                    for (int s = 0; s < KD; ++s) {
                        T1 = CBF_MFMA(mv[s], fmB[s], T1);
                        T2 = CBF_MFMA(sv[s], fvB[s], T2);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) a2bar[i][r] = T1[r] + 2.0 * a2[i][r] * T2[r] - kreg[i][r] * fvsum;
                // 16x16 transposes through this wave's own rows of the A2bar tile (nobody else reads them before the
                // next barrier): C-layout (row g+4r, col nl) -> A-operand layout (row nl, k = 4s+g)
                double a2T[4], abT[4];
                double* own = A2t + 16 * rbs[i] * PD;
#pragma unroll
                for (int r = 0; r < 4; ++r) own[(g + 4 * r) * PD + nl] = a2[i][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < 4; ++s) a2T[s] = own[nl * PD + 4 * s + g];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) own[(g + 4 * r) * PD + nl] = a2bar[i][r];   // stays: A2bar tile of phase F
                if constexpr (XW) flag_release(xflag + rbs[i], step + 1);               // the extra wave may take this block
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    gMu[i] = CBF_MFMA(a2T[s], fmT[s], gMu[i]);                  // mubar[m][d] += A2[m][n] Fm[d][n]
                    gS2[i] = CBF_MFMA(a2T[s] * a2T[s], fvT[s], gS2[i]);         // s2bar[m][d] += A2[m][n]^2 Fv[d][n]
                }
                if constexpr (STASH && SW) {
                    // (the stash-writer wave copies both operand images of this step from the LDS tiles)
                } else if constexpr (STASH) {
                    // A2bar^T and K^T of this row block as the MFMA operand images of Kinvbar += A2bar K^T (exactly
                    // what the in-register variant below feeds its MFMAs): slot = (workgroup, step), per slot and
                    // row block 4 x 64 doubles each; cbfssm_stash_contract_f64 contracts them after the launch
                    __builtin_amdgcn_wave_barrier();
                    const int64_t slot = wg_linear * a.chunk_steps + step;
                    double* pa = a.stash_a + (slot * NBLK + rbs[i]) * 256 + l;
                    double* pk = a.stash_k + (slot * NBLK + rbs[i]) * 256 + l;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        pa[s * 64] = own[nl * PD + 4 * s + g];                               // A[row m][k = chain]
                        pk[s * 64] = Kt[(16 * rbs[i] + nl) * PD + 4 * s + g];                // B[k = chain][col m]
                    }
                } else {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int s = 0; s < 4; ++s) abT[s] = own[nl * PD + 4 * s + g];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const double kT = Kt[(16 * cb + nl) * PD + 4 * s + g];
                            gB[i][cb] = CBF_MFMA(abT[s], kT, gB[i][cb]);        // Kinvbar[m'][m] += A2bar[m'][n] K[m][n]
                        }
                    }
                }
            }
        }
        CBF_STAMP_MARK(5);
        // next step's inputs: loaded at the top of the step, their latency is behind phases B and E by now
        // (xqn was last read in phase G of the previous step, it is next read after the barrier that ends this step)
        if (has_next) store_inputs(xqn, hnext, auxn);
        CBF_STAMP_BARRIER(4);

        // ---- F: Kbar, Ebar, input adjoint partials, Zbar~
        CBF_STAMP_MARK0();
        double eps_n, ytil_n[QPW], fm_n[QPW], fv_n[QPW];     // phase D inputs of the next step, consumed after phase G
        if (has_next) epilogue_load(tn, eps_n, ytil_n, fm_n, fv_n);
        // y_t of this lane for the log-likelihood term of phase G: issued here, not on the carry chain
        double ycur[QPW];
        if (MODE == MODE_FWD) {
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                ycur[qi] = (act[qi] && d < a.dim_y) ? a.y[(int64_t(bq) * T + t) * a.dim_y + d] : 0.0;
            }
        }
        // (Z~)^T operands of the input-adjoint product below: issued here, in flight under the K^-1 A2bar loop (with the
        // eighth wave the row-block waves have the registers for it; the product was 7 % of a step waiting for L2)
        constexpr bool ZTPRE = XW && RB == 1;
        double ztv[ZTPRE ? JB : 1][4];
        if constexpr (ZTPRE) {
            const double* ZTp0 = a.rk.ZT + rbs[0] * JB * 256 + l;
#pragma unroll
            for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ztv[jb][r] = ZTp0[(jb * 4 + r) * 64];
        }
        d4 ebar[RB];
        {
            d4 acc[RB][2];
#pragma unroll
            for (int i = 0; i < RB; ++i) { acc[i][0] = d4{0, 0, 0, 0}; acc[i][1] = d4{0, 0, 0, 0}; }
            if constexpr (BLDS && RB == 1) {
                // Both operands come from LDS.  Left to the compiler the loop is read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs
                // per iteration (half rate, measured): its wait insertion drains the counter on every back edge, also
                // for a hand-unrolled two-set source.  So the loop over whole groups of four k-steps is ONE asm
                // statement: two operand sets, the reads of the next pair in flight under the MFMAs of the current one,
                // counted waits.  (Scalar loads also count on lgkmcnt but only make a wait longer.)  The trailing
                // s_nops are what the compiler puts between an MFMA and a VALU read of its result.
                const int nquad = KSr >> 2;
                if (nquad > 0) {
                    uint32_t adrA = lds_addr(bop[0]);               // + 512 bytes per k-step
                    uint32_t adrB = lds_addr(A2t + g * PD + nl);    // + 4 * 17 * 8 = 544 bytes per k-step
                    int cnt = nquad;
                    double a0, a1, a2, a3, b0, b1, b2, b3;
                    asm volatile(
                        "ds_read_b64 %[a0], %[pa]\n\t"
                        "ds_read_b64 %[b0], %[pb]\n\t"
                        "ds_read_b64 %[a1], %[pa] offset:512\n\t"
                        "ds_read_b64 %[b1], %[pb] offset:544\n"
                        "1:\n\t"
                        "ds_read_b64 %[a2], %[pa] offset:1024\n\t"
                        "ds_read_b64 %[b2], %[pb] offset:1088\n\t"
                        "ds_read_b64 %[a3], %[pa] offset:1536\n\t"
                        "ds_read_b64 %[b3], %[pb] offset:1632\n\t"
                        "s_waitcnt lgkmcnt(4)\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c0], %[a0], %[b0], %[c0]\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c1], %[a1], %[b1], %[c1]\n\t"
                        "s_sub_u32 %[n], %[n], 1\n\t"
                        "v_add_u32 %[pa], 0x800, %[pa]\n\t"
                        "v_add_u32 %[pb], 0x880, %[pb]\n\t"
                        "s_cmp_eq_u32 %[n], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "ds_read_b64 %[a0], %[pa]\n\t"
                        "ds_read_b64 %[b0], %[pb]\n\t"
                        "ds_read_b64 %[a1], %[pa] offset:512\n\t"
                        "ds_read_b64 %[b1], %[pb] offset:544\n\t"
                        "s_waitcnt lgkmcnt(4)\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c0], %[a2], %[b2], %[c0]\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c1], %[a3], %[b3], %[c1]\n\t"
                        "s_branch 1b\n"
                        "2:\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c0], %[a2], %[b2], %[c0]\n\t"
                        "v_mfma_f64_16x16x4_f64 %[c1], %[a3], %[b3], %[c1]\n\t"
                        "s_nop 15\n\t"
                        "s_nop 2"
                        : [c0] "+v"(acc[0][0]), [c1] "+v"(acc[0][1]), [pa] "+v"(adrA), [pb] "+v"(adrB), [n] "+s"(cnt),
                          [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
                          [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
                        :
                        : "scc", "memory");
                }
                {   // the 0..3 k-steps left (static accumulator indices: a runtime one makes the compiler index registers)
                    const int s0 = 4 * nquad, rem = KSr - s0;
                    if (rem > 0) acc[0][0] = CBF_MFMA(bop[0][s0 * 64], A2t[(4 * s0 + g) * PD + nl], acc[0][0]);
                    if (rem > 1) acc[0][1] = CBF_MFMA(bop[0][(s0 + 1) * 64], A2t[(4 * s0 + 4 + g) * PD + nl], acc[0][1]);
                    if (rem > 2) acc[0][0] = CBF_MFMA(bop[0][(s0 + 2) * 64], A2t[(4 * s0 + 8 + g) * PD + nl], acc[0][0]);
                }
            } else if constexpr (BLDS) {
                int s = 0;
#pragma unroll 2
                for (; s + 1 < KSr; s += 2) {
                    const double b0 = A2t[(4 * s + g) * PD + nl], b1 = A2t[(4 * s + 4 + g) * PD + nl];
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        if (ok[i]) {
                            acc[i][0] = CBF_MFMA(bop[i][s * 64], b0, acc[i][0]);
                            acc[i][1] = CBF_MFMA(bop[i][(s + 1) * 64], b1, acc[i][1]);
                        }
                    }
                }
                if (s < KSr) {
                    const double b0 = A2t[(4 * s + g) * PD + nl];
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (ok[i]) acc[i][0] = CBF_MFMA(bop[i][s * 64], b0, acc[i][0]);
                }
            } else if constexpr (RB == 2 && NBLK >= 13) {
                // (measured: -6 % on the C5 step at NBLK = 20, -2 % on the C4 step at NBLK = 13: the loop itself runs 25-30 %
                // faster, the 48 fixed registers give some of it back as spills in the other phases)
                // K^-1 streams from L2 through the hand-scheduled loop (cbfssm_kernels.hpp); the image is zero-padded to
                // KS = 4 NBLK k-steps, the tile rows beyond M are finite, a non-existent second row block is dropped
                stream_kinv_rb2<4 * PD * 8>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], bop[0], bop[1],
                                            lds_addr(A2t + g * PD + nl), (KSr + 3) >> 2);
            } else {
                // K^-1 streams from L2: issue the operand loads of four k-steps together, ahead of their MFMAs
#pragma unroll 1
                for (int s0 = 0; s0 < KSr; s0 += 4) {
                    double b[4], aop[RB][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int i = 0; i < RB; ++i) aop[i][j] = bop[i][(s0 + j) * 64];
                        b[j] = A2t[(4 * (s0 + j) + g) * PD + nl];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (ok[i]) acc[i][j & 1] = CBF_MFMA(aop[i][j], b[j], acc[i][j & 1]);
                }
            }
            if constexpr (STASH) {
                // the kernel tile and (when the forward evaluation kept it) A2 are re-read here instead of staying in
                // 32 registers across phases E and F: K from this wave's own rows of the LDS tile, A2 from L2
                const bool a2_saved = (a.a2s != nullptr);
                const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                const double* ap = a2_saved ? a.a2s + (slot * G16 + (c0 >> 4)) * TS + l : nullptr;
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double kv = Kt[(16 * rbs[i] + 4 * r + g) * PD + nl];
                        const double av = a2_saved ? ap[rbs[i] * 256 + r * 64] : a2[i][r];
                        ebar[i][r] = (acc[i][0][r] + acc[i][1][r] - av * fvsum) * kv;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ebar[i][r] = (acc[i][0][r] + acc[i][1][r] - a2[i][r] * fvsum) * kreg[i][r];
            }
        }
        CBF_STAMP_MARK(6);
        d4 xp[JB];
        {
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) xp[jb] = d4{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (ok[i]) {
                    const double* ZTp = (ZTLDS ? ZTl : a.rk.ZT) + rbs[i] * JB * 256 + l;
#pragma unroll
                    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            xp[jb] = CBF_MFMA(ZTPRE ? ztv[jb][r] : ZTp[(jb * 4 + r) * 64], ebar[i][r], xp[jb]);   // rows j, k = m of this block
                }
            }
            if constexpr (!PALIAS && !XW) {
#pragma unroll
                for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[w * PSL + (jb * 4 + r) * 64 + l] = xp[jb][r];
            }
        }
        CBF_STAMP_MARK(7);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (ok[i]) {
                double ebT[4];
                // the K tile is dead after phase E: reuse own rows for Ebar^T -- unless the extra wave is still reading
                // it: then this wave's (not yet written) slot of the partial tiles is the scratch
                double* ownk = XW ? (part + w * PSL) : (Kt + 16 * rbs[i] * PD);
#pragma unroll
                for (int r = 0; r < 4; ++r) ownk[(g + 4 * r) * PD + nl] = ebar[i][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < 4; ++s) ebT[s] = ownk[nl * PD + 4 * s + g];
#pragma unroll
                for (int jb = 0; jb < JB; ++jb) {
                    const int j = 16 * jb + nl;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        double xT = (j < 4 * DK) ? xq[j * PD + 4 * s + g] : 0.0;
                        if (j == D) xT = 1.0;                                         // ones column: row sums of Ebar
                        gZ[i][jb] = CBF_MFMA(ebT[s], xT, gZ[i][jb]);                  // Zbar~[m][j] += Ebar[m][n] x~[j][n]
                    }
                }
            }
        }
        CBF_STAMP_MARK(8);
        if constexpr (XW) {
            __builtin_amdgcn_wave_barrier();     // the transposes above went through this slot
#pragma unroll
            for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[w * PSL + (jb * 4 + r) * 64 + l] = xp[jb][r];
        }
        if constexpr (PALIAS) {
            __syncthreads();                 // every wave is done with its rows of the K tile (Ebar, the transposes)
#pragma unroll
            for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[w * PSL + (jb * 4 + r) * 64 + l] = xp[jb][r];
        }
        CBF_STAMP_BARRIER(5);
        if (has_next) load_saved(tn);      // (KSV) a2 / kreg were last read in phase F

        // ---- G: input adjoint, carried to the next reverse step
        double esum = 0.0;   // colsum of Ebar for this lane's chain = row D of the xbar tile
        {
            const int jbD = D >> 4, qD = (D >> 2) & 3, gD = D & 3;
#pragma unroll
            for (int ww = 0; ww < W; ++ww) esum += part[ww * PSL + (jbD * 4 + qD) * 64 + gD * 16 + nl];
        }
#pragma unroll
        for (int k2 = 0; k2 < GPW; ++k2) {
            // (with four or more waves the later rounds start at the last wave: waves 0-3 also carry phase D; with fewer,
            //  the groups gi < 4 of later rounds must stay on the wave that owns their phase-D lanes)
            const int gi = (k2 == 0 || W < 4) ? (w + k2 * W) : (k2 * W + (W - 1 - w));
            if (gi < NG) {
                const int jb = gi >> 2, q = gi & 3;
                const int j = 16 * jb + 4 * q + g;
                double xb = 0.0;
#pragma unroll
                for (int ww = 0; ww < W; ++ww) xb += part[ww * PSL + (jb * 4 + q) * 64 + l];
                if (j < D && cvalid) {
                    const double xt = xq[j * PD + nl];
                    xb -= xt * esum;
                    glx[k2] += xb * xt;                                            // lengthscale adjoint (inputs)
                }
                if (j == D && cvalid) glogsig += xb;
                // state rows hand their adjoint to the phase-D lanes of the same (d, chain): identical lanes when
                // jb == 0 and this wave owns group q in both phases (gi = q for gi < 4)
                if (jb == 0) {
#pragma unroll
                    for (int qi = 0; qi < QPW; ++qi) {
                        if (w + qi * W == q) {
                            double gin = 0.0;
                            if (act[qi] && cvalid) gin = gdir[qi] + xb * il[qi];
                            if (MODE == MODE_FWD) {
                                // gin = d loss/d x_t ; add the log-likelihood's own term for x_t (t >= 1)
                                const int d = 4 * q + g;
                                if (act[qi] && cvalid) {
                                    if ((t >= 1 || a.half) && d < a.dim_y) {
                                        gin += -a.cL * (ycur[qi] - hcur[qi]) * ivy[qi];
                                    }
                                    if (t == 0) {
                                        if (a.half) a.gx0[int64_t(c) * a.dim_x + d] = gin;       // x_0 = recognition model
                                        else if (d >= a.dim_y) a.gy2[int64_t(c) * dob + (d - a.dim_y)] = gin;   // x_0 = y_tilde_0
                                    }
                                }
                                gcar[qi] = gin;
                            } else {
                                gcar[qi] = resample_t ? 0.0 : gin;                 // h_t = out_{t+1} unless resampled
                            }
                        }
                    }
                }
            }
        }
        // ---- D of the next step: same lanes as the carried adjoint just produced (Fm/Fv were last read in phase E)
        CBF_STAMP_MARK0();
        if (has_next) {
            epilogue_adjoint(tn, tmn, eps_n, ytil_n, fm_n, fv_n);
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) hcur[qi] = hnext[qi];
        }
        CBF_STAMP_MARK(3);
        tmod = tmn;
        CBF_STAMP_BARRIER(6);
        CBF_STAMP_MARK0();
    }

    if (MODE == MODE_FWD) {
        if (nsteps == 0 && a.t_hi < 0) {
            // T == 1: x_0 = y_tilde_0 only feeds the log-likelihood through its observed dims -> no gradient to y2
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                if (act[qi] && cvalid) {
                    if (a.half) {
                        double gv = 0.0;
                        if (d < a.dim_y) gv = -a.cL * (a.y[(int64_t(bq) * T) * a.dim_y + d] - a.x[int64_t(c) * a.dim_x + d]) / vy[qi];
                        a.gx0[int64_t(c) * a.dim_x + d] = gv;
                    } else if (d >= a.dim_y) {
                        a.gy2[int64_t(c) * dob + (d - a.dim_y)] = 0.0;
                    }
                }
            }
        }
        if (a.gx_carry && a.t_lo > 0 && nsteps > 0) {
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) {
                const int d = 4 * (w + qi * W) + g;
                if (act[qi] && cvalid) a.gx_carry[int64_t(c) * a.dim_x + d] = gcar[qi];
            }
        }
    }
    if constexpr (STASH) {
        // unused step slots of this workgroup's range must read as zero in the contraction
        for (int step = nsteps; step < a.chunk_steps; ++step) {
            const int64_t slot = wg_linear * a.chunk_steps + step;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (ok[i]) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        a.stash_a[(slot * NBLK + rbs[i]) * 256 + s * 64 + l] = 0.0;
                        a.stash_k[(slot * NBLK + rbs[i]) * 256 + s * 64 + l] = 0.0;
                    }
                }
            }
        }
    }

    // ---- write this workgroup's slab
    double* slab = a.gpart + wg_linear * a.slab;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if (ok[i]) {
            const int rb = rbs[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                slab[SL::gMu + rb * 256 + r * 64 + l] = gMu[i][r];
                slab[SL::gS2 + rb * 256 + r * 64 + l] = gS2[i][r];
                if constexpr (!STASH) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) slab[SL::gB + (rb * NBLK + cb) * 256 + r * 64 + l] = gB[i][cb][r];
                }
#pragma unroll
                for (int jb = 0; jb < JB; ++jb) slab[SL::gZ + (rb * JB + jb) * 256 + r * 64 + l] = gZ[i][jb][r];
            }
        }
    }
    // small per-dimension sums: reduce over the 16 chains of each 16-lane group, lane nl == 0 writes
    for (int i = tid; i < 192; i += NT) slab[SL::small + i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int q = w + qi * W;
        double v1 = gvx[qi], v2 = gvy[qi];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
        if (q < 4 && nl == 0) {
            slab[SL::small + 4 * q + g] = v1;
            slab[SL::small + 16 + 4 * q + g] = v2;
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < GPW; ++k2) {
        const int gi = (k2 == 0 || W < 4) ? (w + k2 * W) : (k2 * W + (W - 1 - w));
        double v = glx[k2];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (gi < NG && nl == 0) slab[SL::small + 32 + 4 * gi + g] = v;
    }
    const double s1 = block_sum(gsig, red, tid, NT);
    const double s2 = block_sum(glogsig, red, tid, NT);
    if (tid == 0) {
        slab[SL::small + 96] = s1;
        slab[SL::small + 97] = s2;
    }
#ifdef CBF_STAMP_ALLWAVES
    // diagnostic: the barrier waits of EVERY row-block wave (who is last at which barrier), 12 slots per wave from 100:
    // wait[0..6], then the sum of the compute shares (profiles/tools/rev_barrier_waits.py)
    if (l == 0 && w < 7) {
        double cs = 0.0;
        for (int i = 0; i < 7; ++i) {
            slab[SL::small + 100 + 12 * w + i] = double(st_w[i]);
            cs += double(st_c[i]);
        }
        slab[SL::small + 100 + 12 * w + 7] = cs;
    }
#elif defined(CBF_REV_STAMPS)
#ifndef CBF_STAMP_WAVE
#define CBF_STAMP_WAVE (W - 1)          // the second wave whose phase shares are recorded (-DCBF_STAMP_WAVE=k picks another)
#endif
    if (l == 0 && (w == 0 || w == CBF_STAMP_WAVE)) {
        const int o = (w == 0) ? 100 : 114;
        for (int i = 0; i < 7; ++i) {
            slab[SL::small + o + i] = double(st_c[i]);
            slab[SL::small + o + 7 + i] = double(st_w[i]);
        }
#ifdef CBF_STAMP_MARKS_LAST
        if (w == CBF_STAMP_WAVE) for (int i = 0; i < 12; ++i) slab[SL::small + 128 + i] = double(st_m[i]);   // sub-phase marks of that wave
#else
        if (w == 0) for (int i = 0; i < 12; ++i) slab[SL::small + 128 + i] = double(st_m[i]);
#endif
        if (w == 0) {
            // in-kernel clock: shader cycles per 100 MHz real-time tick (MI355X_MICROARCH.md, DVFS give-back item 6)
            slab[SL::small + 140] = double(__builtin_amdgcn_s_memtime() - clk0);
            slab[SL::small + 141] = double(__builtin_amdgcn_s_memrealtime() - rt0);
        }
    }
#endif
}

}  // namespace cbfssm
