// Stash-mode contraction  Kinvbar += sum over (workgroup, step) slots of  A2bar K^T   (cbfssm_adjoint.hpp, STASH).
//
// The adjoint kernels of the tall tiles (M > 112) cannot keep the M x M accumulator of d loss / d K^-1 in registers; they
// write, per step, the two MFMA operand images the in-register variant would have consumed (A2bar^T as A operand, K^T as
// B operand: [slot][NBLK][4][64] doubles each).  This kernel is that accumulation as a split-K product: a workgroup of up to
// eight waves owns up to eight 16-row blocks of the output (one per wave: NBLK accumulator tiles in VGPRs) and a
// contiguous range of slots; the B image of a slot is shared through LDS (double-buffered), the A image of the wave's
// row block comes straight from HBM in 512-byte wave loads.  Partial results leave as C-layout images, summed in a
// fixed order by a second kernel (no atomics: reproducible).
//
// It replaces a float64 library GEMM whose kernel choice depended erratically on the number of stashed columns
// (measured at Mp = 208: 38 TFLOP/s at K = 524 288 and 983 040, 1.2 TFLOP/s at K = 32 768 ... 327 680).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cbfssm_hip.h"

namespace cbfssm {

int fail(int rc, const char* fmt, ...);   // cbfssm_api.hip

typedef double d4 __attribute__((ext_vector_type(4)));
#define CBF_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// waves per workgroup: eight, or the even split of the row blocks over the same number of workgroups when that drops
// whole idle waves (every workgroup stages the full B image).  Measured: NBLK = 20 as 3 x 7 instead of 8 + 8 + 4 is
// -0.8 % on the C5 step; NBLK = 13 as 2 x 7 instead of 8 + 5 is +1.5 % on the C4 step (seven waves load the four SIMDs
// 2-2-2-1), so odd splits below 16 row blocks stay at eight.
constexpr int contract_waves(int nblk)
{
    const int groups = (nblk + 7) / 8, even = (nblk + groups - 1) / groups;
    return (nblk < 16 && (even & 1)) ? 8 : even;
}

template <int NBLK>
__global__ __launch_bounds__(64 * contract_waves(NBLK)) void stash_contract_kernel(const double* __restrict__ sa,
                                                                             const double* __restrict__ sk, int64_t nslots,
                                                                             int64_t slots_per_wg, double* __restrict__ part)
{
    constexpr int IMG = NBLK * 256;                     // doubles of one operand image (one slot)
    constexpr int CONTRACT_WAVES = contract_waves(NBLK);
    constexpr int NT = 64 * CONTRACT_WAVES;
    constexpr int LPT = (IMG + NT - 1) / NT;            // B-image doubles staged per thread
    extern __shared__ double lds[];                     // [2][IMG]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int rb = blockIdx.y * CONTRACT_WAVES + w;
    const bool valid = rb < NBLK;
    const int rbc = valid ? rb : NBLK - 1;
    const int64_t s_begin = int64_t(blockIdx.x) * slots_per_wg;
    const int64_t s_end = (s_begin + slots_per_wg < nslots) ? s_begin + slots_per_wg : nslots;

    d4 acc[NBLK];
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb) acc[cb] = d4{0, 0, 0, 0};

    if (s_begin < s_end) {
        // prologue: B image of the first slot -> LDS buffer 0
        for (int i = tid; i < IMG; i += NT) lds[i] = sk[s_begin * IMG + i];
        double an[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) an[s] = sa[(s_begin * NBLK + rbc) * 256 + s * 64 + l];
        for (int64_t slot = s_begin; slot < s_end; ++slot) {
            const int buf = int(slot - s_begin) & 1;
            const bool has_next = slot + 1 < s_end;
            // next slot: B image into registers (stored to the other LDS buffer after this slot's MFMAs), A operands
            double bn[LPT], ac[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ac[s] = an[s];
            if (has_next) {
#pragma unroll
                for (int k = 0; k < LPT; ++k) {
                    const int i = tid + k * NT;
                    bn[k] = (i < IMG) ? sk[(slot + 1) * IMG + i] : 0.0;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) an[s] = sa[((slot + 1) * NBLK + rbc) * 256 + s * 64 + l];
            }
            __syncthreads();                             // this slot's B image is complete in lds[buf]
            const double* Bl = lds + buf * IMG + l;
#pragma unroll
            for (int cb = 0; cb < NBLK; ++cb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[cb] = CBF_MFMA(ac[s], Bl[(cb * 4 + s) * 64], acc[cb]);
            }
            if (has_next) {
                double* Bn = lds + (buf ^ 1) * IMG;      // last read two slots ago: ordered by the barrier above
#pragma unroll
                for (int k = 0; k < LPT; ++k) {
                    const int i = tid + k * NT;
                    if (i < IMG) Bn[i] = bn[k];
                }
            }
        }
    }
    if (valid) {
        double* o = part + (int64_t(blockIdx.x) * NBLK * NBLK + int64_t(rb) * NBLK) * 256 + l;
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[cb * 256 + r * 64] = acc[cb][r];
    }
}


// Second form, for the tile heights whose accumulator fits ONE workgroup (NBLK <= 13: two row blocks per wave, 2 x NBLK
// accumulator tiles = 208 VGPRs at NBLK = 13): every slot's two operand images (A2bar^T and K^T, 2 x NBLK x 2 KiB) are read
// from HBM exactly once -- the row-group form above reads the K^T image once per row group, 78 KB per slot instead of 53 KB
// at NBLK = 13, and the contraction runs against the adjoint kernels' own stash writes on the other stream: it is
// bandwidth-bound there (1.5 TB/s of reads next to 1 TB/s of writes).  Both images arrive by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging registers -- the accumulators own the register file),
// two LDS buffers, the copies of slot s+1 in flight under the 2 x 4 x NBLK MFMAs of slot s; every K^T operand read from LDS
// feeds both row blocks of the wave.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int NBLK>
__global__ __launch_bounds__(64 * ((NBLK + 1) / 2)) void stash_contract2_kernel(const double* __restrict__ sa,
                                                                              const double* __restrict__ sk, int64_t nslots,
                                                                              int64_t slots_per_wg, double* __restrict__ part)
{
    constexpr int W = (NBLK + 1) / 2;
    constexpr int IMG = NBLK * 256;                     // doubles of one operand image (one slot)
    constexpr int PIMG = IMG / 128;                     // 1-KiB pieces per image
    extern __shared__ double lds[];                     // [2][A image | B image]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int rb0 = 2 * w, rb1 = min(2 * w + 1, NBLK - 1);      // (the last wave of an odd NBLK repeats its block; dropped)
    const bool valid1 = 2 * w + 1 < NBLK;
    const int64_t s_begin = int64_t(blockIdx.x) * slots_per_wg;
    const int64_t s_end = (s_begin + slots_per_wg < nslots) ? s_begin + slots_per_wg : nslots;

    d4 acc0[NBLK], acc1[NBLK];
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb) { acc0[cb] = d4{0, 0, 0, 0}; acc1[cb] = d4{0, 0, 0, 0}; }

    auto stage = [&](int64_t slot, int buf) {
        double* dst = lds + buf * 2 * IMG;
#pragma unroll
        for (int k = 0; k < (2 * PIMG + W - 1) / W; ++k) {
            const int p = w + k * W;                    // piece of this wave: A image pieces first, then the B image
            if (p < 2 * PIMG) {
                const double* src = (p < PIMG) ? sa + slot * IMG + p * 128 : sk + slot * IMG + (p - PIMG) * 128;
                __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 2 * l), (lds_void_t*)(dst + p * 128), 16, 0, 0);
            }
        }
    };

    if (s_begin < s_end) {
        stage(s_begin, 0);
        __syncthreads();                                 // (its fence waits for the LDS-DMA: vmcnt(0))
        for (int64_t slot = s_begin; slot < s_end; ++slot) {
            const int buf = int(slot - s_begin) & 1;
            if (slot + 1 < s_end) stage(slot + 1, buf ^ 1);       // read last in the previous iteration, before its barrier
            const double* A = lds + buf * 2 * IMG + l;
            const double* B = A + IMG;
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {                // (not unrolled: the B reads of one k-step ahead of their MFMAs
                                                         //  are what fits next to 208 accumulator registers)
                const double a0 = A[rb0 * 256 + s * 64], a1 = A[rb1 * 256 + s * 64];
#pragma unroll
                for (int cb = 0; cb < NBLK; ++cb) {
                    const double b = B[(cb * 4 + s) * 64];
                    acc0[cb] = CBF_MFMA(a0, b, acc0[cb]);
                    acc1[cb] = CBF_MFMA(a1, b, acc1[cb]);
                }
            }
            __syncthreads();                             // next slot's images landed (vmcnt(0)), this buffer free
        }
    }
    double* o0 = part + (int64_t(blockIdx.x) * NBLK * NBLK + int64_t(rb0) * NBLK) * 256 + l;
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[cb * 256 + r * 64] = acc0[cb][r];
    if (valid1) {
        double* o1 = part + (int64_t(blockIdx.x) * NBLK * NBLK + int64_t(rb1) * NBLK) * 256 + l;
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) o1[cb * 256 + r * 64] = acc1[cb][r];
    }
}

// Third form: only the SYMMETRIC PART of the sum is ever used -- the train tail maps d loss / d K^-1 to d loss / d K_mm =
// -K^-1 (.) K^-1 and every consumer contracts that with a symmetric dK_mm/dtheta (cbfssm_tail.hip, cbfssm.py:273-275 through
// gp_tf.py:129-130) -- so this form accumulates the lower-triangular 16 x 16 blocks of  S = sum (A2bar K^T + K A2bar^T)  only:
// NBLK (NBLK + 1) / 2 accumulator tiles instead of NBLK^2 (91 instead of 169 at NBLK = 13, 210 instead of 400 at 20) for the
// same number of MFMAs (eight per off-diagonal block: both images have the SAME lane layout, X[m = lane & 15][n = 4 s +
// (lane >> 4)], which is at once the A operand of X and the B operand of X^T, so  S_ij += A_i K_j^T + K_i A_j^T  is two
// MFMAs per k-step on one accumulator; the diagonal blocks take A_i K_i^T only and are symmetrised by the reduction).
// What the halved accumulator buys: ONE workgroup holds the whole triangle at every tile height (so each slot's two images
// are read from HBM exactly once -- the row-group form above reads the K^T image once per row group), and ANY wave can take
// ANY block (all operands are in LDS), so the blocks are dealt to the waves by MFMA count: every SIMD issues the same
// number -- the two-row-blocks-per-wave form loads the four SIMDs 4-4-3-2 at 13 row blocks (its own bound: 5.2 ms of the C4
// step against 4.2 ms here).
constexpr int tri_row(int b) { int i = 0; while ((i + 1) * (i + 2) / 2 <= b) ++i; return i; }
constexpr int tri_col(int b) { return b - tri_row(b) * (tri_row(b) + 1) / 2; }
#ifndef CBF_SYM_WAVES13
#define CBF_SYM_WAVES13 8
#endif
constexpr int sym_waves(int nblk) { return nblk <= 13 ? CBF_SYM_WAVES13 : 8; }
// workgroups per slot slice: one holds the whole triangle up to 16 row blocks (17 accumulator tiles per wave); at 20 the 210
// tiles go to two workgroups of eight waves (13-14 tiles each: 27 per wave would spill), the first of which needs -- and
// stages -- only the image rows of its own block rows
constexpr int sym_groups(int nblk) { return nblk >= 20 ? 2 : 1; }
template <int NBLK, int NW>
constexpr int sym_start(int w)                   // first block of wave w: the blocks in row-major order, cut at equal MFMA counts
{
    constexpr int NB = NBLK * (NBLK + 1) / 2, TOT = NBLK * NBLK;
    if (w >= NW) return NB;
    int b = 0, i = 0, j = 0, wsum = 0;              // wsum: MFMAs / 4 of the blocks [0, b): two per off-diagonal block, one per diagonal one
    while (b < NB && wsum * NW < TOT * w) {
        wsum += (i == j) ? 1 : 2;
        ++b;
        if (j == i) { ++i; j = 0; } else ++j;
    }
    return b;
}
template <int NBLK, int NW>
constexpr int sym_maxcount()
{
    int m = 0;
    for (int w = 0; w < NW; ++w) {
        const int c = sym_start<NBLK, NW>(w + 1) - sym_start<NBLK, NW>(w);
        if (c > m) m = c;
    }
    return m;
}

// blocks [B, END) of one wave, two at a time where they share a row (their MFMAs alternate between two accumulators)
template <int NBLK, int START, int END, int B, int MAXC>
__device__ __forceinline__ void sym_blocks(d4 (&acc)[MAXC], const double* Al, const double* Kl, double (&ra)[4], double (&rk)[4])
{
    if constexpr (B < END) {
        constexpr int i = tri_row(B), j = tri_col(B);
        if constexpr (B == START || tri_row(B - 1) != i) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { ra[s] = Al[(i * 4 + s) * 64]; rk[s] = Kl[(i * 4 + s) * 64]; }
        }
        constexpr bool PAIR = (B + 1 < END) && (tri_row(B + 1) == i);
        if constexpr (PAIR) {
            constexpr int j1 = j + 1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double k0 = Kl[(j * 4 + s) * 64], k1 = Kl[(j1 * 4 + s) * 64];
                acc[B - START] = CBF_MFMA(ra[s], k0, acc[B - START]);
                acc[B + 1 - START] = CBF_MFMA(ra[s], k1, acc[B + 1 - START]);
                const double a0 = Al[(j * 4 + s) * 64];
                acc[B - START] = CBF_MFMA(rk[s], a0, acc[B - START]);                      // (j < j1 <= i: off the diagonal)
                if constexpr (j1 != i) {
                    const double a1 = Al[(j1 * 4 + s) * 64];
                    acc[B + 1 - START] = CBF_MFMA(rk[s], a1, acc[B + 1 - START]);
                }
            }
            sym_blocks<NBLK, START, END, B + 2, MAXC>(acc, Al, Kl, ra, rk);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double k0 = Kl[(j * 4 + s) * 64];
                acc[B - START] = CBF_MFMA(ra[s], k0, acc[B - START]);
                if constexpr (j != i) {
                    const double a0 = Al[(j * 4 + s) * 64];
                    acc[B - START] = CBF_MFMA(rk[s], a0, acc[B - START]);
                }
            }
            sym_blocks<NBLK, START, END, B + 1, MAXC>(acc, Al, Kl, ra, rk);
        }
    }
}

// One wave's whole pass: its own accumulator array (statically indexed: it stays in registers), the slot loop, the store.
// NW: waves of a workgroup, NV = NW x groups: virtual waves the triangle is dealt to, WV: this wave's index among them.
template <int NBLK, int NW, int NV, int WV>
__device__ __forceinline__ void sym_run(const double* __restrict__ sa, const double* __restrict__ sk, int64_t s_begin, int64_t s_end,
                                        double* lds, int l, double* __restrict__ o)
{
    constexpr int S0 = sym_start<NBLK, NV>(WV), S1 = sym_start<NBLK, NV>(WV + 1), CNT = S1 - S0;
    constexpr int IMG = NBLK * 256;                     // doubles of one operand image (one slot)
    constexpr int PIMG = IMG / 128;                     // 1-KiB pieces per image (two per row block)
    constexpr int GRP = WV / NW;                        // this wave's workgroup among the groups of a slice
    constexpr int ROWS = tri_row(sym_start<NBLK, NV>((GRP + 1) * NW) - 1) + 1;     // image row blocks this workgroup reads
    constexpr int PROW = 2 * ROWS;                      // ... = pieces of each image it stages
    d4 acc[CNT > 0 ? CNT : 1];
#pragma unroll
    for (int k = 0; k < CNT; ++k) acc[k] = d4{0, 0, 0, 0};

    auto stage = [&](int64_t slot, int buf) {
        double* dst = lds + buf * 2 * IMG;
#pragma unroll
        for (int k = 0; k < (2 * PROW + NW - 1) / NW; ++k) {
            const int q = (WV % NW) + k * NW;           // piece of this wave: A image pieces first, then the K image
            if (q < 2 * PROW) {
                const int p = (q < PROW) ? q : PIMG + (q - PROW);
                const double* src = (q < PROW) ? sa + slot * IMG + q * 128 : sk + slot * IMG + (q - PROW) * 128;
                __builtin_amdgcn_global_load_lds((glb_void_t*)(src + 2 * l), (lds_void_t*)(dst + p * 128), 16, 0, 0);
            }
        }
    };

    if (s_begin < s_end) {
        stage(s_begin, 0);
        __syncthreads();                                 // (its fence waits for the LDS-DMA: vmcnt(0))
        for (int64_t slot = s_begin; slot < s_end; ++slot) {
            const int buf = int(slot - s_begin) & 1;
            if (slot + 1 < s_end) stage(slot + 1, buf ^ 1);       // read last in the previous iteration, before its barrier
            const double* Al = lds + buf * 2 * IMG + l;
            double ra[4], rk[4];
            sym_blocks<NBLK, S0, S1, S0, (CNT > 0 ? CNT : 1)>(acc, Al, Al + IMG, ra, rk);
            __syncthreads();                             // next slot's images landed (vmcnt(0)), this buffer free
        }
    }
#pragma unroll
    for (int k = 0; k < CNT; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(S0 + k) * 256 + r * 64 + l] = acc[k][r];
}

template <int NBLK, int NW, int NV, int WV>
__device__ __forceinline__ void sym_dispatch(int gw, const double* sa, const double* sk, int64_t s_begin, int64_t s_end, double* lds,
                                             int l, double* o)
{
    if constexpr (WV < NV) {
        if (gw == WV) sym_run<NBLK, NW, NV, WV>(sa, sk, s_begin, s_end, lds, l, o);    // (gw is wave-uniform: a scalar branch)
        else sym_dispatch<NBLK, NW, NV, WV + 1>(gw, sa, sk, s_begin, s_end, lds, l, o);
    }
}

template <int NBLK>
__global__ __launch_bounds__(64 * sym_waves(NBLK)) void stash_contract_sym_kernel(const double* __restrict__ sa,
                                                                                   const double* __restrict__ sk, int64_t nslots,
                                                                                   int64_t slots_per_wg, double* __restrict__ part)
{
    constexpr int NW = sym_waves(NBLK), NV = NW * sym_groups(NBLK);
    constexpr int NB = NBLK * (NBLK + 1) / 2;
    extern __shared__ double lds[];                     // [2][A image | K image]
    const int tid = threadIdx.x, l = tid & 63;
    const int gw = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * NW + (tid >> 6));
    const int64_t s_begin = int64_t(blockIdx.x) * slots_per_wg;
    const int64_t s_end = (s_begin + slots_per_wg < nslots) ? s_begin + slots_per_wg : nslots;
    // every wave of a workgroup runs the same slot loop with the same two barriers per slot, each on its own blocks
    sym_dispatch<NBLK, NW, NV, 0>(gw, sa, sk, s_begin, s_end, lds, l, part + int64_t(blockIdx.x) * NB * 256);
}

// out (the full [NBLK][NBLK] C-layout image) += the symmetric part of the sum: (S + S^T) / 2 with S the lower-triangular
// partial images, summed over the slices in a fixed order.  C layout of a block: row = (lane >> 4) + 4 reg, column = lane & 15.
__global__ void contract_reduce_sym_kernel(const double* part, int nblk, int nsplit, double* out)
{
    const int64_t n = int64_t(nblk) * nblk * 256, nb256 = int64_t(nblk) * (nblk + 1) / 2 * 256;
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int e = int(idx & 255), blk = int(idx >> 8), I = blk / nblk, J = blk % nblk;
    const int rr = ((e & 63) >> 4) + 4 * (e >> 6), cc = e & 15;
    const int eT = (cc >> 2) * 64 + (cc & 3) * 16 + rr;             // the element (cc, rr) of a block
    const int hi = I > J ? I : J, lo = I > J ? J : I;
    const int64_t b = (int64_t(hi) * (hi + 1) / 2 + lo) * 256;
    double s = 0.0;
    if (I > J) {
        for (int k = 0; k < nsplit; ++k) s += part[k * nb256 + b + e];
    } else if (I < J) {
        for (int k = 0; k < nsplit; ++k) s += part[k * nb256 + b + eT];
    } else {
        for (int k = 0; k < nsplit; ++k) s += part[k * nb256 + b + e] + part[k * nb256 + b + eT];
    }
    out[idx] += 0.5 * s;
}

// out[i] += sum_k part[k][i] in a fixed order
__global__ void contract_reduce_kernel(const double* part, int64_t n, int nsplit, double* out)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += part[int64_t(k) * n + i];
    out[i] += s;
}

static int contract_split(int64_t nslots)
{
    // ~256 slices (two row groups => ~512 workgroups), at least 8 slots each.  A multiple of 8: the grid is (slice, row
    // group) and workgroups go to the eight XCDs round-robin by linear id, so the row-group workgroups of one slice
    // (ids x, x + nsplit, ...) land on the SAME XCD and the second one finds the slice's B images in that XCD's L2.
    // (Measured neutral at C4: 42.48 ms with 256 slices, 42.47 ms with 255.)
    int64_t n = nslots / 8;
    if (n > 256) n = 256;
    if (n >= 8) n &= ~int64_t(7);
    if (n < 1) n = 1;
    if (const char* e = getenv("CBFSSM_CONTRACT_SPLIT")) {          // measurement switch
        const long v = atol(e);
        if (v >= 1 && v <= nslots) n = v;
    }
    return int(n);
}

template <int NBLK>
static int launch_contract(const double* sa, const double* sk, int64_t nslots, double* work, double* out, hipStream_t st)
{
    const int nsplit = contract_split(nslots);
    const int64_t per = (nslots + nsplit - 1) / nsplit;
    constexpr int CONTRACT_WAVES = contract_waves(NBLK);
    const int nrg = (NBLK + CONTRACT_WAVES - 1) / CONTRACT_WAVES;
    if (!getenv("CBFSSM_CONTRACT_FULL")) {              // (measurement switch: the full-matrix forms below)
        const size_t lds3 = size_t(4) * NBLK * 256 * sizeof(double);
        auto k3 = stash_contract_sym_kernel<NBLK>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds3));
        if (e != hipSuccess) return -int(e) - 1000;
        hipLaunchKernelGGL(k3, dim3(unsigned(nsplit), unsigned(sym_groups(NBLK))), dim3(64 * sym_waves(NBLK)), lds3, st, sa, sk, nslots, per, work);
        const int64_t n3 = int64_t(NBLK) * NBLK * 256;
        hipLaunchKernelGGL(contract_reduce_sym_kernel, dim3(unsigned((n3 + 255) / 256)), dim3(256), 0, st, (const double*)work,
                           NBLK, nsplit, out);
        e = hipGetLastError();
        return e == hipSuccess ? 0 : -int(e) - 1000;
    }
    if constexpr (NBLK <= 13) {
        if (!getenv("CBFSSM_CONTRACT_V1")) {            // (measurement switch: the row-group form)
            const size_t lds2 = size_t(4) * NBLK * 256 * sizeof(double);
            auto k2 = stash_contract2_kernel<NBLK>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds2));
            if (e != hipSuccess) return -int(e) - 1000;
            hipLaunchKernelGGL(k2, dim3(unsigned(nsplit), 1), dim3(64 * ((NBLK + 1) / 2)), lds2, st, sa, sk, nslots, per, work);
            const int64_t n2 = int64_t(NBLK) * NBLK * 256;
            hipLaunchKernelGGL(contract_reduce_kernel, dim3(unsigned((n2 + 255) / 256)), dim3(256), 0, st, (const double*)work, n2,
                               nsplit, out);
            e = hipGetLastError();
            return e == hipSuccess ? 0 : -int(e) - 1000;
        }
    }
    const size_t lds = size_t(2) * NBLK * 256 * sizeof(double);
    auto k = stash_contract_kernel<NBLK>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return -int(e) - 1000;
    }
    hipLaunchKernelGGL(k, dim3(unsigned(nsplit), unsigned(nrg)), dim3(64 * CONTRACT_WAVES), lds, st, sa, sk, nslots, per, work);
    const int64_t n = int64_t(NBLK) * NBLK * 256;
    hipLaunchKernelGGL(contract_reduce_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, (const double*)work, n, nsplit,
                       out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

}  // namespace cbfssm

using namespace cbfssm;

extern "C" {

int64_t cbfssm_stash_contract_work_elems(const cbfssm_pack_layout* L, int64_t nslots)
{
    if (!L || nslots < 0) return -1;
    return int64_t(contract_split(nslots)) * L->NBLK * L->NBLK * 256;
}

int cbfssm_stash_contract_f64(const cbfssm_pack_layout* L, const double* stash_a, const double* stash_k, int64_t nslots,
                              double* work, double* ginv_image, void* stream)
{
    if (!L || !stash_a || !stash_k || !work || !ginv_image) return fail(-1, "null pointer");
    if (!L->rev_stash) return fail(-3, "M=%d does not run in stash mode", L->M);
    if (nslots <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (L->NBLK) {
        case 10: rc = launch_contract<10>(stash_a, stash_k, nslots, work, ginv_image, st); break;
        case 13: rc = launch_contract<13>(stash_a, stash_k, nslots, work, ginv_image, st); break;
        case 16: rc = launch_contract<16>(stash_a, stash_k, nslots, work, ginv_image, st); break;
        case 20: rc = launch_contract<20>(stash_a, stash_k, nslots, work, ginv_image, st); break;
        default: return fail(-3, "no contraction kernel for tile height %d", L->NBLK);
    }
    return rc ? fail(rc, "stash contraction launch failed (NBLK=%d rc=%d)", L->NBLK, rc) : 0;
}

}  // extern "C"
