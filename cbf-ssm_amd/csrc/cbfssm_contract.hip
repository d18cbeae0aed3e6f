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
