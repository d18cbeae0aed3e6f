// Shared pieces of the float32-arithmetic kernels (cbfssm_f32.hip: forward evaluation; cbfssm_rev32.hip: adjoint).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include "../../include/cbfssm_hip.h"
#include "cbfssm_kernels.hpp"

namespace cbfssm {

int fail(int code, const char* fmt, ...);

namespace f32 {

typedef float f4 __attribute__((ext_vector_type(4)));
#define CBF_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// float -> the nearest bfloat16 value (round to nearest even on the dropped 16 mantissa bits), kept in a float: the
// bf16-operand mode of the sweep rounds the two operands of the K^-1 K contraction this way and multiplies them on the
// float32 MFMA -- products of bf16 values are exact in float32, so this is the arithmetic of a bf16-operand /
// float32-accumulate MFMA up to the order of the accumulation (it measures the precision, it is not a throughput path).
__host__ __device__ __forceinline__ float round_bf16(float x)
{
    union { float f; uint32_t u; } v;
    v.f = x;
    v.u = (v.u + 0x7FFFu + ((v.u >> 16) & 1u)) & 0xFFFF0000u;
    return v.f;
}

struct Pack32 {
    const float* Bp;     // [NBLK][KS][64]  K^-1, A-operand image, permuted k order
    const float* Zp;     // [NBLK][DK][64]  Z / lengthscale, A-operand image (natural k order: the inputs' dimensions)
    const float* cz;     // [Mp]
    const float* mu;     // [NBLK][4][64]   zeta_mean, A-operand image, permuted k order
    const float* s2;     // [NBLK][4][64]   zeta_var
    const float* invl;   // [Dp]
    const float* scal;   // [0] = sigma^2, [1] = 1 when the contraction operands are rounded to bf16
    // adjoint kernels (cbfssm_rev32.hip): B operands come from LDS tiles there, so these images keep the NATURAL k order
    const float* BpN;    // [NBLK][KS][64]     K^-1, A-operand image
    const float* muB;    // [NBLK][4][64]      zeta_mean as A[row m][k = d]
    const float* s2B;    // [NBLK][4][64]      zeta_var
    const float* ZTq;    // [NBLK][JB][4][64]  (Z / lengthscale)^T as A[row j][k = m], row D = ones; PERMUTED k order (its
                         //                    B operand is an accumulator: register r of lane group g is row 4 g + r)
    // two-triangular GP form (gp_tf.py:137,145): W = L^-1 and W^T as A-operand images, permuted k order
    const float* Wp;     // [NBLK][KS][64]
    const float* WTp;    // [NBLK][KS][64]
    const float* WpN;    // the same two images in natural k order (adjoint)
    const float* WTpN;
};

struct Off32 {
    int64_t Bp, Zp, cz, mu, s2, invl, scal, BpN, muB, s2B, ZTq, Wp, WTp, WpN, WTpN, total;
};

static Off32 pack32_offsets(const cbfssm_pack_layout* L)
{
    Off32 o;
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 63) / 64 * 64; return r; };
    o.Bp = take(int64_t(L->NBLK) * L->KS * 64);
    o.Zp = take(int64_t(L->NBLK) * L->DK * 64);
    o.cz = take(L->Mp);
    o.mu = take(int64_t(L->NBLK) * 256);
    o.s2 = take(int64_t(L->NBLK) * 256);
    o.invl = take(L->Dp);
    o.scal = take(64);
    o.BpN = take(int64_t(L->NBLK) * L->KS * 64);
    o.muB = take(int64_t(L->NBLK) * 256);
    o.s2B = take(int64_t(L->NBLK) * 256);
    o.ZTq = take(int64_t(L->NBLK) * L->JB * 256);
    o.Wp = take(int64_t(L->NBLK) * L->KS * 64);
    o.WTp = take(int64_t(L->NBLK) * L->KS * 64);
    o.WpN = take(int64_t(L->NBLK) * L->KS * 64);
    o.WTpN = take(int64_t(L->NBLK) * L->KS * 64);
    o.total = p;
    return o;
}

static Pack32 pack32_ptrs(const cbfssm_pack_layout* L, const float* p)
{
    const Off32 o = pack32_offsets(L);
    Pack32 k;
    k.Bp = p + o.Bp; k.Zp = p + o.Zp; k.cz = p + o.cz; k.mu = p + o.mu; k.s2 = p + o.s2; k.invl = p + o.invl;
    k.scal = p + o.scal;
    k.BpN = p + o.BpN; k.muB = p + o.muB; k.s2B = p + o.s2B; k.ZTq = p + o.ZTq; k.Wp = p + o.Wp; k.WTp = p + o.WTp;
    k.WpN = p + o.WpN; k.WTpN = p + o.WTpN;
    return k;
}

__device__ __forceinline__ float rcp32(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ float rsqrt32(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    return fmaf(y, fmaf(-0.5f * x * y, y, 0.5f), y);
}

}  // namespace f32
}  // namespace cbfssm
