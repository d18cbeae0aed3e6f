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
#include <type_traits>

namespace cbfssm {

typedef double d4 __attribute__((ext_vector_type(4)));

#define CBF_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Reciprocal and reciprocal square root for the step epilogues and their adjoints (they sit on the serial path of a
// time step): the
// hardware seed and two Newton steps, 1-2 ulp, about half the dependent instructions of the IEEE division / sqrt
// sequences.  Arguments are variances: positive, finite, far from the subnormal range.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return fma(y, fma(-hx * y, y, 0.5), y);
}

// exp for the kernel tiles (4 per lane, row block and step: 105 cycles per wave-call with the library routine, which
// also handles overflow, NaN and the subnormal range).  The argument here is E = z.x - .5|z|^2 - .5|x|^2 + log sigma^2 <=
// log sigma^2, or the -1e30 of a padding row: no overflow case.  Cody-Waite reduction x = k ln2 + r, |r| <= 0.3466, the
// degree-13 Taylor polynomial in Horner form (truncation 4e-18 relative), ldexp; x is clamped where exp underflows to
// zero.  Measured against long double on 5e6 arguments in [-60, 2]: 1.23 ulp at worst.
__device__ __forceinline__ double tile_exp(double x)
{
    x = fmax(x, -746.0);
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.60590438368216145994e-10;            // 1/13!
    p = fma(p, r, 2.08767569878680989792e-09);        // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);        // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);        // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);        // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);        // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);        // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);        // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);        // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);        // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, int(k));
}

// The four exponentials of a lane's kernel-tile rows, stage by stage: each stage is four independent instructions, so
// the four dependent chains of tile_exp interleave in program order (left to the scheduler, the four inlined copies
// are issued one after the other and every multiply-add waits for its predecessor's result).  Same operations per
// element as tile_exp: bitwise the same result.  (The clamp as one v_max_f64: fmax() also emits a canonicalising
// v_max_f64 x, x in front of it -- the argument is an MFMA result or the -1e30 row constant, never a signalling NaN.)
__device__ __forceinline__ d4 tile_exp4(d4 x)
{
    double k[4], r[4], p[4];
    const double lo = -746.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double y;
        asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(x[i]), "v"(lo));
        x[i] = y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = __builtin_rint(x[i] * 1.4426950408889634074);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma(-k[i], 6.93147180369123816490e-01, x[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma(-k[i], 1.90821492927058770002e-10, r[i]);
    constexpr double c[13] = {2.08767569878680989792e-09, 2.50521083854417187751e-08, 2.75573192239858906526e-07,
                              2.75573192239858906526e-06, 2.48015873015873015873e-05, 1.98412698412698412698e-04,
                              1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02,
                              1.66666666666666666667e-01, 0.5, 1.0, 1.0};
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = 1.60590438368216145994e-10;
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = fma(p[i], r[i], c[j]);
    d4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_ldexp(p[i], int(k[i]));
    return o;
}

enum { MODE_FWD = 0, MODE_BWD = 1 };

// Diagnostic build only (-DCBF_REV_STAMPS): per-phase cycle shares of the adjoint step, compute vs barrier wait, summed
// over the steps of a pass by lane 0 of every wave into otherwise unused slots of the slab's scalar block
// (wave 0 -> compute[7] at 100.., wait[7] at 107..; last wave -> 114.., 121..; wave 0 sub-phase marks at 128..).
// Never defined in the shipped library.
#ifdef CBF_REV_STAMPS
#define CBF_STAMP_DECL                                                                                      \
    unsigned long long st_prev, st_mprev = 0, st_c[7] = {0, 0, 0, 0, 0, 0, 0}, st_w[7] = {0, 0, 0, 0, 0, 0, 0}, \
                                              st_m[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CBF_STAMP_READ(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define CBF_STAMP_START() CBF_STAMP_READ(st_prev)
#define CBF_STAMP_BARRIER(i)                 \
    {                                        \
        unsigned long long t1_, t2_;         \
        CBF_STAMP_READ(t1_);                 \
        __syncthreads();                     \
        CBF_STAMP_READ(t2_);                 \
        st_c[i] += t1_ - st_prev;            \
        st_w[i] += t2_ - t1_;                \
        st_prev = t2_;                       \
    }
#define CBF_STAMP_MARK(i)                    \
    {                                        \
        unsigned long long t1_;              \
        CBF_STAMP_READ(t1_);                 \
        st_m[i] += t1_ - st_mprev;           \
        st_mprev = t1_;                      \
    }
#define CBF_STAMP_MARK0() CBF_STAMP_READ(st_mprev)
#define CBF_STAMP_GP_BARRIER() __syncthreads()
#else
#define CBF_STAMP_DECL
#define CBF_STAMP_START()
#define CBF_STAMP_BARRIER(i) __syncthreads()
#define CBF_STAMP_MARK(i)
#define CBF_STAMP_MARK0()
#define CBF_STAMP_GP_BARRIER() __syncthreads()
#endif



// 32-bit LDS address of a pointer into the dynamic shared array (for the hand-scheduled loops)
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ uint32_t lds_addr(const double* p) { return uint32_t(uintptr_t((lds_cdouble*)p)); }

// acc[i][h] += sum over `ngrp` groups of four k-steps of A_i[s] x B[s]   (i = the wave's two row blocks, h = s & 1)
//   A_i: lane-linear K^-1 operand image in global memory / L2, 512 bytes per k-step, this lane's first element at pa_i
//   B:   operand tile in LDS, BSTRIDE bytes per k-step, this lane's first element at LDS address ldsb
// Hand-scheduled: written as a source loop, hipcc's wait insertion drains vmcnt/lgkmcnt on the loop back edge, so every
// group of four k-steps waits out an L2 latency before its MFMAs.  Here two operand sets (fixed registers v120..v167,
// declared as clobbers: inside the 168-register budget of the 10-wave workgroups) alternate: the twelve loads of group g+1 are in flight under the eight MFMAs of group g, the
// waits are counted.  The images are zero-padded to whole groups; the trailing s_nops are what the compiler puts
// between an MFMA and a VALU read of its result.
template <int BSTRIDE>
__device__ __forceinline__ void stream_kinv_rb2(d4& c00, d4& c01, d4& c10, d4& c11, const double* pa0, const double* pa1,
                                                uint32_t ldsb, int ngrp)
{
    static_assert(7 * BSTRIDE < 65536, "ds_read offset field");
    if (ngrp <= 0) return;
    const unsigned long long step = 4096;           // bytes of K^-1 image per two groups
    asm volatile(
        "global_load_dwordx2 v[120:121], %[pa0], off offset:0\n\t"
        "global_load_dwordx2 v[128:129], %[pa1], off offset:0\n\t"
        "ds_read_b64 v[136:137], %[pb] offset:%[bs0]\n\t"
        "global_load_dwordx2 v[122:123], %[pa0], off offset:512\n\t"
        "global_load_dwordx2 v[130:131], %[pa1], off offset:512\n\t"
        "ds_read_b64 v[138:139], %[pb] offset:%[bs1]\n\t"
        "global_load_dwordx2 v[124:125], %[pa0], off offset:1024\n\t"
        "global_load_dwordx2 v[132:133], %[pa1], off offset:1024\n\t"
        "ds_read_b64 v[140:141], %[pb] offset:%[bs2]\n\t"
        "global_load_dwordx2 v[126:127], %[pa0], off offset:1536\n\t"
        "global_load_dwordx2 v[134:135], %[pa1], off offset:1536\n\t"
        "ds_read_b64 v[142:143], %[pb] offset:%[bs3]\n\t"
        "1:\n"
        "s_cmp_lt_u32 %[n], 3\n\t"
        "s_cbranch_scc1 2f\n\t"
        "global_load_dwordx2 v[144:145], %[pa0], off offset:2048\n\t"
        "global_load_dwordx2 v[152:153], %[pa1], off offset:2048\n\t"
        "ds_read_b64 v[160:161], %[pb] offset:%[bs4]\n\t"
        "global_load_dwordx2 v[146:147], %[pa0], off offset:2560\n\t"
        "global_load_dwordx2 v[154:155], %[pa1], off offset:2560\n\t"
        "ds_read_b64 v[162:163], %[pb] offset:%[bs5]\n\t"
        "global_load_dwordx2 v[148:149], %[pa0], off offset:3072\n\t"
        "global_load_dwordx2 v[156:157], %[pa1], off offset:3072\n\t"
        "ds_read_b64 v[164:165], %[pb] offset:%[bs6]\n\t"
        "global_load_dwordx2 v[150:151], %[pa0], off offset:3584\n\t"
        "global_load_dwordx2 v[158:159], %[pa1], off offset:3584\n\t"
        "ds_read_b64 v[166:167], %[pb] offset:%[bs7]\n\t"
        "s_waitcnt vmcnt(8) lgkmcnt(4)\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[120:121], v[136:137], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[128:129], v[136:137], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[122:123], v[138:139], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[130:131], v[138:139], %[c11]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[124:125], v[140:141], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[132:133], v[140:141], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[126:127], v[142:143], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[134:135], v[142:143], %[c11]\n\t"
        "v_lshl_add_u64 %[pa0], %[pa0], 0, %[step]\n\t"
        "v_lshl_add_u64 %[pa1], %[pa1], 0, %[step]\n\t"
        "v_add_u32 %[pb], %[bs8], %[pb]\n\t"
        "s_sub_u32 %[n], %[n], 2\n\t"
        "global_load_dwordx2 v[120:121], %[pa0], off offset:0\n\t"
        "global_load_dwordx2 v[128:129], %[pa1], off offset:0\n\t"
        "ds_read_b64 v[136:137], %[pb] offset:%[bs0]\n\t"
        "global_load_dwordx2 v[122:123], %[pa0], off offset:512\n\t"
        "global_load_dwordx2 v[130:131], %[pa1], off offset:512\n\t"
        "ds_read_b64 v[138:139], %[pb] offset:%[bs1]\n\t"
        "global_load_dwordx2 v[124:125], %[pa0], off offset:1024\n\t"
        "global_load_dwordx2 v[132:133], %[pa1], off offset:1024\n\t"
        "ds_read_b64 v[140:141], %[pb] offset:%[bs2]\n\t"
        "global_load_dwordx2 v[126:127], %[pa0], off offset:1536\n\t"
        "global_load_dwordx2 v[134:135], %[pa1], off offset:1536\n\t"
        "ds_read_b64 v[142:143], %[pb] offset:%[bs3]\n\t"
        "s_waitcnt vmcnt(8) lgkmcnt(4)\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[144:145], v[160:161], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[152:153], v[160:161], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[146:147], v[162:163], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[154:155], v[162:163], %[c11]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[148:149], v[164:165], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[156:157], v[164:165], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[150:151], v[166:167], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[158:159], v[166:167], %[c11]\n\t"
        "s_branch 1b\n\t"
        "2:\n"
        "s_cmp_eq_u32 %[n], 2\n\t"
        "s_cbranch_scc0 3f\n\t"
        "global_load_dwordx2 v[144:145], %[pa0], off offset:2048\n\t"
        "global_load_dwordx2 v[152:153], %[pa1], off offset:2048\n\t"
        "ds_read_b64 v[160:161], %[pb] offset:%[bs4]\n\t"
        "global_load_dwordx2 v[146:147], %[pa0], off offset:2560\n\t"
        "global_load_dwordx2 v[154:155], %[pa1], off offset:2560\n\t"
        "ds_read_b64 v[162:163], %[pb] offset:%[bs5]\n\t"
        "global_load_dwordx2 v[148:149], %[pa0], off offset:3072\n\t"
        "global_load_dwordx2 v[156:157], %[pa1], off offset:3072\n\t"
        "ds_read_b64 v[164:165], %[pb] offset:%[bs6]\n\t"
        "global_load_dwordx2 v[150:151], %[pa0], off offset:3584\n\t"
        "global_load_dwordx2 v[158:159], %[pa1], off offset:3584\n\t"
        "ds_read_b64 v[166:167], %[pb] offset:%[bs7]\n\t"
        "s_waitcnt vmcnt(8) lgkmcnt(4)\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[120:121], v[136:137], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[128:129], v[136:137], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[122:123], v[138:139], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[130:131], v[138:139], %[c11]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[124:125], v[140:141], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[132:133], v[140:141], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[126:127], v[142:143], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[134:135], v[142:143], %[c11]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[144:145], v[160:161], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[152:153], v[160:161], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[146:147], v[162:163], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[154:155], v[162:163], %[c11]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[148:149], v[164:165], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[156:157], v[164:165], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[150:151], v[166:167], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[158:159], v[166:167], %[c11]\n\t"
        "s_branch 4f\n\t"
        "3:\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[120:121], v[136:137], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[128:129], v[136:137], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[122:123], v[138:139], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[130:131], v[138:139], %[c11]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c00], v[124:125], v[140:141], %[c00]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c10], v[132:133], v[140:141], %[c10]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c01], v[126:127], v[142:143], %[c01]\n\t"
        "v_mfma_f64_16x16x4_f64 %[c11], v[134:135], v[142:143], %[c11]\n\t"
        "4:\n"
        "s_nop 15\n\t"
        "s_nop 2\n\t"
        : [c00] "+v"(c00), [c01] "+v"(c01), [c10] "+v"(c10), [c11] "+v"(c11), [pa0] "+v"(pa0), [pa1] "+v"(pa1),
          [pb] "+v"(ldsb), [n] "+s"(ngrp)
        : [step] "s"(step), [bs0] "n"(0 * BSTRIDE), [bs1] "n"(1 * BSTRIDE), [bs2] "n"(2 * BSTRIDE), [bs3] "n"(3 * BSTRIDE),
          [bs4] "n"(4 * BSTRIDE), [bs5] "n"(5 * BSTRIDE), [bs6] "n"(6 * BSTRIDE), [bs7] "n"(7 * BSTRIDE),
          [bs8] "n"(8 * BSTRIDE)
        : "scc", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
}

struct PackPtrs {
    const double* Bp;
    const double* Zp;
    const double* cz;
    const double* muA;
    const double* s2A;
    const double* invl;
    const double* scal;
    int KSr;
    const double* Wp;    // two-triangular form: W = L^-1 as A-operand image [NBLK][KS][64] (zero above the diagonal)
    const double* WTp;   //                      W^T as A-operand image [NBLK][KS][64] (zero below the diagonal)
};

// Workgroup-scope LDS flags of the two-triangular GP form: wave p publishes the rows of A = L^-1 k it owns and raises
// flag[p] to the step's epoch; a consumer polls before it reads those rows.  Every wave raises its flag in every step
// before it waits on anybody (no cycle), and the poll is bounded.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void flag_release(int* f, int v)
{
    __hip_atomic_store((lds_int*)f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A poll that runs out (never in a correct launch: the bound only keeps a broken one from hanging) raises the marker in
// slot 96 of the flag array; the pass kernel turns it into a NaN partial sum, so a broken hand-off shows up as a NaN loss
// instead of a silently wrong one.
#define CBF_FLAG_TIMEOUT_SLOT 96
__device__ __forceinline__ void flag_wait(int* flags, int idx, int v)
{
    int spins = 0;
    while (__hip_atomic_load((lds_int*)(flags + idx), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {
            __hip_atomic_store((lds_int*)(flags + CBF_FLAG_TIMEOUT_SLOT), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
}

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
    double* dbg;           // diagnostic builds: [workgroup][32] stamp sums (null otherwise)
    int half;              // CBFSSMHALF forward pass (cbfssmhalf.py:117-172): x_0 from x0, Kalman update on d < dim_y only
    const double* x0;      // half: (B, dim_x) recognition-model output
    int group0, gtotal;    // this launch covers workgroup tiles [group0, group0 + gridDim.x) of gtotal (chain-group split)
    double* a2s;           // optional: per-step A2 = K^-1 k tiles kept for the adjoint, [slot][16-chain group][NBLK*256]
                           // in MFMA C-layout (slot = t for fwd, run*T + t for bwd)
    double* fmv;           // optional: per-step (fmean, fvar) after residual / process noise, kept for the adjoint:
                           // fwd [(T-1)][N][dim_x][2], bwd [2][T][N][dim_x-dim_y][2]
    int tri;               // 1: GP conditional in the reference's two-triangular form (gp_tf.py:137-145), 0: K^-1 contraction
    int ksave;             // 1: the kernel tile K = k(Z, x_t) of every step is kept next to its A2 tile: a saved record is
                           // [A2: NBLK*256][K: NBLK*256] and the adjoint reads K instead of recomputing it (MFMA + exp)
};

// doubles per saved record (one step, one 16-chain group) -- the pass kernels (writers) and the adjoint (reader) agree on it
__host__ __device__ constexpr int saved_tile_stride(int nblk, int ksave) { return nblk * 256 * (ksave ? 2 : 1); }

struct PredictArgs {
    PackPtrs pk;
    const double* X;   // (npts, D)
    int64_t npts;
    int D, Do;
    double* fmean;     // (npts, Do)
    double* fvar;
    int tri;
    double* a2o;       // optional: the A2 = K^-1 k tiles of every 16-point group, [group][NBLK*256] in MFMA C-layout
};

// TRI: the GP conditional in the reference's own two-triangular form (gp_tf.py:137-145) instead of the K^-1 contraction:
//   A = W K (W = L^-1, lower triangular), fvar_0 = sigma^2 - colsum(A o A), A2 = W^T A (upper triangular)
// -- the same M^2 multiply-adds per point when the zero blocks are skipped, and sigma^2 - |L^-1 k|^2 does not cancel the
// way sigma^2 - k.(K^-1 k) does on an ill-conditioned K_mm.  Row block rb of A needs the k-blocks 0..rb, row block rb of
// A2 the blocks rb..NBLK-1 of A: NBLK + 1 blocks per row block in total, whoever owns it.  The A rows travel through
// an LDS tile; a consumer waits for the producing wave's flag, not for a workgroup barrier, so wave rb starts its second
// product when its own rows are done and meets the rows of the later blocks as they appear.
// KT: compile-time trim of the register-resident tiles.  KT >= 0 promises that exactly KS - KT k-steps carry data (M in
// (4 (KS - KT - 1), 4 (KS - KT)]): the loops end there with no runtime guard -- a guard on an MFMA cuts the straight-line
// sequence into basic blocks and the LDS reads are no longer issued ahead (measured: forward kernels 2.6 -> 3.9 ms at C3).
// KT = -1: not specialised, all KS k-steps of the zero-padded images (the small tiles; every streamed tile).
template <int NBLK, int RB, int DK, bool BREG, bool TRI = false, int KT = -1>
struct Tile {
    static constexpr int W = (NBLK + RB - 1) / RB;   // waves per workgroup
    static constexpr int NT = 64 * W;
    static constexpr int MP = 16 * NBLK;
    static constexpr int KS = MP / 4;                // k-steps of the K^-1 K product (stride of the operand images)
    static constexpr bool EXACT = (KT >= 0);
    static constexpr int KSE = EXACT ? KS - KT : KS;  // k-steps the register-resident loops execute
    static_assert(!EXACT || (BREG && KT < 4), "the trim applies to register-resident tiles, within the last row block");
    static constexpr int QPW = (4 + W - 1) / W;      // state-row groups (4 rows each) per wave in phase 3
    static constexpr int TRI_LDS = TRI ? MP * 16 + 64 : 0;                 // A = L^-1 k tile + the waves' flags
    static constexpr int LDS_DOUBLES = DK * 64 + MP * 16 + W * 512 + 64 + TRI_LDS;   // per column block (+64 once)
    static constexpr int NBR = BREG ? (TRI ? KSE + 4 : KSE) : 1;          // loop-invariant matrix operands in VGPRs
    static_assert(!(TRI && BREG) || RB == 1, "register-resident triangular operands: one row block per wave");

    // Row block(s) of wave w.  Dense form: w RB + i.  Two-triangular form with register operands: row block rb costs
    // 4 (rb + 1) k-steps in the first product and KS - 4 rb in the second, so which waves SHARE A SIMD matters (waves go
    // to the four SIMDs round robin: w & 3).  At seven row blocks the pairs (5,0), (4,1), (3,2) carry 28 k-steps of the
    // first product each and the longest row block (6) sits alone on the fourth SIMD: every A row block is complete
    // after about 28 MFMA slots instead of about 50 with the identity map, and the second product rarely waits.
    __device__ __forceinline__ static int rb_of(int w, int i)
    {
        if constexpr (TRI && BREG && NBLK == 7) {
            return (w == 3) ? 6 : ((w < 3) ? 5 - w : w - 4);       // w: 0..6 -> 5, 4, 3, 6, 0, 1, 2
        } else {
            return w * RB + i;
        }
    }
    // pass_kernel<NC = 1>, streamed K^-1: how many of the mean / variance operand images fit into LDS next to the tiles
    static constexpr int EPI_LDS_N = BREG ? 0 : (LDS_DOUBLES + 2 * NBLK * 256 <= 20480 ? 2 : (LDS_DOUBLES + NBLK * 256 <= 20480 ? 1 : 0));
    static constexpr bool EPI_LDS = EPI_LDS_N > 0;
    static constexpr int EPI_LDS_DOUBLES = EPI_LDS_N * NBLK * 256;

    // loop-invariant MFMA A operands of this wave
    double Zreg[RB][DK];
    double czr[RB][4];
    double muA[RB][4];
    double s2A[RB][4];
    double Breg[BREG ? RB : 1][NBR];
    const double* Bp;
    const double* Wp;
    const double* WTp;
    const double* muAg;
    const double* s2Ag;
    int lane_;
    double sigma2;
    int KSr;       // k-steps of K^-1 that carry data: ceil(M/4) <= KS
    int koff;      // > 0: the kernel tile is kept too, koff doubles behind the A2 tile of its record (PassArgs::ksave)

    template <bool WITH_EPI = true>
    __device__ __forceinline__ void load_operands(const PackPtrs& pk, int w, int l)
    {
        Bp = pk.Bp;
        Wp = pk.Wp;
        WTp = pk.WTp;
        muAg = pk.muA;
        s2Ag = pk.s2A;
        lane_ = l;
        sigma2 = pk.scal[0];
        KSr = pk.KSr;
        koff = 0;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = rb_of(w, i);
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
            if constexpr (BREG && !TRI) {
#pragma unroll
                for (int s = 0; s < KSE; ++s) Breg[i][s] = ok ? pk.Bp[(rbc * KS + s) * 64 + l] : 0.0;
            }
        }
        if constexpr (BREG && TRI) load_tri_dispatch<0>(pk, __builtin_amdgcn_readfirstlane(rb_of(w, 0)), l);
    }

    // register-resident triangular operands of the wave that owns row block RBI: 4 (RBI + 1) k-steps of W's row block
    // (blocks 0..RBI), then KS - 4 RBI k-steps of W^T's row block (blocks RBI..NBLK-1): KS + 4 in total for every wave.
    // The indices must be compile-time constants (a register array), hence one instantiation per row block.
    template <int RBI>
    __device__ __forceinline__ void load_tri_reg(const PackPtrs& pk, int l)
    {
        constexpr int N1 = (4 * (RBI + 1) < KSE) ? 4 * (RBI + 1) : KSE;
#pragma unroll
        for (int s = 0; s < N1; ++s) Breg[0][s] = pk.Wp[(RBI * KS + s) * 64 + l];
#pragma unroll
        for (int j = 0; j < KSE - 4 * RBI; ++j) Breg[0][N1 + j] = pk.WTp[(RBI * KS + 4 * RBI + j) * 64 + l];
    }
    template <int I>
    __device__ __forceinline__ void load_tri_dispatch(const PackPtrs& pk, int wu, int l)
    {
        if constexpr (I < NBLK) {
            if (wu == I) load_tri_reg<I>(pk, l);
            else load_tri_dispatch<I + 1>(pk, wu, l);
        }
    }

    // ---- phase 2 in the two-triangular form.  epilogue shared by both operand sources: a2 rows -> P1/P2 partials
    // slot: this wave's slot of the partial tiles (its index); rb0: its first row block
    __device__ __forceinline__ void tri_epilogue(const d4 (&a2)[RB], double q, double* part, int slot, int rb0, int l,
                                                 double* a2o)
    {
        const int w = slot;
        d4 P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = rb0 + i;
            if (rb < NBLK) {
                if (a2o) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a2o[rb * 256 + r * 64 + l] = a2[i][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double mu_op = BREG ? muA[i][r] : muAg[(rb * 4 + r) * 64 + lane_];
                    const double s2_op = BREG ? s2A[i][r] : s2Ag[(rb * 4 + r) * 64 + lane_];
                    P1 = CBF_MFMA(mu_op, a2[i][r], P1);
                    P2 = CBF_MFMA(s2_op, a2[i][r] * a2[i][r], P2);
                }
            }
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            part[((w * 2 + 0) * 4 + r) * 64 + l] = P1[r];
            part[((w * 2 + 1) * 4 + r) * 64 + l] = P2[r] - q;       // fvar_0 - sigma^2 = -colsum(A o A)   (gp_tf.py:140)
        }
    }

    // operands in VGPRs (NBLK <= 7, one row block per wave); RBI = the wave's row block.  Straight-line code: only the
    // k-steps of the LAST row block are guarded by the number of k-steps that carry data (the images are zero beyond M,
    // so the others are at worst products with zeros, and a guard per MFMA would cut the loop into basic blocks whose
    // LDS reads cannot be issued ahead).
    template <int RBI>
    __device__ __forceinline__ void phase2_tri_reg(const double* Kt, double* At, int* flag, int epoch, double* part, int l,
                                                   int slot, double* a2o)
    {
        constexpr int N1 = (4 * (RBI + 1) < KSE) ? 4 * (RBI + 1) : KSE;
        // without the compile-time trim the k-steps of the LAST row block are guarded by the count that carry data
        constexpr int NFULL = (!EXACT && RBI == NBLK - 1) ? N1 - 4 : N1;
        // A rows of this block: W[RBI, 0..RBI] K[0..RBI]                                         (gp_tf.py:137)
        d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
            const double b = Kt[64 * s + l];
            if (s & 1) acc1 = CBF_MFMA(Breg[0][s], b, acc1);
            else acc0 = CBF_MFMA(Breg[0][s], b, acc0);
        }
        if constexpr (NFULL < N1) {
#pragma unroll
            for (int s = NFULL; s < N1; ++s) {
                if (s < KSr) {
                    const double b = Kt[64 * s + l];
                    if (s & 1) acc1 = CBF_MFMA(Breg[0][s], b, acc1);
                    else acc0 = CBF_MFMA(Breg[0][s], b, acc0);
                }
            }
        }
        const d4 A = acc0 + acc1;
        double q = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            At[256 * RBI + 64 * r + l] = A[r];
            q = fma(A[r], A[r], q);
        }
        flag_release(flag + RBI, epoch);
        // A2 rows of this block: W^T[RBI, RBI..] A[RBI..]                                        (gp_tf.py:145)
        // (the accumulator of this wave's own A rows is already the B operand of their k-steps: C layout = B layout)
        d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (EXACT) {
                if (4 * RBI + r < KSE) {
                    if (r & 1) c1 = CBF_MFMA(Breg[0][N1 + r], A[r], c1);
                    else c0 = CBF_MFMA(Breg[0][N1 + r], A[r], c0);
                }
            } else {
                if (RBI < NBLK - 1 || 4 * RBI + r < KSr) {
                    if (r & 1) c1 = CBF_MFMA(Breg[0][N1 + r], A[r], c1);
                    else c0 = CBF_MFMA(Breg[0][N1 + r], A[r], c0);
                }
            }
        }
#pragma unroll
        for (int kb = RBI + 1; kb < NBLK; ++kb) {
            if (4 * kb < KSE) {
                flag_wait(flag, kb, epoch);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = 4 * kb + r;
                    bool go;
                    if constexpr (EXACT) go = s < KSE;
                    else go = (kb < NBLK - 1) || (s < KSr);
                    if (go) {
                        const double b = At[64 * s + l];
                        if (r & 1) c1 = CBF_MFMA(Breg[0][N1 + 4 * (kb - RBI) + r], b, c1);
                        else c0 = CBF_MFMA(Breg[0][N1 + 4 * (kb - RBI) + r], b, c0);
                    }
                }
            }
        }
        d4 a2[RB];
        a2[0] = c0 + c1;
        tri_epilogue(a2, q, part, slot, RBI, l, a2o);
    }
    template <int I>
    __device__ __forceinline__ void phase2_tri_dispatch(const double* Kt, double* At, int* flag, int epoch, double* part,
                                                        int rbu, int slot, int l, double* a2o)
    {
        if constexpr (I < NBLK) {
            if (rbu == I) phase2_tri_reg<I>(Kt, At, flag, epoch, part, l, slot, a2o);
            else phase2_tri_dispatch<I + 1>(Kt, At, flag, epoch, part, rbu, slot, l, a2o);
        }
    }

    // operands streamed from L2 (NBLK >= 10): W and W^T as lane-linear A-operand images, zero blocks skipped
    __device__ __forceinline__ void phase2_tri_stream(const double* Kt, double* At, int* flag, int epoch, double* part,
                                                      int w, int l, double* a2o)
    {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                const int n1 = min(4 * (rb + 1), KSr);
                const double* ap = Wp + (rb * KS) * 64 + l;
#pragma unroll 1
                for (int s0 = 0; s0 < n1; s0 += 4) {
                    double b[4], aop[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        aop[j] = ap[(s0 + j) * 64];
                        b[j] = Kt[64 * (s0 + j) + l];
                    }
                    acc0 = CBF_MFMA(aop[0], b[0], acc0);
                    acc1 = CBF_MFMA(aop[1], b[1], acc1);
                    acc0 = CBF_MFMA(aop[2], b[2], acc0);
                    acc1 = CBF_MFMA(aop[3], b[3], acc1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double av = acc0[r] + acc1[r];
                    At[256 * rb + 64 * r + l] = av;
                    q = fma(av, av, q);
                }
            }
        }
        flag_release(flag + w, epoch);
        d4 c[RB][2];
#pragma unroll
        for (int i = 0; i < RB; ++i) { c[i][0] = d4{0, 0, 0, 0}; c[i][1] = d4{0, 0, 0, 0}; }
        const int kbe = (KSr + 3) >> 2;                       // k-blocks that carry data
#pragma unroll 1
        for (int kb = w * RB; kb < kbe; ++kb) {
            // W^T operands of this k-block first: they do not depend on the producer's flag
            double aop[RB][4];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int rb = min(w * RB + i, NBLK - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) aop[i][j] = WTp[(rb * KS + 4 * kb + j) * 64 + l];
            }
            const int pw = kb / RB;
            if (pw != w) flag_wait(flag, pw, epoch);
            double b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = At[64 * (4 * kb + j) + l];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int rb = w * RB + i;
                if (rb < NBLK && rb <= kb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[i][j & 1] = CBF_MFMA(aop[i][j], b[j], c[i][j & 1]);
                }
            }
        }
        d4 a2[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) a2[i] = c[i][0] + c[i][1];
        tri_epilogue(a2, q, part, w, w * RB, l, a2o);
    }

    __device__ __forceinline__ void phase2_tri(const double* Kt, double* At, int* flag, int epoch, double* part, int w, int l,
                                               double* a2o = nullptr)
    {
        if constexpr (BREG) phase2_tri_dispatch<0>(Kt, At, flag, epoch, part, __builtin_amdgcn_readfirstlane(rb_of(w, 0)), w, l, a2o);
        else phase2_tri_stream(Kt, At, flag, epoch, part, w, l, a2o);
    }

    // phases 1 and 2 for NC column blocks of 16 points whose scaled inputs sit in xq[c]; leaves the P1/P2 partials
    // of this wave in part[c].  Contains one workgroup barrier; the caller must barrier before reading `part` and
    // before rewriting xq.  With NC = 2 every K^-1 operand feeds two MFMAs (two independent accumulator chains).
    template <int NC>
    __device__ __forceinline__ void gp_phases(const double* xq, double* Kt, double* part, int w, int l,
                                              double* a2o = nullptr, int ncol_ok = NC)
    {
        constexpr int XS = DK * 64, KTS = MP * 16, PS = W * 512;
        // ---- phase 1: kernel tile rows of this wave
        double kreg[NC][RB][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double bx[DK];
            double xx = 0.0;
#pragma unroll
            for (int s = 0; s < DK; ++s) {
                bx[s] = xq[c * XS + 64 * s + l];
                xx = fma(bx[s], bx[s], xx);
            }
            xx += __shfl_xor(xx, 16);
            xx += __shfl_xor(xx, 32);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int rb = w * RB + i;
                if (rb < NBLK) {
                    d4 e;
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[r] = czr[i][r] - 0.5 * xx;
#pragma unroll
                    for (int s = 0; s < DK; ++s) e = CBF_MFMA(Zreg[i][s], bx[s], e);
                    const d4 ev = tile_exp4(e);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        kreg[c][i][r] = ev[r];
                        Kt[c * KTS + 256 * rb + 64 * r + l] = kreg[c][i][r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) kreg[c][i][r] = 0.0;
                }
            }
        }
        CBF_STAMP_GP_BARRIER();

        // ---- phase 2: A2 rows of this wave, then the P1/P2 products
        constexpr int two = (RB * NC == 1) ? 1 : 0;
        d4 acc[NC][RB][2];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < RB; ++i) { acc[c][i][0] = d4{0, 0, 0, 0}; acc[c][i][1] = d4{0, 0, 0, 0}; }
        if constexpr (BREG) {
#pragma unroll
            for (int s = 0; s < KSE; ++s) {
                double b[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) b[c] = Kt[c * KTS + 64 * s + l];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int rb = w * RB + i;
                    if (rb < NBLK) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            acc[c][i][(s & 1) * two] = CBF_MFMA(Breg[i][s], b[c], acc[c][i][(s & 1) * two]);
                    }
                }
            }
        } else {
            // K^-1 streamed from L2 as a lane-linear A-operand image: 512 contiguous bytes per (row block, k-step)
            static_assert(KS % 4 == 0, "KS must be a multiple of 4");
#pragma unroll 1
            for (int s0 = 0; s0 < KSr; s0 += 4) {
                double b[NC][4], aop[RB][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) b[c][j] = Kt[c * KTS + 64 * (s0 + j) + l];
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
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                acc[c][i][(j & 1) * two] = CBF_MFMA(aop[i][j], b[c][j], acc[c][i][(j & 1) * two]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            d4 P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0};
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int rb = w * RB + i;
                if (rb < NBLK) {
                    const d4 a2 = acc[c][i][0] + acc[c][i][1];
                    if (a2o && c < ncol_ok) {
                        double* rec = a2o + c * (NBLK * 256 + koff) + rb * 256 + l;
#pragma unroll
                        for (int r = 0; r < 4; ++r) rec[r * 64] = a2[r];
                        if (koff) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) rec[koff + r * 64] = kreg[c][i][r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double mu_op = BREG ? muA[i][r] : muAg[(rb * 4 + r) * 64 + lane_];
                        const double s2_op = BREG ? s2A[i][r] : s2Ag[(rb * 4 + r) * 64 + lane_];
                        P1 = CBF_MFMA(mu_op, a2[r], P1);
                        P2 = CBF_MFMA(s2_op, a2[r] * a2[r], P2);
                        q = fma(kreg[c][i][r], a2[r], q);
                    }
                }
            }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                part[c * PS + ((w * 2 + 0) * 4 + r) * 64 + l] = P1[r];
                part[c * PS + ((w * 2 + 1) * 4 + r) * 64 + l] = P2[r] - q;
            }
        }
    }

    // ---- the same two phases as separate calls on ONE column block (used by the skewed two-group pipeline)
    __device__ __forceinline__ void phase1(const double* xq, double* Kt, double (&kreg)[RB][4], int w, int l)
    {
        double bx[DK];
        double xx = 0.0;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            bx[s] = xq[64 * s + l];
            xx = fma(bx[s], bx[s], xx);
        }
        xx += __shfl_xor(xx, 16);
        xx += __shfl_xor(xx, 32);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = rb_of(w, i);
            if (rb < NBLK) {
                d4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = czr[i][r] - 0.5 * xx;
#pragma unroll
                for (int s = 0; s < DK; ++s) e = CBF_MFMA(Zreg[i][s], bx[s], e);
                const d4 ev = tile_exp4(e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kreg[i][r] = ev[r];
                    Kt[256 * rb + 64 * r + l] = kreg[i][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) kreg[i][r] = 0.0;
            }
        }
    }

    __device__ __forceinline__ void phase2(const double* Kt, double* part, const double (&kreg)[RB][4], int w, int l,
                                           double* a2o = nullptr)
    {
        d4 acc[RB][2];
#pragma unroll
        for (int i = 0; i < RB; ++i) { acc[i][0] = d4{0, 0, 0, 0}; acc[i][1] = d4{0, 0, 0, 0}; }
        if constexpr (BREG) {
#pragma unroll
            for (int s = 0; s < KSE; ++s) {
                const double b = Kt[64 * s + l];
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    if (w * RB + i < NBLK) acc[i][s & 1] = CBF_MFMA(Breg[i][s], b, acc[i][s & 1]);
            }
        } else if constexpr (RB == 2) {
            // (a wave whose second row block does not exist runs it on a copy of the last block; the result is dropped)
            const int rb0 = min(w * 2, NBLK - 1), rb1 = min(w * 2 + 1, NBLK - 1);
            stream_kinv_rb2<512>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], Bp + (rb0 * KS) * 64 + l,
                                 Bp + (rb1 * KS) * 64 + l, lds_addr(Kt + l), (KSr + 3) >> 2);
        } else {
#pragma unroll 1
            for (int s0 = 0; s0 < KSr; s0 += 4) {
                double b[4], aop[RB][4];
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
                        if (w * RB + i < NBLK) acc[i][j & 1] = CBF_MFMA(aop[i][j], b[j], acc[i][j & 1]);
            }
        }
        d4 P1 = {0, 0, 0, 0}, P2 = {0, 0, 0, 0};
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int rb = w * RB + i;
            if (rb < NBLK) {
                const d4 a2 = acc[i][0] + acc[i][1];
                if (a2o) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a2o[rb * 256 + r * 64 + l] = a2[r];
                    if (koff) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) a2o[koff + rb * 256 + r * 64 + l] = kreg[i][r];
                    }
                }
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
template <int NBLK, int RB, int DK, bool BREG, bool TRI = false, int KT = -1>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void predict_kernel(PredictArgs a)
{
    typedef Tile<NBLK, RB, DK, BREG, TRI, KT> TT;
    extern __shared__ double lds[];
    double* xq = lds;
    double* Kt = xq + DK * 64;
    double* part = Kt + TT::MP * 16;
    double* At = part + TT::W * 512 + 64;            // TRI: A = L^-1 k tile, then the waves' flags
    int* flag = reinterpret_cast<int*>(At + TT::MP * 16);
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
    if constexpr (TRI) {
        if (tid < 64) { flag[tid] = 0; flag[tid + 64] = 0; }
    }
    __syncthreads();
    double kr[RB][4];
    tile.phase1(xq, Kt, kr, w, l);
    __syncthreads();
    double* a2o = a.a2o ? a.a2o + int64_t(blockIdx.x) * (NBLK * 256) : nullptr;
    if constexpr (TRI) tile.phase2_tri(Kt, At, flag, 1, part, w, l, a2o);
    else tile.phase2(Kt, part, kr, w, l, a2o);
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
// NC: column blocks (16 chains each) per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
template <int NBLK, int RB, int DK, bool BREG, int MODE, int NC, bool TRI = false, int KT = -1>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void pass_kernel(PassArgs a)
{
    static_assert(!TRI || NC == 1, "the two-triangular form runs one column block per workgroup");
    typedef Tile<NBLK, RB, DK, BREG, TRI, KT> TT;
    constexpr int W = TT::W, NT = TT::NT;
    constexpr int NTASK = 4 * NC;                    // (state-row group q, column block) pairs of phase 3
    constexpr int QPW = (NTASK + W - 1) / W;
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    constexpr int XS = DK * 64, KTS = TT::MP * 16, PS = W * 512;
    extern __shared__ double lds[];
    double* xq = lds;                                // [NC][DK*64]
    double* Kt = xq + NC * XS;                       // [NC][MP*16]
    double* part = Kt + NC * KTS;                    // [NC][W][512]
    double* red = part + NC * PS;
    double* At = red + 64 + ((TT::EPI_LDS && NC == 1) ? TT::EPI_LDS_DOUBLES : 0);   // TRI: A = L^-1 k tile, then the flags
    int* flag = reinterpret_cast<int*>(At + TT::MP * 16);

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, nl = l & 15;
    const int N = a.N, S = a.S, T = a.T, Do = a.Do;
    const int naux = a.D - Do;                       // rows of the GP input that are not chain state
    const int gx = blockIdx.x + a.group0;            // chain group of this workgroup
    const int c0 = gx * 16 * NC;
    const int G16 = (N + 15) >> 4;

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
            if (tid == 0) {   // partial sums are indexed per 16-chain group, whatever the kernel variant
                const int G16p = (a.N + 15) >> 4;
                for (int cq = 0; cq < NC; ++cq)
                    if (gx * NC + cq < G16p) a.part_out[blockIdx.y * G16p + gx * NC + cq] = (cq == 0) ? 0.0 : 0.0;
            }
            return;
        }
    }

    TT tile;
    tile.load_operands(a.pk, w, l);
    tile.koff = a.ksave ? NBLK * 256 : 0;
    if constexpr (TT::EPI_LDS && NC == 1) {
        // streamed-K^-1 tiles re-read the mean / variance operand images every step: from LDS (100 KB are free next to
        // the tiles) instead of through an L1 that the K^-1 stream keeps flushing
        double* mul = red + 64;
        for (int i = tid; i < NBLK * 256; i += NT) mul[i] = a.pk.muA[i];
        tile.muAg = mul;
        if constexpr (TT::EPI_LDS_N == 2) {
            double* s2l = mul + NBLK * 256;
            for (int i = tid; i < NBLK * 256; i += NT) s2l[i] = a.pk.s2A[i];
            tile.s2Ag = s2l;
        }
        // (the barrier that publishes the first step's inputs below also orders these writes)
    }

    // per-lane constants of the phase-3 tasks of this wave: task tk = w + qi*W -> (q = tk & 3, column block tk >> 2)
    double vx[QPW], vy[QPW], il[QPW];
    double hcur[QPW];
    double lin[QPW];
    LogProd lp[QPW];
    bool act[QPW], cval[QPW];
    int cc[QPW], bqv[QPW], qv[QPW], cbv[QPW];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int tk = w + qi * W;
        qv[qi] = tk & 3;
        cbv[qi] = (tk >> 2) < NC ? (tk >> 2) : (NC - 1);
        const int d = 4 * qv[qi] + g;
        const int cg = c0 + cbv[qi] * 16 + nl;
        cc[qi] = min(cg, N - 1);
        cval[qi] = cg < N;
        bqv[qi] = cc[qi] / S;
        act[qi] = (tk < NTASK) && (d < Do);
        const int dc = act[qi] ? d : 0;
        vx[qi] = a.var_x[dc];
        vy[qi] = (MODE == MODE_FWD) ? a.var_y[(a.half && dc >= a.dim_y) ? 0 : dc] : 0.0;   // half: var_y has dim_y entries
        il[qi] = a.pk.invl[dc];
        lin[qi] = 0.0;
        lp[qi].init();
        hcur[qi] = 0.0;
    }

    // auxiliary (non-state) input rows: fwd u_t; bwd [u_t, y_t]           (cbfssm.py:137,197)
    // auxiliary input rows of this thread: base pointer, time stride and 1/lengthscale are fixed for the whole pass
    const double* auxp[NC][AUXR];
    int auxs[NC][AUXR];
    double auxl[NC][AUXR];
#pragma unroll
    for (int cb = 0; cb < NC; ++cb)
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT, ja = i >> 4, n = i & 15;
            auxp[cb][k2] = a.pk.invl; auxs[cb][k2] = 0; auxl[cb][k2] = 0.0;   // (no row: a valid dummy address, factor 0)
            if (i < 16 * naux) {
                const int b = min(c0 + cb * 16 + n, N - 1) / S;
                if (ja < a.dim_u) { auxp[cb][k2] = a.u + int64_t(b) * T * a.dim_u + ja; auxs[cb][k2] = a.dim_u; }
                else { auxp[cb][k2] = a.y + int64_t(b) * T * a.dim_y + (ja - a.dim_u); auxs[cb][k2] = a.dim_y; }
                auxl[cb][k2] = a.pk.invl[Do + ja];
            }
        }
    // The raw value: the 1/lengthscale factor is applied where the row is written to LDS.  (A multiply right behind the
    // load makes the compiler wait for it -- and, vmcnt being in-order, for the noise / observation loads issued before
    // it -- at the top of every step, on the waves everybody then waits for at the barrier.)
    auto aux_load = [&](int cb, int k2, int t) -> double { return auxp[cb][k2][int64_t(t) * auxs[cb][k2]]; };

    // ---- initial state and first input
    // xq rows [0,Do) carry the chain state, rows [Do,D) the auxiliary inputs, rows [D,4*DK) stay zero
    for (int i = tid; i < NC * XS; i += NT) xq[i] = 0.0;
    if constexpr (TRI) {
        if (tid < 64) { flag[tid] = 0; flag[tid + 64] = 0; }
    }
    __syncthreads();
    const bool resample0 = (MODE == MODE_BWD) && (((t_first + 1 + run * R) % P) == 0);
    const int tm0 = (MODE == MODE_BWD) ? (t_first % P) : 0;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int d = 4 * qv[qi] + g;
        const int c = cc[qi];
        if (act[qi]) {
            double v;
            if (MODE == MODE_FWD) {
                // x_0 = y_tilde[:, 0] = [y_0, y2_0]  (cbfssm.py:97,168); half: recognition model (cbfssmhalf.py:106)
                if (a.half) v = a.x0[int64_t(bqv[qi]) * a.dim_x + d];
                else v = (d < a.dim_y) ? a.y[(int64_t(bqv[qi]) * T) * a.dim_y + d]
                                       : a.y2_in[int64_t(c) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                if (cval[qi]) a.x_out[int64_t(c) * a.dim_x + d] = v;
            } else {
                v = resample0 ? a.hid[(int64_t(run) * T + t_first) * N + c] : 0.0;   // cbfssm.py:106,133-136
            }
            hcur[qi] = v;
            xq[cbv[qi] * XS + 64 * qv[qi] + l] = v * il[qi];
        }
    }
    double auxr[NC][AUXR];
#pragma unroll
    for (int cb = 0; cb < NC; ++cb)
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT;
            if (i < 16 * naux) xq[cb * XS + 16 * Do + i] = aux_load(cb, k2, t_first) * auxl[cb][k2];
        }

    CBF_STAMP_DECL;
    CBF_STAMP_START();
    for (int step = 0; step < nsteps; ++step) {
        const int t = t_first + dir * step;
        const int tn = t + dir;                       // time index of the next GP input
        // t mod 2R and (t - 1) mod 2R without a runtime modulo per step (a segment has at most 2R steps)
        int tmod = tm0 - step; if (tmod < 0) tmod += P;
        const int tmn = (tmod == 0) ? P - 1 : tmod - 1;
        const bool has_next = (step + 1 < nsteps);
        CBF_STAMP_BARRIER(0);                         // xq complete

        // ---- prefetch this step's epilogue inputs and the next step's auxiliary rows
        double eps_t[QPW], ytil[QPW], hidn[QPW];
        bool resample_n = false;
        if (MODE == MODE_BWD) resample_n = has_next && (tmn + 1 + run * R == P);            // cbfssm.py:124,127
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int c = cc[qi];
            const int d = 4 * qv[qi] + g;
            ytil[qi] = 0.0; hidn[qi] = 0.0;
            if (MODE == MODE_FWD) {
                eps_t[qi] = a.eps[int64_t(t) * N + c];                                       // cbfssm.py:209
                if (act[qi]) {
                    if (d < a.dim_y) ytil[qi] = a.y[(int64_t(bqv[qi]) * T + (t + 1)) * a.dim_y + d];      // cbfssm.py:196
                    else if (!a.half) ytil[qi] = a.y2_in[(int64_t(t + 1) * N + c) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                }
            } else {
                eps_t[qi] = a.eps[(int64_t(run) * T + t) * N + c];                           // cbfssm.py:149
                if (resample_n) hidn[qi] = a.hid[(int64_t(run) * T + tn) * N + c];
            }
        }
#pragma unroll
        for (int cb = 0; cb < NC; ++cb)
#pragma unroll
            for (int k2 = 0; k2 < AUXR; ++k2) {
                const int i = tid + k2 * NT;
                auxr[cb][k2] = has_next ? aux_load(cb, k2, tn) : 0.0;
            }

        double* a2o = nullptr;                        // this step's A2 tiles, kept for the adjoint
        if (a.a2s) {
            const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
            a2o = a.a2s + (slot * G16 + int64_t(gx) * NC) * (NBLK * 256 + tile.koff);
        }
        if constexpr (NC == 1) {
            double kr[RB][4];
            tile.phase1(xq, Kt, kr, w, l);
            // The epilogue inputs prefetched above are made to land HERE, a phase after they were issued and in front of
            // the tile stores of phase 2.  vmcnt retires in order and counts stores: left to the first use in phase 3, the
            // wait the compiler can prove (a count that must also hold when no tiles are kept) waits in a train step for
            // six of this step's eight tile stores to complete -- on the lanes that carry the step's serial chain.
#pragma unroll
            for (int qi = 0; qi < QPW; ++qi) asm volatile("" : "+v"(eps_t[qi]), "+v"(ytil[qi]), "+v"(hidn[qi]));
#pragma unroll
            for (int cb = 0; cb < NC; ++cb)
#pragma unroll
                for (int k2 = 0; k2 < AUXR; ++k2) asm volatile("" : "+v"(auxr[cb][k2]));
            if constexpr (TRI) {
                // (the dense form's phase 2 writes the kernel tile next to its A2 rows; the two-triangular one does not see it)
                if (a2o && tile.koff) {
#pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        const int rb = tile.rb_of(w, i);
                        if (rb < NBLK) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) a2o[tile.koff + rb * 256 + r * 64 + l] = kr[i][r];
                        }
                    }
                }
            }
            CBF_STAMP_BARRIER(1);
            if constexpr (TRI) tile.phase2_tri(Kt, At, flag, step + 1, part, w, l, a2o);
            else tile.phase2(Kt, part, kr, w, l, a2o);
        } else {
            tile.template gp_phases<NC>(xq, Kt, part, w, l, a2o, G16 - gx * NC);
        }
        CBF_STAMP_BARRIER(2);                         // part complete; xq and Kt free

        // ---- phase 3
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int tk = w + qi * W;
            if (tk < NTASK) {
                const int q = qv[qi], cb = cbv[qi], c = cc[qi];
                double fm, fv;
                tile.gather(part + cb * PS, q, l, fm, fv);
                const int d = 4 * q + g;
                double outv = 0.0;
                if (act[qi]) {
                    const double fmean = fm + hcur[qi];                                // cbfssm.py:145,205
                    const double fvar = fv + vx[qi];                                   // cbfssm.py:146,206
                    if (a.fmv && cval[qi]) {
                        const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                        double* o = a.fmv + ((slot * N + c) * Do + d) * 2;
                        o[0] = fmean; o[1] = fvar;
                    }
                    if (MODE == MODE_FWD) {
                        const double vyt = vy[qi] + (a.k_factor - 1.0) * fvar;         // cbfssm.py:212-214
                        const double s = vyt + fvar;                                   // :216
                        const double rs = fast_rcp(s);
                        const double kk = fvar * rs;                                   // :217
                        const double ydiff = ytil[qi] - fmean;                         // :215
                        const double dm = kk * ydiff;
                        const double mu = fmean + dm;                                  // :218
                        // sig = (1-k)^2 fvar + k^2 v with 1 - k = v / s  ==  fvar v / s = k v             (:219-220)
                        const double sig = kk * vyt;
                        // half: the hidden dims (d >= dim_y) get no Kalman update: k = 0, mu = fmean, sig = fvar, KL = 0
                        const bool do_cond = (a.condition || (t < R - 1)) && !(a.half && d >= a.dim_y);     // :227
                        outv = do_cond ? (mu + eps_t[qi] * (sig * fast_rsqrt(sig)))                         // :221-229
                                       : (fmean + eps_t[qi] * (fvar * fast_rsqrt(fvar)));
                        if (do_cond && cval[qi]) {
                            // kl_reg = log fvar - log sig + (sig + (mu - fmean)^2)/fvar - 1        (:232)
                            // with sig / fvar = v / s = 1 - k and (mu - fmean)^2 / fvar = k ydiff^2 / s:
                            lin[qi] += kk * (ydiff * ydiff * rs - 1.0);
                            lp[qi].mul(vyt * rs);
                        }
                        if (cval[qi]) a.x_out[(int64_t(t + 1) * N + c) * a.dim_x + d] = outv;       // :229
                    } else {
                        outv = fmean + eps_t[qi] * (fvar * fast_rsqrt(fvar));          // cbfssm.py:150
                        const bool write = (run == 0) ? (tmod < R) : (tmod >= R);             // :125,128
                        if (cval[qi]) {
                            if (write) {
                                a.y2_out[(int64_t(t) * N + c) * Do + d] = outv;        // :151
                                lp[qi].mul(fvar);                                      // :154-156
                                lin[qi] += 1.0;
                            }
                            if (a.h_all) a.h_all[((int64_t(run) * T + t) * N + c) * Do + d] = outv;
                        }
                    }
                }
                const double hn = (MODE == MODE_BWD && resample_n) ? hidn[qi] : outv;  // cbfssm.py:133-136,158
                hcur[qi] = act[qi] ? hn : 0.0;
                if (has_next && act[qi]) xq[cb * XS + 64 * q + l] = hn * il[qi];
            }
        }
        if (has_next) {
#pragma unroll
            for (int cb = 0; cb < NC; ++cb)
#pragma unroll
                for (int k2 = 0; k2 < AUXR; ++k2) {
                    const int i = tid + k2 * NT;
                    if (i < 16 * naux) xq[cb * XS + 16 * Do + i] = auxr[cb][k2] * auxl[cb][k2];
                }
        }
    }

    // ---- per-workgroup partial of the regulariser
    double v = 0.0;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        if (act[qi] && cval[qi]) {
            if (MODE == MODE_FWD) v += 0.5 * (lin[qi] - lp[qi].log());
            else v += 0.5 * (lin[qi] * 2.8378770664093453391 + lp[qi].log());          // log(2 pi e)
        }
    }
    double tot = block_sum(v, red, tid, NT);
    if constexpr (TRI) {
        if (tid == 0 && flag[CBF_FLAG_TIMEOUT_SLOT] != 0) tot = __builtin_nan("");   // a hand-off poll ran out (see flag_wait)
    }
    if (tid == 0) {   // partial sums are indexed per 16-chain group, whatever the kernel variant
        const int G16p = (a.N + 15) >> 4;
        for (int cq = 0; cq < NC; ++cq)
            if (gx * NC + cq < G16p) a.part_out[blockIdx.y * G16p + gx * NC + cq] = (cq == 0) ? tot : 0.0;
    }
#ifdef CBF_REV_STAMPS
#ifndef CBF_STAMP_WAVE_PASS
#define CBF_STAMP_WAVE_PASS (W - 1)     // the second wave whose phase shares are recorded (-DCBF_STAMP_WAVE_PASS=k picks another)
#endif
    if (a.dbg && l == 0 && (w == 0 || w == CBF_STAMP_WAVE_PASS)) {
        double* o = a.dbg + (int64_t(blockIdx.y) * a.gtotal + gx) * 64 + (w == 0 ? 0 : 32);
        for (int i = 0; i < 7; ++i) { o[i] = double(st_c[i]); o[7 + i] = double(st_w[i]); }
        for (int i = 0; i < 12; ++i) o[14 + i] = double(st_m[i]);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Skewed two-group pass kernel: one workgroup carries TWO column blocks (A, B: 16 chains each) half a step apart, so
// that every barrier interval pairs matrix work of one group with the VALU-heavy epilogue / kernel-tile work of the
// other (3 barriers per step of both groups instead of 6):
//     alpha:  phase2(A, s)              ||  phase3(B, s-1)
//     beta :  phase3(A, s)              ||  phase1(B, s)
//     gamma:  phase1(A, s+1)            ||  phase2(B, s)
// Phase-3 work of A sits on the low waves, of B on the high waves (SIMD partners w, w+4 get different mixes).
// ---------------------------------------------------------------------------------------------------------------------
template <int NBLK, int RB, int DK, bool BREG, int MODE, int KT = -1>
__global__ __launch_bounds__(64 * ((NBLK + RB - 1) / RB)) void pass_kernel_skew(PassArgs a)
{
    typedef Tile<NBLK, RB, DK, BREG, false, KT> TT;
    constexpr int W = TT::W, NT = TT::NT;
    constexpr int QPW = TT::QPW;                     // phase-3 tasks per wave and group
    constexpr int AUXR = (DK * 64 + NT - 1) / NT;
    constexpr int XS = DK * 64, KTS = TT::MP * 16, PS = W * 512;
    extern __shared__ double lds[];
    double* xq = lds;                                // [2][DK*64]
    double* Kt = xq + 2 * XS;                        // [2][MP*16]
    double* part = Kt + 2 * KTS;                     // [2][W][512]
    double* red = part + 2 * PS;

    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, nl = l & 15;
    const int N = a.N, S = a.S, T = a.T, Do = a.Do;
    const int naux = a.D - Do;
    const int gx = blockIdx.x + a.group0;
    const int c0 = gx * 32;

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
            if (tid == 0) {   // partial sums are indexed per 16-chain group, whatever the kernel variant
                const int G16p = (a.N + 15) >> 4;
                for (int cq = 0; cq < 2; ++cq)
                    if (gx * 2 + cq < G16p) a.part_out[blockIdx.y * G16p + gx * 2 + cq] = (cq == 0) ? 0.0 : 0.0;
            }
            return;
        }
    }

    TT tile;
    tile.load_operands(a.pk, w, l);
    tile.koff = a.ksave ? NBLK * 256 : 0;

    // phase-3 tasks: group c, row group q = wq + qi*W (< 4) with wq = w for group A and W-1-w for group B
    double vx[2][QPW], vy[2][QPW], il[2][QPW], hcur[2][QPW], lin[2][QPW];
    LogProd lp[2][QPW];
    bool act[2][QPW], cval[2];
    int cc[2], bqv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int cg = c0 + c * 16 + nl;
        cc[c] = min(cg, N - 1);
        cval[c] = cg < N;
        bqv[c] = cc[c] / S;
        const int wq = (c == 0) ? w : (W - 1 - w);
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = wq + qi * W;
            const int d = 4 * q + g;
            act[c][qi] = (q < 4) && (d < Do);
            const int dc = act[c][qi] ? d : 0;
            vx[c][qi] = a.var_x[dc];
            vy[c][qi] = (MODE == MODE_FWD) ? a.var_y[dc] : 0.0;
            il[c][qi] = a.pk.invl[dc];
            lin[c][qi] = 0.0;
            lp[c][qi].init();
            hcur[c][qi] = 0.0;
        }
    }

    // auxiliary input rows of this thread: base pointer, time stride and 1/lengthscale are fixed for the whole pass
    const double* auxp[2][AUXR];
    int auxs[2][AUXR];
    double auxl[2][AUXR];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT, ja = i >> 4, n = i & 15;
            auxp[cb][k2] = a.pk.invl; auxs[cb][k2] = 0; auxl[cb][k2] = 0.0;   // (no row: a valid dummy address, factor 0)
            if (i < 16 * naux) {
                const int b = min(c0 + cb * 16 + n, N - 1) / S;
                if (ja < a.dim_u) { auxp[cb][k2] = a.u + int64_t(b) * T * a.dim_u + ja; auxs[cb][k2] = a.dim_u; }
                else { auxp[cb][k2] = a.y + int64_t(b) * T * a.dim_y + (ja - a.dim_u); auxs[cb][k2] = a.dim_y; }
                auxl[cb][k2] = a.pk.invl[Do + ja];
            }
        }
    // (the raw value; the 1/lengthscale factor is applied at the LDS write, see pass_kernel)
    auto aux_load = [&](int cb, int k2, int t) -> double { return auxp[cb][k2][int64_t(t) * auxs[cb][k2]]; };

    // ---- initial state and first input of both groups
    for (int i = tid; i < 2 * XS; i += NT) xq[i] = 0.0;
    __syncthreads();
    const bool resample0 = (MODE == MODE_BWD) && (((t_first + 1 + run * R) % P) == 0);
    const int tm0 = (MODE == MODE_BWD) ? (t_first % P) : 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int wq = (c == 0) ? w : (W - 1 - w);
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = wq + qi * W;
            const int d = 4 * q + g;
            if (act[c][qi]) {
                double v;
                if (MODE == MODE_FWD) {
                    v = (d < a.dim_y) ? a.y[(int64_t(bqv[c]) * T) * a.dim_y + d]
                                      : a.y2_in[int64_t(cc[c]) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                    if (cval[c]) a.x_out[int64_t(cc[c]) * a.dim_x + d] = v;
                } else {
                    v = resample0 ? a.hid[(int64_t(run) * T + t_first) * N + cc[c]] : 0.0;
                }
                hcur[c][qi] = v;
                xq[c * XS + 64 * q + l] = v * il[c][qi];
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT;
            if (i < 16 * naux) xq[c * XS + 16 * Do + i] = aux_load(c, k2, t_first) * auxl[c][k2];
        }
    }
    __syncthreads();

    double kregA[RB][4], kregB[RB][4];

    // phase 3 of group c for step index s (time t): epilogue + next input of that group
    auto phase3 = [&](auto cidx, int s) {
        constexpr int c = decltype(cidx)::value;
        const int t = t_first + dir * s;
        const int tn = t + dir;
        int tmod = tm0 - s; if (tmod < 0) tmod += P;      // t mod 2R, (t - 1) mod 2R (a segment has at most 2R steps)
        const int tmn = (tmod == 0) ? P - 1 : tmod - 1;
        const bool has_next = (s + 1 < nsteps);
        const int wq = (c == 0) ? w : (W - 1 - w);
        const int cch = cc[c];
        bool resample_n = false;
        if (MODE == MODE_BWD) resample_n = has_next && (tmn + 1 + run * R == P);
        double eps_t, hidn = 0.0;
        if (MODE == MODE_FWD) eps_t = a.eps[int64_t(t) * N + cch];
        else {
            eps_t = a.eps[(int64_t(run) * T + t) * N + cch];
            if (resample_n) hidn = a.hid[(int64_t(run) * T + tn) * N + cch];
        }
        double auxn[AUXR];
#pragma unroll
        for (int k2 = 0; k2 < AUXR; ++k2) {
            const int i = tid + k2 * NT;
            auxn[k2] = has_next ? aux_load(c, k2, tn) : 0.0;
        }
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = wq + qi * W;
            if (q < 4) {
                const int d = 4 * q + g;
                double ytil = 0.0;
                if (MODE == MODE_FWD && act[c][qi])
                    ytil = (d < a.dim_y) ? a.y[(int64_t(bqv[c]) * T + (t + 1)) * a.dim_y + d]
                                         : a.y2_in[(int64_t(t + 1) * N + cch) * (a.dim_x - a.dim_y) + (d - a.dim_y)];
                double fm, fv;
                tile.gather(part + c * PS, q, l, fm, fv);
                double outv = 0.0;
                if (act[c][qi]) {
                    const double fmean = fm + hcur[c][qi];
                    const double fvar = fv + vx[c][qi];
                    if (a.fmv && cval[c]) {
                        const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
                        double* o = a.fmv + ((slot * N + cch) * Do + d) * 2;
                        o[0] = fmean; o[1] = fvar;
                    }
                    if (MODE == MODE_FWD) {
                        const double vyt = vy[c][qi] + (a.k_factor - 1.0) * fvar;
                        const double sm = vyt + fvar;
                        const double rs = fast_rcp(sm);
                        const double kk = fvar * rs;
                        const double ydiff = ytil - fmean;
                        const double mu = fmean + kk * ydiff;
                        const double sig = kk * vyt;                  // = (1-k)^2 fvar + k^2 v   (1 - k = v / s)
                        const bool do_cond = a.condition || (t < R - 1);
                        outv = do_cond ? (mu + eps_t * (sig * fast_rsqrt(sig))) : (fmean + eps_t * (fvar * fast_rsqrt(fvar)));
                        if (do_cond && cval[c]) {
                            lin[c][qi] += kk * (ydiff * ydiff * rs - 1.0);     // (sig + (mu - fmean)^2) / fvar - 1
                            lp[c][qi].mul(vyt * rs);                           // sig / fvar
                        }
                        if (cval[c]) a.x_out[(int64_t(t + 1) * N + cch) * a.dim_x + d] = outv;
                    } else {
                        outv = fmean + eps_t * (fvar * fast_rsqrt(fvar));
                        const bool write = (run == 0) ? (tmod < R) : (tmod >= R);
                        if (cval[c]) {
                            if (write) {
                                a.y2_out[(int64_t(t) * N + cch) * Do + d] = outv;
                                lp[c][qi].mul(fvar);
                                lin[c][qi] += 1.0;
                            }
                            if (a.h_all) a.h_all[((int64_t(run) * T + t) * N + cch) * Do + d] = outv;
                        }
                    }
                }
                const double hn = (MODE == MODE_BWD && resample_n) ? hidn : outv;
                hcur[c][qi] = act[c][qi] ? hn : 0.0;
                if (has_next && act[c][qi]) xq[c * XS + 64 * q + l] = hn * il[c][qi];
            }
        }
        if (has_next) {
#pragma unroll
            for (int k2 = 0; k2 < AUXR; ++k2) {
                const int i = tid + k2 * NT;
                if (i < 16 * naux) xq[c * XS + 16 * Do + i] = auxn[k2] * auxl[c][k2];
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    const int G16 = (N + 15) >> 4;
    auto a2slot = [&](int s, int c) -> double* {      // A2 tiles of step s, column block c (kept for the adjoint)
        if (!a.a2s || 2 * gx + c >= G16) return nullptr;
        const int t = t_first + dir * s;
        const int64_t slot = (MODE == MODE_FWD) ? int64_t(t) : (int64_t(run) * T + t);
        return a.a2s + (slot * G16 + 2 * gx + c) * (NBLK * 256 + tile.koff);
    };
    tile.phase1(xq, Kt, kregA, w, l);                                   // phase1(A, 0)
    __syncthreads();
    // SIMD partners (waves w and w+4) run the two halves of an interval in opposite order, so that one is in its
    // matrix part while the other is in its VALU part
    const bool hi = (w >= 4);
    for (int s = 0; s < nsteps; ++s) {
        // alpha: phase2(A, s) || phase3(B, s-1)   (B's epilogue lanes sit on the high waves)
        if (hi) {
            if (s > 0) phase3(I1{}, s - 1);
            tile.phase2(Kt, part, kregA, w, l, a2slot(s, 0));
        } else {
            tile.phase2(Kt, part, kregA, w, l, a2slot(s, 0));
            if (s > 0) phase3(I1{}, s - 1);
        }
        __syncthreads();
        // beta: phase1(B, s) || phase3(A, s)      (A's epilogue lanes sit on the low waves)
        if (hi) {
            tile.phase1(xq + XS, Kt + KTS, kregB, w, l);
            phase3(I0{}, s);
        } else {
            phase3(I0{}, s);
            tile.phase1(xq + XS, Kt + KTS, kregB, w, l);
        }
        __syncthreads();
        // gamma: phase2(B, s) || phase1(A, s+1)
        if (hi) {
            tile.phase2(Kt + KTS, part + PS, kregB, w, l, a2slot(s, 1));
            if (s + 1 < nsteps) tile.phase1(xq, Kt, kregA, w, l);
        } else {
            if (s + 1 < nsteps) tile.phase1(xq, Kt, kregA, w, l);
            tile.phase2(Kt + KTS, part + PS, kregB, w, l, a2slot(s, 1));
        }
        __syncthreads();
    }
    phase3(I1{}, nsteps - 1);                                           // phase3(B, last)

    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            if (act[c][qi] && cval[c]) {
                if (MODE == MODE_FWD) v += 0.5 * (lin[c][qi] - lp[c][qi].log());
                else v += 0.5 * (lin[c][qi] * 2.8378770664093453391 + lp[c][qi].log());
            }
        }
    const double tot = block_sum(v, red, tid, NT);
    if (tid == 0) {   // partial sums are indexed per 16-chain group, whatever the kernel variant
        const int G16p = (a.N + 15) >> 4;
        for (int cq = 0; cq < 2; ++cq)
            if (gx * 2 + cq < G16p) a.part_out[blockIdx.y * G16p + gx * 2 + cq] = (cq == 0) ? tot : 0.0;
    }
}

}  // namespace cbfssm
