// Probe: do f64 MFMA and f64 VALU (v_fma_f64) execute concurrently on one SIMD, or do they share the DP datapath?
// Block = 512 threads = 8 waves = 2 per SIMD.  mode 0: all waves MFMA; 1: all waves FMA; 2: waves 0-3 MFMA, 4-7 FMA
// (SIMD partners run different pipes); 3: same as 2 with f32 FMA instead of f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters)
{
    const int w = threadIdx.x >> 6;
    const bool do_mfma = (MODE == 0) || (MODE >= 2 && w < 4);
    const bool do_fma = (MODE == 1) || (MODE == 2 && w >= 4);
    const bool do_fma32 = (MODE == 3 && w >= 4);
    double s = 0;
    if (do_mfma) {
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        }
        s = c0[0] + c1[1];
    } else if (do_fma) {
        double x[8];
        for (int j = 0; j < 8; ++j) x[j] = j + threadIdx.x * 1e-3;
        const double m = 1.0000001, ad = 1e-9;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fma(x[j], m, ad);     // 32 v_fma_f64 per iteration (= 128 cycles)
        }
        for (int j = 0; j < 8; ++j) s += x[j];
    } else if (do_fma32) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = j + threadIdx.x * 1e-3f;
        const float m = 1.0000001f, ad = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], m, ad);    // 64 v_fma_f32 per iteration (= 128 cycles)
        }
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> float run(double* d, int blocks, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, 512>>>(d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<MODE><<<blocks, 512>>>(d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main()
{
    double* d; CK(hipMalloc(&d, sizeof(double) * 512 * 256));
    const int iters = 20000, blocks = 256;
    float t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters), t3 = run<3>(d, blocks, iters);
    // per SIMD: mode 0: 2 waves x 2 MFMA x iters x 64 cyc = 256 cyc/iter ; mode 1: 2 waves x 32 fma x 4 cyc = 256 cyc/iter
    // mode 2: one MFMA wave (128 cyc/iter) + one FMA wave (128 cyc/iter): 128 cyc/iter if concurrent, 256 if shared
    printf("all-MFMA      %.3f ms (expect ~%.3f)\n", t0, iters * 256 / 2.4e6);
    printf("all-FMA64     %.3f ms (expect ~%.3f)\n", t1, iters * 256 / 2.4e6);
    printf("MFMA | FMA64  %.3f ms (%.3f if concurrent, %.3f if the DP datapath is shared)\n", t2, iters * 128 / 2.4e6, iters * 256 / 2.4e6);
    printf("MFMA | FMA32  %.3f ms (%.3f if concurrent)\n", t3, iters * 128 / 2.4e6);
    return 0;
}
