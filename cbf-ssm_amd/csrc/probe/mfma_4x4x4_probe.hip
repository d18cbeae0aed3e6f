// Operand layout and rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (diagnostic; not part of the library).
// For every lane la: A = one-hot at la, B[lb] = 1000 + lb  ->  the lanes ld with D != 0 and the lb they saw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void onehot(int la, double* out)
{
    const int l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0;
    const double b = 1000.0 + l;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

__global__ void rate(double* out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

int main()
{
    double* d;
    hipMalloc(&d, 64 * sizeof(double) * 4096);
    std::vector<double> h(64);
    printf("la -> (ld: lb) pairs\n");
    for (int la = 0; la < 64; ++la) {
        hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, 0, la, d);
        hipMemcpy(h.data(), d, 64 * sizeof(double), hipMemcpyDeviceToHost);
        printf("A lane %2d:", la);
        for (int ld = 0; ld < 64; ++ld)
            if (h[ld] != 0.0) printf(" D%d<-B%d", ld, int(h[ld] - 1000.0 + 0.5));
        printf("\n");
    }
    // rate: one wave per SIMD (4 waves per CU, 256 CUs)
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {
        hipLaunchKernelGGL(rate, dim3(256), dim3(256 * waves), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(256), dim3(256 * waves), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = 4.0 * iters * waves;
        printf("4x4x4_4b f64: %d wave(s) per SIMD: %.1f cycles per MFMA per SIMD at 2.4 GHz\n", waves, ms * 1e-3 * 2.4e9 / mfma_per_simd);
    }
    return 0;
}
