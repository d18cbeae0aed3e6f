// Probe for gfx950: (1) lane layout of v_mfma_f64_16x16x4_f64 checked with asymmetric integer data,
// (2) issue / dependent-chain cost of the f64 MFMA, v_fma_f64 and the software f64 exp/log.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// One wave: D = A(16x4) * B(4x16) with the assumed operand layout; writes every lane's 4 results raw.
__global__ void layout_kernel(const double* A, const double* B, double* raw)
{
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];      // A[i = l&15][k = l>>4], row-major 16x4
    double b = B[(l >> 4) * 16 + (l & 15)];     // B[k = l>>4][j = l&15], row-major 4x16
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) raw[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void mfma_rate_kernel(double* out, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// float32 matrix instruction of the float32-arithmetic path (cbfssm_f32.hip, cbfssm_rev32.hip): v_mfma_f32_16x16x4_f32
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void mfma32_rate_kernel(double* out, int iters)
{
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void fma_rate_kernel(double* out, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
__global__ void func_rate_kernel(double* out, int iters)
{
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = 0.5 + 0.001 * threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (OP == 0) x[i] = exp(-x[i]) + 0.5;
            if (OP == 1) x[i] = log(x[i]) + 2.5;
            if (OP == 2) x[i] = sqrt(x[i]) + 1.5;
            if (OP == 3) x[i] = 1.0 / x[i] + 1.5;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

template <typename F>
static float time_ms(F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    // ---- layout ----
    std::vector<double> A(64), B(64), raw(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;        // asymmetric
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 7;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *draw;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&draw, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, draw);
    CK(hipMemcpy(raw.data(), draw, 256 * 8, hipMemcpyDeviceToHost));
    int bad_guide = 0, bad_alt = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        double v = raw[l * 4 + r];
        int rowg = (l >> 4) + 4 * r, col = l & 15;           // guide: row = (lane>>4) + 4*reg
        int rowa = 4 * (l >> 4) + r;                          // f32-style: row = 4*(lane>>4) + reg
        if (v != ref[rowg * 16 + col]) ++bad_guide;
        if (v != ref[rowa * 16 + col]) ++bad_alt;
    }
    printf("LAYOUT f64 16x16x4: mismatches with row=(lane>>4)+4*reg: %d ; with row=4*(lane>>4)+reg: %d\n", bad_guide, bad_alt);
    if (bad_guide && bad_alt) {
        // print where lane 0..3,16,17 regs land
        for (int l : {0, 1, 16, 17, 32, 48}) for (int r = 0; r < 4; ++r) {
            double v = raw[l * 4 + r]; int fi = -1, fj = -1;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (ref[i * 16 + j] == v) { fi = i; fj = j; }
            printf("  lane %d reg %d -> (row %d, col %d)\n", l, r, fi, fj);
        }
    }

    // ---- rates ----
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    double* dout; CK(hipMalloc(&dout, sizeof(double) * 256 * 1024 * 16));
    int iters = 20000;
    for (int wpc : {4, 8, 16}) {   // waves per CU (block = 64 threads, blocks = cus * wpc)
        int blocks = cus * wpc;
        float m1 = time_ms([&] { mfma_rate_kernel<1><<<blocks, 64>>>(dout, iters); });
        float m2 = time_ms([&] { mfma_rate_kernel<2><<<blocks, 64>>>(dout, iters); });
        float m4 = time_ms([&] { mfma_rate_kernel<4><<<blocks, 64>>>(dout, iters); });
        auto tf = [&](float ms, int nacc) { return 2048.0 * nacc * iters * blocks / (ms * 1e-3) / 1e12; };
        auto cyc = [&](float ms, int nacc) { double wps = wpc / 4.0; return ms * 1e-3 * clk / (double(iters) * nacc * wps); };
        printf("MFMA f64 16x16x4  waves/CU %2d: 1acc %.1f TF (%.1f cyc/mfma/simd@nominal) 2acc %.1f TF (%.1f) 4acc %.1f TF (%.1f)\n",
               wpc, tf(m1, 1), cyc(m1, 1), tf(m2, 2), cyc(m2, 2), tf(m4, 4), cyc(m4, 4));
    }
    for (int wpc : {4, 8, 16}) {
        int blocks = cus * wpc;
        float m2 = time_ms([&] { mfma32_rate_kernel<2><<<blocks, 64>>>(dout, iters); });
        float m4 = time_ms([&] { mfma32_rate_kernel<4><<<blocks, 64>>>(dout, iters); });
        auto tf = [&](float ms, int nacc) { return 2048.0 * nacc * iters * blocks / (ms * 1e-3) / 1e12; };
        auto cyc = [&](float ms, int nacc) { double wps = wpc / 4.0; return ms * 1e-3 * clk / (double(iters) * nacc * wps); };
        printf("MFMA f32 16x16x4  waves/CU %2d: 2acc %.1f TF (%.1f cyc/mfma/simd@nominal) 4acc %.1f TF (%.1f)\n",
               wpc, tf(m2, 2), cyc(m2, 2), tf(m4, 4), cyc(m4, 4));
    }
    for (int wpc : {4, 8, 16}) {
        int blocks = cus * wpc;
        float f1 = time_ms([&] { fma_rate_kernel<1><<<blocks, 64>>>(dout, iters); });
        float f8 = time_ms([&] { fma_rate_kernel<8><<<blocks, 64>>>(dout, iters); });
        auto tf = [&](float ms, int nacc) { return 128.0 * nacc * iters * blocks / (ms * 1e-3) / 1e12; };
        printf("v_fma_f64 waves/CU %2d: 1 chain %.1f TF  8 chains %.1f TF\n", wpc, tf(f1, 1), tf(f8, 8));
    }
    {
        int blocks = cus * 8; int it2 = 2000;
        const char* names[4] = {"exp", "log", "sqrt", "rcp(1/x)"};
        float t[4];
        t[0] = time_ms([&] { func_rate_kernel<0><<<blocks, 64>>>(dout, it2); });
        t[1] = time_ms([&] { func_rate_kernel<1><<<blocks, 64>>>(dout, it2); });
        t[2] = time_ms([&] { func_rate_kernel<2><<<blocks, 64>>>(dout, it2); });
        t[3] = time_ms([&] { func_rate_kernel<3><<<blocks, 64>>>(dout, it2); });
        for (int i = 0; i < 4; ++i) {
            // 2 waves per SIMD, 4 calls per iter per wave
            double cyc_per_call = t[i] * 1e-3 * clk / (double(it2) * 4 * 2);
            printf("f64 %s: %.1f cycles per wave-call per SIMD (nominal clock)\n", names[i], cyc_per_call);
        }
    }
    return 0;
}
