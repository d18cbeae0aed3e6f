// Standard-normal draws for the passes.  The reference draws them inside its graph (tf.random_normal: cbfssm.py:134,149,209;
// cbfssmhalf.py:142; prssm.py:126) -- one normal per (b, s) chain and step; TensorFlow's generator stream cannot be reproduced
// outside TensorFlow, so this is the library's own: the counter-based Philox4x32-10 (Salmon et al., SC'11; the generator family
// TensorFlow and PyTorch use on devices) followed by Box-Muller in float64.  Element i of a draw is a pure function of
// (seed, offset + i): a draw can be split, repeated or resumed anywhere, on any number of devices.
//   pair p = (offset + i) >> 1:  (r0, r1, r2, r3) = philox4x32_10(counter = (p_lo, p_hi, 0, 0), key = (seed_lo, seed_hi))
//   u1 = (((r0 << 32 | r1) >> 11) + 1) 2^-53 in (0, 1],  u2 = ((r2 << 32 | r3) >> 11) 2^-53 in [0, 1)
//   z = sqrt(-2 ln u1) (cos 2 pi u2, sin 2 pi u2)[(offset + i) & 1]
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/cbfssm_hip.h"

namespace cbfssm {
int fail(int code, const char* fmt, ...);   // cbfssm_api.hip

static int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-int(e) - 1000, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&r)[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        const uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        c1 = uint32_t(p1); c3 = uint32_t(p0); c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// one thread per pair; a thread writes the one or two elements of its pair that lie inside [0, n)
__global__ __launch_bounds__(256) void normal_kernel(uint64_t seed, uint64_t offset, int64_t n, double* out, uint32_t* raw)
{
    const uint64_t first = offset >> 1;
    const int64_t npairs = int64_t(((offset + uint64_t(n) + 1) >> 1) - first);
    for (int64_t j = int64_t(blockIdx.x) * 256 + threadIdx.x; j < npairs; j += int64_t(gridDim.x) * 256) {
        const uint64_t p = first + uint64_t(j);
        uint32_t r[4];
        philox4x32_10(uint32_t(p), uint32_t(p >> 32), 0u, 0u, uint32_t(seed), uint32_t(seed >> 32), r);
        if (raw) {                                   // (known-answer tests: the generator's words as they are)
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[4 * j + q] = r[q];
            continue;
        }
        const uint64_t a = (uint64_t(r[0]) << 32) | r[1], b = (uint64_t(r[2]) << 32) | r[3];
        const double u1 = double((a >> 11) + 1) * 0x1.0p-53;
        const double u2 = double(b >> 11) * 0x1.0p-53;
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        const int64_t i0 = int64_t(2 * p - offset);   // element index of the pair's first member (may be -1)
        if (i0 >= 0) out[i0] = rad * cs;
        if (i0 + 1 < n) out[i0 + 1] = rad * sn;
    }
}

}  // namespace cbfssm

using namespace cbfssm;

extern "C" {

int cbfssm_normal_f64(uint64_t seed, uint64_t offset, int64_t n, double* out, void* stream)
{
    if (n < 0) return fail(-1, "n must be >= 0");
    if (n == 0) return 0;
    if (!out) return fail(-1, "null pointer");
    const int64_t npairs = int64_t(((offset + uint64_t(n) + 1) >> 1) - (offset >> 1));
    const int64_t blocks = (npairs + 255) / 256;
    hipLaunchKernelGGL(normal_kernel, dim3(unsigned(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, seed,
                       offset, n, out, nullptr);
    return check_launch("normal");
}

int cbfssm_philox4x32_10_u32(uint64_t seed, uint64_t first_counter, int64_t ncounters, uint32_t* out, void* stream)
{
    if (ncounters < 0) return fail(-1, "ncounters must be >= 0");
    if (ncounters == 0) return 0;
    if (!out) return fail(-1, "null pointer");
    const int64_t blocks = (ncounters + 255) / 256;
    hipLaunchKernelGGL(normal_kernel, dim3(unsigned(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, seed,
                       2 * first_counter, 2 * ncounters, nullptr, out);
    return check_launch("philox");
}

}  // extern "C"
