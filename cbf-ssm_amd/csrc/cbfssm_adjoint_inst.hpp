// Per-NBLK instantiation of the adjoint kernels (tile heights whose K^-1-adjoint accumulator fits the VGPR file).
#pragma once
#include "cbfssm_adjoint.hpp"
#include "cbfssm_inst.hpp"

namespace cbfssm {

template <int NBLK, int DK>
struct RevGeom {
    static constexpr int JB = (4 * DK + 1 + 15) / 16;
    static constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    static constexpr int LDS_BASE = 4 * DK * 17 + 2 * (16 * NBLK) * 17 + 2 * 16 * 17 + NBLK * PSL + 64;
    static constexpr int LDS_LIMIT = 163840 / 8;
    static constexpr int SLAB = Slab<NBLK, JB>::total;
};

template <int NBLK, int DK, int MODE>
int launch_rev_t(const RevArgs& a, dim3 grid, hipStream_t st)
{
    // K^-1 image in LDS when it fits next to the tiles (M <= 104 at NBLK = 7), else streamed from L2; the VGPRs hold
    // the K^-1-adjoint accumulator either way
    typedef RevGeom<NBLK, DK> G;
    const int blds_doubles = NBLK * a.KSr * 64;
    if (G::LDS_BASE + blds_doubles <= G::LDS_LIMIT) {
        const size_t lds = size_t(G::LDS_BASE + blds_doubles) * sizeof(double);
        auto k = rev_kernel<NBLK, DK, true, MODE>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, dim3(64 * NBLK), lds, st, a);
    } else {
        const size_t lds = size_t(G::LDS_BASE) * sizeof(double);
        auto k = rev_kernel<NBLK, DK, false, MODE>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, dim3(64 * NBLK), lds, st, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK>
int launch_rev_n(int DK, int mode, const RevArgs& a, dim3 grid, hipStream_t st)
{
    if (mode == MODE_FWD) {
        switch (DK) {
            case 2: return launch_rev_t<NBLK, 2, MODE_FWD>(a, grid, st);
            case 4: return launch_rev_t<NBLK, 4, MODE_FWD>(a, grid, st);
            case 6: return launch_rev_t<NBLK, 6, MODE_FWD>(a, grid, st);
        }
    } else {
        switch (DK) {
            case 2: return launch_rev_t<NBLK, 2, MODE_BWD>(a, grid, st);
            case 4: return launch_rev_t<NBLK, 4, MODE_BWD>(a, grid, st);
            case 6: return launch_rev_t<NBLK, 6, MODE_BWD>(a, grid, st);
        }
    }
    return -2;
}

template <int NBLK>
int64_t rev_slab_n(int DK)
{
    switch (DK) {
        case 2: return RevGeom<NBLK, 2>::SLAB;
        case 4: return RevGeom<NBLK, 4>::SLAB;
        case 6: return RevGeom<NBLK, 6>::SLAB;
    }
    return -1;
}

}  // namespace cbfssm

#define CBF_REV_DECLARE(NB)                                                                      \
    namespace cbfssm {                                                                           \
    int launch_rev_nb##NB(int DK, int mode, const RevArgs& a, dim3 grid, hipStream_t st);        \
    int64_t rev_slab_nb##NB(int DK);                                                             \
    }

#define CBF_REV_INSTANTIATE(NB)                                                                  \
    namespace cbfssm {                                                                           \
    int launch_rev_nb##NB(int DK, int mode, const RevArgs& a, dim3 grid, hipStream_t st)         \
    {                                                                                            \
        return launch_rev_n<NB>(DK, mode, a, grid, st);                                          \
    }                                                                                            \
    int64_t rev_slab_nb##NB(int DK) { return rev_slab_n<NB>(DK); }                               \
    }

// tile heights with an adjoint kernel (M <= 112 this round; larger M needs the K^-1-adjoint accumulator outside VGPRs)
#define CBF_FOR_EACH_REV_NBLK(X) X(1) X(2) X(4) X(7)
