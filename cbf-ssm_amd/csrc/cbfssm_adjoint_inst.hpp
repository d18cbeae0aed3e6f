// Per-NBLK instantiation of the adjoint kernels.
#pragma once
#include <cstdlib>
#include "cbfssm_adjoint.hpp"
#include "cbfssm_inst.hpp"

namespace cbfssm {

// NBLK <= 7 (M <= 112): one row block per wave, the K^-1-adjoint accumulator (NBLK x 16 x 16 f64 per wave) stays in
// VGPRs for the whole pass.  Larger tiles: two row blocks per wave and the accumulator does not fit the register file
// of a CU any more -- the A2bar / K tiles of every step are stashed in HBM and contracted by one GEMM per launch.
template <int NBLK>
struct RevCfg {
    static constexpr bool STASH = (NBLK > 7);
#ifdef CBF_REV_RB13      // diagnostic builds: row blocks per wave at 13 row blocks (4: four waves, one per SIMD, up to 512 registers)
    static constexpr int RB = (NBLK == 13) ? CBF_REV_RB13 : (STASH ? 2 : 1);
#else
    static constexpr int RB = STASH ? 2 : 1;
#endif
    static constexpr int W = (NBLK + RB - 1) / RB;
};

template <int NBLK, int DK>
struct RevGeom {
    typedef RevCfg<NBLK> C;
    static constexpr int JB = (4 * DK + 1 + 15) / 16;
    static constexpr int PSL = (JB > 2 ? JB : 2) * 256;
    static constexpr int LDS_BASE = 2 * 4 * DK * 17 + 2 * (16 * NBLK) * 17 + 2 * 16 * 17 + C::W * PSL + 64;
    static constexpr int LDS_LIMIT = 163840 / 8;
    static constexpr int SLAB = Slab<NBLK, JB, C::STASH>::total;
};

template <int NBLK, int DK, int MODE, int KD>
int launch_rev_k(const RevArgs& a, dim3 grid, hipStream_t st)
{
    // K^-1 image in LDS when it fits next to the tiles (M <= 104 at NBLK = 7), else streamed from L2
    typedef RevGeom<NBLK, DK> G;
    typedef RevCfg<NBLK> C;
    const int blds_doubles = NBLK * a.KSr * 64;
    const dim3 block(64 * (C::W + (rev_extra_wave(NBLK, C::STASH) ? 1 : 0)));
    if (G::LDS_BASE + blds_doubles <= G::LDS_LIMIT && !getenv("CBFSSM_NO_BLDS")) {
        const size_t lds = size_t(G::LDS_BASE + blds_doubles) * sizeof(double);
        if constexpr (!C::STASH) {
            if (a.ksave && a.a2s) {          // the kernel tiles were kept next to the A2 tiles
                auto k = rev_kernel<NBLK, C::RB, DK, true, C::STASH, MODE, KD, true>;
                int rc = set_lds(k, lds);
                if (rc) return rc;
                hipLaunchKernelGGL(k, grid, block, lds, st, a);
                hipError_t e = hipGetLastError();
                return e == hipSuccess ? 0 : -int(e) - 1000;
            }
        }
        auto k = rev_kernel<NBLK, C::RB, DK, true, C::STASH, MODE, KD>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    } else {
        typedef RevLds<NBLK, C::RB, DK, C::STASH> RL;
        static_assert(RL::BASE_PLAIN == G::LDS_BASE, "LDS layout");
        const size_t lds = size_t(RL::BASE + RL::EXTRA) * sizeof(double);
        if constexpr (!C::STASH) {
            if (a.ksave && a.a2s) {
                auto k = rev_kernel<NBLK, C::RB, DK, false, C::STASH, MODE, KD, true>;
                int rc = set_lds(k, lds);
                if (rc) return rc;
                hipLaunchKernelGGL(k, grid, block, lds, st, a);
                hipError_t e = hipGetLastError();
                return e == hipSuccess ? 0 : -int(e) - 1000;
            }
        }
        auto k = rev_kernel<NBLK, C::RB, DK, false, C::STASH, MODE, KD>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK, int DK, int MODE>
int launch_rev_t(const RevArgs& a, dim3 grid, hipStream_t st)
{
    // (seven row blocks = the Sarcos class: its backward runs carry dim_x - dim_y = 7 <= 8 output dimensions)
    if constexpr (NBLK == 7) {
        if (a.Do <= 8 && !getenv("CBFSSM_REV_KD4")) return launch_rev_k<NBLK, DK, MODE, 2>(a, grid, st);
    }
    return launch_rev_k<NBLK, DK, MODE, 4>(a, grid, st);
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

#define CBF_FOR_EACH_REV_NBLK(X) X(1) X(2) X(4) X(7) X(10) X(13) X(16) X(20)
