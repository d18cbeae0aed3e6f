// Per-NBLK instantiation of the time-loop kernels (one translation unit per NBLK so the build parallelises).
#pragma once
#include "cbfssm_kernels.hpp"

namespace cbfssm {

// How a tile of NBLK 16-row blocks of K^-1 is spread over the waves of a workgroup.
//   NBLK <= 7  (M <= 112): one row block per wave, the wave's K^-1 rows stay in VGPRs for the whole pass.
//   larger:                K^-1 is streamed from L2 as a pre-swizzled A-operand image (it no longer fits the
//                          register file of a CU: 13 blocks x 52 k-steps x 2 VGPRs = 1352 of 2048).
template <int NBLK>
struct Cfg {
    // (measured at NBLK = 7, C3: 4 waves x 2 row blocks with K^-1 in VGPRs needs 380 registers -> one wave per SIMD,
    //  2.6x slower; the same with K^-1 streamed from L2 fits two workgroups per CU but is 10-20 % slower than this)
    static constexpr int RB = (NBLK >= 13) ? 2 : 1;
    static constexpr bool BREG = (NBLK <= 7);
};

template <typename K>
inline int set_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
        if (e != hipSuccess) return -int(e) - 1000;
    }
    return 0;
}

// Register-resident tiles of seven row blocks (M = 97..112, the Sarcos class) are instantiated once per number of
// all-padding k-steps (KT = 0..3) so that their loops end at the last k-step that carries data without a runtime guard;
// everything else runs the unspecialised form (KT = -1).
template <int NBLK>
constexpr bool trim_tiles() { return NBLK == 7; }

template <int NBLK, int DK, int KT>
int launch_predict_k(const PredictArgs& a, hipStream_t st)
{
    typedef Cfg<NBLK> C;
    const unsigned groups = unsigned((a.npts + 15) / 16);
    if (a.tri) {
        typedef Tile<NBLK, C::RB, DK, C::BREG, true, KT> TT;
        const size_t lds = TT::LDS_DOUBLES * sizeof(double);
        auto k = predict_kernel<NBLK, C::RB, DK, C::BREG, true, KT>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(groups), dim3(TT::NT), lds, st, a);
    } else {
        typedef Tile<NBLK, C::RB, DK, C::BREG, false, KT> TT;
        const size_t lds = TT::LDS_DOUBLES * sizeof(double);
        auto k = predict_kernel<NBLK, C::RB, DK, C::BREG, false, KT>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(groups), dim3(TT::NT), lds, st, a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK, int DK>
int launch_predict_t(const PredictArgs& a, hipStream_t st)
{
    if constexpr (trim_tiles<NBLK>()) {
        switch (4 * NBLK - a.pk.KSr) {
            case 0: return launch_predict_k<NBLK, DK, 0>(a, st);
            case 1: return launch_predict_k<NBLK, DK, 1>(a, st);
            case 2: return launch_predict_k<NBLK, DK, 2>(a, st);
            case 3: return launch_predict_k<NBLK, DK, 3>(a, st);
        }
    }
    return launch_predict_k<NBLK, DK, -1>(a, st);
}

template <int NBLK, int DK, int MODE, int KT>
int launch_pass_k(const PassArgs& a, dim3 grid, int nc, hipStream_t st)
{
    typedef Cfg<NBLK> C;
    typedef Tile<NBLK, C::RB, DK, C::BREG, false, KT> TT;
    // nc: 1 = one 16-chain column block per workgroup; 2 = two blocks, skewed by half a step (pass_kernel_skew);
    //     3 = two blocks sharing every K^-1 operand load (pass_kernel<NC = 2>: the streamed-K^-1 tiles)
    const size_t lds = (size_t(nc == 1 ? 1 : 2) * (TT::LDS_DOUBLES - 64) + 64 + (nc == 1 ? TT::EPI_LDS_DOUBLES : 0)) *
                       sizeof(double);
    hipError_t e;
    if (a.tri) {
        // the reference's two-triangular form: one column block per workgroup (the caller passes nc = 1)
        typedef Tile<NBLK, C::RB, DK, C::BREG, true, KT> TR;
        const size_t ldt = (size_t(TR::LDS_DOUBLES) + TR::EPI_LDS_DOUBLES) * sizeof(double);
        auto k = pass_kernel<NBLK, C::RB, DK, C::BREG, MODE, 1, true, KT>;
        int rc = set_lds(k, ldt);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, dim3(TR::NT), ldt, st, a);
    } else if (nc == 3) {
        if constexpr (KT < 0) {
            auto k = pass_kernel<NBLK, C::RB, DK, C::BREG, MODE, 2>;
            int rc = set_lds(k, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
        } else {
            return -2;      // (the shared-operand variant belongs to the streamed tiles: never trimmed)
        }
    } else if (nc == 2) {
        auto k = pass_kernel_skew<NBLK, C::RB, DK, C::BREG, MODE, KT>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
    } else {
        auto k = pass_kernel<NBLK, C::RB, DK, C::BREG, MODE, 1, false, KT>;
        int rc = set_lds(k, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k, grid, dim3(TT::NT), lds, st, a);
    }
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -int(e) - 1000;
}

template <int NBLK, int DK, int MODE>
int launch_pass_t(const PassArgs& a, dim3 grid, int nc, hipStream_t st)
{
    if constexpr (trim_tiles<NBLK>()) {
        if (nc != 3) {
            switch (4 * NBLK - a.pk.KSr) {
                case 0: return launch_pass_k<NBLK, DK, MODE, 0>(a, grid, nc, st);
                case 1: return launch_pass_k<NBLK, DK, MODE, 1>(a, grid, nc, st);
                case 2: return launch_pass_k<NBLK, DK, MODE, 2>(a, grid, nc, st);
                case 3: return launch_pass_k<NBLK, DK, MODE, 3>(a, grid, nc, st);
            }
        }
    }
    return launch_pass_k<NBLK, DK, MODE, -1>(a, grid, nc, st);
}

template <int NBLK>
int launch_predict_n(int DK, const PredictArgs& a, hipStream_t st)
{
    switch (DK) {
        case 2: return launch_predict_t<NBLK, 2>(a, st);
        case 4: return launch_predict_t<NBLK, 4>(a, st);
        case 6: return launch_predict_t<NBLK, 6>(a, st);
    }
    return -2;
}

template <int NBLK>
int launch_pass_n(int DK, int mode, const PassArgs& a, dim3 grid, int nc, hipStream_t st)
{
    if (mode == MODE_FWD) {
        switch (DK) {
            case 2: return launch_pass_t<NBLK, 2, MODE_FWD>(a, grid, nc, st);
            case 4: return launch_pass_t<NBLK, 4, MODE_FWD>(a, grid, nc, st);
            case 6: return launch_pass_t<NBLK, 6, MODE_FWD>(a, grid, nc, st);
        }
    } else {
        switch (DK) {
            case 2: return launch_pass_t<NBLK, 2, MODE_BWD>(a, grid, nc, st);
            case 4: return launch_pass_t<NBLK, 4, MODE_BWD>(a, grid, nc, st);
            case 6: return launch_pass_t<NBLK, 6, MODE_BWD>(a, grid, nc, st);
        }
    }
    return -2;
}

}  // namespace cbfssm

#define CBF_DECLARE(NB)                                                                          \
    namespace cbfssm {                                                                           \
    int launch_predict_nb##NB(int DK, const PredictArgs& a, hipStream_t st);                     \
    int launch_pass_nb##NB(int DK, int mode, const PassArgs& a, dim3 grid, int nc, hipStream_t st); \
    }

#define CBF_INSTANTIATE(NB)                                                                      \
    namespace cbfssm {                                                                           \
    int launch_predict_nb##NB(int DK, const PredictArgs& a, hipStream_t st)                      \
    {                                                                                            \
        return launch_predict_n<NB>(DK, a, st);                                                  \
    }                                                                                            \
    int launch_pass_nb##NB(int DK, int mode, const PassArgs& a, dim3 grid, int nc, hipStream_t st) \
    {                                                                                            \
        return launch_pass_n<NB>(DK, mode, a, grid, nc, st);                                     \
    }                                                                                            \
    }

// the supported tile heights (16-row blocks of inducing points); M is padded up to the next one
#define CBF_FOR_EACH_NBLK(X) X(1) X(2) X(4) X(7) X(10) X(13) X(16) X(20)
