// Analysis column pass of the steerable pyramid on the wave-private FFT engine: ana_cols_kernel (vfi_pyrw_kernels.h)
// instantiated for every column configuration of vfi_wfft_configs.h, plus the column-side lookups of vfi_pyramid_wave.h.
#include "vfi_pyrw_kernels.h"

#include <cmath>

namespace vfi {
namespace pyrw {

#define VFI_COL_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) Cfg<M, L, TEAM, true, PITCH, P0, P1, P2, R0, R1, R2, R3>

namespace {
template <class C>
int ana_dispatch(const AnaColsArgs &a, hipStream_t s) {
    if (a.tb.bluestein) {
        if constexpr (blu_capable(C::M)) return launch_cols<C, true, ana_cols_kernel<C, true>>(a, a.w, a.N * kBands, s);
        return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid columns: engine length %d does not serve Bluestein", C::M);
    }
    return launch_cols<C, false, ana_cols_kernel<C, false>>(a, a.w, a.N * kBands, s);
}
template <class C>
int twiddles_of(float2 *out, int cap) {
    if (C::TW > cap) return -1;
    for_twiddles<C>([&](int idx, int e) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)e / (double)C::M;
        out[idx] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    });
    return C::TW;
}
}  // namespace

int cols_engine_length(int n, int bluestein_m) {
    const int m = bluestein_m ? bluestein_m : n;
    if (bluestein_m && (!blu_capable(m) || 2 * n > m)) return 0;
    switch (m) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return M;
        VFI_WFFT_COL_CONFIGS(X)
#undef X
    }
    return 0;
}

int cols_twiddles(int M, float2 *out, int cap) {
    switch (M) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return twiddles_of<VFI_COL_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3)>(out, cap);
        VFI_WFFT_COL_CONFIGS(X)
#undef X
    }
    return -1;
}

int launch_ana_cols(const AnaColsArgs &a, hipStream_t s) {
    switch (a.tb.M) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return ana_dispatch<VFI_COL_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3)>(a, s);
        VFI_WFFT_COL_CONFIGS(X)
#undef X
    }
    return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid columns: no engine configuration for length %d", a.tb.M);
}

}  // namespace pyrw
}  // namespace vfi
