// Plain column passes on the wave-private FFT engine: gen_cols_kernel (vfi_pyrw_passes.h) for every analysis column
// configuration (the long lengths on a team of waves: 8 columns = 64-byte row segments for 1080 rows).
#include "vfi_pyrw_passes.h"

namespace vfi {
namespace pyrw {

#define VFI_COL_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) Cfg<M, L, TEAM, true, PITCH, P0, P1, P2, R0, R1, R2, R3>

namespace {
template <class C>
int cols_dispatch(const GenColsArgs &a, bool inverse, hipStream_t s) {
    if (a.tb.bluestein) {
        if constexpr (blu_capable(C::M))
            return inverse ? launch_cols<C, true, gen_cols_kernel<C, true, true>>(a, a.cols, a.planes, s)
                           : launch_cols<C, true, gen_cols_kernel<C, true, false>>(a, a.cols, a.planes, s);
        return vfi::fail(VFI_ERR_UNSUPPORTED, "fft columns: engine length %d does not serve Bluestein", C::M);
    }
    return inverse ? launch_cols<C, false, gen_cols_kernel<C, false, true>>(a, a.cols, a.planes, s)
                   : launch_cols<C, false, gen_cols_kernel<C, false, false>>(a, a.cols, a.planes, s);
}
}  // namespace

int launch_gen_cols(const GenColsArgs &a, bool inverse, hipStream_t s) {
    switch (a.tb.M) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return cols_dispatch<VFI_COL_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3)>(a, inverse, s);
        VFI_WFFT_COL_CONFIGS(X)
#undef X
    }
    return vfi::fail(VFI_ERR_UNSUPPORTED, "fft columns: no engine configuration for length %d", a.tb.M);
}

}  // namespace pyrw
}  // namespace vfi
