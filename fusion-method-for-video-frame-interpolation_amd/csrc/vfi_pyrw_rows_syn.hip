// Synthesis row pass of the steerable pyramid on the wave-private FFT engine: rows_from_polar_kernel
// (vfi_pyrw_kernels.h) instantiated for every row configuration of vfi_wfft_configs.h.
#include "vfi_pyrw_kernels.h"

namespace vfi {
namespace pyrw {

#define VFI_ROW_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) Cfg<M, L, TEAM, false, PITCH, P0, P1, P2, R0, R1, R2, R3>

namespace {
template <class C>
int from_polar_dispatch(const RowsArgs &a, hipStream_t s) {
    const int nbatch = (a.planes * a.h + C::L - 1) / C::L;
    if (a.tb.bluestein) {
        if constexpr (blu_capable(C::M)) return launch_rows<C, true, rows_from_polar_kernel<C, true>>(a, nbatch, s);
        return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid rows: engine length %d does not serve Bluestein", C::M);
    }
    return launch_rows<C, false, rows_from_polar_kernel<C, false>>(a, nbatch, s);
}
}  // namespace

int launch_rows_from_polar(const RowsArgs &a, hipStream_t s) {
    switch (a.tb.M) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return from_polar_dispatch<VFI_ROW_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3)>(a, s);
        VFI_WFFT_ROW_CONFIGS(X)
#undef X
    }
    return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid rows: no engine configuration for length %d", a.tb.M);
}

}  // namespace pyrw
}  // namespace vfi
