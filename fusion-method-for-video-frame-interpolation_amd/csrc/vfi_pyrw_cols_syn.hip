// Synthesis column pass of the steerable pyramid on the wave-private FFT engine: syn_cols_kernel (vfi_pyrw_kernels.h)
// instantiated for every column configuration of vfi_wfft_configs.h.
#include "vfi_pyrw_kernels.h"

namespace vfi {
namespace pyrw {

#define VFI_COL_CFG(M, L, PITCH, P0, P1, P2, R0, R1, R2, R3) Cfg<M, L, true, PITCH, P0, P1, P2, R0, R1, R2, R3>

namespace {
template <class C>
int syn_dispatch(const SynColsArgs &a, hipStream_t s) {
    if (a.tb.bluestein) {
        if constexpr (blu_capable(C::M)) return launch_cols<C, true>(syn_cols_kernel<C, true>, a, a.w, a.N, s);
        return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid columns: engine length %d does not serve Bluestein", C::M);
    }
    return launch_cols<C, false>(syn_cols_kernel<C, false>, a, a.w, a.N, s);
}
}  // namespace

int launch_syn_cols(const SynColsArgs &a, hipStream_t s) {
    switch (a.tb.M) {
#define X(M, L, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return syn_dispatch<VFI_COL_CFG(M, L, PITCH, P0, P1, P2, R0, R1, R2, R3)>(a, s);
        VFI_WFFT_COL_CONFIGS(X)
#undef X
    }
    return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid columns: no engine configuration for length %d", a.tb.M);
}

}  // namespace pyrw
}  // namespace vfi
