// Plain row passes on the wave-private FFT engine: gen_rows_kernel (vfi_pyrw_passes.h) for every row configuration.
#include "vfi_pyrw_passes.h"

namespace vfi {
namespace pyrw {

#define VFI_ROW_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) Cfg<M, L, TEAM, false, PITCH, P0, P1, P2, R0, R1, R2, R3>

namespace {
template <class C, bool BLU>
int rows_kind(const GenRowsArgs &a, int load, int store, bool inverse, hipStream_t s) {
    const int nbatch = (a.rows + C::L - 1) / C::L;
    if (load == kGenReal && store == kGenHalf && !inverse)
        return launch_rows<C, BLU, gen_rows_kernel<C, BLU, kGenReal, kGenHalf, false>>(a, nbatch, s);
    if (load == kGenHalf && store == kGenReal && inverse)
        return launch_rows<C, BLU, gen_rows_kernel<C, BLU, kGenHalf, kGenReal, true>>(a, nbatch, s);
    if (load == kGenComplex && store == kGenComplex)
        return inverse ? launch_rows<C, BLU, gen_rows_kernel<C, BLU, kGenComplex, kGenComplex, true>>(a, nbatch, s)
                       : launch_rows<C, BLU, gen_rows_kernel<C, BLU, kGenComplex, kGenComplex, false>>(a, nbatch, s);
    return vfi::fail(VFI_ERR_UNSUPPORTED, "fft row pass: load %d / store %d / inverse %d", load, store, (int)inverse);
}
template <class C>
int rows_dispatch(const GenRowsArgs &a, int load, int store, bool inverse, hipStream_t s) {
    if (a.tb.bluestein) {
        if constexpr (blu_capable(C::M)) return rows_kind<C, true>(a, load, store, inverse, s);
        return vfi::fail(VFI_ERR_UNSUPPORTED, "fft rows: engine length %d does not serve Bluestein", C::M);
    }
    return rows_kind<C, false>(a, load, store, inverse, s);
}
}  // namespace

int launch_gen_rows(const GenRowsArgs &a, int load, int store, bool inverse, hipStream_t s) {
    switch (a.tb.M) {
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) case M: return rows_dispatch<VFI_ROW_CFG(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3)>(a, load, store, inverse, s);
        VFI_WFFT_ROW_CONFIGS(X)
#undef X
    }
    return vfi::fail(VFI_ERR_UNSUPPORTED, "fft rows: no engine configuration for length %d", a.tb.M);
}

}  // namespace pyrw
}  // namespace vfi
