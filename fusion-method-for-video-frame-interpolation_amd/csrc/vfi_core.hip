// Library-wide entry points: ABI version, status strings, thread-local error detail.
#include "vfi_common.h"
#include <cstring>

namespace vfi {
static thread_local char g_last_error[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
}
}  // namespace vfi

extern "C" int vfi_abi_version(void) { return VFI_ABI_VERSION; }

extern "C" const char *vfi_last_error(void) { return vfi::g_last_error; }

extern "C" const char *vfi_status_string(int status) {
    switch (status) {
        case VFI_OK: return "VFI_OK";
        case VFI_ERR_INVALID_ARG: return "VFI_ERR_INVALID_ARG";
        case VFI_ERR_SHAPE: return "VFI_ERR_SHAPE";
        case VFI_ERR_LAUNCH: return "VFI_ERR_LAUNCH";
        case VFI_ERR_UNSUPPORTED: return "VFI_ERR_UNSUPPORTED";
        case VFI_ERR_FFT: return "VFI_ERR_FFT";
        case VFI_ERR_NOMEM: return "VFI_ERR_NOMEM";
        default: return "VFI_ERR_UNKNOWN";
    }
}
