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

// Test aid: every workgroup fills its whole LDS allocation with NaNs and leaves.  LDS is not cleared between workgroups,
// so the kernels launched next start on NaN-filled LDS: anything that depends on LDS it has not written shows.
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(int words) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds_words[i] = 0x7fc00000u;
    __syncthreads();
    if (lds_words[(threadIdx.x * 61u) % words] != 0x7fc00000u) __builtin_trap();     // (keeps the stores alive)
}
}  // namespace

extern "C" int vfi_debug_poison_lds(vfi_stream_t stream) {
    constexpr int kBytes = 160 * 1024;
    static bool done[vfi::kMaxDevices] = {};
    bool &d = done[vfi::current_device()];
    if (!d) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
        d = true;
    }
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    hipLaunchKernelGGL(poison_lds_kernel, dim3(4 * cus), dim3(256), kBytes, vfi::as_stream(stream), kBytes / 4);
    return vfi::check_launch("vfi_debug_poison_lds");
}

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
