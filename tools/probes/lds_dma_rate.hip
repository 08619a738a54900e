// Probe: how many cycles of a CU does one global -> LDS request cost?  One workgroup per CU; every wave issues REQS requests
// per round (its own LDS region, a small L2-resident source), ROUNDS rounds, one wait per round.
//   mode 0: LDS-DMA, 16 bytes per lane (buffer_load_dwordx4 ... lds), 1 KiB per request
//   mode 1: LDS-DMA,  4 bytes per lane (buffer_load_dword   ... lds), 256 B per request
//   mode 2: buffer_load_dwordx4 into registers + ds_write_b128 (1 KiB per request, staged through 4 VGPRs)
//   mode 3: mode 0 with an empty descriptor (no memory behind the request; zeros land in LDS)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int REQS = 8, ROUNDS = 2000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(const float *__restrict__ src, float *__restrict__ sink, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *mine = lds + wave * REQS * 256;                 // REQS KiB per wave
    const __amdgpu_buffer_rsrc_t r = rsrc_of(src, MODE == 3 ? 0u : bytes);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < ROUNDS; ++it) {
        const unsigned base = (unsigned)((it & 7) * 8192);    // 8 KiB windows of a 64 KiB source: L2 hits
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int q = 0; q < REQS; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(mine + 256 * q), 16,
                                                         base + (unsigned)(q * 1024 + lane * 16), 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < REQS; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(mine + 64 * q), 4,
                                                         base + (unsigned)(q * 256 + lane * 4), 0, 0, 0);
        } else {
            float4 v[REQS];
#pragma unroll
            for (int q = 0; q < REQS; ++q) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 t = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, base + (unsigned)(q * 1024 + lane * 16), 0, 0));
                v[q] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int q = 0; q < REQS; ++q) *reinterpret_cast<float4 *>(mine + 256 * q + 4 * lane) = v[q];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const float4 b = *reinterpret_cast<const float4 *>(mine + 4 * lane);
    acc.x += b.x + b.y + b.z + b.w;
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc.x;
}

// Does a request hold its wave back?  Each round: ONE 1 KiB LDS-DMA request (WITH_DMA) and a chain of 64 dependent FMAs
// per lane (~ the arithmetic between two requests of the Winograd kernels); the requests are only waited for every 8 rounds.
template <bool WITH_DMA>
__global__ __launch_bounds__(512) void overlap_kernel(const float *__restrict__ src, float *__restrict__ sink, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *mine = lds + wave * REQS * 256;
    const __amdgpu_buffer_rsrc_t r = rsrc_of(src, bytes);
    float x = (float)lane * 1e-3f, y = 1.0001f;
    for (int it = 0; it < ROUNDS; ++it) {
        if (WITH_DMA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(mine + 256 * (it & 7)), 16,
                                                     (unsigned)((it & 7) * 8192 + lane * 16), 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, y, 1e-7f);
        if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (x == 12345.678f) sink[threadIdx.x] = x + mine[lane];
}

template <bool WITH_DMA>
double run_overlap(const float *src, float *sink, int waves, int cus, double mhz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)8 * REQS * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(overlap_kernel<WITH_DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(overlap_kernel<WITH_DMA>, dim3(cus), dim3(64 * waves), lds, 0, src, sink, 65536u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(overlap_kernel<WITH_DMA>, dim3(cus), dim3(64 * waves), lds, 0, src, sink, 65536u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * mhz * 1e6 / ROUNDS;        // cycles per round
}

template <int MODE>
void run(const char *name, const float *src, float *sink, int waves, int cus, double mhz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)8 * REQS * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(cus), dim3(64 * waves), lds, 0, src, sink, 65536u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(cus), dim3(64 * waves), lds, 0, src, sink, 65536u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double reqs = (double)waves * REQS * ROUNDS;                  // per CU
    const double cyc = ms * 1e-3 * mhz * 1e6 / reqs;
    const double bytes = MODE == 1 ? 256.0 : 1024.0;
    std::printf("%-34s waves %d: %7.1f cycles per request and CU, %6.1f bytes per clock and CU\n", name, waves, cyc, bytes / cyc);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    float *src, *sink;
    hipMalloc(&src, 65536 + 8 * 8192);
    hipMalloc(&sink, 4096);
    hipMemset(src, 0, 65536 + 8 * 8192);
    std::printf("%s: %d CUs, %.0f MHz (nominal; cycles below assume it)\n", p.name, cus, mhz);
    for (int waves : {1, 2, 4, 8}) {
        run<0>("LDS-DMA 16 B/lane", src, sink, waves, cus, mhz);
        run<3>("LDS-DMA 16 B/lane, empty descriptor", src, sink, waves, cus, mhz);
        run<1>("LDS-DMA 4 B/lane", src, sink, waves, cus, mhz);
        run<2>("load x4 -> VGPR -> ds_write_b128", src, sink, waves, cus, mhz);
    }
    for (int waves : {1, 4, 8}) {
        const double a = run_overlap<false>(src, sink, waves, cus, mhz), b = run_overlap<true>(src, sink, waves, cus, mhz);
        std::printf("64 dependent FMAs per round, waves %d: %.0f cycles per round alone, %.0f with one 1 KiB LDS-DMA request per round (+%.0f)\n",
                    waves, a, b, b - a);
    }
    return 0;
}
