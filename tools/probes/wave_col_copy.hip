// Probe: copy rate of the access patterns a wave-private FFT pass would use on a (planes, h, w) complex64 array.
//   cols<L>: one WAVE owns L adjacent columns (row segments of L*8 bytes, lanes run over (column fastest, row)), loads all
//            of its h*L elements into registers (E = ceil(h*L/64) 8-byte loads in flight per lane), then stores them.
//            The four waves of a 256-thread workgroup own 4*L adjacent columns.
//   rows   : one wave owns one row of w elements (8-byte or 16-byte accesses).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wave_col_copy.hip -o /tmp/wave_col_copy && /tmp/wave_col_copy
#include <hip/hip_runtime.h>
#include <cstdio>

template <int L, int E>
__global__ __launch_bounds__(256) void cols_copy(const float2 *__restrict__ src, float2 *__restrict__ dst, int h, int w, int xcd) {
    const int groups = w / (4 * L);                     // workgroups per plane
    int g = blockIdx.x;
    if (xcd) { const int per = (groups + 7) >> 3; g = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    if (g >= groups) return;
    const size_t plane = (size_t)blockIdx.y * h * w;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % L, r0 = lane / L, step = 64 / L, v = (g * 4 + wave) * L + c;
    float2 z[E];
#pragma unroll
    for (int q = 0; q < E; ++q) { const int u = r0 + step * q; z[q] = u < h ? src[plane + (size_t)u * w + v] : make_float2(0, 0); }
#pragma unroll
    for (int q = 0; q < E; ++q) { const int u = r0 + step * q; if (u < h) dst[plane + (size_t)u * w + v] = z[q]; }
}

template <int E, int VEC>
__global__ __launch_bounds__(256) void rows_copy(const float *__restrict__ src, float *__restrict__ dst, long long rows, int w) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    typedef float vec __attribute__((ext_vector_type(VEC)));
    const vec *s = reinterpret_cast<const vec *>(src + row * w * 2);
    vec *d = reinterpret_cast<vec *>(dst + row * w * 2);
    const int n = w * 2 / VEC;
    vec z[E];
#pragma unroll
    for (int q = 0; q < E; ++q) { const int j = lane + 64 * q; if (j < n) z[q] = s[j]; }
#pragma unroll
    for (int q = 0; q < E; ++q) { const int j = lane + 64 * q; if (j < n) d[j] = z[q]; }
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const int planes = 24, h = 1080, w = 1920;
    const size_t n = (size_t)planes * h * w;
    float2 *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    const double gb = 2.0 * n * 8 / 1e9;
#define COLS(L)                                                                                                          \
    for (int xcd = 0; xcd < 2; ++xcd) {                                                                                  \
        constexpr int E = (1080 * L + 63) / 64;                                                                          \
        const int groups = w / (4 * L);                                                                                  \
        dim3 grid(8 * ((groups + 7) / 8), planes);                                                                       \
        const float ms = time_ms([&]() { hipLaunchKernelGGL((cols_copy<L, E>), grid, dim3(256), 0, 0, a, b, h, w, xcd); }); \
        std::printf("wave owns %2d columns (%3d-byte segments, %3d loads per lane), xcd order %d: %.3f ms  %.2f TB/s\n", L, L * 8, E, \
                    xcd, ms, gb / ms);                                                                                   \
    }
    COLS(1) COLS(2) COLS(4) COLS(8) COLS(16)
    {
        const long long rows = (long long)planes * h;
        float ms = time_ms([&]() { hipLaunchKernelGGL((rows_copy<30, 2>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, 0, (const float *)a, (float *)b, rows, w); });
        std::printf("wave owns one row, 8-byte accesses (30 per lane): %.3f ms  %.2f TB/s\n", ms, gb / ms);
        ms = time_ms([&]() { hipLaunchKernelGGL((rows_copy<15, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, 0, (const float *)a, (float *)b, rows, w); });
        std::printf("wave owns one row, 16-byte accesses (15 per lane): %.3f ms  %.2f TB/s\n", ms, gb / ms);
    }
    return 0;
}
