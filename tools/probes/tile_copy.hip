// Probe: how fast can 256-thread workgroups copy a (planes, h, w) complex64 array when each workgroup owns a tile of C
// adjacent columns (the access pattern of the pyramid's column passes: row segments of C*8 bytes, row pitch w*8 bytes)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_copy.hip -o /tmp/tile_copy && /tmp/tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SLOTS>
__global__ __launch_bounds__(256) void tile_copy(const float2 *__restrict__ src, float2 *__restrict__ dst, int h, int w, int shift, int xcd) {
    const int C = 1 << shift, ntiles = (w + C - 1) / C;
    int tile = blockIdx.x;
    if (xcd) { const int per = (ntiles + 7) >> 3; tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    if (tile >= ntiles) return;
    const size_t plane = (size_t)blockIdx.y * h * w;
    const int t = threadIdx.x, cc = t & (C - 1), u0 = t >> shift, step = 256 >> shift, v = tile * C + cc;
    float2 z[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) { const int u = u0 + step * q; z[q] = (u < h && v < w) ? src[plane + (size_t)u * w + v] : make_float2(0, 0); }
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) { const int u = u0 + step * q; if (u < h && v < w) dst[plane + (size_t)u * w + v] = z[q]; }
}

int main() {
    const int planes = 24, h = 1080, w = 1920;
    const size_t n = (size_t)planes * h * w;
    float2 *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int shift = 2; shift <= 5; ++shift) {
            const int C = 1 << shift, ntiles = (w + C - 1) / C, slots = (h * C + 255) / 256;
            dim3 grid(8 * ((ntiles + 7) / 8), planes);
            auto launch = [&]() {
                if (slots <= 17) hipLaunchKernelGGL(tile_copy<17>, grid, dim3(256), 0, 0, a, b, h, w, shift, xcd);
                else if (slots <= 34) hipLaunchKernelGGL(tile_copy<34>, grid, dim3(256), 0, 0, a, b, h, w, shift, xcd);
                else if (slots <= 68) hipLaunchKernelGGL(tile_copy<68>, grid, dim3(256), 0, 0, a, b, h, w, shift, xcd);
                else hipLaunchKernelGGL(tile_copy<135>, grid, dim3(256), 0, 0, a, b, h, w, shift, xcd);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            std::printf("tile of %2d columns (%3d-byte row segments), xcd-aware order %d: %.3f ms  %.2f TB/s (read + write)\n", C, C * 8, xcd,
                        ms, 2.0 * n * 8 / ms / 1e9);
        }
    return 0;
}
