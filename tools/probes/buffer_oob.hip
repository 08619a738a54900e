// Probe: what a 16-byte LDS-DMA buffer load returns when only SOME of its dwords are out of range.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *x, float *y, unsigned nbytes) {
    extern __shared__ float lds[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)nbytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // lane 0: starts 4 bytes before the base (wraps to 0xfffffffc); lane 1: starts 8 bytes before the end;
    // lane 2: fully inside at a 4-byte (not 16-byte) aligned offset; lane 3: fully outside
    unsigned vo = lane == 0 ? 0xfffffffcu : lane == 1 ? nbytes - 8 : lane == 2 ? 36 : nbytes + 64;
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.0f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, vo, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 16; i += 64) y[i] = lds[i];
}
int main() {
    float h[64], *x, *y, out[16];
    for (int i = 0; i < 64; ++i) h[i] = 100.0f + i;
    hipMalloc(&x, 4096); hipMalloc(&y, 64);
    hipMemcpy(x + 256, h, sizeof h, hipMemcpyHostToDevice);     // buffer = x+256 .. +64 floats; x+255 holds 0 (fresh) -> set it
    float before = 77.0f; hipMemcpy(x + 255, &before, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, x + 256, y, 64u * 4u);
    hipMemcpy(out, y, sizeof out, hipMemcpyDeviceToHost);
    for (int l = 0; l < 4; ++l) printf("lane %d: %g %g %g %g\n", l, out[4 * l], out[4 * l + 1], out[4 * l + 2], out[4 * l + 3]);
    return 0;
}
