// Probe: how fast a CU's four waves get 16-byte-per-lane global stores out, by what one wave-instruction's 1 KiB covers:
//   pattern 0: ONE contiguous 1 KiB run (64 lanes x 16 B)
//   pattern 1: 4 runs of 256 B in 4 different planes (the F(4x4) conv epilogue: lane = (tile n16, channel group k4))
//   pattern 2: 16 runs of 64 B, pattern 3: 64 runs of 16 B (a row per lane)
// 256 workgroups x 256 threads, each wave issues NST stores back to back; cycles by s_memtime around the burst + drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NST = 32;
__global__ __launch_bounds__(256, 1) void k(float *y, unsigned long long *t, int pattern, size_t plane) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0x7fffffff, 0x00020000);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 d = {(unsigned)lane, 1u, 2u, 3u};
    unsigned off;
    const unsigned base = (blockIdx.x * 4 + wave) * 4096u * NST;        // this wave's region
    if (pattern == 0) off = lane * 16;
    else if (pattern == 1) off = (lane >> 4) * (unsigned)plane + (lane & 15) * 16;
    else if (pattern == 2) off = (lane >> 2) * (unsigned)(plane / 4) + (lane & 3) * 16;
    else off = lane * (unsigned)(plane / 16);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(d, r, base % (unsigned)plane + off, i * 1024, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { t[(blockIdx.x * 4 + wave) * 2] = t1 - t0; t[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
int main() {
    const size_t plane = 8294400;                 // 1080 x 1920 floats
    float *y; unsigned long long *t;
    hipMalloc(&y, plane * 5 + (64u << 20)); hipMalloc(&t, 256 * 4 * 2 * 8);
    std::vector<unsigned long long> h(256 * 4 * 2);
    for (int pattern = 0; pattern < 4; ++pattern)
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, y, t, pattern, plane);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
            double issue = 0, drain = 0;
            for (int i = 0; i < 256 * 4; ++i) { issue += h[2 * i]; drain += h[2 * i + 1]; }
            issue /= 1024; drain /= 1024;
            printf("pattern %d: %d stores per wave, 4 waves per CU: issue %.0f cycles, issue + drain %.0f cycles -> %.1f B/clk/CU\n", pattern, NST, issue, drain,
                   4.0 * NST * 1024 / drain);
        }
    return 0;
}
