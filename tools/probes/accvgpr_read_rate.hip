// Probe: what a v_accvgpr_read_b32 costs a lone wave (the F(4x4) M = 32 epilogue reads 288 accumulators per item), next to
// v_mov_b32 and v_add_f32, back to back and alternating with a dependent VALU instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int REP = 64, N = 96;
template <int MODE> __global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *t) {
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.001f + i;
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\t"
                 "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0" ::"v"(v[0]) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (MODE == 0) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v[i % 12]) : "i"(i % 8));
            if (MODE == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i % 12]) : "v"(v[(i + 5) % 12]));
            if (MODE == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i % 12]) : "v"(v[(i + 5) % 12]), "v"(v[(i + 7) % 12]));
            if (MODE == 3) { asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v[i % 12]) : "i"(i % 8)); asm volatile("v_add_f32 %0, %1, %1" : "=v"(v[(i + 6) % 12]) : "v"(v[i % 12])); }
            if (MODE == 4) { asm volatile("v_mov_b32 %0, %1" : "=v"(v[i % 12]) : "v"(v[(i + 5) % 12])); asm volatile("v_add_f32 %0, %1, %1" : "=v"(v[(i + 6) % 12]) : "v"(v[i % 12])); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(float *out, unsigned long long *t, const char *what, int per) {
    std::vector<unsigned long long> h(256 * 4);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, t); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    printf("%s: %.2f cycles per %s\n", what, s / h.size() / REP / N, per == 1 ? "instruction" : "pair");
}
int main() {
    float *out; unsigned long long *t;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 256 * 4 * 8);
    run<0>(out, t, "v_accvgpr_read_b32 back to back", 1);
    run<1>(out, t, "v_mov_b32 back to back", 1);
    run<2>(out, t, "v_add_f32 back to back", 1);
    run<3>(out, t, "v_accvgpr_read_b32 + dependent v_add_f32", 2);
    run<4>(out, t, "v_mov_b32 + dependent v_add_f32", 2);
    return 0;
}
