// Probe: do the fp32 matrix pipe and the vector ALU of ONE SIMD run at the same time when the two instruction kinds come
// from DIFFERENT waves?  (From one wave they do not: the F(4x4) M = 32 body is MFMA cycles + transform cycles)
// Per SIMD and "chunk": 72 x v_mfma_f32_16x16x4_f32 (independent accumulators) and 72 x v_pk_fma_f32 -- the M = 32 body's mix.
//   mode 0: one wave per SIMD, MFMAs only            mode 1: one wave per SIMD, packed FMAs only
//   mode 2: one wave per SIMD, both, one after the other (today's kernel)
//   mode 3: one wave per SIMD, both, interleaved 1:1 in program order
//   mode 4: two waves per SIMD, wave A all 72 MFMAs, wave B all 72 packed FMAs (a producer / consumer split)
//   mode 5: two waves per SIMD, each 36 MFMAs then 36 packed FMAs (a symmetric split: verdict r03's lever (b))
//   mode 6: as 5, the two waves in opposite order (A: MFMA then FMA, B: FMA then MFMA)
// Cycles by s_memtime around NCHUNK chunks, maximum over the workgroup's waves, mean over 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NCHUNK = 64;

template <int NM> __device__ __forceinline__ void mfmas(f4 (&acc)[36], float a, float b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 36]) : "v"(a), "v"(b));
}
template <int NV> __device__ __forceinline__ void fmas(f2 (&v)[12], f2 k) {
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i % 12]) : "v"(k), "v"(v[(i + 5) % 12]));
}

template <int MODE> __global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *t, int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= nwaves) return;
    f4 acc[36];
    f2 v[12];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = f2{(float)lane * 1e-3f, (float)i * 1e-3f};
    const float a = 1.0f + lane * 1e-6f, b = 0.5f;
    const f2 kk = {0.999f, 1.001f};
    const bool second = wave >= 4;                 // waves 4..7 share the SIMDs of waves 0..3
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < NCHUNK; ++c) {
        if (MODE == 0) mfmas<72>(acc, a, b);
        if (MODE == 1) fmas<72>(v, kk);
        if (MODE == 2) { mfmas<72>(acc, a, b); fmas<72>(v, kk); }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 72; ++i) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % 36]) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i % 12]) : "v"(kk), "v"(v[(i + 5) % 12]));
            }
        }
        if (MODE == 4) { if (!second) mfmas<72>(acc, a, b); else fmas<72>(v, kk); }
        if (MODE == 5) { mfmas<36>(acc, a, b); fmas<36>(v, kk); }
        if (MODE == 6) { if (!second) { mfmas<36>(acc, a, b); fmas<36>(v, kk); } else { fmas<36>(v, kk); mfmas<36>(acc, a, b); } }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 36; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE> void run(float *out, unsigned long long *t, int nwaves, const char *what) {
    std::vector<unsigned long long> h(256 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(t, 0, h.size() * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, t, nwaves);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int g = 0; g < 256; ++g) {
        unsigned long long m = 0;
        for (int w = 0; w < 8; ++w) m = h[g * 8 + w] > m ? h[g * 8 + w] : m;
        sum += (double)m;
    }
    printf("mode %d (%s): %.0f cycles per chunk of 72 MFMA + 72 pk_fma per SIMD\n", MODE, what, sum / 256 / NCHUNK);
}

int main() {
    float *out; unsigned long long *t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8 * 8);
    run<0>(out, t, 4, "1 wave/SIMD, MFMA only");
    run<1>(out, t, 4, "1 wave/SIMD, pk_fma only");
    run<2>(out, t, 4, "1 wave/SIMD, MFMAs then pk_fmas");
    run<3>(out, t, 4, "1 wave/SIMD, interleaved 1:1");
    run<4>(out, t, 8, "2 waves/SIMD, A = MFMAs, B = pk_fmas");
    run<5>(out, t, 8, "2 waves/SIMD, each 36 MFMA then 36 pk_fma");
    run<6>(out, t, 8, "2 waves/SIMD, opposite order");
    return 0;
}
