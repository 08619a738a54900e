// Probe for DESIGN.md section 9: fp32 products on the bf16 matrix cores by three-way operand splitting ("bf16x3").
//   x = hi + mid + lo with hi = trunc_bf16(x), mid = trunc_bf16(x - hi), lo = trunc_bf16(x - hi - mid)  (each 8 mantissa bits)
//   a*b ~ ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm        (6 bf16 MFMAs, fp32 accumulate; dropped terms <= 2^-24 |ab|)
// Measures (1) the error of a K = 4096 dot product against double for fp32 MFMA, bf16x3 and plain bf16, and
// (2) the issue rate of the two instruction mixes on one wave per SIMD and on a full chip:
//   fp32 : 8 x v_mfma_f32_16x16x4_f32   per K = 32
//   x3   : 6 x v_mfma_f32_16x16x32_bf16 per K = 32, B operand split on the fly (A = weights, split once outside)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bf16x3_mfma.hip -o /tmp/bf16x3 && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __float_as_uint(x[i]);
        hb[i] = xb & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(hb[i]);
        mb[i] = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb[i]);
        lb[i] = __float_as_uint(r2) & 0xffff0000u;
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // two bf16 (the high halves) per dword
        ph[i] = (hb[2 * i] >> 16) | hb[2 * i + 1];
        pm[i] = (mb[2 * i] >> 16) | mb[2 * i + 1];
        pl[i] = (lb[2 * i] >> 16) | lb[2 * i + 1];
    }
    h = __builtin_bit_cast(bf16x8, ph); m = __builtin_bit_cast(bf16x8, pm); l = __builtin_bit_cast(bf16x8, pl);
}

// C (16x16) = A (16xK) * B (Kx16); A row-major [16][K], B column-major [16][K]; one wave.  MODE 0 fp32, 1 bf16x3, 2 bf16.
template <int MODE>
__global__ void gemm16(const float *A, const float *B, float *C, int K, int reps) {
    const int lane = threadIdx.x, r = lane & 15, kb = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep)
        for (int k0 = 0; k0 < K; k0 += 32) {
            if (MODE == 0) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + 4 * s + kb], B[r * K + k0 + 4 * s + kb], acc, 0, 0, 0);
            } else {
                float a[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = A[r * K + k0 + 8 * kb + i]; b[i] = B[r * K + k0 + 8 * kb + i]; }
                bf16x8 ah, am, al, bh, bm, bl;
                split3(a, ah, am, al);
                split3(b, bh, bm, bl);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                if (MODE == 1) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(4 * kb + j) * 16 + r] = acc[j];      // C[m][n]: m = 4*(lane/16) + j, n = lane % 16
}

// issue-rate loops: operands in registers, B re-split every step in MODE 1 (as a convolution would have to)
template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int steps) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};       // 4 independent accumulators (2 output halves x 2)
    float b[8];
    for (int i = 0; i < 8; ++i) b[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    bf16x8 ah, am, al;
    { float a[8]; for (int i = 0; i < 8; ++i) a[i] = 0.5f + 1e-3f * i; split3(a, ah, am, al); }
    for (int s = 0; s < steps; ++s) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t], b[(t + 1) & 7], acc[q], 0, 0, 0);
        } else {
            bf16x8 bh, bm, bl;
            split3(b, bh, bm, bl);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] += 1e-6f;         // new data every step: the split cannot be hoisted
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int j = 0; j < 4; ++j) s += acc[q][j];
    if (s == 123.456f) out[0] = s;
}

int main() {
    const int K = 4096;
    std::vector<float> A(16 * K), B(16 * K);
    unsigned st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) / 16777216.0f - 0.5f) * 2.0f; };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd() * (1.0f + rnd());
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256);
    double scale = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double s = 0, a = 0;
        for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * B[n * K + k]; a += std::fabs((double)A[m * K + k] * B[n * K + k]); }
        ref[m * 16 + n] = s; scale = std::fmax(scale, a);
    }
    const char *names[3] = {"fp32 MFMA 16x16x4", "bf16x3 (6 x 16x16x32_bf16)", "plain bf16 (1 x 16x16x32_bf16)"};
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(gemm16<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1);
        if (mode == 1) hipLaunchKernelGGL(gemm16<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1);
        if (mode == 2) hipLaunchKernelGGL(gemm16<2>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 1);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < 256; ++i) worst = std::fmax(worst, std::fabs(C[i] - ref[i]));
        std::printf("%-34s max |err| of a K=%d dot product: %.3e  (= %.2e of sum |a b|)\n", names[mode], K, worst, worst / scale);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {256, 512, 1024}) {          // 4 / 8 / 16 waves per CU
            float ms;
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(wgs), dim3(256), 0, 0, dC, steps);
                else hipLaunchKernelGGL(rate<1>, dim3(wgs), dim3(256), 0, 0, dC, steps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // per step and wave: 4 accumulators x (16 x 16 x 32) multiply-adds of fp32-equivalent work
            const double flop = 2.0 * 16 * 16 * 32 * 4 * (double)steps * wgs * 4;
            std::printf("%-8s %4d workgroups: %.3f ms  -> %.1f TFLOP/s fp32-equivalent\n", mode ? "bf16x3" : "fp32", wgs, ms, flop / ms / 1e9);
        }
    return 0;
}
