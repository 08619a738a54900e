// Shared by the two convolution translation units (vfi_conv.hip: direct kernels, pack, entry points;
// vfi_conv_winograd.hip: the 3x3 Winograd kernel).
#pragma once
#include "vfi_common.h"

namespace vfi {
namespace conv {

// Exact unsigned division by a run-time constant (Granlund-Montgomery, round-up variant): five scalar instructions
// instead of the ~25 of a 32-bit division; exact for every 32-bit n.
struct FastDiv {
    unsigned m, sh1, sh2;
};
inline FastDiv make_fastdiv(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                  // ceil(log2 d)
    FastDiv f;
    f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__host__ __device__ __forceinline__ int fast_div(int n, const FastDiv &f) {
    const unsigned t = (unsigned)(((unsigned long long)(unsigned)n * f.m) >> 32);      // (one s_mul_hi_u32 / v_mul_hi_u32)
    return (int)((t + (((unsigned)n - t) >> f.sh1)) >> f.sh2);
}

// F(4x4) Winograd kernel (vfi_conv_winograd4.hip; declared here so that tests/native/fastdiv_check.cpp can count the
// requests the kernel's hand-written `s_waitcnt vmcnt` constants assume).  Which 1 KiB piece of a chunk's 18 KiB weight
// slab wave `wave` requests as its t-th (t < 3), or -1.  The 20 input pieces go 3 to each of waves 0..3 and 2 to waves
// 4..7 (piece w + 8k of wave w), so the weight pieces go mostly to the upper waves: 5 | 4 | 5 requests per chunk for
// waves 0,1 | 2,3 | 4..7 (a request holds its wave for 36-70 cycles and the chunk's barrier waits for the slowest wave:
// 3 + 3 on waves 0,1 would cost everyone one more).
__host__ __device__ __forceinline__ int weight_piece(int wave, int t) {
    if (wave >= 4) return (wave - 4) + 4 * t;                 // 0 .. 11
    return t == 0 ? 12 + wave : (t == 1 && wave < 2 ? 16 + wave : -1);
}

struct ConvArgs {
    const float *x;      // (N, Cin, H, W) slice, batch stride x_bs
    const float *wp;     // packed weights [Cin_pad][KS*KS][Cout_pad]
    const float *bias;   // (Cout) or null
    const float *res;    // residual (N, Cout, H, W) slice, batch stride res_bs, or null
    float *y;            // (N, Cout, H, W) slice, batch stride y_bs
    long long x_bs, res_bs, y_bs;
    int Cin, Cin_pad, Cout, Cout_pad, H, W, tiles_x;
    int pad_mode;  // 0 zero, 1 reflect
    int act;       // vfi_act
    long long ws_floats;
    float *ws;     // split-K partial sums [splits][N][Cout][H][W] (splits > 1)
    int splits;    // K (input-channel chunk) range split over `splits` workgroups per output tile
    const float *x2;       // SRC = 2: tensor whose resize forms the first rsz_channels input channels
    long long x2_bs;
    int rsz_channels;      // multiple of CK
    int wino_tiles, wino_items, wino_batch, wino_run;   // Winograd kernel: spatial tiles per sample; work items per K split; N; tiles per XCD run
    FastDiv fd_items, fd_cb, fd_run, fd_tiles, fd_tiles_x, fd_splits;   // ... and their reciprocals (item decoding)
    int Hs, Ws;    // UPS kernels: size of the low-resolution source x (H = 2*Hs, W = 2*Ws)
    float ups_sy, ups_sx;   // (Hs-1)/(H-1), (Ws-1)/(W-1): torch bilinear, align_corners=True
    float *pool;            // Winograd kernel: optional second output, the 2x2 / stride-2 pooled result (N, Cout, H/2, W/2)
    long long pool_bs;
    int pool_max;           // 1: max pooling, 0: average
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.0f);
        // ELU: exp(v) - 1 from the hardware exponential (absolute error <= 1.2e-7, i.e. half an ulp of the -1 it approaches;
        // libm's expm1f keeps RELATIVE accuracy near 0, which an activation does not need, at ~28 instructions and a
        // branch per output against 5)
        // As one median: exp(v) - 1 >= v everywhere, so {v, exp(v) - 1, 0} is ordered v < e < 0 for v < 0 (median e) and
        // 0 < v < e for v > 0 (median v) -- v_med3_f32 instead of a compare and a select, same bits
        case 2: return __builtin_amdgcn_fmed3f(v, __expf(v) - 1.0f, 0.0f);
        case 3: return tanhf(v);
        case 4: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ int reflect_index(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);  // tile overhang beyond the reflected range feeds discarded outputs only
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// vfi_conv.hip
void launch_splitk_reduce(const ConvArgs &a, int N, hipStream_t s);
// vfi_conv_winograd.hip
int launch_winograd(const ConvArgs &a, int N, hipStream_t s);
void launch_pack_winograd(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin, int Cin_pad, int Cout_pad,
                          hipStream_t s);
// vfi_conv_winograd4.hip
bool winograd4_suits(const ConvArgs &a, int N);
int launch_winograd4(const ConvArgs &a, int N, hipStream_t s);
void launch_pack_winograd4(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin, int Cin_pad, int Cout_pad,
                           hipStream_t s);
// vfi_conv_winograd4m.hip: the same work items on one wave per SIMD with 32 output channels per wave (args as prepared by
// launch_winograd4)
bool winograd4m_enabled();
int launch_winograd4m(const ConvArgs &prepared, hipStream_t s);

}  // namespace conv
}  // namespace vfi
