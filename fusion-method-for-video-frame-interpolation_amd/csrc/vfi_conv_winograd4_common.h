// Shared by the two F(4x4,3x3) Winograd translation units (vfi_conv_winograd4.hip: M = 16 output channels per wave, two
// waves per SIMD; vfi_conv_winograd4m.hip: M = 32 per wave, one wave per SIMD, hand-scheduled chunk body): tile geometry
// and LDS layout of a ring slot, the XCD-aware item order, the 6-point transforms.
#pragma once
#include "vfi_conv_common.h"

namespace vfi {
namespace conv {
namespace w4 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Wino4Tile {
    static constexpr int TH = 16, TW = 64, R = TH + 2, CK = 4, BN = 32, THREADS = 512, WAVES = THREADS / 64;
    static constexpr int ROWP = 68;                       // LDS row pitch: 66 columns fetched as 17 float4
    static constexpr int ROW_PIECES = ROWP / 4;           // 17
    static constexpr int PLANE = R * ROWP;                // 1224
    // LDS stride of a channel plane = 0 (mod 64) dwords: the 16-byte patch reads of a 16-lane group (two channels,
    // lanes 16 bytes apart) then cover the 64 banks exactly once
    static constexpr int PLANE_S = (PLANE + 63) / 64 * 64;              // 1280
    static constexpr int PIECES = PLANE_S / 4;            // float4 pieces per plane (320, 306 of them real)
    static constexpr int IN_FLOATS = CK * PLANE_S;        // 5120 = 20 wave-instructions of 16 bytes per lane, 80 of 4 bytes
    static constexpr int IN_WI = IN_FLOATS / 256;         // 16-byte DMA wave-instructions per chunk: wave w issues w, w+8, w+16 (< 20)
    static constexpr int IN_X4 = (IN_WI + WAVES - 1) / WAVES;                       // 3 (waves 4..7: 2)
    static constexpr int IN_X1 = IN_FLOATS / THREADS;     // 4-byte DMA instructions per wave (10)
    static constexpr int NPOS = 36, NGRP = NPOS / 4;
    static constexpr int W_FLOATS = CK * NPOS * BN;       // 4608 = 18 wave-instructions
    static constexpr int W_WI = W_FLOATS / 256;
    static constexpr int W_INSTR = (W_WI + WAVES - 1) / WAVES;                      // 3 (see weight_piece)
    static constexpr int BUF = IN_FLOATS + W_FLOATS;      // 38 KiB
    static constexpr int NBUF = 4;
    static constexpr int BIAS_SLOTS = 8;                  // (> NBUF: the DMA cursor runs up to NBUF one-chunk items ahead)
    static constexpr int BIAS_OFF = NBUF * BUF;           // BIAS_SLOTS x 64 floats
    static constexpr size_t LDS_BYTES = ((size_t)NBUF * BUF + BIAS_SLOTS * 64) * sizeof(float);
    static_assert(IN_X1 * THREADS == IN_FLOATS && IN_FLOATS % 256 == 0 && W_FLOATS % 256 == 0 && PLANE_S % 64 == 0 && ROWP % 4 == 0 &&
                      BIAS_SLOTS > NBUF && LDS_BYTES <= 160 * 1024 && IN_X4 == 3 && W_INSTR == 3, "tile layout");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword data
}
// wave-uniform by construction; pinned to SGPRs (see vfi_conv_winograd.hip)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *uni(const char *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

struct Item {
    int n, x0, y0, nb;
    bool valid;
};

// Same XCD-aware item order as the F(2x2) kernel (vfi_conv_winograd.hip: decode_item), one K split.
__device__ __forceinline__ Item decode_item(const ConvArgs &a, int L) {
    using T = Wino4Tile;
    Item it;
    const int cb = a.Cout_pad / T::BN;
    const int xcd = L & 7, q = L >> 3;
    const int tq = fast_div(q, a.fd_cb), run = a.wino_run;
    it.nb = q - tq * cb;
    const int tr = fast_div(tq, a.fd_run);
    const int tl = (tr * 8 + xcd) * run + (tq - tr * run);
    it.n = fast_div(tl, a.fd_tiles);
    const int t = tl - it.n * a.wino_tiles;
    it.valid = it.n < a.wino_batch;
    const int ty = fast_div(t, a.fd_tiles_x);
    it.x0 = (t - ty * a.tiles_x) * T::TW;
    it.y0 = ty * T::TH;
    return it;
}

// B^T x for the six samples of one line (12 operations)
__device__ __forceinline__ void input_transform6(const float (&x)[6], float (&t)[6]) {
    t[0] = fmaf(4.0f, x[0], fmaf(-5.0f, x[2], x[4]));
    const float p = fmaf(-4.0f, x[2], x[4]), q = fmaf(-4.0f, x[1], x[3]);
    t[1] = p + q;
    t[2] = p - q;
    const float r = x[4] - x[2], s = x[3] - x[1];
    t[3] = fmaf(2.0f, s, r);
    t[4] = fmaf(-2.0f, s, r);
    t[5] = fmaf(4.0f, x[1], fmaf(-5.0f, x[3], x[5]));
}
// A^T m for the six frequency samples of one line (10 operations)
__device__ __forceinline__ void output_transform6(const float (&m)[6], float (&y)[4]) {
    const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = fmaf(2.0f, s, q);
    y[2] = fmaf(4.0f, r, p);
    y[3] = fmaf(8.0f, s, q) + m[5];
}


}  // namespace w4
}  // namespace conv
}  // namespace vfi
