// 3x3 convolutions (stride 1, "same" size, zero or reflect padding): Winograd F(2x2, 3x3) on the gfx950 matrix
// cores, v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).  Same callers as vfi_conv.hip (PhaseNet blocks,
// KernelEstimation U-Net and heads, FusionNet's 3x3 layers).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A : 16 multiply-adds per 2x2 outputs and channel pair instead of 36.
//
//   * a workgroup (256 threads) works on an 8-row x 32-column output tile (4 x 16 Winograd tiles) for 32 output
//     channels; wave w owns tile row w.  M = 16 output channels, N = 16 tiles, K = 4 input channels per MFMA; the 16
//     frequency positions x 2 channel halves are 32 independent accumulators (128 registers) per wave;
//   * per K step each lane reads ONE raw 4x4 input patch (its tile, its channel: 8-byte LDS reads), forms
//     V = B^T d B in registers (32 adds) and feeds V's 16 entries to the 16 x 2 MFMAs; the A operands are the
//     pre-transformed weights U = G g G^T ([cin][row i][cout][column j] in LDS: one ds_read_b128 = four positions);
//   * the 16 position accumulators of one (channel, tile) sit in one lane: the output transform is register-only;
//   * staging is all LDS-DMA (buffer_load ... lds): no staging registers, no ds_write pass, no per-chunk address
//     arithmetic -- per-lane source offsets are tile-invariant, the chunk's channel base and the channel tail live
//     in the (scalar) buffer descriptor, out-of-range offsets (zero padding, tail channels) read as 0.  Interior tiles
//     fetch 16 bytes per lane (2 + 2 DMA instructions per wave and 4-channel chunk), tiles on the left / right image
//     border fetch per element (7 + 2);
//   * a chunk is only 1024 MFMA cycles per wave, far less than the HBM latency, so the chunks flow through a ring of
//     NBUF LDS buffers filled NBUF-1 chunks ahead: ONE raw s_barrier per chunk and a COUNTED s_waitcnt vmcnt;
//   * the fixed cost of a tile (first-chunk latency, 2 x 2 x 8 output rows of stores, workgroup launch) is as long as
//     ~12 chunks when paid serially, so workgroups are PERSISTENT: 2 per CU walk the (tile, channel block) items,
//     the ring keeps running across items (the next item's first chunks are requested during the current item's last
//     MFMAs) and an item's stores drain behind the next item's MFMAs;
//   * items are ordered so that the workgroups of one XCD (own L2) take the channel blocks of the same spatial tile
//     back to back: the input tile comes from HBM once, not Cout/32 times.
#include "vfi_conv_common.h"

using namespace vfi::conv;

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WinoTile {
    static constexpr int TH = 8, TW = 32, R = TH + 2, CK = 4, BN = 32;
    static constexpr int ROWP = 36;                       // LDS row pitch: 34 columns fetched as 9 float4
    static constexpr int PLANE = R * ROWP;                // 360
    // LDS stride of a channel plane = 32 (mod 64) dwords: the four channels of a K step (lanes 16k..16k+15 read
    // channel k) then fall into disjoint bank halves for the 8-byte patch reads
    static constexpr int PLANE_S = (PLANE - 32 + 63) / 64 * 64 + 32;   // 416
    static constexpr int PIECES = PLANE_S / 4;            // float4 pieces per plane (104, 90 of them real)
    static constexpr int IN_X4 = (CK * PIECES + 255) / 256;             // 16-byte DMA instructions per wave (2)
    static constexpr int IN_X1 = (CK * PLANE_S + 255) / 256;            // 4-byte DMA instructions per wave (7)
    static constexpr int IN_FLOATS = IN_X4 * 1024;        // 2048 >= IN_X1 * 256
    static constexpr int W_FLOATS = CK * 16 * BN, W_INSTR = W_FLOATS / 4 / 256;   // 2048 floats, 2 per wave
    static constexpr int BUF = IN_FLOATS + W_FLOATS;      // 16 KiB
    static constexpr int NBUF = 4;
    static constexpr int MIN_LOADS = IN_X4 + W_INSTR;     // fewest DMA instructions a chunk issues per wave
    // the bias of every item that can be in flight: the DMA cursor runs up to NBUF chunks = up to NBUF (one-chunk)
    // items ahead of the item whose epilogue reads its slot
    static constexpr int BIAS_SLOTS = 8;
    static constexpr int BIAS_OFF = NBUF * BUF;          // BIAS_SLOTS x 64 floats
    static constexpr size_t LDS_BYTES = ((size_t)NBUF * BUF + BIAS_SLOTS * 64) * sizeof(float);
    static_assert(IN_X1 * 256 <= IN_FLOATS && PLANE_S % 4 == 0 && ROWP % 4 == 0 && BIAS_SLOTS > NBUF, "tile layout");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);   // raw buffer, dword data
}

// The cursors below are wave-uniform by construction, but the compiler cannot always prove it (and then wraps every DMA
// in a waterfall loop over "divergent" descriptors): pin them to SGPRs.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char *uni(const char *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

// One work item = (K split, sample, spatial tile, 32-channel block).
struct Item {
    int split, n, x0, y0, nb;
    bool valid;
};

__device__ __forceinline__ Item decode_item(const ConvArgs &a, int L) {
    using T = WinoTile;
    Item it;
    const int cb = a.Cout_pad / T::BN;
    it.split = fast_div(L, a.fd_items);
    const int Lr = L - it.split * a.wino_items;
    // XCD-aware order.  Workgroup L runs on XCD L % 8 (own L2 each).  Consecutive slots of one XCD take the channel
    // blocks of the SAME spatial tile, then the next tile of a run of `wino_run` horizontally adjacent tiles: the input
    // tile is fetched once for all channel blocks, and the 128-byte lines a tile shares with its left / right neighbours
    // (a tile is exactly one line wide, its halo touches both neighbouring lines) stay inside one L2 within a run
    // (fabric reads of the 1080p 64->64 layers: 3.4 -> 1.5 GB by FETCH_SIZE).  Runs are dealt round-robin to the XCDs.
    // (run-time divisors through their precomputed reciprocals: the decoding runs once per item and wave, twice over)
    const int xcd = Lr & 7, q = Lr >> 3;
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

// RES: a residual tensor is added after the activation.  Its (ordinary) loads make the compiler drain the DMA ring
// in every item epilogue, so layers without a residual get an instantiation without them.
// ACT: the activation as a compile-time constant (-1: read it from the arguments) -- the epilogue applies it to 64 values
// per lane, and five inlined branches per value are most of the kernel's code size.
// POOL: the lane that holds a 2x2 output block also writes its pooled value (AvgPool2d / MaxPool2d(2) after the layer,
// src/fusion_net/fusion_adacofnet.py:76-89, src/fusion_net/fusion_net.py:41,59): no separate pass re-reads the output.
template <bool RES, int ACT, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_winograd_kernel(const ConvArgs a) {
    using T = WinoTile;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(threadIdx.x >> 6));      // (in an SGPR: LDS-DMA destinations are scalar arithmetic)
    const int n16 = lane & 15, k4 = lane >> 4;       // lane roles in an MFMA: tile column / channel of the K = 4 step
    const int HW = a.H * a.W, G = gridDim.x, N = a.wino_batch;
    const int Ltotal = a.wino_items * a.splits;
    const int nchunks_all = (a.Cin + T::CK - 1) / T::CK;

    auto next_valid = [&](int L) {      // first valid item at or after L in this workgroup's sequence
        while (L < Ltotal && !decode_item(a, L).valid) L += G;
        return L;
    };

    // ------------------------------------------------------------------------------------------------------------
    // DMA side ("issue cursor"): item iL, chunk ich of [.., ich_end); per-lane source offsets of the item's tile
    // ------------------------------------------------------------------------------------------------------------
    // Per chunk the cursor only moves two addresses and two counters (scalar adds): the descriptors are rebuilt from
    // them, nothing is multiplied or divided in the chunk loop.
    int iL = next_valid(blockIdx.x), ileft = 0, iseq = -1, inb = 0;      // ileft: chunks of the item still to request
    unsigned in_bytes_left = 0;                                              // bytes from the chunk's first channel to the input's end
    const char *in_ptr = nullptr, *w_ptr = nullptr;
    bool iborder = false, ifirst = false;
    unsigned voff[T::IN_X1], woff[T::W_INSTR];
    const unsigned in_chunk_bytes = (unsigned)T::CK * (unsigned)HW * 4u, w_chunk_bytes = (unsigned)T::CK * 4u * (unsigned)a.Cout_pad * 16u;
    auto setup_issue = [&]() {
        const Item it = decode_item(a, iL);
        ++iseq;
        ifirst = true;
        inb = it.nb;
        const int ch0 = fast_div(nchunks_all * it.split, a.fd_splits);
        ileft = fast_div(nchunks_all * (it.split + 1), a.fd_splits) - ch0;
        in_ptr = reinterpret_cast<const char *>(a.x + (size_t)it.n * a.x_bs) + (size_t)ch0 * in_chunk_bytes;
        in_bytes_left = (unsigned)(a.Cin - ch0 * T::CK) * (unsigned)HW * 4u;       // (host: Cin*H*W*4 < 2^32)
        w_ptr = reinterpret_cast<const char *>(a.wp) + (size_t)ch0 * w_chunk_bytes;
        iborder = it.x0 == 0 || it.x0 + T::TW >= a.W;      // a fetched 16-byte piece would wrap around an image row
        if (iborder) {
#pragma unroll
            for (int i = 0; i < T::IN_X1; ++i) {
                const int e = 64 * (wave + 4 * i) + lane;            // LDS dword inside the buffer's input part
                const int c = e / T::PLANE_S, rem = e % T::PLANE_S, r = rem / T::ROWP, xx = rem % T::ROWP;
                int gy = it.y0 - 1 + r, gx = it.x0 - 1 + xx;
                bool ok = c < T::CK && rem < T::PLANE && xx < T::TW + 2;
                if (a.pad_mode == 1) {
                    gy = reflect_index(gy, a.H);
                    gx = reflect_index(gx, a.W);
                } else {
                    ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                }
                voff[i] = ok ? (unsigned)(c * HW + gy * a.W + gx) * 4u : 0xffffffffu;   // out of range -> the DMA writes 0
            }
        } else {
#pragma unroll
            for (int t = 0; t < T::IN_X4; ++t) {
                const int f = 64 * (wave + 4 * t) + lane;            // float4 piece inside the buffer's input part
                const int c = f / T::PIECES, rem = f % T::PIECES, r = rem / (T::ROWP / 4), q = rem % (T::ROWP / 4);
                int gy = it.y0 - 1 + r;
                bool ok = c < T::CK && rem < T::PLANE / 4;
                if (a.pad_mode == 1) gy = reflect_index(gy, a.H);
                else ok = ok && gy >= 0 && gy < a.H;
                voff[t] = ok ? (unsigned)(c * HW + gy * a.W + it.x0 - 1 + 4 * q) * 4u : 0xffffffffu;
            }
        }
#pragma unroll
        for (int t = 0; t < T::W_INSTR; ++t) {
            const int f = 64 * (wave + 4 * t) + lane;            // float4 inside the slab [4 cin][4 i][32 cout][4 j]
            woff[t] = (unsigned)(((f / T::BN) * a.Cout_pad + it.nb * T::BN + f % T::BN) * 16);
        }
    };

    // Requests the next chunk of this workgroup's item sequence into ring slot `slot` and advances the cursor.  Past
    // the end it still issues MIN_LOADS (empty) DMA instructions so that the counted wait below stays valid.
    auto issue_next = [&](int slot) {
        float *b = lds + uni(slot) * T::BUF;
        const bool live = uni(iL) < Ltotal;
        const bool border = uni(iborder ? 1 : 0) != 0, first = uni(ifirst ? 1 : 0) != 0;
        // (the channel tail and everything past the end read as 0: zero-sized / shortened buffers)
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(uni(in_ptr), live ? (unsigned)uni((int)in_bytes_left) : 0u);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(uni(w_ptr), live ? w_chunk_bytes : 0u);
        if (live && border) {
#pragma unroll
            for (int i = 0; i < T::IN_X1; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(b + 64 * (wave + 4 * i)),
                                                         4, voff[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < T::IN_X4; ++t)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(b + 256 * (wave + 4 * t)),
                                                         16, voff[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T::W_INSTR; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(b + T::IN_FLOATS + 256 * (wave + 4 * t)),
                                                     16, woff[t], 0, 0, 0);
        // The item's bias goes through LDS as well: an ordinary load in the epilogue would make the compiler drain
        // the whole DMA ring (vmcnt(0)) before its first use.  (More DMA instructions only make the counted wait
        // earlier; a null / short bias reads as 0.)
        if (live && first && uni(wave) == 0) {
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bias ? a.bias : a.x, a.bias ? (unsigned)a.Cout * 4u : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void *)(lds + T::BIAS_OFF + (uni(iseq) & (T::BIAS_SLOTS - 1)) * 64),
                                                     4, lane < T::BN ? (unsigned)(uni(inb) * T::BN + lane) * 4u : 0xffffffffu, 0, 0, 0);
        }
        ifirst = false;
        if (live) {
            in_ptr += in_chunk_bytes;
            w_ptr += w_chunk_bytes;
            in_bytes_left = in_bytes_left > in_chunk_bytes ? in_bytes_left - in_chunk_bytes : 0u;
            if (--ileft == 0) {
                iL = next_valid(iL + G);
                if (iL < Ltotal) setup_issue();
            }
        }
    };
    if (iL < Ltotal) setup_issue();
#pragma unroll
    for (int p = 0; p < T::NBUF; ++p) issue_next(p);

    // ------------------------------------------------------------------------------------------------------------
    // MFMA side: a software pipeline over the workgroup's flattened (item, chunk) sequence, in half chunks (rows 0-1
    // and rows 2-3 of the 4x4 frequency grid, 16 MFMAs each).  While one half's MFMAs issue, the operands of the next
    // half are fetched from LDS:
    //     [fetch U rows 2-3 of c]   MFMA rows 0-1 of c
    //     wait + barrier            (chunk c+1 visible to everyone, everyone's reads of chunk c retired)
    //     DMA chunk c+NBUF -> slot of c ; [fetch patch + U rows 0-1 of c+1]   MFMA rows 2-3 of c ; V(c+1) = B^T d B
    // so neither the LDS latency nor the DMA issue sits in front of an MFMA, and one barrier per chunk remains.
    // ------------------------------------------------------------------------------------------------------------
    const int b_base = k4 * T::PLANE_S + (2 * wave) * T::ROWP + 2 * n16;
    const int a_base = T::IN_FLOATS + (k4 * 4 * T::BN + n16) * 4;
    auto fetch_u = [&](float4 (&u)[2][2], int slot, int i_begin) {       // U rows i_begin, i_begin+1 for both channel halves
        const float *w_s = lds + slot * T::BUF + a_base;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float *pu = w_s + ((i_begin + i) * T::BN + mb * 16) * 4;      // (scalar-typed: merged to ds_read_b128)
                u[mb][i] = make_float4(pu[0], pu[1], pu[2], pu[3]);
            }
    };
    auto fetch_d = [&](float (&d)[4][4], int slot) {                    // this lane's raw 4x4 patch
        const float *in_s = lds + slot * T::BUF + b_base;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[r][j] = in_s[r * T::ROWP + j];      // (merged to 8-byte reads)
    };
    auto transform = [&](const float (&d)[4][4], float (&v)[4][4]) {    // V = B^T d B
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i][0] = t[i][0] - t[i][2]; v[i][1] = t[i][1] + t[i][2];
            v[i][2] = t[i][2] - t[i][1]; v[i][3] = t[i][1] - t[i][3];
        }
    };
    f32x4 acc[2][16];     // [16-channel half][frequency position 4*i + j]
    auto zero_acc = [&]() {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[mb][p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    auto mfma_rows = [&](const float4 (&u)[2][2], const float (&v)[4][4], int i_begin) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int r = i_begin + i;
                acc[mb][4 * r + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[mb][i].x, v[r][0], acc[mb][4 * r + 0], 0, 0, 0);
                acc[mb][4 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[mb][i].y, v[r][1], acc[mb][4 * r + 1], 0, 0, 0);
                acc[mb][4 * r + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[mb][i].z, v[r][2], acc[mb][4 * r + 2], 0, 0, 0);
                acc[mb][4 * r + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[mb][i].w, v[r][3], acc[mb][4 * r + 3], 0, 0, 0);
            }
    };

    int cL = next_valid(blockIdx.x);
    if (cL >= Ltotal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    Item it = decode_item(a, cL);
    int ch = fast_div(nchunks_all * it.split, a.fd_splits), ch_end = fast_div(nchunks_all * (it.split + 1), a.fd_splits);
    int slot = 0, cseq = 0;
    // The oldest chunk in flight has landed once no more than the DMA instructions of the younger chunks in flight are
    // outstanding (>= MIN_LOADS each; an item's stores in between only make the wait earlier).  The barrier makes every
    // wave's part visible.
    static_assert((T::NBUF - 1) * T::MIN_LOADS == 12 && (T::NBUF - 2) * T::MIN_LOADS == 8, "update the counted waits");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float4 u01[2][2], u23[2][2];
    float v[4][4];
    {
        float d[4][4];
        fetch_d(d, 0);
        fetch_u(u01, 0, 0);
        transform(d, v);
    }
    zero_acc();
    while (true) {
        fetch_u(u23, slot, 2);
        __builtin_amdgcn_sched_barrier(0);      // (keep the LDS reads in front of the MFMAs they overlap with)
        mfma_rows(u01, v, 0);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's reads of the current slot are done; after the barrier everyone's are, and chunk c+1 is visible
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int cur_slot = slot;
        slot = slot + 1 == T::NBUF ? 0 : slot + 1;
        float d[4][4];
        fetch_d(d, slot);              // (after the workgroup's last chunk: an empty look-ahead slot, unused)
        fetch_u(u01, slot, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(u23, v, 2);
        __builtin_amdgcn_sched_barrier(0);
        // The non-MFMA stretch of a chunk (DMA requests, cursor, transform) runs at raised priority: the sooner this wave
        // is back at its MFMAs, the less often both waves of a SIMD are away from the matrix pipe at once (+2 %).
        __builtin_amdgcn_s_setprio(3);
        issue_next(cur_slot);          // its issue cost overlaps the matrix-core work queued above
        transform(d, v);
        __builtin_amdgcn_s_setprio(0);
        const bool last_of_item = ch + 1 == ch_end;
        const int nL = last_of_item ? next_valid(cL + G) : cL;
        const bool more = nL < Ltotal;
        ++ch;
        if (last_of_item) {
            // ---- item epilogue: Y = A^T M A per lane: channel co = nb*32 + mb*16 + 4*k4 + j, tile (row `wave`, column n16) ----
            const int gxw = it.x0 + 2 * n16, gy0 = it.y0 + 2 * wave;
            const bool split_out = a.splits > 1;   // split-K: raw partial sums; bias / activation / residual in the reduce kernel
            const float *__restrict__ resp = (RES && !split_out && a.res) ? a.res + (size_t)it.n * a.res_bs : nullptr;
            float *__restrict__ yp = split_out ? a.ws + ((size_t)it.split * N + it.n) * a.Cout * HW : a.y + (size_t)it.n * a.y_bs;
            const int act = ACT >= 0 ? ACT : (split_out ? 0 : a.act);
            const int co0 = it.nb * T::BN + 4 * k4;            // this lane's channel for (mb, j) = (0, 0)
            const float *bias_l = lds + T::BIAS_OFF + (cseq & (T::BIAS_SLOTS - 1)) * 64 + 4 * k4;
            auto outputs = [&](int mb, int j, float2 (&o)[2]) {     // the 2x2 outputs of (mb, j), before bias
                float s0[4], s1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = acc[mb][c][j] + acc[mb][4 + c][j] + acc[mb][8 + c][j];
                    s1[c] = acc[mb][4 + c][j] - acc[mb][8 + c][j] - acc[mb][12 + c][j];
                }
                o[0] = make_float2(s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3]);
                o[1] = make_float2(s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]);
            };
            // Fast path (wave-uniform, straight-line): the tile lies inside the image and rows are 16-byte aligned.  Lane
            // pairs (even / odd tile column) swap halves so that each lane stores ONE float4 per channel (the even lane
            // the upper output row of both tiles, the odd lane the lower row): 8 stores per lane instead of 16.
            const bool vec4 = it.x0 + T::TW <= a.W && it.y0 + T::TH <= a.H && (a.W % 4 == 0) && ((reinterpret_cast<size_t>(yp) & 15) == 0) &&
                              (!resp || (reinterpret_cast<size_t>(resp) & 15) == 0);
            // pooled output: this lane's 2x2 block -> pixel (gy0/2, gxw/2) of the (H/2, W/2) plane
            const int Hq = a.H >> 1, Wq = a.W >> 1;
            float *__restrict__ pq = POOL ? a.pool + (size_t)it.n * a.pool_bs + (size_t)co0 * Hq * Wq + (size_t)(gy0 >> 1) * Wq + (gxw >> 1) : nullptr;
            const bool pool_ok = POOL && (gy0 >> 1) < Hq && (gxw >> 1) < Wq;
            auto pooled = [&](const float2 (&o)[2], float b) {
                const float p0 = apply_act(o[0].x + b, act), p1 = apply_act(o[0].y + b, act);
                const float p2 = apply_act(o[1].x + b, act), p3 = apply_act(o[1].y + b, act);
                return a.pool_max ? fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)) : (p0 + p1 + p2 + p3) * 0.25f;      // (same order as vfi_pool2)
            };
            if (vec4) {
                const bool odd = n16 & 1;
                const size_t off = (size_t)co0 * HW + (size_t)(gy0 + (odd ? 1 : 0)) * a.W + (gxw & ~3);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int dc = mb * 16 + j;            // channel step from co0
                        float2 o[2];
                        outputs(mb, j, o);
                        if (POOL && pool_ok && co0 + dc < a.Cout) pq[(size_t)dc * Hq * Wq] = pooled(o, bias_l[dc]);
                        // quad_perm [1,0,3,2]: exchange with the neighbouring tile column
                        const float2 give = odd ? o[0] : o[1];
                        float2 got;
                        got.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.x), 0xB1, 0xf, 0xf, false));
                        got.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give.y), 0xB1, 0xf, 0xf, false));
                        float4 q = odd ? make_float4(got.x, got.y, o[1].x, o[1].y) : make_float4(o[0].x, o[0].y, got.x, got.y);
                        const float b = split_out ? 0.0f : bias_l[dc];
                        q.x = apply_act(q.x + b, act); q.y = apply_act(q.y + b, act);
                        q.z = apply_act(q.z + b, act); q.w = apply_act(q.w + b, act);
                        if (co0 + dc < a.Cout) {
                            if (RES && resp) {
                                const float4 r = *reinterpret_cast<const float4 *>(resp + off + (size_t)dc * HW);
                                q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
                            }
                            *reinterpret_cast<float4 *>(yp + off + (size_t)dc * HW) = q;
                        }
                    }
            } else if (gy0 < a.H && gxw < a.W) {      // image border / unaligned rows: element-wise
                const size_t off = (size_t)co0 * HW + (size_t)gy0 * a.W + gxw;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int dc = mb * 16 + j;
                        if (co0 + dc >= a.Cout) continue;
                        float2 o[2];
                        outputs(mb, j, o);
                        const float b = split_out ? 0.0f : bias_l[dc];
                        if (POOL && pool_ok) pq[(size_t)dc * Hq * Wq] = pooled(o, b);
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
                            if (gy0 + dy < a.H) {
                                const size_t e = off + (size_t)dc * HW + (size_t)dy * a.W;
                                float vx = apply_act(o[dy].x + b, act), vy = apply_act(o[dy].y + b, act);
                                if (RES && resp) {
                                    vx += resp[e];
                                    if (gxw + 1 < a.W) vy += resp[e + 1];
                                }
                                yp[e] = vx;
                                if (gxw + 1 < a.W) yp[e + 1] = vy;
                            }
                    }
            }
            if (!more) break;
            cL = nL;
            ++cseq;
            it = decode_item(a, cL);
            ch = fast_div(nchunks_all * it.split, a.fd_splits);
            ch_end = fast_div(nchunks_all * (it.split + 1), a.fd_splits);
            zero_acc();
        }
    }
    // the (empty) look-ahead DMAs must have retired before the workgroup's LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 3x3 OIHW -> Winograd F(2x2,3x3) weights U = G g G^T as [Cin_pad][4 (row i)][Cout_pad][4 (column j)], BatchNorm
// scale folded.  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; evaluated in double, rounded once.
__global__ void conv2d_pack_winograd_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                            float *__restrict__ out, int Cout, int Cin, int Cin_pad, int Cout_pad) {
    const size_t total = (size_t)Cin_pad * 16 * Cout_pad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = e & 3, co = (e >> 2) % Cout_pad, i = (e / ((size_t)4 * Cout_pad)) & 3, ci = e / ((size_t)16 * Cout_pad);
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) u += G[i][r] * (double)g[r * 3 + c] * G[j][c];
            if (scale) u *= (double)scale[co];
        }
        out[e] = (float)u;
    }
}

}  // namespace

void vfi::conv::launch_pack_winograd(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin, int Cin_pad,
                                     int Cout_pad, hipStream_t s) {
    const long long total = (long long)Cin_pad * 16 * Cout_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv2d_pack_winograd_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, scale, packed, Cout, Cin, Cin_pad, Cout_pad);
}

int vfi::conv::launch_winograd(const ConvArgs &a, int N, hipStream_t s) {
    using T = WinoTile;
    static int resident_dev[vfi::kMaxDevices] = {};   // persistent grid: 2 workgroups per CU; per device, idempotent
    int &resident = resident_dev[vfi::current_device()];
    if (!resident) {
        hipError_t e = hipSuccess;
        for (const void *k : {reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 0>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 1>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 2>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 3>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 4>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<false, 1, true>),
                              reinterpret_cast<const void *>(conv3x3_winograd_kernel<true, -1>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        int dev = 0, cus = 0;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || cus <= 0) return vfi::fail(VFI_ERR_LAUNCH, "vfi_conv2d: Winograd kernel setup: %s", hipGetErrorString(e));
        resident = 2 * cus;
    }
    ConvArgs b = a;
    const int cb = a.Cout_pad / T::BN;
    b.wino_tiles = a.tiles_x * vfi::ceil_div(a.H, T::TH);
    b.wino_batch = N;
    // run length of x-adjacent tiles per XCD: as long as every XCD still gets >= 32 runs (<= 3 % imbalance)
    b.wino_run = 1;
    while (b.wino_run < 8 && (long long)b.wino_tiles * N >= 256ll * 2 * b.wino_run) b.wino_run *= 2;
    b.wino_items = round_up(b.wino_tiles * N, 8 * b.wino_run) * cb;
    // Split-K: few, long items (deep U-Net levels) leave most of the resident workgroups idle; splitting the channel
    // loop S ways makes S times as many items of 1/S the length (partial sums reduced deterministically afterwards).
    b.splits = 1;
    const int nchunks = vfi::ceil_div(a.Cin, T::CK);
    const long long out_floats = (long long)N * a.Cout * a.H * a.W;
    // Cost model (microseconds, measured constants): rounds of resident workgroups x ~1.1 us per chunk of an item, plus
    // the reduce pass that reads S partial tensors and writes one (~2.5 TB/s, + a launch).
    if (a.ws && nchunks >= 16 && b.wino_items < 4 * resident) {
        auto cost = [&](int S) {
            const double rounds = (double)(((long long)b.wino_items * S + resident - 1) / resident);
            const double reduce = S > 1 ? 4.0 + (double)(S + 1) * out_floats * 4.0 / 2.5e6 : 0.0;
            return rounds * (1.1 * nchunks / S) + reduce;
        };
        int best = 1;
        for (int S = 2; S <= 16; S *= 2)
            if (nchunks / S >= 8 && out_floats * S <= a.ws_floats && cost(S) < cost(best)) best = S;
        if (cost(best) <= 0.9 * cost(1)) b.splits = best;
    }
    b.fd_items = make_fastdiv((unsigned)b.wino_items);
    b.fd_cb = make_fastdiv((unsigned)cb);
    b.fd_run = make_fastdiv((unsigned)b.wino_run);
    b.fd_tiles = make_fastdiv((unsigned)b.wino_tiles);
    b.fd_tiles_x = make_fastdiv((unsigned)b.tiles_x);
    b.fd_splits = make_fastdiv((unsigned)b.splits);
    const long long items = (long long)b.wino_items * b.splits;
    dim3 grid((unsigned)(items < resident ? items : resident));
    const int act = b.splits > 1 ? 0 : b.act;      // split-K: the reduce kernel applies bias / activation / residual
    // pooled second output: in the epilogue when the layer is a plain ReLU layer in one piece, else a pass afterwards
    const bool pool_fused = b.pool && b.splits == 1 && !b.res && act == 1;
    if (pool_fused) {
        hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 1, true>), grid, dim3(256), T::LDS_BYTES, s, b);
        return vfi::check_launch("vfi_conv2d");
    }
    if (b.res && b.splits == 1) hipLaunchKernelGGL((conv3x3_winograd_kernel<true, -1>), grid, dim3(256), T::LDS_BYTES, s, b);
    else if (act == 0) hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 0>), grid, dim3(256), T::LDS_BYTES, s, b);
    else if (act == 1) hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 1>), grid, dim3(256), T::LDS_BYTES, s, b);
    else if (act == 2) hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 2>), grid, dim3(256), T::LDS_BYTES, s, b);
    else if (act == 3) hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 3>), grid, dim3(256), T::LDS_BYTES, s, b);
    else hipLaunchKernelGGL((conv3x3_winograd_kernel<false, 4>), grid, dim3(256), T::LDS_BYTES, s, b);
    if (b.splits > 1) launch_splitk_reduce(b, N, s);
    if (b.pool) {
        const int rc = vfi_pool2(b.y, b.y_bs, b.pool, b.pool_bs, N, b.Cout, b.H, b.W, b.pool_max, s);
        if (rc) return rc;
    }
    return vfi::check_launch("vfi_conv2d");
}
