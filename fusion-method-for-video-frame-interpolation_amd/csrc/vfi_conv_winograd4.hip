// 3x3 convolutions (stride 1, "same" size, zero or reflect padding) with the larger Winograd tile F(4x4, 3x3) on the
// gfx950 matrix cores (v_mfma_f32_16x16x4_f32): 36 multiply-adds per 4x4 outputs and channel pair instead of the 64 of
// F(2x2, 3x3) (vfi_conv_winograd.hip), i.e. 1.78x fewer MFMAs, for layers without a K split that are large enough to fill
// the chip; everything else stays with the F(2x2) kernel.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A     with the 6x6 transforms of the points {0, +-1, +-2, inf}
//
//   * one 512-thread workgroup per CU works on a 16-row x 64-column output tile (4 x 16 Winograd tiles) for 32 output
//     channels: wave w owns tile row w & 3 and the 16-channel half w >> 2 (M = 16 channels, N = 16 tiles, K = 4 input
//     channels per MFMA; the 36 frequency positions are 36 independent accumulators, 144 registers);
//   * per K step a lane reads ONE raw 6x6 input patch (its tile, its channel: a 16-byte and an 8-byte LDS read per row),
//     forms V = B^T d B in registers (12 one-dimensional transforms of 12 operations) and feeds V's 36 entries to 36
//     MFMAs whose A operands are the pre-transformed weights U = G g G^T ([cin][group of 4 positions][cout][4] in LDS:
//     one ds_read_b128 = four positions);
//   * the 36 position accumulators of one (channel, tile) sit in one lane: the output transform is register-only and
//     a lane stores its 4x4 block as four 16-byte row segments per channel (16 lanes = one 256-byte run per row);
//   * staging is all LDS-DMA (buffer_load ... lds) as in the F(2x2) kernel: tile-invariant per-lane source offsets,
//     chunk base and channel tail in the buffer descriptor, out-of-range offsets read as 0; interior tiles fetch 16
//     bytes per lane (38 DMA instructions per workgroup and 4-channel chunk), tiles on the left / right image border per
//     element (10 + 3); a ring of 4 buffers of 38 KiB (the weight slab of a chunk is 18 KiB: 36 positions x 32
//     channels x 4) filled three chunks ahead, ONE barrier per chunk and a counted s_waitcnt vmcnt; the requests of a
//     chunk are spread over the six rows of MFMAs of an earlier one (all eight waves of the CU run in step);
//   * persistent workgroups walk the (tile, channel block) items in the XCD-aware order of the F(2x2) kernel.
// Accuracy: the transforms' constants (up to 8 in A, 5 in B, 1/24 in G) cost about a decimal digit -- rms error 1.9e-6 of
// the output rms against 2.7e-7 for F(2x2) (tools/probes/winograd_f43_error.py).
#include "vfi_conv_winograd4_common.h"

#include <cstdlib>

using namespace vfi::conv;

namespace {

using namespace vfi::conv::w4;

// ACT: the activation as a compile-time constant (-1: read it from the arguments).  RES: a residual tensor is added after
// the activation.  POOL: the lane that holds a 4x4 output block also writes its four 2x2-pooled values (AvgPool2d /
// MaxPool2d(2) after the layer: no separate pass re-reads the output; same evaluation order as vfi_pool2).
template <int ACT, bool RES = false, bool POOL = false>
__global__ __launch_bounds__(512, 2) void conv3x3_winograd4_kernel(const ConvArgs a) {
    using T = Wino4Tile;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni((int)(threadIdx.x >> 6));      // (the wave index lives in an SGPR:
                                                                                             //  LDS-DMA destinations are scalar arithmetic)
    const int n16 = lane & 15, k4 = lane >> 4;       // lane roles in an MFMA: tile column / channel of the K = 4 step
    const int trow = wave & 3, chalf = wave >> 2;    // wave roles: tile row / 16-channel half
    const int HW = a.H * a.W, G = gridDim.x;
    const int Ltotal = a.wino_items;
    const int nchunks = (a.Cin + T::CK - 1) / T::CK;

    auto next_valid = [&](int L) __attribute__((always_inline)) {      // first valid item at or after L in this workgroup's sequence
        while (L < Ltotal && !decode_item(a, L).valid) L += G;
        return L;
    };

    // ------------------------------------------------------------------------------------------------------------
    // DMA side ("issue cursor"), as in the F(2x2) kernel
    // ------------------------------------------------------------------------------------------------------------
    int iL = next_valid(blockIdx.x), ileft = 0, iseq = -1, inb = 0;
    unsigned in_bytes_left = 0;
    const char *in_ptr = nullptr, *w_ptr = nullptr;
    bool iborder = false, ifirst = false;
    unsigned voff[T::IN_X1], woff[T::W_INSTR];
    const unsigned in_chunk_bytes = (unsigned)T::CK * (unsigned)HW * 4u, w_chunk_bytes = (unsigned)T::CK * (unsigned)T::NPOS * 4u * (unsigned)a.Cout_pad;
    auto setup_issue = [&]() __attribute__((always_inline)) {
        const Item it = decode_item(a, iL);
        ++iseq;
        ifirst = true;
        inb = it.nb;
        ileft = nchunks;
        in_ptr = reinterpret_cast<const char *>(a.x + (size_t)it.n * a.x_bs);
        in_bytes_left = (unsigned)a.Cin * (unsigned)HW * 4u;       // (host: Cin*H*W*4 < 2^32)
        w_ptr = reinterpret_cast<const char *>(a.wp);
        iborder = it.x0 == 0 || it.x0 + T::TW >= a.W;      // a fetched 16-byte piece with columns in use would wrap around an image row
        if (iborder) {
#pragma unroll
            for (int i = 0; i < T::IN_X1; ++i) {
                const int e = 64 * (wave + T::WAVES * i) + lane;     // LDS dword inside the buffer's input part
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
                const int f = 64 * (wave + T::WAVES * t) + lane;     // float4 piece inside the buffer's input part
                const int c = f / T::PIECES, rem = f % T::PIECES, r = rem / T::ROW_PIECES, q = rem % T::ROW_PIECES;
                int gy = it.y0 - 1 + r;
                bool ok = c < T::CK && rem < T::PLANE / 4;
                if (a.pad_mode == 1) gy = reflect_index(gy, a.H);
                else ok = ok && gy >= 0 && gy < a.H;
                voff[t] = ok ? (unsigned)(c * HW + gy * a.W + it.x0 - 1 + 4 * q) * 4u : 0xffffffffu;
            }
        }
#pragma unroll
        for (int t = 0; t < T::W_INSTR; ++t) {
            const int wp = weight_piece(wave, t);
            const int f = 64 * (wp < 0 ? 0 : wp) + lane;             // float4 inside the slab [4 cin][9 groups][32 cout][4]
            woff[t] = (unsigned)(((f / T::BN) * a.Cout_pad + it.nb * T::BN + f % T::BN) * 16);      // (issued only for wp >= 0)
        }
    };
    // The DMA requests of the next chunk of this workgroup's item sequence into ring slot `slot`.  One workgroup per CU
    // means that all eight waves reach this point together: issued in one go, 38 requests queue up in front of the
    // texture addresser and every wave sits behind them, so the requests of an interior tile are handed out ONE AT A TIME
    // (the caller spreads them over the chunk's rows of MFMAs: wave-instruction w + 8 t of the input part for t < 3, of
    // the weight part for t >= 3); only the per-element requests of an image-border tile go out at once.  Past the end of
    // the sequence the requests are still issued, empty, so that the counted wait below stays valid.
    struct Dma {
        float *buf;
        bool live, x4;
    };
    __amdgpu_buffer_rsrc_t rin_c = make_rsrc(a.x, 0u), rw_c = make_rsrc(a.wp, 0u);     // this chunk's descriptors (dma_begin)
    auto dma_begin = [&](int slot) __attribute__((always_inline)) {
        float *b = lds + uni(slot) * T::BUF;
        const bool live = uni(iL) < Ltotal, border = live && uni(iborder ? 1 : 0) != 0;
        if (border) {
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(uni(in_ptr), (unsigned)uni((int)in_bytes_left));
#pragma unroll
            for (int i = 0; i < T::IN_X1; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(b + 64 * (wave + T::WAVES * i)),
                                                         4, voff[i], 0, 0, 0);
        }
        // the item's bias goes through LDS as well (an ordinary load would make the compiler drain the DMA ring)
        if (live && uni(ifirst ? 1 : 0) != 0 && uni(wave) == 0) {
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bias ? a.bias : a.x, a.bias ? (unsigned)a.Cout * 4u : 0u);
            // (the lane index is recomputed here: kept in a register it gets spilled, and its reload -- an ordinary vector
            // memory load -- has to wait for the whole DMA ring, vmcnt(0))
            const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void *)(lds + T::BIAS_OFF + (uni(iseq) & (T::BIAS_SLOTS - 1)) * 64),
                                                     4, ln < (unsigned)T::BN ? ((unsigned)uni(inb) * T::BN + ln) * 4u : 0xffffffffu, 0, 0, 0);
        }
        ifirst = false;
        rin_c = make_rsrc(uni(in_ptr), live ? (unsigned)uni((int)in_bytes_left) : 0u);
        rw_c = make_rsrc(uni(w_ptr), live ? w_chunk_bytes : 0u);
        return Dma{b, live, !border};
    };
    auto dma_piece = [&](const Dma &d, int k) __attribute__((always_inline)) {      // k = 0 .. 5 (a constant after unrolling)
        if (k < T::IN_X4) {
            if (d.x4 && uni(wave) + T::WAVES * k < T::IN_WI) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin_c, (__attribute__((address_space(3))) void *)(d.buf + 256 * (wave + T::WAVES * k)),
                                                         16, voff[k], 0, 0, 0);
            }
        } else {
            const int t = k - T::IN_X4;
            const int wp = weight_piece(uni(wave), t);
            if (wp >= 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_c, (__attribute__((address_space(3))) void *)(d.buf + T::IN_FLOATS + 256 * wp),
                                                         16, woff[t], 0, 0, 0);
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {       // moves the cursor past the chunk just requested
        if (uni(iL) < Ltotal) {
            in_ptr += in_chunk_bytes;
            w_ptr += w_chunk_bytes;
            in_bytes_left = in_bytes_left > in_chunk_bytes ? in_bytes_left - in_chunk_bytes : 0u;
            if (--ileft == 0) {
                iL = next_valid(iL + G);
                if (iL < Ltotal) setup_issue();
            }
        }
    };
    auto issue_next = [&](int slot) __attribute__((always_inline)) {      // (prologue: nothing to interleave with)
        const Dma d = dma_begin(slot);
#pragma unroll
        for (int k = 0; k < T::IN_X4 + T::W_INSTR; ++k) dma_piece(d, k);
        advance();
    };
    if (iL < Ltotal) setup_issue();
#pragma unroll
    for (int p = 0; p < T::NBUF - 1; ++p) issue_next(p);

    // ------------------------------------------------------------------------------------------------------------
    // MFMA side
    // ------------------------------------------------------------------------------------------------------------
    const int b_base = k4 * T::PLANE_S + (4 * trow) * T::ROWP + 4 * n16;
    const int a_base = T::IN_FLOATS + (k4 * T::NGRP * T::BN + chalf * 16 + n16) * 4;
    f32x4 acc[T::NPOS];     // frequency position 6*i + j
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < T::NPOS; ++p) acc[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };

    int cL = next_valid(blockIdx.x);
    if (cL >= Ltotal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    Item it = decode_item(a, cL);
    int ch = 0, slot = 0, cseq = 0;
    zero_acc();
    // Interior requests per wave and chunk: waves 0,1: 3 + 2, waves 2,3: 3 + 1, waves 4..7: 2 + 3 (border tiles: more; an
    // item's stores in between only make a counted wait earlier).
    static_assert(T::NBUF == 4 && T::IN_WI == 20 && T::W_WI == 18, "update the counted waits");
    auto read_patch = [&](float (&d)[6][6], int sl) __attribute__((always_inline)) {       // this lane's raw 6x6 patch
        const float *in_s = lds + sl * T::BUF + b_base;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[r][j] = in_s[r * T::ROWP + j];      // (element-wise: merged to 16- and 8-byte reads, see below)
        }
    };
    float t[6][6];          // the current chunk's patch, fetched at the end of the chunk before; B^T d in place
    if (wave < 2 || wave >= 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");       // chunk 0 landed (chunks 1, 2 may be in flight)
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_patch(t, 0);
    while (true) {
        // Chunk c+1 (whose patch is fetched at the end of this chunk) has landed once only the requests of chunk c+2 are
        // outstanding; the barrier makes every wave's part visible and retires everyone's reads of chunk c-1, whose slot the
        // requests of chunk c+3 then take.
        if (wave < 2 || wave >= 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const Dma dma = dma_begin(slot == 0 ? T::NBUF - 1 : slot - 1);
        __builtin_amdgcn_sched_barrier(0);
        // t = B^T d by columns, then V = t B row by row: the MFMAs of a row of positions go out as soon as that row of V
        // exists, and the next row's 13 operations fill their shadow
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]};
            float o[6];
            input_transform6(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = o[i];
        }
        {
            const float *w_s = lds + slot * T::BUF + a_base;
            float4 u[T::NGRP];
#pragma unroll
            for (int g = 0; g < T::NGRP; ++g) {
                // (element-wise: merged to ds_read_b128; NOT through a float4 pointer -- an LDS read the compiler can name waits for
                //  every outstanding LDS-DMA request, vmcnt(0), because it cannot tell that they do not overlap)
                const float *pu = w_s + g * T::BN * 4;
                u[g] = make_float4(pu[0], pu[1], pu[2], pu[3]);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float v[6];
                input_transform6(t[i], v);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int p = 6 * i + j;
                    const float4 &ug = u[p / 4];
                    const float ue = (p % 4) == 0 ? ug.x : (p % 4) == 1 ? ug.y : (p % 4) == 2 ? ug.z : ug.w;
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(ue, v[j], acc[p], 0, 0, 0);
                }
                dma_piece(dma, i);          // one request of chunk c+3 behind every row of MFMAs
            }
        }
        slot = slot + 1 == T::NBUF ? 0 : slot + 1;
        read_patch(t, slot);                // (behind the last row's MFMAs; after the workgroup's last chunk: an empty slot, unused)
        advance();
        ++ch;
        if (ch == nchunks) {
            // ---- item epilogue: Y = A^T M A per lane: channel co = nb*32 + chalf*16 + 4*k4 + j, tile (row trow, column n16) ----
            const int gx = it.x0 + 4 * n16, gy0 = it.y0 + 4 * trow;
            float *__restrict__ yp = a.y + (size_t)it.n * a.y_bs;
            const int act = ACT >= 0 ? ACT : a.act;
            const int co0 = it.nb * T::BN + chalf * 16 + 4 * k4;
            const float *bias_l = lds + T::BIAS_OFF + (cseq & (T::BIAS_SLOTS - 1)) * 64 + chalf * 16 + 4 * k4;
            const float *__restrict__ resp = RES ? a.res + (size_t)it.n * a.res_bs : nullptr;
            const bool vec4 = (a.W % 4 == 0) && ((reinterpret_cast<size_t>(yp) & 15) == 0) && (!RES || (reinterpret_cast<size_t>(resp) & 15) == 0);
            const int Hq = a.H >> 1, Wq = a.W >> 1;       // pooled plane; this lane's block -> rows gy0/2 .., columns gx/2 ..
            float *__restrict__ pq = POOL ? a.pool + (size_t)it.n * a.pool_bs + (size_t)(gy0 >> 1) * Wq + (gx >> 1) : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float tt[4][6];        // A^T M: 4 x 6
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float col[6] = {acc[c][j], acc[6 + c][j], acc[12 + c][j], acc[18 + c][j], acc[24 + c][j], acc[30 + c][j]};
                    float o[4];
                    output_transform6(col, o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tt[r][c] = o[r];
                }
                const float b = bias_l[j];
                if (co0 + j < a.Cout) {
                    const size_t plane = (size_t)(co0 + j) * HW;
                    float *pc = yp + plane;
                    float prev[4];         // (POOL) the activated row above
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float o[4];
                        output_transform6(tt[r], o);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = apply_act(o[c] + b, act);
                        const int gy = gy0 + r;
                        if (POOL) {
                            if (r & 1) {
                                const int qy = (gy0 >> 1) + (r >> 1);
#pragma unroll
                                for (int cc = 0; cc < 2; ++cc) {
                                    const float p0 = prev[2 * cc], p1 = prev[2 * cc + 1], p2 = o[2 * cc], p3 = o[2 * cc + 1];
                                    const float pv = a.pool_max ? fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)) : (p0 + p1 + p2 + p3) * 0.25f;
                                    if (qy < Hq && (gx >> 1) + cc < Wq) pq[(size_t)(co0 + j) * Hq * Wq + (size_t)(r >> 1) * Wq + cc] = pv;
                                }
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c) prev[c] = o[c];
                            }
                        }
                        if (gy < a.H) {
                            if (vec4) {
                                if (gx < a.W) {
                                    float4 q = make_float4(o[0], o[1], o[2], o[3]);
                                    if (RES) {
                                        const float4 rr = *reinterpret_cast<const float4 *>(resp + plane + (size_t)gy * a.W + gx);
                                        q.x += rr.x; q.y += rr.y; q.z += rr.z; q.w += rr.w;
                                    }
                                    *reinterpret_cast<float4 *>(pc + (size_t)gy * a.W + gx) = q;
                                }
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (gx + c < a.W) pc[(size_t)gy * a.W + gx + c] = o[c] + (RES ? resp[plane + (size_t)gy * a.W + gx + c] : 0.0f);
                            }
                        }
                    }
                }
            }
            const int nL = next_valid(cL + G);
            if (nL >= Ltotal) break;
            cL = nL;
            ++cseq;
            it = decode_item(a, cL);
            ch = 0;
            zero_acc();
        }
    }
    // the (empty) look-ahead DMAs must have retired before the workgroup's LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 3x3 OIHW -> F(4x4,3x3) weights U = G g G^T (6x6) as [Cin_pad][9 groups of 4 positions][Cout_pad][4], BatchNorm scale
// folded.  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]; in double, rounded once.
__global__ void conv2d_pack_winograd4_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                             float *__restrict__ out, int Cout, int Cin, int Cin_pad, int Cout_pad) {
    const size_t total = (size_t)Cin_pad * 36 * Cout_pad;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j4 = e & 3, co = (e >> 2) % Cout_pad, g = (e / ((size_t)4 * Cout_pad)) % 9, ci = e / ((size_t)36 * Cout_pad);
        const int p = 4 * g + j4, i = p / 6, j = p % 6;
        double u = 0.0;
        if (co < Cout && ci < Cin) {
            const float *k = w + ((size_t)co * Cin + ci) * 9;
            const double Gm[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                     {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) u += Gm[i][r] * (double)k[r * 3 + c] * Gm[j][c];
            if (scale) u *= (double)scale[co];
        }
        out[e] = (float)u;
    }
}

}  // namespace

void vfi::conv::launch_pack_winograd4(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin, int Cin_pad,
                                      int Cout_pad, hipStream_t s) {
    const long long total = (long long)Cin_pad * 36 * Cout_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv2d_pack_winograd4_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, scale, packed, Cout, Cin, Cin_pad, Cout_pad);
}

// Whether a layer goes to the F(4x4) kernel: plain (the caller checks residual / pooled output), one K split's worth of
// items for every resident workgroup several times over, 32-bit offsets.
bool vfi::conv::winograd4_suits(const ConvArgs &a, int N) {
    using T = Wino4Tile;
    // VFI_CONV_WINOGRAD4: 0 = never (A/B aid: the F(2x2) kernel takes everything), 2 = every plain layer whatever its size
    // (fuzzing small shapes through this kernel); default 1
    static const int mode = getenv("VFI_CONV_WINOGRAD4") ? atoi(getenv("VFI_CONV_WINOGRAD4")) : 1;
    if (!mode) return false;
    if (a.pool && (a.res || a.act != 1)) return false;          // pooled output: plain ReLU layers only (as the F(2x2) kernel fuses it)
    const long long items = (long long)vfi::ceil_div(a.W, T::TW) * vfi::ceil_div(a.H, T::TH) * N * (a.Cout_pad / T::BN);
    if (items >= (1ll << 28)) return false;
    if (mode == 2) return true;
    // Which of the two Winograd kernels is faster for this layer: a cost model of both, in microseconds, fitted to the
    // per-layer A/B of every 3x3 layer of the 1080p frame (profiles/r04_conv_layers.txt: for each of the 67 shapes it picks the
    // measured winner, or a kernel within 5 % of it).  Both kernels are persistent: time = rounds of resident workgroups x
    // time of one work item (chunks of 4 input channels + epilogue).
    //   F(4x4), M = 32 (vfi_conv_winograd4m.hip): 16 x 64 x 32-channel items, one workgroup per CU, 1.45 us per chunk
    //     (2 930 cycles: 72 MFMAs + 72 packed transform operations + requests) + 3.8 us per item (output transform, stores,
    //     item set-up; 5.0 before the straight-line epilogue);
    //   F(2x2) (vfi_conv_winograd.hip): 8 x 32 x 32-channel items, two workgroups per CU, 1.1 us per chunk + 2.7 us per item,
    //     and its K split for few, long items (same rule as launch_winograd).
    static int cus_dev[vfi::kMaxDevices] = {};
    int &cus = cus_dev[vfi::current_device()];
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    const int nchunks = vfi::ceil_div(a.Cin, T::CK), cb = a.Cout_pad / T::BN;
    const double cost4 = (double)((items + cus - 1) / cus) * (1.45 * nchunks + 3.8);
    const long long items2 = (long long)vfi::ceil_div(a.W, 32) * vfi::ceil_div(a.H, 8) * N * cb, resident2 = 2ll * cus;
    const long long out_floats = (long long)N * a.Cout * a.H * a.W;
    auto cost2 = [&](int S) {
        const double rounds = (double)((items2 * S + resident2 - 1) / resident2);
        const double reduce = S > 1 ? 4.0 + (double)(S + 1) * out_floats * 4.0 / 2.5e6 : 0.0;
        return rounds * (1.1 * nchunks / S + 2.7) + reduce;
    };
    double best2 = cost2(1);
    if (a.ws && nchunks >= 16 && items2 < 4 * resident2)
        for (int S = 2; S <= 16; S *= 2)
            if (nchunks / S >= 8 && out_floats * S <= a.ws_floats && cost2(S) < best2) best2 = cost2(S);
    return cost4 < best2;
}

int vfi::conv::launch_winograd4(const ConvArgs &a, int N, hipStream_t s) {
    using T = Wino4Tile;
    static int resident_dev[vfi::kMaxDevices] = {};   // persistent grid: 1 workgroup per CU; per device, idempotent
    int &resident = resident_dev[vfi::current_device()];
    if (!resident) {
        hipError_t e = hipSuccess;
        for (const void *k : {reinterpret_cast<const void *>(conv3x3_winograd4_kernel<0>), reinterpret_cast<const void *>(conv3x3_winograd4_kernel<1>),
                              reinterpret_cast<const void *>(conv3x3_winograd4_kernel<2>), reinterpret_cast<const void *>(conv3x3_winograd4_kernel<3>),
                              reinterpret_cast<const void *>(conv3x3_winograd4_kernel<4>), reinterpret_cast<const void *>(conv3x3_winograd4_kernel<-1, true>),
                              reinterpret_cast<const void *>(conv3x3_winograd4_kernel<1, false, true>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        int dev = 0, cus = 0;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || cus <= 0) return vfi::fail(VFI_ERR_LAUNCH, "vfi_conv2d: Winograd F(4x4) kernel setup: %s", hipGetErrorString(e));
        resident = cus;
    }
    ConvArgs b = a;
    const int cb = a.Cout_pad / T::BN;
    b.tiles_x = vfi::ceil_div(a.W, T::TW);
    b.wino_tiles = b.tiles_x * vfi::ceil_div(a.H, T::TH);
    b.wino_batch = N;
    b.wino_run = 1;
    while (b.wino_run < 8 && (long long)b.wino_tiles * N >= 256ll * 2 * b.wino_run) b.wino_run *= 2;
    b.wino_items = round_up(b.wino_tiles * N, 8 * b.wino_run) * cb;
    b.splits = 1;
    b.fd_items = make_fastdiv((unsigned)b.wino_items);
    b.fd_cb = make_fastdiv((unsigned)cb);
    b.fd_run = make_fastdiv((unsigned)b.wino_run);
    b.fd_tiles = make_fastdiv((unsigned)b.wino_tiles);
    b.fd_tiles_x = make_fastdiv((unsigned)b.tiles_x);
    b.fd_splits = make_fastdiv(1u);
    if (winograd4m_enabled()) return launch_winograd4m(b, s);
    dim3 grid((unsigned)(b.wino_items < resident ? b.wino_items : resident));
    if (b.pool) hipLaunchKernelGGL((conv3x3_winograd4_kernel<1, false, true>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else if (b.res) hipLaunchKernelGGL((conv3x3_winograd4_kernel<-1, true>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else if (b.act == 0) hipLaunchKernelGGL((conv3x3_winograd4_kernel<0>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else if (b.act == 1) hipLaunchKernelGGL((conv3x3_winograd4_kernel<1>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else if (b.act == 2) hipLaunchKernelGGL((conv3x3_winograd4_kernel<2>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else if (b.act == 3) hipLaunchKernelGGL((conv3x3_winograd4_kernel<3>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    else hipLaunchKernelGGL((conv3x3_winograd4_kernel<4>), grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
    return vfi::check_launch("vfi_conv2d");
}
