// Dense 2-D convolution (stride 1, "same" size, zero or reflect padding) on the gfx950 matrix cores,
// exact fp32 (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate; there is no xf32 on gfx950).
//
// Replaces the cuDNN convolutions behind the reference's three networks:
//   PhaseNetBlock            reference src/phase_net/phase_net.py:190-207   (1x1 / 3x3 reflect, BN folded, ELU/tanh)
//   KernelEstimation         reference src/fusion_net/fusion_adacofnet.py:18-107 (3x3 zero pad, ReLU/sigmoid)
//   FusionNet                reference src/fusion_net/fusion_net.py:24-41    (5x5 / 3x3 / 1x1 reflect, ReLU)
//
// Roofline: MFMA fp32 (157.3 TFLOP/s dense).  Direct (implicit-GEMM) formulation per workgroup:
//   D[cout][pixel] += sum_{cin,ky,kx} Wt[cout][(cin,ky,kx)] * X[(cin,ky,kx)][pixel]
//   * M = 32 output channels, N = 32 consecutive pixels of one output row, K = 2 input channels
//     of the same tap per MFMA (lanes 0-31 hold k = 0, lanes 32-63 hold k = 1);
//   * a 256-thread workgroup (4 waves, one per SIMD) owns an 8-row x 32-column output tile for
//     BN = 32*NT output channels; each wave owns 2 rows x BN channels = 2*NT independent 32x32
//     accumulators, so the 64-cycle MFMA issues back to back;
//   * the K loop walks the input channels in chunks of CK; a chunk's input tile (with its KS-1
//     halo, padding resolved while loading) and weight slab are staged global -> registers -> LDS,
//     double buffered, one barrier per chunk; the loads of chunk c+1 are in flight during the
//     MFMAs of chunk c;
//   * every LDS fragment read is one ds_read_b32 whose 32-lane groups touch 32 consecutive dwords
//     (conflict free); tap / channel offsets are compile-time immediates;
//   * accumulator layout puts the pixel on the lane and the channel on the register, so each
//     epilogue store instruction writes two full 128-B row segments of the NCHW output;
//   * bias, activation (ReLU / ELU / tanh / sigmoid) and an optional residual add are fused into
//     the epilogue; batch strides are explicit so inputs / outputs may be channel slices of larger
//     (concatenated) tensors without a copy.
#include "vfi_conv_common.h"

#include <cstdlib>

using namespace vfi::conv;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;


// Padding / channel-tail elements of the input tile are loaded from here instead of being selected to zero
// after the load: a select would make the loaded value "used" right away and force s_waitcnt vmcnt(0)
// BEFORE the chunk's MFMAs, i.e. no overlap of the prefetch with compute.
__device__ float g_zero_word = 0.0f;   // (non-const: stays in the global address space -> global_load, not flat_load)



template <int KS, int CK, int NT, int RW = 2>
struct ConvTile {
    static constexpr int TH = 4 * RW, TW = 32, PADK = (KS - 1) / 2;   // 4 waves x RW rows
    static constexpr int R = TH + KS - 1, PW = TW + KS - 1, PLANE = R * PW, TAPS = KS * KS, BN = 32 * NT;
    static constexpr int IN_ELEMS = CK * PLANE;
    static constexpr int IN_ELEMS_PAD = (IN_ELEMS + 3) / 4 * 4;  // keeps the weight slab 16-B aligned
    static constexpr int W_ELEMS = CK * TAPS * BN;
    static constexpr int IN_PER_THREAD = (IN_ELEMS + 255) / 256;
    static constexpr int W4_PER_THREAD = (W_ELEMS / 4 + 255) / 256;
    static constexpr int W_ALLOC = (W_ELEMS + 255) / 256 * 256;   // LDS-DMA writes whole 1-KiB wave pieces
    static constexpr int BUF = IN_ELEMS_PAD + W_ALLOC;
    static constexpr size_t LDS_BYTES = 2ull * BUF * sizeof(float);
};

// source coordinate of output index g: SRC 1 = align_corners=True (scale*(g)), SRC 2 = align_corners=False
template <int SRC>
__device__ __forceinline__ float src_coord(float scale, int g) {
    return SRC == 1 ? scale * (float)g : fmaxf(scale * ((float)g + 0.5f) - 0.5f, 0.0f);
}

// UPS: the conv input is nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)(x) and is never
// materialised -- the tile loader interpolates it from the low-resolution x (4 loads + lerp per element).
// SRC = 1 (see above); SRC = 2: the first `rsz_channels` input channels are torch's bilinear resize
// (align_corners=False, arbitrary ratio) of a second tensor x2 (N, rsz_channels, Hs, Ws), the remaining channels
// come from x itself -- PhaseNet's `cat(Upsample(feature), phase, amp, Upsample(prediction))` block input
// (src/phase_net/phase_net.py:138-141) without materialising the resized maps.
template <int KS, int CK, int NT, int SRC = 0, int RW = 2>
__global__ __launch_bounds__(256, 2) void conv2d_mfma_kernel(const ConvArgs a) {
    using T = ConvTile<KS, CK, NT, RW>;
    constexpr bool UPS = SRC != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int tile_x = blockIdx.x % a.tiles_x, tile_y = blockIdx.x / a.tiles_x;
    const int nb = blockIdx.y, n = blockIdx.z / a.splits, split = blockIdx.z % a.splits;
    const int x0 = tile_x * T::TW, y0 = tile_y * T::TH;
    const int HW = a.H * a.W;

    // ---- per-thread staging map (identical for every chunk) -------------------------------------
    int in_off[T::IN_PER_THREAD];   // offset inside the chunk's channel block, or -1 (zero fill)
    int in_c[T::IN_PER_THREAD];     // channel inside the chunk (for the Cin tail)
#pragma unroll
    for (int i = 0; i < T::IN_PER_THREAD; ++i) {
        const int e = tid + 256 * i;
        const int c = e / T::PLANE, rem = e % T::PLANE;
        const int r = rem / T::PW, xx = rem % T::PW;
        int gy = y0 - T::PADK + r, gx = x0 - T::PADK + xx;
        bool ok = e < T::IN_ELEMS;
        if (a.pad_mode == 1) {
            gy = reflect_index(gy, a.H);
            gx = reflect_index(gx, a.W);
        } else {
            ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        if (UPS) in_off[i] = ok ? ((gy << 16) | gx) : -1;     // packed output-grid coordinates
        else     in_off[i] = ok ? c * HW + gy * a.W + gx : -1;
        in_c[i] = c;
    }
    const int HWs = UPS ? a.Hs * a.Ws : HW;   // channel stride of x
    // UPS: chunk-invariant top-left source offset (relative to the chunk's first channel) per element, with the
    // +1 column / +1 row steps packed beside the channel index: a chunk's staging then costs ~25 VALU per
    // element instead of ~55 (the loader's VALU work does not overlap the same wave's MFMAs)
    int uo[UPS ? T::IN_PER_THREAD : 1];
    if constexpr (UPS) {
#pragma unroll
        for (int i = 0; i < T::IN_PER_THREAD; ++i) {
            const int pk = max(in_off[i], 0);
            const int gy = pk >> 16, gx = pk & 0xffff;
            const int sy0 = min((int)src_coord<SRC>(a.ups_sy, gy), a.Hs - 1), sx0 = min((int)src_coord<SRC>(a.ups_sx, gx), a.Ws - 1);
            const int dx = sx0 + 1 <= a.Ws - 1 ? 1 : 0, dy = sy0 + 1 <= a.Hs - 1 ? 1 : 0;
            uo[i] = in_c[i] * HWs + sy0 * a.Ws + sx0;
            in_c[i] |= (dx << 8) | (dy << 9);
        }
    }
    const float *xn = a.x + (size_t)n * a.x_bs;
    const float *x2n = SRC == 2 ? a.x2 + (size_t)n * a.x2_bs : xn;   // source of the resized channels
    const float *wn = a.wp + (size_t)nb * T::BN;

    float in_reg[UPS ? 4 : 1][T::IN_PER_THREAD];

    // Input tile: global -> registers (branch-free: every load is issued unconditionally from a valid
    // address and the value selected afterwards, so all loads of a chunk are in flight together and the
    // staging array stays in VGPRs) -> LDS after the chunk's MFMAs.
    auto load_inputs = [&](int ch) {
        const int cbase = ch * CK;
        const bool direct = SRC == 2 && cbase >= a.rsz_channels;          // wave-uniform
        const float *xc = (SRC == 2 ? x2n : xn) + (size_t)ch * CK * HWs;
#pragma unroll
        for (int i = 0; i < T::IN_PER_THREAD; ++i) {
            const bool ok = in_off[i] >= 0 && (cbase + (in_c[i] & 0xff)) < a.Cin;
            if constexpr (UPS) {
                // padding / channel-tail elements read a valid address and are zeroed when the tile is written to
                // LDS; SRC = 2 chunks past the resized prefix read x itself (same 4 loads, zero steps: no branch)
                const int pk = max(in_off[i], 0);
                const int od = (cbase + (in_c[i] & 0xff)) * HW + (pk >> 16) * a.W + (pk & 0xffff);
                const int o0 = ok ? (direct ? od : uo[i]) : 0;
                const int ddx = (ok && !direct) ? ((in_c[i] >> 8) & 1) : 0;
                const int ddy = (ok && !direct && ((in_c[i] >> 9) & 1)) ? a.Ws : 0;
                const float *pb = direct ? xn : xc;
                in_reg[0][i] = pb[o0];
                in_reg[1][i] = pb[o0 + ddx];
                in_reg[2][i] = pb[o0 + ddy];
                in_reg[3][i] = pb[o0 + ddy + ddx];
            } else {
                const float *p = ok ? xc + in_off[i] : &g_zero_word;
                in_reg[0][i] = *p;
            }
        }
    };
    auto store_inputs = [&](float *buf, int ch) {
#pragma unroll
        for (int i = 0; i < T::IN_PER_THREAD; ++i) {
            const int e = tid + 256 * i;
            float v = in_reg[0][i];
            if constexpr (UPS) {   // torch upsample_bilinear2d: h0l*(w0l*v00 + w1l*v01) + h1l*(w0l*v10 + w1l*v11)
                if (!(SRC == 2 && ch * CK >= a.rsz_channels)) {
                    const int pk = max(in_off[i], 0);
                    const float fy = src_coord<SRC>(a.ups_sy, pk >> 16), fx = src_coord<SRC>(a.ups_sx, pk & 0xffff);
                    const float ly = fy - (float)min((int)fy, a.Hs - 1), lx = fx - (float)min((int)fx, a.Ws - 1);
                    v = (1.0f - ly) * ((1.0f - lx) * in_reg[0][i] + lx * in_reg[1][i]) +
                        ly * ((1.0f - lx) * in_reg[2][i] + lx * in_reg[3][i]);
                }
                v = (in_off[i] >= 0 && ch * CK + (in_c[i] & 0xff) < a.Cin) ? v : 0.0f;
            }
            if (T::IN_ELEMS % 256 == 0 || e < T::IN_ELEMS) buf[e] = v;
        }
    };
    // Weight slab: global -> LDS directly (LDS-DMA, 16 B per lane = 1 KiB per wave-instruction, no staging
    // registers, no ds_write pass).  The slab is linear in LDS in exactly the order the lanes are numbered.
    auto load_weights_async = [&](int ch, float *buf) {
        float *wb = buf + T::IN_ELEMS_PAD;
#pragma unroll
        for (int i = 0; i < T::W4_PER_THREAD; ++i) {
            const int f0 = 256 * i + wave * 64;               // wave-uniform first float4 of this piece
            if (f0 < T::W_ELEMS / 4) {
                const int f = min(f0 + lane, T::W_ELEMS / 4 - 1);
                const int row = f / (T::BN / 4), col4 = f % (T::BN / 4);
                const float *g = wn + ((size_t)ch * CK * T::TAPS + row) * a.Cout_pad + col4 * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)(wb + f0 * 4), 16, 0, 0);
            }
        }
    };

    f32x16 acc[NT][RW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nt][rr][q] = 0.0f;

    // this workgroup's range of input-channel chunks (all of them unless the launch is split over K)
    const int nchunks_all = a.Cin_pad / CK;
    const int ch_begin = (int)((long long)nchunks_all * split / a.splits);
    const int nchunks = (int)((long long)nchunks_all * (split + 1) / a.splits);
    load_weights_async(ch_begin, lds);
    load_inputs(ch_begin);
    store_inputs(lds, ch_begin);
    __syncthreads();

    const int b_base = khalf * T::PLANE + (RW * wave) * T::PW + l31;
    const int a_base = khalf * T::TAPS * T::BN + l31;

    for (int ch = ch_begin; ch < nchunks; ++ch) {
        const float *buf = lds + ((ch - ch_begin) & 1) * T::BUF;
        float *nxt = lds + ((ch - ch_begin + 1) & 1) * T::BUF;
        if (ch + 1 < nchunks) {
            load_weights_async(ch + 1, nxt);
            load_inputs(ch + 1);
        }
        const float *in_s = buf + b_base;
        const float *w_s = buf + T::IN_ELEMS_PAD + a_base;
        // K loop over (channel pair, tap), fully unrolled; fragments of step s+1 are read from LDS while the
        // MFMAs of step s issue (explicit two-deep register pipeline).  Pairs beyond the real Cin (tail of
        // the last chunk) are skipped: their weights are zero.
        constexpr int NSTEP = (CK / 2) * T::TAPS;
        const int valid_pairs = min(CK / 2, (a.Cin - ch * CK + 1) / 2);
        float af[2][NT], bf[2][RW];
        auto frag = [&](int sidx, float (&fa)[NT], float (&fb)[RW]) {
            const int c2 = sidx / T::TAPS, tap = sidx % T::TAPS, ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) fb[rr] = in_s[2 * c2 * T::PLANE + (rr + ky) * T::PW + kx];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fa[nt] = w_s[(2 * c2 * T::TAPS + tap) * T::BN + nt * 32];
        };
        frag(0, af[0], bf[0]);
#pragma unroll
        for (int sidx = 0; sidx < NSTEP; ++sidx) {
            if (sidx % T::TAPS == 0 && sidx / T::TAPS >= valid_pairs) break;   // wave-uniform
            if (sidx + 1 < NSTEP) frag(sidx + 1, af[(sidx + 1) & 1], bf[(sidx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int rr = 0; rr < RW; ++rr)
                    acc[nt][rr] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sidx & 1][nt], bf[sidx & 1][rr], acc[nt][rr], 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_inputs(nxt, ch + 1);
        __syncthreads();
    }

    // ---- epilogue: bias + activation (+ residual), 128-B row segments per store ------------------
    // All bias / residual loads of a 32x32 tile are issued together before the first store (a load inside
    // the store loop is waited for individually: 16*NT serialized round trips per lane).
    const int gx = x0 + l31;
    if (a.splits > 1) {   // split-K: raw partial sums; bias / activation / residual are applied by the reduce kernel
        float *__restrict__ wsp = a.ws + ((size_t)split * gridDim.z / a.splits + n) * a.Cout * HW;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int gy = y0 + RW * wave + rr;
            if (gy >= a.H || gx >= a.W) continue;
            const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = nb * T::BN + nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
                    if (co < a.Cout) wsp[(size_t)co * HW + pix] = acc[nt][rr][q];
                }
        }
        return;
    }
    const float *__restrict__ biasp = a.bias;
    const float *__restrict__ resp = a.res ? a.res + (size_t)n * a.res_bs : nullptr;
    float *__restrict__ yp = a.y + (size_t)n * a.y_bs;
    float bv[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = nb * T::BN + nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
            bv[nt][q] = biasp ? biasp[min(co, a.Cout - 1)] : 0.0f;
        }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int gy = y0 + RW * wave + rr;
        if (gy >= a.H || gx >= a.W) continue;
        const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float rv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = nb * T::BN + nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
                rv[q] = resp ? resp[(size_t)min(co, a.Cout - 1) * HW + pix] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = nb * T::BN + nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * khalf;
                if (co < a.Cout) yp[(size_t)co * HW + pix] = apply_act(acc[nt][rr][q] + bv[nt][q], a.act) + rv[q];
            }
        }
    }
}

// y = act(sum_s ws[s] + bias) + residual   (deterministic split-K reduction: fixed summation order)
// grid: x over the plane's pixels (VEC per thread), y = output channel, z = sample.
template <int VEC>
__global__ void conv2d_splitk_reduce_kernel(const float *__restrict__ ws, int splits, const float *__restrict__ bias,
                                            const float *__restrict__ res, long long res_bs, float *__restrict__ y,
                                            long long y_bs, int Cout, int HW, int act) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (p >= HW) return;
    const int co = blockIdx.y, n = blockIdx.z;
    const size_t e = (size_t)co * HW + p, total = (size_t)gridDim.z * Cout * HW;
    const float *w = ws + (size_t)n * Cout * HW + e;
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.0f;
    for (int sidx = 0; sidx < splits; ++sidx, w += total) {
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(w);
            v[0] += t.x; v[1 % VEC] += t.y; v[2 % VEC] += t.z; v[3 % VEC] += t.w;
        } else {
            v[0] += w[0];
        }
    }
    const float b = bias ? bias[co] : 0.0f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = apply_act(v[k] + b, act);
    if (VEC == 4) {
        if (res) {
            const float4 r = *reinterpret_cast<const float4 *>(res + (size_t)n * res_bs + e);
            v[0] += r.x; v[1 % VEC] += r.y; v[2 % VEC] += r.z; v[3 % VEC] += r.w;
        }
        *reinterpret_cast<float4 *>(y + (size_t)n * y_bs + e) = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
    } else {
        if (res) v[0] += res[(size_t)n * res_bs + e];
        y[(size_t)n * y_bs + e] = v[0];
    }
}

// OIHW -> [Cin_pad][KS*KS][Cout_pad] with optional per-output-channel scale (folded BatchNorm).
__global__ void conv2d_pack_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                   float *__restrict__ out, int Cout, int Cin, int taps, int Cin_pad,
                                   int Cout_pad) {
    const size_t total = (size_t)Cin_pad * taps * Cout_pad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = i % Cout_pad;
        const int t = (i / Cout_pad) % taps;
        const int ci = i / ((size_t)Cout_pad * taps);
        float v = 0.0f;
        if (co < Cout && ci < Cin) {
            v = w[((size_t)co * Cin + ci) * taps + t];
            if (scale) v *= scale[co];
        }
        out[i] = v;
    }
}


template <int KS, int CK, int NT, int UPS = 0, int RW = 2>
int launch_conv(const ConvArgs &a, int N, hipStream_t s) {
    using T = ConvTile<KS, CK, NT, RW>;
    static bool attr_done[vfi::kMaxDevices] = {};  // per device, idempotent
    const int dev = vfi::current_device();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv2d_mfma_kernel<KS, CK, NT, UPS, RW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        if (e != hipSuccess) return vfi::fail(VFI_ERR_LAUNCH, "vfi_conv2d: set LDS size: %s", hipGetErrorString(e));
        attr_done[dev] = true;
    }
    const int tiles_y = vfi::ceil_div(a.H, T::TH);
    const long long blocks = (long long)a.tiles_x * tiles_y * (a.Cout_pad / T::BN) * N;
    // Split-K: 2 workgroups are resident per CU (512 slots); a launch of B workgroups takes ceil(B/512) rounds, so
    // few / long workgroups (deep U-Net levels) leave most of a round idle.  Splitting the channel loop S ways
    // makes S*B workgroups of 1/S the length; partial sums go to the caller's workspace and are reduced
    // deterministically.  Pick the S with the fewest full-round equivalents, if it saves >= 15 %.
    ConvArgs b = a;
    b.splits = 1;
    const int nchunks = a.Cin_pad / CK;
    const long long out_floats = (long long)N * a.Cout * a.H * a.W;
    if (a.ws && nchunks >= 8 && blocks < 4 * 512) {
        auto cost = [&](int S) { return (double)((blocks * S + 511) / 512) / S; };
        int best = 1;
        for (int S = 2; S <= 16; S *= 2)
            if (nchunks / S >= 2 && out_floats * S <= a.ws_floats && cost(S) < cost(best) - 1e-9) best = S;
        if (cost(best) <= 0.85 * cost(1)) b.splits = best;
    }
    dim3 grid(a.tiles_x * tiles_y, a.Cout_pad / T::BN, N * b.splits);
    hipLaunchKernelGGL((conv2d_mfma_kernel<KS, CK, NT, UPS, RW>), grid, dim3(256), T::LDS_BYTES, s, b);
    if (b.splits > 1) launch_splitk_reduce(b, N, s);
    return vfi::check_launch("vfi_conv2d");
}

}  // namespace

// ---- 1x1 layers with at most sixteen output channels (PhaseNet's 64 -> 8 prediction maps and 64 -> 1 low-pass map, the
// 64 -> 9 tap maps of the occlusion head, FusionNet's 32 -> 3 tail): 2*Cout flop per input byte, i.e. bound by reading the
// input once.  A streaming kernel: a thread owns VEC consecutive pixels, walks the channel planes with eight VEC*4-byte
// loads in flight, keeps NOUT x VEC sums in registers; the weights of a channel ([c][Cout_pad] in the direct packing:
// consecutive floats) come by scalar loads.  No LDS, no matrix cores (at NOUT = 8, 32 FMAs per 16 loaded bytes is a quarter
// of the vector ALU's rate at the HBM rate).  5.1 TB/s on 3 x 64 -> 8 at 1080p against 3.3 for the matrix-core kernel.
template <int ACT, int NOUT, int VEC>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const float *__restrict__ x, long long x_bs, const float *__restrict__ wp,
                                                             const float *__restrict__ bias, float *__restrict__ y, long long y_bs,
                                                             int Cin, int Cout, int Cout_pad, int HW, int act) {
    typedef float vec __attribute__((ext_vector_type(VEC)));
    const int q = blockIdx.x * 256 + threadIdx.x;          // pixel group
    if (q * VEC >= HW) return;
    const vec *xp = reinterpret_cast<const vec *>(x + (size_t)blockIdx.y * x_bs) + q;
    const size_t plane = (size_t)HW / VEC;
    float acc[NOUT][VEC];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[o][k] = 0.0f;
    int c = 0;
    for (; c + 8 <= Cin; c += 8) {
        vec v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = xp[(size_t)(c + i) * plane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float *w = wp + (size_t)(c + i) * Cout_pad;          // (uniform: scalar loads)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float wo = w[o];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[o][k] = fmaf(wo, v[i][k], acc[o][k]);
            }
        }
    }
    for (; c < Cin; ++c) {
        const vec v = xp[(size_t)c * plane];
        const float *w = wp + (size_t)c * Cout_pad;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float wo = w[o];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[o][k] = fmaf(wo, v[k], acc[o][k]);
        }
    }
    const int a_ = ACT >= 0 ? ACT : act;
    vec *yp = reinterpret_cast<vec *>(y + (size_t)blockIdx.y * y_bs) + q;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        if (o < Cout) {                                              // (uniform)
            const float b = bias ? bias[o] : 0.0f;
            vec r;
#pragma unroll
            for (int k = 0; k < VEC; ++k) r[k] = apply_act(acc[o][k] + b, a_);
            yp[(size_t)o * plane] = r;
        }
    }
}

// vector width of the streaming kernel for a layer: 4, 2 or 0 (not its layer)
static int conv1x1_stream_vec(const ConvArgs &a, int KS) {
    static const bool on = !(getenv("VFI_CONV_STREAM1X1") && atoi(getenv("VFI_CONV_STREAM1X1")) == 0);      // (A/B aid)
    const long long HW = (long long)a.H * a.W;
    if (!on || KS != 1 || a.Cout > 16 || a.res || HW < 4096) return 0;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y) | (uintptr_t)(a.x_bs * 4) | (uintptr_t)(a.y_bs * 4) |
                           (uintptr_t)(HW * 4);
    return (bits & 15u) == 0 ? 4 : (bits & 7u) == 0 ? 2 : 0;
}

template <int NOUT, int VEC>
static int launch_conv1x1_stream(const ConvArgs &a, int N, hipStream_t s) {
    const int HW = a.H * a.W;
    const dim3 grid((unsigned)vfi::ceil_div(HW / VEC, 256), (unsigned)N);
    if (a.act == 3)
        hipLaunchKernelGGL((conv1x1_stream_kernel<3, NOUT, VEC>), grid, dim3(256), 0, s, a.x, a.x_bs, a.wp, a.bias, a.y, a.y_bs, a.Cin, a.Cout, a.Cout_pad, HW, a.act);
    else
        hipLaunchKernelGGL((conv1x1_stream_kernel<-1, NOUT, VEC>), grid, dim3(256), 0, s, a.x, a.x_bs, a.wp, a.bias, a.y, a.y_bs, a.Cin, a.Cout, a.Cout_pad, HW, a.act);
    return vfi::check_launch("vfi_conv2d");
}

// ---- PhaseNet's prediction head of one level in one pass (phase_net.py:149-168, 190-207 + reverse_normalize :80-90): the
// 64 -> 8 1x1 map with tanh (conv1x1_stream_kernel's loop), written out because the next level resizes it, and from the same
// registers the level's outputs: phase = pred[0:4] * pi, amp = (b * amp_in[4:8] + (1 - b) * amp_in[0:4]) * max[n] with
// b = (pred[4:8] + 1) / 2 -- vfi_phasenet_emit's arithmetic to the bit, without reading pred back.
template <int VEC>
__global__ __launch_bounds__(256) void phasenet_predict_kernel(const float *__restrict__ x, long long x_bs, const float *__restrict__ wp,
                                                               const float *__restrict__ bias, int Cout_pad, const float *__restrict__ amp_in,
                                                               long long amp_bs, const float *__restrict__ maxv, float *__restrict__ pred,
                                                               long long pred_bs, float *__restrict__ phase_out, float *__restrict__ amp_out,
                                                               int Cin, int HW) {
    typedef float vec __attribute__((ext_vector_type(VEC)));
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * VEC >= HW) return;
    const int n = blockIdx.y;
    const vec *xp = reinterpret_cast<const vec *>(x + (size_t)n * x_bs) + q;
    const size_t plane = (size_t)HW / VEC;
    float acc[8][VEC];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[o][k] = 0.0f;
    int c = 0;
    for (; c + 8 <= Cin; c += 8) {
        vec v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = xp[(size_t)(c + i) * plane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float *w = wp + (size_t)(c + i) * Cout_pad;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const float wo = w[o];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[o][k] = fmaf(wo, v[i][k], acc[o][k]);
            }
        }
    }
    for (; c < Cin; ++c) {
        const vec v = xp[(size_t)c * plane];
        const float *w = wp + (size_t)c * Cout_pad;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float wo = w[o];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[o][k] = fmaf(wo, v[k], acc[o][k]);
        }
    }
    vec *pp = reinterpret_cast<vec *>(pred + (size_t)n * pred_bs) + q;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const float b = bias ? bias[o] : 0.0f;
        vec r;
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = acc[o][k] = apply_act(acc[o][k] + b, 3);
        pp[(size_t)o * plane] = r;
    }
    const vec *ap = reinterpret_cast<const vec *>(amp_in + (size_t)n * amp_bs) + q;
    vec *pho = reinterpret_cast<vec *>(phase_out + (size_t)n * 4 * HW) + q, *amo = reinterpret_cast<vec *>(amp_out + (size_t)n * 4 * HW) + q;
    const float mx = maxv[n];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const vec a0 = ap[(size_t)b * plane], a1 = ap[(size_t)(4 + b) * plane];
        vec ph, am;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float beta = (acc[4 + b][k] + 1.0f) / 2.0f;
            const float a = fmaf(beta, a1[k], (1.0f - beta) * a0[k]);
            ph[k] = acc[b][k] * 3.14159265358979323846f;
            am[k] = a * mx;
        }
        pho[(size_t)b * plane] = ph;
        amo[(size_t)b * plane] = am;
    }
}

void vfi::conv::launch_splitk_reduce(const ConvArgs &b, int N, hipStream_t s) {
    const int HW = b.H * b.W;
    const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(b.y) & 15u) == 0 && b.y_bs % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(b.ws) & 15u) == 0 &&
                     (!b.res || ((reinterpret_cast<uintptr_t>(b.res) & 15u) == 0 && b.res_bs % 4 == 0));
    if (vec)
        hipLaunchKernelGGL(conv2d_splitk_reduce_kernel<4>, dim3(vfi::ceil_div(HW / 4, 256), b.Cout, N), dim3(256), 0, s, b.ws,
                           b.splits, b.bias, b.res, b.res_bs, b.y, b.y_bs, b.Cout, HW, b.act);
    else
        hipLaunchKernelGGL(conv2d_splitk_reduce_kernel<1>, dim3(vfi::ceil_div(HW, 256), b.Cout, N), dim3(256), 0, s, b.ws,
                           b.splits, b.bias, b.res, b.res_bs, b.y, b.y_bs, b.Cout, HW, b.act);
}

extern "C" long long vfi_conv2d_packed_floats(int Cout, int Cin, int KS) {
    if (Cout <= 0 || Cin <= 0 || KS <= 0) return -1;
    // 3x3: the direct layout [Cin_pad][9][Cout_pad] is followed by the Winograd banks: F(2x2) [Cin_pad][4][Cout_pad][4]
    // and F(4x4) [Cin_pad][9][Cout_pad][4]
    return (long long)round_up(Cin, 8) * (KS * KS + (KS == 3 ? 16 + 36 : 0)) * round_up(Cout, 32);
}

extern "C" int vfi_conv2d_pack(const float *w_oihw, const float *scale, float *packed, int Cout, int Cin,
                               int KS, vfi_stream_t stream) {
    VFI_REQUIRE(w_oihw && packed, VFI_ERR_INVALID_ARG, "vfi_conv2d_pack: null pointer");
    VFI_REQUIRE(Cout > 0 && Cin > 0 && (KS == 1 || KS == 3 || KS == 5), VFI_ERR_INVALID_ARG,
                "vfi_conv2d_pack: bad shape Cout=%d Cin=%d KS=%d", Cout, Cin, KS);
    const int Cin_pad = round_up(Cin, 8), Cout_pad = round_up(Cout, 32);
    const long long total = (long long)Cin_pad * KS * KS * Cout_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3(blocks), dim3(256), 0, vfi::as_stream(stream), w_oihw, scale,
                       packed, Cout, Cin, KS * KS, Cin_pad, Cout_pad);
    if (KS == 3) {
        launch_pack_winograd(w_oihw, scale, packed + total, Cout, Cin, Cin_pad, Cout_pad, vfi::as_stream(stream));
        launch_pack_winograd4(w_oihw, scale, packed + total + (long long)Cin_pad * 16 * Cout_pad, Cout, Cin, Cin_pad, Cout_pad,
                              vfi::as_stream(stream));
    }
    return vfi::check_launch("vfi_conv2d_pack");
}

static int conv2d_impl(const float *x, long long x_bstride, const float *packed_w, const float *bias,
                       const float *residual, long long res_bstride, float *y, long long y_bstride, int N,
                       int Cin, int H, int W, int Cout, int KS, int pad_mode, int act, bool ups, float *workspace,
                       long long workspace_floats, vfi_stream_t stream, const float *x2 = nullptr, long long x2_bstride = 0,
                       int rsz_channels = 0, int Hs = 0, int Ws = 0, float *pooled = nullptr, long long pooled_bstride = 0,
                       int pool_max = 0) {
    VFI_REQUIRE(x && packed_w && y, VFI_ERR_INVALID_ARG, "vfi_conv2d: null pointer");
    VFI_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, VFI_ERR_INVALID_ARG, "vfi_conv2d: non-positive size");
    VFI_REQUIRE(KS == 1 || KS == 3 || KS == 5, VFI_ERR_UNSUPPORTED, "vfi_conv2d: kernel size %d", KS);
    VFI_REQUIRE(pad_mode == 0 || pad_mode == 1, VFI_ERR_INVALID_ARG, "vfi_conv2d: pad_mode %d", pad_mode);
    VFI_REQUIRE(act >= 0 && act <= 4, VFI_ERR_INVALID_ARG, "vfi_conv2d: act %d", act);
    // torch's reflect padding requires pad < size
    VFI_REQUIRE(pad_mode == 0 || ((KS - 1) / 2 < H && (KS - 1) / 2 < W), VFI_ERR_SHAPE,
                "vfi_conv2d: reflect padding %d needs a larger input than %dx%d", (KS - 1) / 2, H, W);
    VFI_REQUIRE((long long)Cin * H * W < (1ll << 31) && (long long)Cout * H * W < (1ll << 31), VFI_ERR_UNSUPPORTED,
                "vfi_conv2d: per-sample tensor too large for 32-bit offsets");
    VFI_REQUIRE(N <= 65535, VFI_ERR_UNSUPPORTED, "vfi_conv2d: batch %d", N);
    ConvArgs a;
    a.x = x; a.wp = packed_w; a.bias = bias; a.res = residual; a.y = y;
    a.x_bs = x_bstride; a.res_bs = res_bstride; a.y_bs = y_bstride;
    a.Cin = Cin; a.Cin_pad = round_up(Cin, 8); a.Cout = Cout; a.Cout_pad = round_up(Cout, 32);
    a.H = H; a.W = W; a.tiles_x = vfi::ceil_div(W, 32);
    a.pad_mode = pad_mode; a.act = act;
    a.ws = workspace_floats > 0 ? workspace : nullptr; a.ws_floats = workspace ? workspace_floats : 0; a.splits = 1;
    a.x2 = nullptr; a.x2_bs = 0; a.rsz_channels = 0; a.wino_tiles = 0; a.wino_items = 0; a.wino_batch = 0; a.wino_run = 1;
    a.Hs = H / 2; a.Ws = W / 2;
    a.pool = nullptr; a.pool_bs = 0; a.pool_max = pool_max;
    a.ups_sy = H > 1 ? (float)(a.Hs - 1) / (float)(H - 1) : 0.0f;
    a.ups_sx = W > 1 ? (float)(a.Ws - 1) / (float)(W - 1) : 0.0f;
    hipStream_t s = vfi::as_stream(stream);
    const bool wide = (a.Cout_pad % 64 == 0);
    if (ups) {
        VFI_REQUIRE(KS == 3 && pad_mode == 0 && H % 2 == 0 && W % 2 == 0 && H < 65536 && W < 65536, VFI_ERR_UNSUPPORTED,
                    "vfi_conv2d_upsample2x: needs KS=3, zero padding, even output size (got KS=%d pad=%d %dx%d)", KS, pad_mode, H, W);
        return wide ? launch_conv<3, 8, 2, 1>(a, N, s) : launch_conv<3, 8, 1, 1>(a, N, s);
    }
    if (x2) {
        VFI_REQUIRE(KS == 3 && wide && rsz_channels > 0 && rsz_channels % 8 == 0 && rsz_channels <= Cin && Hs > 0 && Ws > 0 &&
                    H < 65536 && W < 65536, VFI_ERR_UNSUPPORTED,
                    "vfi_conv2d_resized_prefix: needs KS=3, Cout%%64==0, prefix channels a multiple of 8 (got KS=%d Cout=%d prefix=%d)",
                    KS, Cout, rsz_channels);
        a.x2 = x2; a.x2_bs = x2_bstride; a.rsz_channels = rsz_channels; a.Hs = Hs; a.Ws = Ws;
        a.ups_sy = (float)Hs / (float)H; a.ups_sx = (float)Ws / (float)W;
        return launch_conv<3, 8, 2, 2>(a, N, s);
    }
    // Measured on MI355X and NOT adopted (kept out of the build): 16-row tiles (4 rows per wave) -3..5 %;
    // CK=4 with 3 workgroups per CU +-5 % by shape; staggering the two co-resident workgroups by half a
    // workgroup: no change; a persistent tile loop (512 resident workgroups walking the tiles, next tile's first
    // chunk requested before the epilogue): -8 % (per-tile setup + spills outweigh the saved dispatch gaps).
    // Pseudo-random within-chunk start offsets for the first-round workgroups (de-phasing chunk boundaries): 0 %.
    // MFMA pipe utilisation of this structure is 70-76 % (PMC) at ~2.18 GHz.
    // 3x3: Winograd F(2x2,3x3) unless VFI_CONV_WINOGRAD=0 (tuning / A-B aid: the direct kernels stay built), the
    // sample's input exceeds the 32-bit byte range of a buffer descriptor, or the work-item count a 32-bit index
    static const bool wino_on = !(getenv("VFI_CONV_WINOGRAD") && atoi(getenv("VFI_CONV_WINOGRAD")) == 0);
    const long long wino_work_items = (long long)a.tiles_x * vfi::ceil_div(H, 8) * N * (a.Cout_pad / 32) * 16;   // (x max. K split)
    if (KS == 3 && wino_on && (long long)Cin * H * W * 4 < (1ll << 32) && wino_work_items < (1ll << 30)) {
        a.pool = pooled; a.pool_bs = pooled_bstride;
        if (winograd4_suits(a, N)) {        // plain, large layers: the F(4x4) tile
            a.wp = packed_w + (size_t)a.Cin_pad * (9 + 16) * a.Cout_pad;
            return launch_winograd4(a, N, s);
        }
        a.wp = packed_w + (size_t)a.Cin_pad * 9 * a.Cout_pad;
        return launch_winograd(a, N, s);
    }
    if (pooled) {       // not a Winograd layer: the convolution, then the pooling pass
        const int rc = conv2d_impl(x, x_bstride, packed_w, bias, residual, res_bstride, y, y_bstride, N, Cin, H, W, Cout, KS, pad_mode,
                                   act, false, workspace, workspace_floats, stream);
        return rc ? rc : vfi_pool2(y, y_bstride, pooled, pooled_bstride, N, Cout, H, W, pool_max, stream);
    }
    if (const int vec = conv1x1_stream_vec(a, KS)) {
        if (a.Cout <= 8) return vec == 4 ? launch_conv1x1_stream<8, 4>(a, N, s) : launch_conv1x1_stream<8, 2>(a, N, s);
        return vec == 4 ? launch_conv1x1_stream<16, 4>(a, N, s) : launch_conv1x1_stream<16, 2>(a, N, s);
    }
    if (KS == 3) return wide ? launch_conv<3, 8, 2>(a, N, s) : launch_conv<3, 8, 1>(a, N, s);
    if (KS == 5) return wide ? launch_conv<5, 4, 2>(a, N, s) : launch_conv<5, 4, 1>(a, N, s);
    return wide ? launch_conv<1, 8, 2>(a, N, s) : launch_conv<1, 8, 1>(a, N, s);
}

// Which kernel vfi_conv2d / vfi_conv2d_pool2 would run for a layer (labels and flop counts of profiling tools: the one
// place the selection lives is here, next to the dispatch above).
extern "C" int vfi_conv2d_algo(int N, int Cin, int H, int W, int Cout, int KS, int has_residual, int pooled, int act) {
    if (N < 1 || Cin < 1 || H < 1 || W < 1 || Cout < 1) return VFI_ERR_INVALID_ARG;
    if (KS != 1 && KS != 3 && KS != 5) return VFI_ERR_UNSUPPORTED;      // what vfi_conv2d answers for the same layer
    if (KS == 1) {       // (16-byte-aligned dense operands assumed: the query has no pointers)
        ConvArgs q{};
        q.Cout = Cout; q.H = H; q.W = W;
        static float dummy1;
        q.res = has_residual ? &dummy1 : nullptr;
        return conv1x1_stream_vec(q, KS) ? VFI_CONV_ALGO_STREAM1X1 : VFI_CONV_ALGO_DIRECT;
    }
    static const bool wino_on = !(getenv("VFI_CONV_WINOGRAD") && atoi(getenv("VFI_CONV_WINOGRAD")) == 0);
    ConvArgs a{};
    a.Cin = Cin; a.Cout = Cout; a.Cout_pad = round_up(Cout, 32); a.H = H; a.W = W; a.act = act; a.tiles_x = vfi::ceil_div(W, 32);
    static float dummy;
    a.res = has_residual ? &dummy : nullptr;
    a.pool = pooled ? &dummy : nullptr;
    const long long wino_work_items = (long long)a.tiles_x * vfi::ceil_div(H, 8) * N * (a.Cout_pad / 32) * 16;
    if (KS == 3 && wino_on && (long long)Cin * H * W * 4 < (1ll << 32) && wino_work_items < (1ll << 30))
        return winograd4_suits(a, N) ? VFI_CONV_ALGO_WINOGRAD4 : VFI_CONV_ALGO_WINOGRAD2;
    return VFI_CONV_ALGO_DIRECT;
}

extern "C" int vfi_conv2d(const float *x, long long x_bstride, const float *packed_w, const float *bias,
                          const float *residual, long long res_bstride, float *y, long long y_bstride, int N,
                          int Cin, int H, int W, int Cout, int KS, int pad_mode, int act, float *workspace,
                          long long workspace_floats, vfi_stream_t stream) {
    return conv2d_impl(x, x_bstride, packed_w, bias, residual, res_bstride, y, y_bstride, N, Cin, H, W, Cout, KS,
                       pad_mode, act, false, workspace, workspace_floats, stream);
}

extern "C" int vfi_phasenet_predict(const float *feat, long long feat_bstride, const float *packed_w, const float *bias,
                                    const float *amp_in, long long amp_bstride, const float *max_amp, float *pred, long long pred_bstride,
                                    float *phase_out, float *amp_out, int N, int Cin, int H, int W, vfi_stream_t stream) {
    VFI_REQUIRE(feat && packed_w && amp_in && max_amp && pred && phase_out && amp_out, VFI_ERR_INVALID_ARG, "vfi_phasenet_predict: null pointer");
    VFI_REQUIRE(N > 0 && N <= 65535 && Cin > 0 && H > 0 && W > 0, VFI_ERR_INVALID_ARG, "vfi_phasenet_predict: bad sizes");
    const long long HW = (long long)H * W;
    VFI_REQUIRE((long long)Cin * HW < (1ll << 31), VFI_ERR_UNSUPPORTED, "vfi_phasenet_predict: per-sample tensor too large for 32-bit offsets");
    ConvArgs a{};
    a.x = feat; a.y = pred; a.x_bs = feat_bstride; a.y_bs = pred_bstride; a.Cout = 8; a.H = H; a.W = W;
    int vec = conv1x1_stream_vec(a, 1);
    const uintptr_t bits = reinterpret_cast<uintptr_t>(amp_in) | reinterpret_cast<uintptr_t>(phase_out) | reinterpret_cast<uintptr_t>(amp_out) | (uintptr_t)(amp_bstride * 4);
    if (vec == 4 && (bits & 15u)) vec = (bits & 7u) ? 0 : 2;
    else if (vec == 2 && (bits & 7u)) vec = 0;
    if (!vec) {        // small or odd levels: the two launches this entry point replaces
        const int rc = conv2d_impl(feat, feat_bstride, packed_w, bias, nullptr, 0, pred, pred_bstride, N, Cin, H, W, 8, 1, 0, 3, false, nullptr, 0, stream);
        return rc ? rc : vfi_phasenet_emit(pred, pred_bstride, amp_in, amp_bstride, max_amp, phase_out, amp_out, N, (int)HW, stream);
    }
    const dim3 grid((unsigned)vfi::ceil_div((int)(HW / vec), 256), (unsigned)N);
    hipStream_t s = vfi::as_stream(stream);
    if (vec == 4)
        hipLaunchKernelGGL(phasenet_predict_kernel<4>, grid, dim3(256), 0, s, feat, feat_bstride, packed_w, bias, 32, amp_in, amp_bstride, max_amp, pred,
                           pred_bstride, phase_out, amp_out, Cin, (int)HW);
    else
        hipLaunchKernelGGL(phasenet_predict_kernel<2>, grid, dim3(256), 0, s, feat, feat_bstride, packed_w, bias, 32, amp_in, amp_bstride, max_amp, pred,
                           pred_bstride, phase_out, amp_out, Cin, (int)HW);
    return vfi::check_launch("vfi_phasenet_predict");
}

extern "C" int vfi_conv2d_pool2(const float *x, long long x_bstride, const float *packed_w, const float *bias, float *y,
                                long long y_bstride, float *pooled, long long pooled_bstride, int is_max, int N, int Cin, int H, int W,
                                int Cout, int KS, int pad_mode, int act, float *workspace, long long workspace_floats,
                                vfi_stream_t stream) {
    VFI_REQUIRE(pooled, VFI_ERR_INVALID_ARG, "vfi_conv2d_pool2: null pooled output");
    VFI_REQUIRE(H >= 2 && W >= 2, VFI_ERR_INVALID_ARG, "vfi_conv2d_pool2: %dx%d cannot be pooled", H, W);
    return conv2d_impl(x, x_bstride, packed_w, bias, nullptr, 0, y, y_bstride, N, Cin, H, W, Cout, KS, pad_mode, act, false,
                       workspace, workspace_floats, stream, nullptr, 0, 0, 0, 0, pooled, pooled_bstride, is_max ? 1 : 0);
}

extern "C" int vfi_conv2d_upsample2x(const float *x_lowres, long long x_bstride, const float *packed_w, const float *bias,
                                     const float *residual, long long res_bstride, float *y, long long y_bstride, int N,
                                     int Cin, int H, int W, int Cout, int KS, int pad_mode, int act, float *workspace,
                                     long long workspace_floats, vfi_stream_t stream) {
    return conv2d_impl(x_lowres, x_bstride, packed_w, bias, residual, res_bstride, y, y_bstride, N, Cin, H, W, Cout, KS,
                       pad_mode, act, true, workspace, workspace_floats, stream);
}

extern "C" int vfi_conv2d_resized_prefix(const float *x, long long x_bstride, const float *x2, long long x2_bstride,
                                         int prefix_channels, int Hs, int Ws, const float *packed_w, const float *bias,
                                         float *y, long long y_bstride, int N, int Cin, int H, int W, int Cout, int KS,
                                         int pad_mode, int act, float *workspace, long long workspace_floats,
                                         vfi_stream_t stream) {
    VFI_REQUIRE(x2, VFI_ERR_INVALID_ARG, "vfi_conv2d_resized_prefix: null x2");
    return conv2d_impl(x, x_bstride, packed_w, bias, nullptr, 0, y, y_bstride, N, Cin, H, W, Cout, KS, pad_mode, act, false,
                       workspace, workspace_floats, stream, x2, x2_bstride, prefix_channels, Hs, Ws);
}
