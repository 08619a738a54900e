// AdaCoF deformable sampling for gfx950.
//
//   vfi_adacof_forward : FunctionAdaCoF.forward (reference src/adacof/cupy_module/adacof.py:313-361,
//                        kernel text :6-65)
//   vfi_adacof_fused   : both sampling sides + occlusion blend + flow-variance mask of
//                        AdaCoFNet.forward (reference src/fusion_net/fusion_adacofnet.py:195-213)
//
// Roofline: HBM.  Per output pixel the kernel streams F*F (w, alpha, beta) triples per side
// (600 B/px for F=5, two sides) and writes 12..40 B; the frame gathers are served by L1/L2
// (a wave touches a few rows of the frame).  Design:
//   * one thread owns VEC=4 consecutive pixels, so every (w, alpha, beta) plane is read with
//     16 B/lane (1 KiB per wave-instruction, fully coalesced), ONCE for all colour channels
//     (the reference launches one thread per channel and re-reads the triple 3x);
//   * a 64x4 thread block covers a 256x4 pixel tile; blocks are walked row-major so that the
//     gathers of neighbouring blocks share L2 lines;
//   * the replication pad of the reference is folded into the tap clamp (fused entry point), so
//     no padded copy of the frame is ever materialised;
//   * the flow-variance statistics reuse the same (w, alpha, beta) registers (one pass, pivoted
//     second moments), so the mask costs no extra HBM traffic;
//   * (rgbx entry point) frames are pixel-interleaved (16-B gathers) and the workgroup's frame neighbourhood
//     (tile + tap reach + a 4-pixel offset margin) is staged in LDS, so almost every bilinear corner is an LDS read;
//     measured at 1088x1920 (offsets ~N(0,2) clipped to +-8): planar gathers 1.29 ms -> rgbx 0.94 ms -> LDS window 0.56 ms.
#include "vfi_common.h"

#include <cstdlib>

namespace {

using vfi::ceil_div;

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float *p) { v[0] = *p; }
};
template <>
struct Vec<4> {
    float v[4];
    __device__ __forceinline__ void load(const float *p) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

template <>
struct Vec<2> {
    float v[2];
    __device__ __forceinline__ void load(const float *p) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
};

template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
        p[0] = v[0];
    }
}

// One tap of adacof.py:24-60 for C channels.  `row`/`col` already contain i + k*dilation (minus the
// folded replication pad in the fused variant).  (int) is truncation toward zero; each tap index
// is clamped on its own; the bilinear weights use the un-clamped fractions (may extrapolate).
template <int C>
__device__ __forceinline__ void tap_accumulate(const float *__restrict__ in, size_t plane, int Hin,
                                               int Win, int row, int col, float w, float alpha,
                                               float beta, float (&acc)[C]) {
    const int A = (int)alpha;
    const int B = (int)beta;
    const float fa = alpha - (float)A;
    const float fb = beta - (float)B;
    const int i0 = min(max(row + A, 0), Hin - 1);
    const int i1 = min(max(row + A + 1, 0), Hin - 1);
    const int j0 = min(max(col + B, 0), Win - 1);
    const int j1 = min(max(col + B + 1, 0), Win - 1);
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
    const int o00 = i0 * Win + j0, o10 = i1 * Win + j0, o01 = i0 * Win + j1, o11 = i1 * Win + j1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float *p = in + (size_t)c * plane;
        const float v = p[o00] * w00 + p[o10] * w10 + p[o01] * w01 + p[o11] * w11;
        acc[c] += w * v;
    }
}

// Same tap on a pixel-interleaved frame (H, W, 4 floats: r, g, b, unused): one 16-B gather per corner
// instead of one 4-B gather per corner and channel (the gathers, not HBM, bound this kernel).
__device__ __forceinline__ void tap_accumulate_rgbx(const float4 *__restrict__ in, int Hin, int Win, int row,
                                                    int col, float w, float alpha, float beta, float (&acc)[3]) {
    const int A = (int)alpha;
    const int B = (int)beta;
    const float fa = alpha - (float)A;
    const float fb = beta - (float)B;
    const int i0 = min(max(row + A, 0), Hin - 1);
    const int i1 = min(max(row + A + 1, 0), Hin - 1);
    const int j0 = min(max(col + B, 0), Win - 1);
    const int j1 = min(max(col + B + 1, 0), Win - 1);
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
    const float4 v00 = in[i0 * Win + j0], v10 = in[i1 * Win + j0], v01 = in[i0 * Win + j1], v11 = in[i1 * Win + j1];
    acc[0] += w * (v00.x * w00 + v10.x * w10 + v01.x * w01 + v11.x * w11);
    acc[1] += w * (v00.y * w00 + v10.y * w10 + v01.y * w01 + v11.y * w11);
    acc[2] += w * (v00.z * w00 + v10.z * w10 + v01.z * w01 + v11.z * w11);
}

// Same tap with the frame neighbourhood of the workgroup staged in LDS: `win` holds frame[clamp(yb+u)][clamp(xb+v)]
// for u < WH, v < WW, so a tap whose 2x2 footprint lies inside the window is four 16-B LDS reads (the border clamp
// is already baked into the staged pixels); the rare tap that leaves the window falls back to global gathers.
__device__ __forceinline__ void tap_accumulate_win(const float4 *__restrict__ win, int WH, int WW, int yb, int xb,
                                                   const float4 *__restrict__ in, int Hin, int Win, int row, int col,
                                                   float w, float alpha, float beta, float (&acc)[3]) {
    const int A = (int)alpha;
    const int B = (int)beta;
    const float fa = alpha - (float)A;
    const float fb = beta - (float)B;
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
    const int u0 = row + A - yb, v0 = col + B - xb;
    float4 v00, v10, v01, v11;
    if ((unsigned)u0 < (unsigned)(WH - 1) && (unsigned)v0 < (unsigned)(WW - 1)) {
        const float4 *p = win + u0 * WW + v0;
        v00 = p[0]; v01 = p[1]; v10 = p[WW]; v11 = p[WW + 1];
    } else {
        const int i0 = min(max(row + A, 0), Hin - 1), i1 = min(max(row + A + 1, 0), Hin - 1);
        const int j0 = min(max(col + B, 0), Win - 1), j1 = min(max(col + B + 1, 0), Win - 1);
        v00 = in[i0 * Win + j0]; v10 = in[i1 * Win + j0]; v01 = in[i0 * Win + j1]; v11 = in[i1 * Win + j1];
    }
    acc[0] += w * (v00.x * w00 + v10.x * w10 + v01.x * w01 + v11.x * w11);
    acc[1] += w * (v00.y * w00 + v10.y * w10 + v01.y * w01 + v11.y * w11);
    acc[2] += w * (v00.z * w00 + v10.z * w10 + v01.z * w01 + v11.z * w11);
}

// ---- plain forward ------------------------------------------------------------------------
template <int C, int VEC>
__global__ __launch_bounds__(256) void adacof_forward_kernel(
    const float *__restrict__ input, const float *__restrict__ weight,
    const float *__restrict__ offset_i, const float *__restrict__ offset_j,
    float *__restrict__ output, int Ctot, int Hin, int Win, int H, int W, int F, int dil) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * VEC;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z / ceil_div(Ctot, C);
    const int c0 = (blockIdx.z % ceil_div(Ctot, C)) * C;
    if (x0 >= W || y >= H) return;
    const size_t plane = (size_t)H * W;
    const size_t in_plane = (size_t)Hin * Win;
    const float *in = input + ((size_t)n * Ctot + c0) * in_plane;
    const size_t pix = (size_t)y * W + x0;
    const size_t tbase = (size_t)n * F * F * plane + pix;

    float acc[VEC][C];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[v][c] = 0.0f;

    for (int k = 0; k < F; ++k)
        for (int l = 0; l < F; ++l) {
            const size_t t = tbase + (size_t)(k * F + l) * plane;
            Vec<VEC> w, a, b;
            w.load(weight + t);
            a.load(offset_i + t);
            b.load(offset_j + t);
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                tap_accumulate<C>(in, in_plane, Hin, Win, y + k * dil, x0 + v + l * dil, w.v[v],
                                  a.v[v], b.v[v], acc[v]);
        }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (c0 + c < Ctot) {
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = acc[v][c];
            store_vec<VEC>(output + ((size_t)n * Ctot + c0 + c) * plane + pix, o);
        }
    }
}

// ---- fused: two sides + occlusion blend + flow-variance mask ---------------------------------
struct FlowStats {  // pivoted weighted moments of one offset plane (alpha or beta)
    float m, q;
};

// SOFTMAX: w1/w2 hold the LOGITS of Subnet_weight (fusion_adacofnet.py:46-57); the softmax over the F*F taps is
// folded into the accumulation (online max rescaling), so the normalised weights are never written to HBM.
// WIN (rgbx, VEC = 1): the 64x4-pixel workgroup first stages, per side, the (4 + (F-1)d + 2M + 1) x (64 + (F-1)d + 2M + 1)
// pixel neighbourhood of its tile (M = margin for the offsets) into LDS with coalesced 16-B row loads.
template <int C, int VEC, int FT, int UNROLL_L, int MIN_WAVES, bool RGBX = false, bool SOFTMAX = false, bool WIN = false>
__global__ __launch_bounds__(256, MIN_WAVES) void adacof_fused_kernel(
    const float *__restrict__ frame0, const float *__restrict__ frame2,
    const float *__restrict__ w1, const float *__restrict__ a1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ a2, const float *__restrict__ b2,
    const float *__restrict__ occ, float *__restrict__ out_t1, float *__restrict__ out_t2,
    float *__restrict__ out_frame, float *__restrict__ out_mask, int H, int W, int Frt, int dil, int margin) {
    const int F = FT > 0 ? FT : Frt;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * VEC;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z;
    extern __shared__ float4 win_lds[];
    const int WH = 4 + (F - 1) * dil + 2 * margin + 1, WW = 64 + (F - 1) * dil + 2 * margin + 1;
    const int yb = blockIdx.y * 4 - ((F - 1) * dil) / 2 - margin, xb = blockIdx.x * 64 - ((F - 1) * dil) / 2 - margin;
    if constexpr (WIN) {
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const float4 *g0 = reinterpret_cast<const float4 *>(frame0) + (size_t)n * H * W;
        const float4 *g2 = reinterpret_cast<const float4 *>(frame2) + (size_t)n * H * W;
        for (int e = tid; e < WH * WW; e += 256) {
            const int gy = min(max(yb + e / WW, 0), H - 1), gx = min(max(xb + e % WW, 0), W - 1);
            win_lds[e] = g0[gy * W + gx];
            win_lds[WH * WW + e] = g2[gy * W + gx];
        }
        __syncthreads();
    }
    if (x0 >= W || y >= H) return;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)y * W + x0;
    const size_t tbase = (size_t)n * F * F * plane + pix;
    const int pad = ((F - 1) * dil) / 2;  // ReplicationPad2d(kernel_pad) folded into the clamp

    float res[2][VEC][C];
    float var[2][VEC];
    // per-side running state
    float acc[VEC][C], s[VEC], pa[VEC], pb[VEC], mx[VEC];
    FlowStats sa[VEC], sb[VEC];
    auto side_begin = [&](const Vec<VEC> &a0, const Vec<VEC> &b0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
#pragma unroll
            for (int c = 0; c < C; ++c) acc[v][c] = 0.0f;
            s[v] = 0.0f;
            mx[v] = -INFINITY;
            sa[v] = {0.0f, 0.0f};
            sb[v] = {0.0f, 0.0f};
            pa[v] = a0.v[v];      // pivot = offsets of tap 0 (keeps the second moments well conditioned)
            pb[v] = b0.v[v];
        }
    };
    auto tap = [&](int side, const float *__restrict__ in, int k, int l, Vec<VEC> w, const Vec<VEC> &a, const Vec<VEC> &b) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if constexpr (SOFTMAX) {   // w = exp(logit - running max)
                if constexpr (FT > 0) {
                    w.v[v] = expf(w.v[v] - mx[v]);      // (the row's maximum is already folded into mx: raise_max)
                } else {                                // rescale what was accumulated so far
                    const float mnew = fmaxf(mx[v], w.v[v]);
                    const float sc = expf(mx[v] - mnew);
                    w.v[v] = expf(w.v[v] - mnew);
                    mx[v] = mnew;
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[v][c] *= sc;
                    s[v] *= sc;
                    sa[v].m *= sc; sa[v].q *= sc; sb[v].m *= sc; sb[v].q *= sc;
                }
            }
            if constexpr (WIN)
                tap_accumulate_win(win_lds + side * WH * WW, WH, WW, yb, xb, reinterpret_cast<const float4 *>(in), H, W,
                                   y + k * dil - pad, x0 + v + l * dil - pad, w.v[v], a.v[v], b.v[v], acc[v]);
            else if constexpr (RGBX)
                tap_accumulate_rgbx(reinterpret_cast<const float4 *>(in), H, W, y + k * dil - pad,
                                    x0 + v + l * dil - pad, w.v[v], a.v[v], b.v[v], acc[v]);
            else
                tap_accumulate<C>(in, plane, H, W, y + k * dil - pad, x0 + v + l * dil - pad,
                                  w.v[v], a.v[v], b.v[v], acc[v]);
            const float da = a.v[v] - pa[v], db = b.v[v] - pb[v];
            s[v] += w.v[v];
            sa[v].m += w.v[v] * da;
            sa[v].q += w.v[v] * da * da;
            sb[v].m += w.v[v] * db;
            sb[v].q += w.v[v] * db * db;
        }
    };
    // online softmax, one step per ROW of taps: the running maximum takes in the row's F logits at once and what was
    // accumulated so far is rescaled once (F + 1 exponentials per row instead of 2 F)
    auto raise_max = [&](const float (&row_max)[VEC]) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float mnew = fmaxf(mx[v], row_max[v]);
            const float sc = expf(mx[v] - mnew);
            mx[v] = mnew;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[v][c] *= sc;
            s[v] *= sc;
            sa[v].m *= sc; sa[v].q *= sc; sb[v].m *= sc; sb[v].q *= sc;
        }
    };
    // Var = sum_k W (Mean - x)^2 with Mean = sum_k W x   (fusion_adacofnet.py:204-208),
    // evaluated from moments pivoted at x_p: x = x' + x_p, c = x_p (S - 1):
    //   Var = M'^2 (S - 2) + Q' + 2 c M' (S - 1) + c^2 S
    auto side_end = [&](float (&r)[VEC][C], float (&vr)[VEC]) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if constexpr (SOFTMAX) {   // normalise: every accumulated sum is linear in the weights
                const float inv = 1.0f / s[v];
#pragma unroll
                for (int c = 0; c < C; ++c) acc[v][c] *= inv;
                sa[v].m *= inv; sa[v].q *= inv; sb[v].m *= inv; sb[v].q *= inv;
                s[v] = 1.0f;
            }
            const float S = s[v];
            const float ca = pa[v] * (S - 1.0f), cb = pb[v] * (S - 1.0f);
            const float va = sa[v].m * sa[v].m * (S - 2.0f) + sa[v].q + 2.0f * ca * sa[v].m * (S - 1.0f) + ca * ca * S;
            const float vb = sb[v].m * sb[v].m * (S - 2.0f) + sb[v].q + 2.0f * cb * sb[v].m * (S - 1.0f) + cb * cb * S;
            vr[v] = va + vb;
#pragma unroll
            for (int c = 0; c < C; ++c) r[v][c] = acc[v][c];
        }
    };
    if constexpr (FT > 0) {
        // The 2 F rows of taps (side 0, then side 2) as ONE pipeline: the F (w, alpha, beta) triples of the next row are
        // requested before the current row's gathers and arithmetic start, so a wave always has loads in flight.
        auto load_row = [&](int r, Vec<VEC> (&w)[FT], Vec<VEC> (&a)[FT], Vec<VEC> (&b)[FT]) {
            const int side = r >= FT ? 1 : 0, k = r - side * FT;
            const float *__restrict__ wp = side ? w2 : w1;
            const float *__restrict__ ap = side ? a2 : a1;
            const float *__restrict__ bp = side ? b2 : b1;
#pragma unroll
            for (int l = 0; l < FT; ++l) {
                const size_t t = tbase + (size_t)(k * FT + l) * plane;
                w[l].load(wp + t);
                a[l].load(ap + t);
                b[l].load(bp + t);
            }
        };
        Vec<VEC> cw[FT], ca[FT], cb[FT];
        load_row(0, cw, ca, cb);
#pragma unroll 1
        for (int r = 0; r < 2 * FT; ++r) {
            Vec<VEC> nw[FT], na[FT], nb[FT];
            load_row(r + 1 < 2 * FT ? r + 1 : r, nw, na, nb);      // (the last iteration re-requests its own row: no branch)
            const int side = r >= FT ? 1 : 0, k = r - side * FT;
            const float *__restrict__ in = (side ? frame2 : frame0) + (size_t)n * (RGBX ? 4 : C) * plane;
            if (k == 0) side_begin(ca[0], cb[0]);
            if constexpr (SOFTMAX) {
                float row_max[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    row_max[v] = cw[0].v[v];
#pragma unroll
                    for (int l = 1; l < FT; ++l) row_max[v] = fmaxf(row_max[v], cw[l].v[v]);
                }
                raise_max(row_max);
            }
#pragma unroll
            for (int l = 0; l < FT; ++l) tap(side, in, k, l, cw[l], ca[l], cb[l]);
            if (k == FT - 1) {
                if (side == 0) side_end(res[0], var[0]);
                else side_end(res[1], var[1]);
            }
#pragma unroll
            for (int l = 0; l < FT; ++l) { cw[l] = nw[l]; ca[l] = na[l]; cb[l] = nb[l]; }
        }
    } else {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const float *__restrict__ in = (side ? frame2 : frame0) + (size_t)n * (RGBX ? 4 : C) * plane;
            const float *__restrict__ wp = side ? w2 : w1;
            const float *__restrict__ ap = side ? a2 : a1;
            const float *__restrict__ bp = side ? b2 : b1;
            {
                Vec<VEC> a, b;
                a.load(ap + tbase);
                b.load(bp + tbase);
                side_begin(a, b);
            }
#pragma unroll 1
            for (int k = 0; k < F; ++k)
#pragma unroll UNROLL_L
                for (int l = 0; l < F; ++l) {
                    const size_t t = tbase + (size_t)(k * F + l) * plane;
                    Vec<VEC> w, a, b;
                    w.load(wp + t);
                    a.load(ap + t);
                    b.load(bp + t);
                    tap(side, in, k, l, w, a, b);
                }
            side_end(res[side], var[side]);
        }
    }

    Vec<VEC> o;
    o.load(occ + (size_t)n * plane + pix);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t q = ((size_t)n * C + c) * plane + pix;
        float f[VEC], t1[VEC], t2[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            t1[v] = res[0][v][c];
            t2[v] = res[1][v][c];
            f[v] = o.v[v] * t1[v] + (1.0f - o.v[v]) * t2[v];  // fusion_adacofnet.py:198
        }
        store_vec<VEC>(out_frame + q, f);
        if (out_t1) store_vec<VEC>(out_t1 + q, t1);
        if (out_t2) store_vec<VEC>(out_t2 + q, t2);
    }
    if (out_mask) {
        float m[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float mx = fmaxf(var[0][v], var[1][v]);
            m[v] = fminf(fmaxf(mx, 0.0f), 20.0f) / 20.0f;  // fusion_adacofnet.py:211-212
        }
        store_vec<VEC>(out_mask + (size_t)n * plane + pix, m);
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int vfi_adacof_forward(const float *input, const float *weight, const float *offset_i,
                                  const float *offset_j, float *output, int N, int C, int Hin,
                                  int Win, int H, int W, int F, int dilation, vfi_stream_t stream) {
    VFI_REQUIRE(input && weight && offset_i && offset_j && output, VFI_ERR_INVALID_ARG,
                "vfi_adacof_forward: null pointer");
    VFI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && F > 0 && dilation > 0, VFI_ERR_INVALID_ARG,
                "vfi_adacof_forward: non-positive size");
    // adacof.py:326-327
    VFI_REQUIRE(Hin - ((F - 1) * dilation + 1) == H - 1 && Win - ((F - 1) * dilation + 1) == W - 1,
                VFI_ERR_SHAPE, "vfi_adacof_forward: input %dx%d does not match output %dx%d for F=%d dilation=%d",
                Hin, Win, H, W, F, dilation);
    VFI_REQUIRE((long long)Hin * Win < (1ll << 31) && (long long)N * F * F * H * W < (1ll << 40),
                VFI_ERR_UNSUPPORTED, "vfi_adacof_forward: tensor too large");
    const bool vec = (W % 4 == 0) && aligned16(weight) && aligned16(offset_i) && aligned16(offset_j) &&
                     aligned16(output);
    hipStream_t s = vfi::as_stream(stream);
    const int ct = (C % 3 == 0) ? 3 : 1;  // channels per thread
    dim3 block(64, 4);
    dim3 grid(vfi::ceil_div(W, 64 * (vec ? 4 : 1)), vfi::ceil_div(H, 4), N * vfi::ceil_div(C, ct));
    VFI_REQUIRE(grid.z <= 65535 && grid.y <= 65535, VFI_ERR_UNSUPPORTED, "vfi_adacof_forward: grid too large");
#define LAUNCH(CT, VEC)                                                                         \
    hipLaunchKernelGGL((adacof_forward_kernel<CT, VEC>), grid, block, 0, s, input, weight, offset_i, \
                       offset_j, output, C, Hin, Win, H, W, F, dilation)
    if (ct == 3) { if (vec) LAUNCH(3, 4); else LAUNCH(3, 1); }
    else         { if (vec) LAUNCH(1, 4); else LAUNCH(1, 1); }
#undef LAUNCH
    return vfi::check_launch("vfi_adacof_forward");
}

static int adacof_fused_impl(const float *frame0, const float *frame2, const float *w1,
                             const float *a1, const float *b1, const float *w2, const float *a2,
                             const float *b2, const float *occ, float *out_t1, float *out_t2,
                             float *out_frame, float *out_mask, int N, int C, int H, int W, int F,
                             int dilation, bool rgbx, bool softmax, vfi_stream_t stream) {
    VFI_REQUIRE(frame0 && frame2 && w1 && a1 && b1 && w2 && a2 && b2 && occ && out_frame,
                VFI_ERR_INVALID_ARG, "vfi_adacof_fused: null pointer");
    VFI_REQUIRE(N > 0 && H > 0 && W > 0 && F > 0 && dilation > 0, VFI_ERR_INVALID_ARG,
                "vfi_adacof_fused: non-positive size");
    VFI_REQUIRE(C == 3, VFI_ERR_UNSUPPORTED, "vfi_adacof_fused: C=%d (only 3 colour channels)", C);
    VFI_REQUIRE(((F - 1) * dilation) % 2 == 0, VFI_ERR_SHAPE,
                "vfi_adacof_fused: (F-1)*dilation must be even (F=%d dilation=%d)", F, dilation);
    VFI_REQUIRE((long long)H * W < (1ll << 31), VFI_ERR_UNSUPPORTED, "vfi_adacof_fused: frame too large");
    bool al16 = aligned16(occ) && aligned16(out_frame);
    for (const void *p : {(const void *)w1, (const void *)a1, (const void *)b1, (const void *)w2,
                          (const void *)a2, (const void *)b2})
        al16 = al16 && aligned16(p);
    for (const void *p : {(const void *)out_t1, (const void *)out_t2, (const void *)out_mask})
        al16 = al16 && (p == nullptr || aligned16(p));
    static const int variant = getenv("VFI_ADACOF_VARIANT") ? atoi(getenv("VFI_ADACOF_VARIANT")) : 2;  // tuning aid
    // measured on MI355X at 1088x1920: VEC=1 1.29 ms < VEC=2 1.49 ms < VEC=4 1.95 ms (gather-issue bound)
    int vec = (al16 && W % 4 == 0) ? 4 : ((al16 && W % 2 == 0) ? 2 : 1);
    if (variant == 1 && vec == 4) vec = 2;
    if (variant == 2 || rgbx) vec = 1;
    if (rgbx) VFI_REQUIRE(aligned16(frame0) && aligned16(frame2), VFI_ERR_INVALID_ARG, "vfi_adacof_fused_rgbx: frames must be 16-B aligned");
    hipStream_t s = vfi::as_stream(stream);
    dim3 block(64, 4);
    dim3 grid(vfi::ceil_div(W, 64 * vec), vfi::ceil_div(H, 4), N);
    VFI_REQUIRE(grid.z <= 65535 && grid.y <= 65535, VFI_ERR_UNSUPPORTED, "vfi_adacof_fused: grid too large");
#define LAUNCH(VEC, FT, UN, MW)                                                                          \
    hipLaunchKernelGGL((adacof_fused_kernel<3, VEC, FT, UN, MW>), grid, block, 0, s, frame0, frame2, w1, a1, b1, \
                       w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, 0)
    if (softmax) VFI_REQUIRE(rgbx, VFI_ERR_UNSUPPORTED, "vfi_adacof_fused: softmax folding needs the rgbx entry point");
    static const int win_margin = getenv("VFI_ADACOF_MARGIN") ? atoi(getenv("VFI_ADACOF_MARGIN")) : 4;   // tuning aid; 0 = off
    const int wh = 4 + (F - 1) * dilation + 2 * win_margin + 1, ww = 64 + (F - 1) * dilation + 2 * win_margin + 1;
    const size_t win_bytes = 2ull * wh * ww * sizeof(float4);
    if (rgbx && win_margin > 0 && win_bytes <= 64 * 1024) {
        // LDS-window variants (<= 64 KiB of LDS: the default launch limit, two workgroups per CU)
#define LAUNCH_WIN(FT, UN, SM)                                                                                       \
    hipLaunchKernelGGL((adacof_fused_kernel<3, 1, FT, UN, 3, true, SM, true>), grid, block, win_bytes, s, frame0, frame2, w1, \
                       a1, b1, w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, win_margin)
        if (F == 5) { if (softmax) LAUNCH_WIN(5, 5, true); else LAUNCH_WIN(5, 5, false); }
        else        { if (softmax) LAUNCH_WIN(0, 1, true); else LAUNCH_WIN(0, 1, false); }
#undef LAUNCH_WIN
    } else if (rgbx) {
        if (F == 5 && softmax)
            hipLaunchKernelGGL((adacof_fused_kernel<3, 1, 5, 5, 3, true, true>), grid, block, 0, s, frame0, frame2, w1, a1, b1,
                               w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, 0);
        else if (softmax)
            hipLaunchKernelGGL((adacof_fused_kernel<3, 1, 0, 1, 4, true, true>), grid, block, 0, s, frame0, frame2, w1, a1, b1,
                               w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, 0);
        else if (F == 5)
            hipLaunchKernelGGL((adacof_fused_kernel<3, 1, 5, 5, 3, true>), grid, block, 0, s, frame0, frame2, w1, a1, b1,
                               w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, 0);
        else
            hipLaunchKernelGGL((adacof_fused_kernel<3, 1, 0, 1, 4, true>), grid, block, 0, s, frame0, frame2, w1, a1, b1,
                               w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, H, W, F, dilation, 0);
    } else if (F == 5) {
        if (vec == 4) LAUNCH(4, 5, 1, 3);
        else if (vec == 2) LAUNCH(2, 5, 1, 4);
        else LAUNCH(1, 5, 5, 3);
    } else {
        if (vec == 4) LAUNCH(4, 0, 1, 3); else if (vec == 2) LAUNCH(2, 0, 1, 4); else LAUNCH(1, 0, 1, 4);
    }
#undef LAUNCH
    return vfi::check_launch("vfi_adacof_fused");
}

extern "C" int vfi_adacof_fused(const float *frame0, const float *frame2, const float *w1,
                                const float *a1, const float *b1, const float *w2, const float *a2,
                                const float *b2, const float *occ, float *out_t1, float *out_t2,
                                float *out_frame, float *out_mask, int N, int C, int H, int W, int F,
                                int dilation, vfi_stream_t stream) {
    return adacof_fused_impl(frame0, frame2, w1, a1, b1, w2, a2, b2, occ, out_t1, out_t2, out_frame, out_mask, N, C,
                             H, W, F, dilation, false, false, stream);
}

extern "C" int vfi_adacof_fused_rgbx(const float *frame0_rgbx, const float *frame2_rgbx, const float *w1,
                                     const float *a1, const float *b1, const float *w2, const float *a2,
                                     const float *b2, const float *occ, float *out_t1, float *out_t2,
                                     float *out_frame, float *out_mask, int N, int H, int W, int F,
                                     int dilation, int weights_are_logits, vfi_stream_t stream) {
    return adacof_fused_impl(frame0_rgbx, frame2_rgbx, w1, a1, b1, w2, a2, b2, occ, out_t1, out_t2, out_frame,
                             out_mask, N, 3, H, W, F, dilation, true, weights_are_logits != 0, stream);
}
