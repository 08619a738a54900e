// HBM-bound glue kernels between the convolutions of the three networks (gfx950).
// Each one replaces a torch op of the reference's forward passes; all take explicit batch strides
// so they can read / write channel slices of wider NCHW tensors (no concat / split copies).
// One thread per output element, lanes along x (coalesced 256 B per wave-instruction).
#include "vfi_common.h"

namespace {

using vfi::ceil_div;

constexpr int kThreads = 256;
inline int blocks_for(long long n) {
    long long b = (n + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > 8 * 2048 ? 8 * 2048 : b));  // grid-stride beyond 16k blocks
}

// ---- AdaCoFNet.forward prologue ---------------------------------------------------------------
// reflect-pad bottom/right to (Hp, Wp) (fusion_adacofnet.py:182-192), keep the raw padded frames for
// the sampler and emit cat(frame0 - mean, frame2 - mean) (utility.py:86-87, fusion_adacofnet.py:110).
__global__ void adacof_prepare_kernel(const float *__restrict__ f0, const float *__restrict__ f2,
                                      float *__restrict__ p0, float *__restrict__ p2, float *__restrict__ x6,
                                      int N, int H, int W, int Hp, int Wp, int rgbx) {
    const long long total = (long long)N * 3 * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = i % Wp, y = (i / Wp) % Hp, c = (i / ((long long)Wp * Hp)) % 3, n = i / ((long long)Wp * Hp * 3);
        const int sy = y < H ? y : 2 * (H - 1) - y, sx = x < W ? x : 2 * (W - 1) - x;
        const size_t s = (((size_t)n * 3 + c) * H + sy) * W + sx;
        const float mean = c == 0 ? 0.4631f : (c == 1 ? 0.4352f : 0.3990f);
        const float a = f0[s], b = f2[s];
        const size_t plane = (size_t)Hp * Wp, pix = (size_t)y * Wp + x;
        if (rgbx) {  // pixel-interleaved (N, Hp, Wp, 4) for the sampler's 16-B gathers; 4th float unused
            p0[((size_t)n * plane + pix) * 4 + c] = a;
            p2[((size_t)n * plane + pix) * 4 + c] = b;
        } else {
            p0[i] = a;
            p2[i] = b;
        }
        x6[((size_t)n * 6 + c) * plane + pix] = a - mean;
        x6[((size_t)n * 6 + 3 + c) * plane + pix] = b - mean;
    }
}

// ---- 2x2 stride-2 pooling (AvgPool2d: fusion_adacofnet.py:76-89; MaxPool2d: fusion_net.py:41,59) -----
template <bool MAX>
__global__ void pool2_kernel(const float *__restrict__ x, long long x_bs, float *__restrict__ y, long long y_bs,
                             int N, int C, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * C * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = i % Wo, yo = (i / Wo) % Ho, c = (i / ((long long)Wo * Ho)) % C, n = i / ((long long)Wo * Ho * C);
        const float *p = x + (size_t)n * x_bs + ((size_t)c * H + 2 * yo) * W + 2 * xo;
        const float2 r0 = *reinterpret_cast<const float2 *>(p);
        const float2 r1 = *reinterpret_cast<const float2 *>(p + W);
        const float v = MAX ? fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y)) : (r0.x + r0.y + r1.x + r1.y) * 0.25f;
        y[(size_t)n * y_bs + ((size_t)c * Ho + yo) * Wo + xo] = v;
    }
}
template <bool MAX>
__global__ void pool2_kernel_unaligned(const float *__restrict__ x, long long x_bs, float *__restrict__ y,
                                       long long y_bs, int N, int C, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * C * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = i % Wo, yo = (i / Wo) % Ho, c = (i / ((long long)Wo * Ho)) % C, n = i / ((long long)Wo * Ho * C);
        const float *p = x + (size_t)n * x_bs + ((size_t)c * H + 2 * yo) * W + 2 * xo;
        const float a = p[0], b = p[1], cc = p[W], d = p[W + 1];
        const float v = MAX ? fmaxf(fmaxf(a, b), fmaxf(cc, d)) : (a + b + cc + d) * 0.25f;
        y[(size_t)n * y_bs + ((size_t)c * Ho + yo) * Wo + xo] = v;
    }
}

// ---- bilinear resize (torch upsample_bilinear2d semantics) -------------------------------------------
// align_corners=True, x2: fusion_adacofnet.py:30,42,54,68; align_corners=False to an arbitrary size:
// phase_net.py:138-139; align_corners=False, x2 after ReLU, + skip: fusion_net.py:65-67.
// Launch geometry of both forms: blockIdx.z = n*C + c (one plane), blockIdx.y * 4 + threadIdx.y = output row,
// blockIdx.x * 64 + threadIdx.x = output column (or column quad): no integer division per element.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ x, long long x_bs,
                                                              const float *__restrict__ res, long long res_bs,
                                                              float *__restrict__ y, long long y_bs, int C, int Hi, int Wi,
                                                              int Ho, int Wo, int align_corners, int relu_in) {
    const int xo = blockIdx.x * 64 + threadIdx.x, yo = blockIdx.y * 4 + threadIdx.y;
    if (xo >= Wo || yo >= Ho) return;
    const int n = blockIdx.z / C, c = blockIdx.z - n * C;
    const float sy = align_corners ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f) : (float)Hi / (float)Ho;
    const float sx = align_corners ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f) : (float)Wi / (float)Wo;
    const float fy = align_corners ? sy * yo : fmaxf(sy * (yo + 0.5f) - 0.5f, 0.0f);
    const float fx = align_corners ? sx * xo : fmaxf(sx * (xo + 0.5f) - 0.5f, 0.0f);
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float *p = x + (size_t)n * x_bs + (size_t)c * Hi * Wi;
    float v00 = p[(size_t)y0 * Wi + x0], v01 = p[(size_t)y0 * Wi + x1];
    float v10 = p[(size_t)y1 * Wi + x0], v11 = p[(size_t)y1 * Wi + x1];
    if (relu_in) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
    float v = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    const size_t o = ((size_t)c * Ho + yo) * Wo + xo;
    if (res) v += res[(size_t)n * res_bs + o];
    y[(size_t)n * y_bs + o] = v;
}

// four consecutive output columns per thread, one 16-byte store
__global__ __launch_bounds__(256) void resize_bilinear_vec4_kernel(const float *__restrict__ x, long long x_bs,
                                                                   const float *__restrict__ res, long long res_bs,
                                                                   float *__restrict__ y, long long y_bs, int C, int Hi, int Wi,
                                                                   int Ho, int Wo, int align_corners, int relu_in) {
    const int xq = blockIdx.x * 64 + threadIdx.x, yo = blockIdx.y * 4 + threadIdx.y;
    if (xq * 4 >= Wo || yo >= Ho) return;
    const int n = blockIdx.z / C, c = blockIdx.z - n * C;
    const float sy = align_corners ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f) : (float)Hi / (float)Ho;
    const float sx = align_corners ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f) : (float)Wi / (float)Wo;
    const float fy = align_corners ? sy * yo : fmaxf(sy * (yo + 0.5f) - 0.5f, 0.0f);
    const int y0 = min((int)fy, Hi - 1), y1 = min(y0 + 1, Hi - 1);
    const float ly = fy - (float)y0;
    const float *p0 = x + (size_t)n * x_bs + ((size_t)c * Hi + y0) * Wi;
    const float *p1 = x + (size_t)n * x_bs + ((size_t)c * Hi + y1) * Wi;
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xo = xq * 4 + k;
        const float fx = align_corners ? sx * xo : fmaxf(sx * (xo + 0.5f) - 0.5f, 0.0f);
        const int x0 = min((int)fx, Wi - 1), x1 = min(x0 + 1, Wi - 1);
        const float lx = fx - (float)x0;
        float v00 = p0[x0], v01 = p0[x1], v10 = p1[x0], v11 = p1[x1];
        if (relu_in) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
        out[k] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    }
    const size_t o = ((size_t)c * Ho + yo) * Wo + (size_t)xq * 4;
    float4 v = make_float4(out[0], out[1], out[2], out[3]);
    if (res) {
        const float4 r = *reinterpret_cast<const float4 *>(res + (size_t)n * res_bs + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4 *>(y + (size_t)n * y_bs + o) = v;
}

// Up-scaling (source step <= 1 in both directions): a workgroup stages the source window of a 16-row x 256-column
// output tile in LDS with coalesced loads (each source element is fetched once per tile instead of once per output
// that uses it); every thread writes four groups of 4 consecutive columns (VEC: as float4 -- rows 16-byte aligned).
// Same arithmetic as above.
constexpr int kRsTileY = 16, kRsTileX = 256, kRsMaxRows = kRsTileY + 2, kRsMaxCols = kRsTileX + 2;
template <bool VEC>
__global__ __launch_bounds__(256) void resize_bilinear_tile_kernel(const float *__restrict__ x, long long x_bs,
                                                                   const float *__restrict__ res, long long res_bs,
                                                                   float *__restrict__ y, long long y_bs, int C, int Hi, int Wi,
                                                                   int Ho, int Wo, int align_corners, int relu_in) {
    __shared__ float tile[kRsMaxRows][kRsMaxCols + 2];
    const int X0 = blockIdx.x * kRsTileX, Y0 = blockIdx.y * kRsTileY;
    const int n = blockIdx.z / C, c = blockIdx.z - n * C;
    const float sy = align_corners ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f) : (float)Hi / (float)Ho;
    const float sx = align_corners ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f) : (float)Wi / (float)Wo;
    auto src_y = [&](int yo) { return align_corners ? sy * yo : fmaxf(sy * (yo + 0.5f) - 0.5f, 0.0f); };
    auto src_x = [&](int xo) { return align_corners ? sx * xo : fmaxf(sx * (xo + 0.5f) - 0.5f, 0.0f); };
    const int ylo = min((int)src_y(Y0), Hi - 1), yhi = min(min((int)src_y(min(Y0 + kRsTileY - 1, Ho - 1)), Hi - 1) + 1, Hi - 1);
    const int xlo = min((int)src_x(X0), Wi - 1), xhi = min(min((int)src_x(min(X0 + kRsTileX - 1, Wo - 1)), Wi - 1) + 1, Wi - 1);
    const int nrows = yhi - ylo + 1, ncols = xhi - xlo + 1;          // <= kRsMaxRows x kRsMaxCols (host checks the scale)
    const float *p = x + (size_t)n * x_bs + (size_t)c * Hi * Wi;
    for (int r = threadIdx.x >> 6; r < nrows; r += 4)
        for (int q = threadIdx.x & 63; q < ncols; q += 64) {
            const float v = p[(size_t)(ylo + r) * Wi + xlo + q];
            tile[r][q] = relu_in ? fmaxf(v, 0.0f) : v;
        }
    __syncthreads();
    const int xo0 = X0 + 4 * (threadIdx.x & 63);
    if (xo0 >= Wo) return;
    int x0[4], x1[4];
    float lx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float fx = src_x(xo0 + k);
        const int xs = min((int)fx, Wi - 1);
        lx[k] = fx - (float)xs;
        x0[k] = xs - xlo;
        x1[k] = min(xs + 1, Wi - 1) - xlo;
    }
#pragma unroll
    for (int rr = 0; rr < kRsTileY / 4; ++rr) {
        const int yo = Y0 + (threadIdx.x >> 6) + 4 * rr;
        if (yo >= Ho) break;
        const float fy = src_y(yo);
        const int ys = min((int)fy, Hi - 1);
        const float ly = fy - (float)ys;
        const float *t0 = tile[ys - ylo], *t1 = tile[min(ys + 1, Hi - 1) - ylo];
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out[k] = (1.0f - ly) * ((1.0f - lx[k]) * t0[x0[k]] + lx[k] * t0[x1[k]]) +
                     ly * ((1.0f - lx[k]) * t1[x0[k]] + lx[k] * t1[x1[k]]);
        const size_t o = ((size_t)c * Ho + yo) * Wo + xo0;
        if (VEC) {
            float4 v = make_float4(out[0], out[1], out[2], out[3]);
            if (res) {
                const float4 r = *reinterpret_cast<const float4 *>(res + (size_t)n * res_bs + o);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            *reinterpret_cast<float4 *>(y + (size_t)n * y_bs + o) = v;
        } else if (xo0 + 3 < Wo) {
            // rows of a width that is no multiple of four (the 1358- and 679-pixel pyramid levels): still one 16-byte store
            // per thread -- the memory pipeline needs 4-byte alignment only, which the compiler cannot be told through a
            // float4 pointer
            if (res) {
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] += res[(size_t)n * res_bs + o + k];
            }
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 v = {out[0], out[1], out[2], out[3]};
            asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(y + (size_t)n * y_bs + o), "v"(v) : "memory");
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (xo0 + k < Wo) y[(size_t)n * y_bs + o + k] = out[k] + (res ? res[(size_t)n * res_bs + o + k] : 0.0f);
        }
    }
}

// ---- single-output-channel `Upsample(x2, align_corners=True) -> Conv2d(C, 1, 3, padding=1)` tail ------------------
// (Subnet_occlusion: fusion_adacofnet.py:68-70).  Both maps are linear, so conv(U(x)) = sum_t shift_t(U(m_t)) with
// m_t = sum_c w[c][t] x_c a 1x1 convolution at LOW resolution (done by vfi_conv2d).  This kernel finishes:
// out(y,x) = act(bias + sum_t bilinear(m_t; y+dy_t, x+dx_t)), taps falling outside the output are zero (padding).
__global__ void upsample_tapsum_kernel(const float *__restrict__ m, float *__restrict__ out, int N, int Hs, int Ws,
                                       float bias, int act) {
    const int H = 2 * Hs, W = 2 * Ws;
    const float sy = H > 1 ? (float)(Hs - 1) / (float)(H - 1) : 0.0f, sx = W > 1 ? (float)(Ws - 1) / (float)(W - 1) : 0.0f;
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = i % W, y = (i / W) % H, n = i / ((long long)W * H);
        const float *mp = m + (size_t)n * 9 * Hs * Ws;
        float acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= H) continue;
            const float fy = sy * (float)yy;
            const int y0 = (int)fy, y1 = min(y0 + 1, Hs - 1);
            const float ly = fy - (float)y0;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const float fx = sx * (float)xx;
                const int x0 = (int)fx, x1 = min(x0 + 1, Ws - 1);
                const float lx = fx - (float)x0;
                const float *p = mp + (size_t)(ky * 3 + kx) * Hs * Ws;
                acc += (1.0f - ly) * ((1.0f - lx) * p[y0 * Ws + x0] + lx * p[y0 * Ws + x1]) +
                       ly * ((1.0f - lx) * p[y1 * Ws + x0] + lx * p[y1 * Ws + x1]);
            }
        }
        out[i] = act == 4 ? 1.0f / (1.0f + expf(-acc)) : (act == 1 ? fmaxf(acc, 0.0f) : acc);
    }
}

// ---- softmax over the channel axis (Subnet_weight: fusion_adacofnet.py:56) -----------------------------
__global__ void softmax_channels_kernel(const float *__restrict__ x, long long x_bs, float *__restrict__ y,
                                        long long y_bs, int N, int C, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, n = i / HW;
        const float *xp = x + (size_t)n * x_bs + p;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, xp[(size_t)c * HW]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += expf(xp[(size_t)c * HW] - m);
        float *yp = y + (size_t)n * y_bs + p;
        for (int c = 0; c < C; ++c) yp[(size_t)c * HW] = expf(xp[(size_t)c * HW] - m) / s;
    }
}

// ---- dst = src (/ div[n]) * mul over a per-sample block of `count` floats ----------------------------------
// slice copies into concat buffers; phase / pi and amplitude / max of PhaseNet.normalize_vals (phase_net.py:61-70).
__global__ void affine_slice_kernel(const float *__restrict__ src, long long src_bs, float *__restrict__ dst,
                                    long long dst_bs, int N, long long count, const float *__restrict__ div,
                                    float mul) {
    const long long total = (long long)N * count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i % count;
        const int n = i / count;
        float v = src[(size_t)n * src_bs + e];
        if (div) v = v / div[n];
        dst[(size_t)n * dst_bs + e] = v * mul;
    }
}

// ---- per-sample maximum (+ eps) : phase_net.py:55,69 ---------------------------------------------------------
__device__ __forceinline__ unsigned enc_ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ void batch_max_kernel(const float *__restrict__ x, long long x_bs, long long count,
                                 unsigned *__restrict__ enc) {
    const int n = blockIdx.y;
    const float *p = x + (size_t)n * x_bs;
    float m = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, p[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));  // 64-lane butterfly
    __shared__ float part[kThreads / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = part[0];
        for (int w = 1; w < kThreads / 64; ++w) r = fmaxf(r, part[w]);
        atomicMax(enc + n, enc_ordered(r));
    }
}
__global__ void batch_max_finish_kernel(const unsigned *__restrict__ enc, float *__restrict__ out, int N, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = dec_ordered(enc[n]) + eps;
}

// ---- PhaseNet per-level outputs -----------------------------------------------------------------------------
// phase_net.py:155-168 + reverse_normalize :80-90: beta = (pred[:,4:8]+1)/2 ; amp = beta*amp_in[:,4:8] +
// (1-beta)*amp_in[:,0:4] ; phase_out = pred[:,0:4]*pi ; amp_out = amp*max[n].  Outputs are (N,4,h,w) dense
// == the per-image layout (N*4,1,h,w) with index colour*4+band.
__global__ void phasenet_emit_kernel(const float *__restrict__ pred, long long pred_bs, const float *__restrict__ amp_in,
                                     long long amp_bs, const float *__restrict__ maxv, float *__restrict__ phase_out,
                                     float *__restrict__ amp_out, int N, int HW) {
    const long long total = (long long)N * 4 * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, b = (i / HW) % 4, n = i / ((long long)HW * 4);
        const float *pr = pred + (size_t)n * pred_bs + p;
        const float *am = amp_in + (size_t)n * amp_bs + p;
        const float beta = (pr[(size_t)(4 + b) * HW] + 1.0f) / 2.0f;
        const float a = fmaf(beta, am[(size_t)(4 + b) * HW], (1.0f - beta) * am[(size_t)b * HW]);      // (spelled out: vfi_phasenet_predict computes the same bits)
        phase_out[i] = pr[(size_t)b * HW] * 3.14159265358979323846f;
        amp_out[i] = a * maxv[n];
    }
}
// phase_net.py:113-116 + :96-98: alpha = (pred+1)/2 ; low = (alpha*low[:,0] + (1-alpha)*low[:,1]) * max_low[n]
__global__ void phasenet_emit_low_kernel(const float *__restrict__ pred, long long pred_bs, const float *__restrict__ low,
                                         long long low_bs, const float *__restrict__ maxv, float *__restrict__ out,
                                         int N, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, n = i / HW;
        const float alpha = (pred[(size_t)n * pred_bs + p] + 1.0f) / 2.0f;
        const float *l = low + (size_t)n * low_bs + p;
        out[i] = (alpha * l[0] + (1.0f - alpha) * l[HW]) * maxv[n];
    }
}

// ---- FusionNet tail: clamp(base + tanh(x), 0, 1)  (fusion_net.py:70-77) ----------------------------------------
__global__ void tanh_residual_clamp_kernel(const float *__restrict__ x, const float *__restrict__ base,
                                           float *__restrict__ y, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        y[i] = fminf(fmaxf(base[i] + tanhf(x[i]), 0.0f), 1.0f);
}

}  // namespace

#define LAUNCH_1D(kernel, total, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(blocks_for(total)), dim3(kThreads), 0, vfi::as_stream(stream), __VA_ARGS__)

extern "C" int vfi_adacof_prepare(const float *frame0, const float *frame2, float *pad0, float *pad2, float *x6,
                                  int N, int H, int W, int Hp, int Wp, int rgbx, vfi_stream_t stream) {
    VFI_REQUIRE(frame0 && frame2 && pad0 && pad2 && x6, VFI_ERR_INVALID_ARG, "vfi_adacof_prepare: null pointer");
    VFI_REQUIRE(N > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, VFI_ERR_INVALID_ARG, "vfi_adacof_prepare: bad sizes");
    VFI_REQUIRE(Hp - H < H && Wp - W < W, VFI_ERR_SHAPE, "vfi_adacof_prepare: reflect pad %dx%d needs a larger frame than %dx%d",
                Hp - H, Wp - W, H, W);
    LAUNCH_1D(adacof_prepare_kernel, (long long)N * 3 * Hp * Wp, stream, frame0, frame2, pad0, pad2, x6, N, H, W, Hp, Wp, rgbx);
    return vfi::check_launch("vfi_adacof_prepare");
}

extern "C" int vfi_pool2(const float *x, long long x_bstride, float *y, long long y_bstride, int N, int C, int H,
                         int W, int is_max, vfi_stream_t stream) {
    VFI_REQUIRE(x && y, VFI_ERR_INVALID_ARG, "vfi_pool2: null pointer");
    VFI_REQUIRE(N > 0 && C > 0 && H >= 2 && W >= 2, VFI_ERR_INVALID_ARG, "vfi_pool2: bad sizes");
    const long long total = (long long)N * C * (H / 2) * (W / 2);
    const bool al = (W % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 7u) == 0) && (x_bstride % 2 == 0);
    if (al) {
        if (is_max) LAUNCH_1D(pool2_kernel<true>, total, stream, x, x_bstride, y, y_bstride, N, C, H, W);
        else LAUNCH_1D(pool2_kernel<false>, total, stream, x, x_bstride, y, y_bstride, N, C, H, W);
    } else {
        if (is_max) LAUNCH_1D(pool2_kernel_unaligned<true>, total, stream, x, x_bstride, y, y_bstride, N, C, H, W);
        else LAUNCH_1D(pool2_kernel_unaligned<false>, total, stream, x, x_bstride, y, y_bstride, N, C, H, W);
    }
    return vfi::check_launch("vfi_pool2");
}

extern "C" int vfi_resize_bilinear(const float *x, long long x_bstride, const float *residual, long long res_bstride,
                                   float *y, long long y_bstride, int N, int C, int Hin, int Win, int Hout, int Wout,
                                   int align_corners, int relu_input, vfi_stream_t stream) {
    VFI_REQUIRE(x && y, VFI_ERR_INVALID_ARG, "vfi_resize_bilinear: null pointer");
    VFI_REQUIRE(N > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, VFI_ERR_INVALID_ARG,
                "vfi_resize_bilinear: bad sizes");
    const bool vec = Wout % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0 && y_bstride % 4 == 0 &&
                     (!residual || ((reinterpret_cast<uintptr_t>(residual) & 15u) == 0 && res_bstride % 4 == 0));
    VFI_REQUIRE((long long)N * C <= 65535, VFI_ERR_UNSUPPORTED, "vfi_resize_bilinear: N*C = %lld planes", (long long)N * C);
    const dim3 block(64, 4), grid(ceil_div(vec ? Wout / 4 : Wout, 64), ceil_div(Hout, 4), N * C);
    if (Hin <= Hout && Win <= Wout && Hout >= kRsTileY && Wout >= 64) {      // up-scaling: LDS-staged tiles
        const dim3 tgrid(ceil_div(Wout, kRsTileX), ceil_div(Hout, kRsTileY), N * C);
        if (vec)
            hipLaunchKernelGGL(resize_bilinear_tile_kernel<true>, tgrid, dim3(256), 0, vfi::as_stream(stream), x, x_bstride, residual,
                               res_bstride, y, y_bstride, C, Hin, Win, Hout, Wout, align_corners, relu_input);
        else
            hipLaunchKernelGGL(resize_bilinear_tile_kernel<false>, tgrid, dim3(256), 0, vfi::as_stream(stream), x, x_bstride, residual,
                               res_bstride, y, y_bstride, C, Hin, Win, Hout, Wout, align_corners, relu_input);
    } else if (vec)
        hipLaunchKernelGGL(resize_bilinear_vec4_kernel, grid, block, 0, vfi::as_stream(stream), x, x_bstride, residual,
                           res_bstride, y, y_bstride, C, Hin, Win, Hout, Wout, align_corners, relu_input);
    else
        hipLaunchKernelGGL(resize_bilinear_kernel, grid, block, 0, vfi::as_stream(stream), x, x_bstride, residual, res_bstride,
                           y, y_bstride, C, Hin, Win, Hout, Wout, align_corners, relu_input);
    return vfi::check_launch("vfi_resize_bilinear");
}

extern "C" int vfi_softmax_channels(const float *x, long long x_bstride, float *y, long long y_bstride, int N, int C,
                                    int HW, vfi_stream_t stream) {
    VFI_REQUIRE(x && y, VFI_ERR_INVALID_ARG, "vfi_softmax_channels: null pointer");
    VFI_REQUIRE(N > 0 && C > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_softmax_channels: bad sizes");
    LAUNCH_1D(softmax_channels_kernel, (long long)N * HW, stream, x, x_bstride, y, y_bstride, N, C, HW);
    return vfi::check_launch("vfi_softmax_channels");
}

extern "C" int vfi_affine_slice(const float *src, long long src_bstride, float *dst, long long dst_bstride, int N,
                                long long count, const float *div_per_sample, float mul, vfi_stream_t stream) {
    VFI_REQUIRE(src && dst, VFI_ERR_INVALID_ARG, "vfi_affine_slice: null pointer");
    VFI_REQUIRE(N > 0 && count > 0, VFI_ERR_INVALID_ARG, "vfi_affine_slice: bad sizes");
    LAUNCH_1D(affine_slice_kernel, (long long)N * count, stream, src, src_bstride, dst, dst_bstride, N, count,
              div_per_sample, mul);
    return vfi::check_launch("vfi_affine_slice");
}

extern "C" int vfi_batch_max(const float *x, long long x_bstride, int N, long long count, float eps, float *out_max,
                             void *workspace_u32, vfi_stream_t stream) {
    VFI_REQUIRE(x && out_max && workspace_u32, VFI_ERR_INVALID_ARG, "vfi_batch_max: null pointer");
    VFI_REQUIRE(N > 0 && N <= 65535 && count > 0, VFI_ERR_INVALID_ARG, "vfi_batch_max: bad sizes");
    hipStream_t s = vfi::as_stream(stream);
    hipError_t e = hipMemsetAsync(workspace_u32, 0, sizeof(unsigned) * N, s);
    if (e != hipSuccess) return vfi::fail(VFI_ERR_LAUNCH, "vfi_batch_max: memset: %s", hipGetErrorString(e));
    long long b = (count + kThreads * 8 - 1) / (kThreads * 8);
    dim3 grid((unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b)), N);
    hipLaunchKernelGGL(batch_max_kernel, grid, dim3(kThreads), 0, s, x, x_bstride, count,
                       static_cast<unsigned *>(workspace_u32));
    hipLaunchKernelGGL(batch_max_finish_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, s,
                       static_cast<const unsigned *>(workspace_u32), out_max, N, eps);
    return vfi::check_launch("vfi_batch_max");
}

extern "C" int vfi_phasenet_emit(const float *pred, long long pred_bstride, const float *amp_in, long long amp_bstride,
                                 const float *max_amp, float *phase_out, float *amp_out, int N, int HW,
                                 vfi_stream_t stream) {
    VFI_REQUIRE(pred && amp_in && max_amp && phase_out && amp_out, VFI_ERR_INVALID_ARG, "vfi_phasenet_emit: null pointer");
    VFI_REQUIRE(N > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_phasenet_emit: bad sizes");
    LAUNCH_1D(phasenet_emit_kernel, (long long)N * 4 * HW, stream, pred, pred_bstride, amp_in, amp_bstride, max_amp,
              phase_out, amp_out, N, HW);
    return vfi::check_launch("vfi_phasenet_emit");
}

extern "C" int vfi_phasenet_emit_low(const float *pred, long long pred_bstride, const float *low_in, long long low_bstride,
                                     const float *max_low, float *low_out, int N, int HW, vfi_stream_t stream) {
    VFI_REQUIRE(pred && low_in && max_low && low_out, VFI_ERR_INVALID_ARG, "vfi_phasenet_emit_low: null pointer");
    VFI_REQUIRE(N > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_phasenet_emit_low: bad sizes");
    LAUNCH_1D(phasenet_emit_low_kernel, (long long)N * HW, stream, pred, pred_bstride, low_in, low_bstride, max_low,
              low_out, N, HW);
    return vfi::check_launch("vfi_phasenet_emit_low");
}

extern "C" int vfi_tanh_residual_clamp(const float *x, const float *base, float *y, long long count,
                                       vfi_stream_t stream) {
    VFI_REQUIRE(x && base && y, VFI_ERR_INVALID_ARG, "vfi_tanh_residual_clamp: null pointer");
    VFI_REQUIRE(count > 0, VFI_ERR_INVALID_ARG, "vfi_tanh_residual_clamp: bad size");
    LAUNCH_1D(tanh_residual_clamp_kernel, count, stream, x, base, y, count);
    return vfi::check_launch("vfi_tanh_residual_clamp");
}

extern "C" int vfi_upsample2x_tapsum(const float *taps_lowres, float *out, int N, int Hs, int Ws, float bias, int act,
                                     vfi_stream_t stream) {
    VFI_REQUIRE(taps_lowres && out, VFI_ERR_INVALID_ARG, "vfi_upsample2x_tapsum: null pointer");
    VFI_REQUIRE(N > 0 && Hs > 0 && Ws > 0, VFI_ERR_INVALID_ARG, "vfi_upsample2x_tapsum: bad sizes");
    VFI_REQUIRE(act == 0 || act == 1 || act == 4, VFI_ERR_UNSUPPORTED, "vfi_upsample2x_tapsum: act %d", act);
    LAUNCH_1D(upsample_tapsum_kernel, (long long)N * 4 * Hs * Ws, stream, taps_lowres, out, N, Hs, Ws, bias, act);
    return vfi::check_launch("vfi_upsample2x_tapsum");
}
