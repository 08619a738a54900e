// Complex steerable pyramid (frequency domain, scale_factor-generalised) for gfx950:
// the arithmetic behind Pyramid.filter / inv_filter (reference call sites src/train/pyramid.py:35-46;
// adapters coeff_to_values / values_to_coeff src/train/pyramid.py:48-112 are fused in).
//
// The reference delegates this arithmetic to the third-party `steerable.SCFpyr_PyTorch` (absent, fork
// unknown: see DESIGN.md, "pyramid spec"); the spec implemented here is the one restated in
// oracle/pyramid_cpu.py: level k works on the centred ceil(H/s^k) x ceil(W/s^k) window of the
// spectrum, radial raised-cosine transition shifted by log2(s) per level, nbands oriented analytic bands.
//
// Roofline: HBM (target).  No FFT library is linked: the large levels run on the wave-private register FFT engine of
// vfi_wfft.h (kernels in vfi_pyrw_kernels.h), every length that engine has no configuration for on the generic LDS
// engine of vfi_fft.h (mixed-radix Stockham + Bluestein).  Each pyramid level is two fused kernels per direction, so that
// no intermediate of the reference's op-by-op chain (mask products, crops, shifts, deepcopy, band spectra, band
// coefficients, 2*L*N*4 atan2/abs launches) is materialised except ONE half-transformed array T:
//   analysis  : the low-pass chain of the build is a product of real masks, so every level reads the R2C half spectrum S
//               of the image directly: level k, band b = IFFT2( i * window_k(S) * Q_k[b] ) with the table
//               Q_k[b] = lo0 * prod_{j<k} lomask_j * himask_k * anglemask_b folded at plan time (double precision): the
//               levels do not depend on each other and a level mask simply skips levels;
//               column kernel : (column tile, image, band): window of S (Hermitian half expanded and ifftshift done by
//                               index arithmetic) * Q, inverse column FFT -> T;
//               row kernel    : rows of T -> inverse row FFT -> (phase, amplitude) written straight into the caller's
//                               layout (per-image planes or PhaseNet's concat buffers), optional phase scale;
//   synthesis : row kernel (A cos p, A sin p -> forward row FFT -> T), column kernel (forward column FFT, sum of
//               rotated, masked band spectra + embedded low-pass of the coarser level); levels without bands only embed
//               (pyr_combine_kernel).
// All mask tables are precomputed once per plan in double precision, stored in the unshifted (FFT-native)
// index order so every table read is coalesced with the spectrum access.
#include "vfi_common.h"
#include "vfi_fft.h"
#include "vfi_pyramid_wave.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <new>
#include <vector>

namespace {

using vfi::ceil_div;
constexpr int kMaxLevels = 40;
constexpr int kMaxImages = 16;
constexpr double kPi = 3.14159265358979323846;

struct Level {
    int h, w;          // window size
    float *P_a;        // [nb][h][w] analysis  : lo0 * prod_{j<k} lomask_j * himask_k * angle mask (one sided), unshifted order
    float *P_s;        // [nb][h][w] synthesis : angle mask (two sided) * himask
    float *lomask;     // [h2][w2]  low-pass applied to the NEXT level's window, unshifted order of that window
};

}  // namespace

struct vfi_pyr_plan {
    int H, W, height, nbands, nlev, max_images;
    double scale;
    std::vector<Level> lev;      // nlev band levels
    int hl, wl;                  // low residual size
    float *lo0 = nullptr, *hi0 = nullptr;   // [H][W] unshifted
    float *low_gain = nullptr;   // [hl][wl] lo0 * prod_j lomask_j on the low residual's window, unshifted
    int tpitch_max = 0;          // row pitch of T the workspace is sized for (W rounded up to 16)
    std::map<int, const float2 *> wave_tw[3];   // [pass kind] engine length -> stage twiddles (vfi_wfft.h)   // engine length -> stage twiddles (vfi_wfft.h)
    // workspace (complex64 unless noted)
    float2 *half0 = nullptr;     // N x H x (W/2+1)   R2C spectrum of the input / FFT of high on synthesis
    float2 *half_hi = nullptr;   // N x H x (W/2+1)   high-pass half spectrum (C2R input)
    float2 *bands = nullptr;     // N x nb x H x W    band spectra / coefficients of the current level
    float2 *lod[2] = {nullptr, nullptr};   // N x H x W each: running low-pass spectrum (ping-pong)
    std::map<int, vfi::fft::Plan1D> fft1d;   // transform length -> tables (vfi_fft.h)
    unsigned *amp_bits = nullptr;            // kMaxLevels * 4 words: vfi_pyr_analyze_max
    std::vector<void *> allocs;
    // kept for vfi_pyr_plan_prepare_filter
    std::vector<double> log_rad, xr0, yr, yir;
    std::vector<float *> filters;   // [id] -> H x (W/2+1) radial gain tables
};

namespace {

// ---- host-side mask construction (double precision, numpy semantics) -----------------------------------
double interp(double x, const std::vector<double> &xp, const std::vector<double> &fp) {
    const size_t n = xp.size();
    if (x <= xp[0]) return fp[0];
    if (x >= xp[n - 1]) return fp[n - 1];
    size_t lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    const double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

std::vector<double> linspace_grid(int m) {  // prepare_grid axis
    std::vector<double> v(m);
    const double start = -(double)(m / 2) / (m / 2.0);
    const double stop = (double)(m / 2) / (m / 2.0) - (1 - m % 2) * 2.0 / m;
    const double step = m > 1 ? (stop - start) / (m - 1) : 0.0;
    for (int i = 0; i < m; ++i) v[i] = start + i * step;
    if (m > 1) v[m - 1] = stop;
    return v;
}

inline int level_size(int d, double s, int k) { return (int)std::ceil(d / std::pow(s, k) - 1e-9); }

template <typename T>
int dev_upload(vfi_pyr_plan *p, const std::vector<T> &host, T **dev) {
    if (hipMalloc((void **)dev, host.size() * sizeof(T)) != hipSuccess) return VFI_ERR_NOMEM;
    p->allocs.push_back(*dev);
    if (hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return VFI_ERR_LAUNCH;
    return VFI_OK;
}

int dev_alloc(vfi_pyr_plan *p, void **dev, size_t bytes) {
    if (hipMalloc(dev, bytes) != hipSuccess) return VFI_ERR_NOMEM;
    p->allocs.push_back(*dev);
    return VFI_OK;
}

// shifted-window index (DC at h/2) for an unshifted index u of a length-h axis
inline int shifted_of(int u, int h) { return (u + h / 2) % h; }

int build_tables(vfi_pyr_plan *p) {
    const int H = p->H, W = p->W, nb = p->nbands;
    const std::vector<double> gy = linspace_grid(H), gx = linspace_grid(W);
    // log_rad / angle on the full shifted grid
    std::vector<double> log_rad((size_t)H * W), angle((size_t)H * W);
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            angle[(size_t)i * W + j] = std::atan2(gy[i], gx[j]);
            log_rad[(size_t)i * W + j] = std::sqrt(gx[j] * gx[j] + gy[i] * gy[i]);
        }
    if (W > 1) log_rad[(size_t)(H / 2) * W + W / 2] = log_rad[(size_t)(H / 2) * W + W / 2 - 1];
    for (auto &v : log_rad) v = std::log2(v);
    // rcosFn(1, -0.5)
    const int n = 256;
    std::vector<double> xr(n + 3), yr(n + 3), yir(n + 3);
    for (int i = 0; i < n + 3; ++i) {
        const double x = kPi * (double)(i - n - 1) / 2.0 / n;
        xr[i] = x;
        yr[i] = std::cos(x) * std::cos(x);
    }
    yr[0] = yr[1];
    yr[n + 2] = yr[n + 1];
    for (int i = 0; i < n + 3; ++i) {
        xr[i] = -0.5 + 2.0 / kPi * (xr[i] + kPi / 4.0);
        yr[i] = std::sqrt(yr[i]);
        yir[i] = std::sqrt(std::fabs(1.0 - yr[i] * yr[i]));
    }
    // angular LUTs
    const int lut = 1024, order = nb - 1;
    const int nl = 3 * lut + 3;
    std::vector<double> xc(nl), ya(nl), ys(nl);
    double fact_o = 1, fact_2o = 1;
    for (int i = 2; i <= order; ++i) fact_o *= i;
    for (int i = 2; i <= 2 * order; ++i) fact_2o *= i;
    const double cst = std::pow(2.0, 2 * order) * fact_o * fact_o / (nb * fact_2o);
    for (int i = 0; i < nl; ++i) {
        xc[i] = kPi * (double)(i - (2 * lut + 1)) / lut;
        double alpha = std::fmod(xc[i] + kPi, 2 * kPi);
        if (alpha < 0) alpha += 2 * kPi;
        alpha -= kPi;
        const double c = std::pow(std::cos(xc[i]), order);
        ya[i] = 2.0 * std::sqrt(cst) * c * (std::fabs(alpha) < kPi / 2 ? 1.0 : 0.0);
        ys[i] = std::sqrt(cst) * c;
    }
    std::vector<double> xcb(nl);
    p->log_rad = log_rad; p->xr0 = xr; p->yr = yr; p->yir = yir;

    std::vector<float> t((size_t)H * W), t2((size_t)H * W);
    for (int u = 0; u < H; ++u)
        for (int v = 0; v < W; ++v) {
            const size_t s = (size_t)shifted_of(u, H) * W + shifted_of(v, W);
            t[(size_t)u * W + v] = (float)interp(log_rad[s], xr, yir);
            t2[(size_t)u * W + v] = (float)interp(log_rad[s], xr, yr);
        }
    int rc;
    if ((rc = dev_upload(p, t, &p->lo0)) || (rc = dev_upload(p, t2, &p->hi0))) return rc;

    // chain(s, k) = lo0 * prod_{j<k} lomask_j at the full-grid (shifted) position s: what the build's running low-pass
    // spectrum has been multiplied by when level k reads it (`lodft = dft * lo0mask`, then `lodft * lomask` per level)
    const double ls = std::log2(p->scale);
    std::vector<std::vector<double>> xr_of(p->nlev + 1, p->xr0);      // xr_of[j] = xr0 - j * log2(scale)
    for (int j = 1; j <= p->nlev; ++j)
        for (auto &x : xr_of[j]) x -= j * ls;
    auto chain = [&](size_t spos, int k) {
        double c = interp(log_rad[spos], xr_of[0], yir);
        for (int j = 1; j <= k; ++j) c *= interp(log_rad[spos], xr_of[j], yir);
        return c;
    };
    for (int k = 0; k < p->nlev; ++k) {
        Level &L = p->lev[k];
        for (auto &x : xr) x -= ls;
        const int h = L.h, w = L.w, sy = H / 2 - h / 2, sx = W / 2 - w / 2;
        std::vector<float> pa((size_t)nb * h * w), ps((size_t)nb * h * w);
        std::vector<float> hm((size_t)h * w);
        std::vector<double> ch((size_t)h * w);
        for (int u = 0; u < h; ++u)
            for (int v = 0; v < w; ++v) {
                const size_t s = (size_t)(sy + shifted_of(u, h)) * W + (sx + shifted_of(v, w));
                hm[(size_t)u * w + v] = (float)interp(log_rad[s], xr, yr);
                ch[(size_t)u * w + v] = chain(s, k);
            }
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < nl; ++i) xcb[i] = xc[i] + kPi * b / nb;
            for (int u = 0; u < h; ++u)
                for (int v = 0; v < w; ++v) {
                    const size_t s = (size_t)(sy + shifted_of(u, h)) * W + (sx + shifted_of(v, w));
                    const size_t o = ((size_t)b * h + u) * w + v;
                    // float32 tables of the oracle multiplied in fp32 there; here folded in double
                    pa[o] = (float)((double)(float)interp(angle[s], xcb, ya) * (double)hm[(size_t)u * w + v] * ch[(size_t)u * w + v]);
                    ps[o] = (float)((double)(float)interp(angle[s], xcb, ys) * (double)hm[(size_t)u * w + v]);
                }
        }
        const int h2 = k + 1 < p->nlev ? p->lev[k + 1].h : p->hl, w2 = k + 1 < p->nlev ? p->lev[k + 1].w : p->wl;
        const int sy2 = H / 2 - h2 / 2, sx2 = W / 2 - w2 / 2;
        std::vector<float> lm((size_t)h2 * w2);
        for (int u = 0; u < h2; ++u)
            for (int v = 0; v < w2; ++v) {
                const size_t s = (size_t)(sy2 + shifted_of(u, h2)) * W + (sx2 + shifted_of(v, w2));
                lm[(size_t)u * w2 + v] = (float)interp(log_rad[s], xr, yir);
            }
        if ((rc = dev_upload(p, pa, &L.P_a)) || (rc = dev_upload(p, ps, &L.P_s)) || (rc = dev_upload(p, lm, &L.lomask)))
            return rc;
    }
    {   // low residual: real(ifft2(window(dft) * lo0 * prod_j lomask_j))
        const int h2 = p->hl, w2 = p->wl, sy2 = H / 2 - h2 / 2, sx2 = W / 2 - w2 / 2;
        std::vector<float> lg((size_t)h2 * w2);
        for (int u = 0; u < h2; ++u)
            for (int v = 0; v < w2; ++v)
                lg[(size_t)u * w2 + v] = (float)chain((size_t)(sy2 + shifted_of(u, h2)) * W + (sx2 + shifted_of(v, w2)), p->nlev);
        if ((rc = dev_upload(p, lg, &p->low_gain))) return rc;
    }
    return VFI_OK;
}

// ---- device kernels ------------------------------------------------------------------------------------
using vfi::pyrw::PlaneMap;      // where image d's band-0 plane goes (vfi_pyramid_wave.h)

__device__ __forceinline__ int signed_freq(int u, int h) { return u <= h - h / 2 - 1 ? u : u - h; }

// high-pass half spectrum: hi_half = half * hi0 / (H W) (C2R input: `hi0dft = dft * hi0mask`, real(ifft2) of it)
__global__ __launch_bounds__(256) void pyr_high_kernel(const float2 *__restrict__ half, float2 *__restrict__ hi_half,
                                                       const float *__restrict__ hi0, int N, int H, int W, float inv_hw) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y, wh = W / 2 + 1;
    if (v >= wh) return;
    const float g = hi0[(size_t)u * W + v] * inv_hw;
    for (int n = 0; n < N; ++n) {
        const float2 z = half[((size_t)n * H + u) * wh + v];
        hi_half[((size_t)n * H + u) * wh + v] = make_float2(z.x * g, z.y * g);
    }
}

// low residual spectrum: the (hl x wl) window of the expanded half spectrum * low_gain (unshifted order)
__global__ __launch_bounds__(256) void pyr_low_kernel(const float2 *__restrict__ half, float2 *__restrict__ low,
                                                      const float *__restrict__ gain, int N, int H, int W, int hl, int wl) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= hl * wl) return;
    const int u = e / wl, v = e - u * wl, wh = W / 2 + 1;
    const int fy = signed_freq(u, hl), fx = signed_freq(v, wl);
    const int U = fx < 0 ? (fy > 0 ? H - fy : -fy) : (fy < 0 ? fy + H : fy), V = fx < 0 ? -fx : fx;
    const float g = gain[e];
    for (int n = 0; n < N; ++n) {
        float2 z = half[((size_t)n * H + U) * wh + V];
        if (fx < 0) z.y = -z.y;
        low[(size_t)n * hl * wl + e] = make_float2(z.x * g, z.y * g);
    }
}

// cur = sum_b (-i) * FFT(band_b) * P_s[b]  +  embed(res * lomask)     (reconstruct: orientdft + resdft)
template <int NB>
__global__ __launch_bounds__(256) void pyr_combine_kernel(const float2 *__restrict__ band, const float2 *__restrict__ res,
                                                          float2 *__restrict__ cur, const float *__restrict__ P_s,
                                                          const float *__restrict__ lomask, int N, int h, int w,
                                                          int h2, int w2, int have_bands) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int u = blockIdx.y;
    if (v >= w) return;
    const size_t hw = (size_t)h * w, o = (size_t)u * w + v;
    float ps[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ps[b] = have_bands ? P_s[(size_t)b * hw + o] : 0.0f;
    const int fy = signed_freq(u, h), fx = signed_freq(v, w);
    const bool inside = fy >= -(h2 / 2) && fy <= h2 - h2 / 2 - 1 && fx >= -(w2 / 2) && fx <= w2 - w2 / 2 - 1;
    const int u2 = fy < 0 ? fy + h2 : fy, v2 = fx < 0 ? fx + w2 : fx;
    const float lom = inside ? lomask[(size_t)u2 * w2 + v2] : 0.0f;
    for (int n = 0; n < N; ++n) {
        float2 acc = make_float2(0.0f, 0.0f);
        if (have_bands) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {  // * (-i) : (re, im) -> (im, -re)
                const float2 z = band[((size_t)n * NB + b) * hw + o];
                acc.x += z.y * ps[b];
                acc.y -= z.x * ps[b];
            }
        }
        if (inside) {
            const float2 r = res[((size_t)n * h2 + u2) * w2 + v2];
            acc.x += r.x * lom;
            acc.y += r.y * lom;
        }
        cur[(size_t)n * hw + o] = acc;
    }
}

// real image -> complex (imag 0) ; complex -> real * scale
__global__ void real_to_complex_kernel(const float *__restrict__ x, float2 *__restrict__ z, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        z[i] = make_float2(x ? x[i] : 0.0f, 0.0f);
}
__global__ void complex_real_kernel(const float2 *__restrict__ z, float *__restrict__ x, long long total, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        x[i] = z[i].x * scale;
}
// out = cur * lo0 + expand(hi_half) * hi0     (reconstruct: `tempdft * lo0mask + hidft * hi0mask`)
__global__ __launch_bounds__(256) void pyr_final_kernel(float2 *__restrict__ cur, const float2 *__restrict__ hi_half,
                                                        const float *__restrict__ lo0, const float *__restrict__ hi0,
                                                        int N, int h, int w) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int u = blockIdx.y;
    if (v >= w) return;
    const size_t hw = (size_t)h * w, o = (size_t)u * w + v;
    const float l0 = lo0[o], h0 = hi0[o];
    const int wh = w / 2 + 1;
    for (int n = 0; n < N; ++n) {
        float2 z = cur[(size_t)n * hw + o];
        z.x *= l0; z.y *= l0;
        if (hi_half) {
            float2 c;
            if (v < wh) c = hi_half[((size_t)n * h + u) * wh + v];
            else { c = hi_half[((size_t)n * h + (u ? h - u : 0)) * wh + (w - v)]; c.y = -c.y; }
            z.x += c.x * h0; z.y += c.y * h0;
        }
        cur[(size_t)n * hw + o] = z;
    }
}

// a = a * ga + b * gb on half spectra (gains already include 1/(H*W))
__global__ void pyr_gain_pair_kernel(float2 *__restrict__ a, const float2 *__restrict__ b, const float *__restrict__ ga,
                                     const float *__restrict__ gb, int N, long long per_image) {
    const long long total = (long long)N * per_image;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long j = i % per_image;
        const float x = ga[j], y = gb[j];
        const float2 za = a[i], zb = b[i];
        a[i] = make_float2(za.x * x + zb.x * y, za.y * x + zb.y * y);
    }
}

// half spectrum *= gain (already includes 1/(H*W) for the un-normalised C2R)
__global__ void pyr_gain_kernel(float2 *__restrict__ half, const float *__restrict__ gain, int N, long long per_image) {
    const long long total = (long long)N * per_image;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float g = gain[i % per_image];
        float2 z = half[i];
        z.x *= g; z.y *= g;
        half[i] = z;
    }
}

// =====================================================================================================================
// Fused level kernels on the LDS FFT engine (vfi_fft.h): the band spectra and the band coefficients never exist in HBM.
//   analysis  level k :  cols kernel : (tile of columns, image): running low-pass spectrum -> for each band
//                                      i * z * P_a[b] -> inverse column FFT -> T[n][b]   (+ the next level's low-pass;
//                                      level 0 also expands the R2C half spectrum and emits the high-pass half spectrum)
//                        rows kernel : rows of T -> inverse row FFT -> 1/(hw) -> atan2 / hypot -> the caller's planes
//   synthesis level k :  rows kernel : (phase, amplitude) rows -> A cos p, A sin p -> forward row FFT -> T
//                        cols kernel : (tile, image): sum_b (-i) * FFTcol(T[n][b]) * P_s[b] + embedded coarser level
// One intermediate (T: 8 bytes per coefficient written and read once) instead of the five full passes of the op-by-op
// form (band spectrum, 2 x 2 FFT passes, polar).
// =====================================================================================================================
using vfi::fft::Plan1D;
using vfi::fft::kThreads;
using vfi::fft::mul24;

struct LevelColsArgs {
    Plan1D ph;                    // column transform (length h)
    const float2 *src;            // R2C half spectrum of the images, N x H x (W/2+1)
    float2 *T;                    // N x NB x h x w
    const float *P;               // Q_k: [NB][h][w]
    int h, w, H, W, tile;
    int bands_per_pass;           // 1, 2 or 4: how many bands go through ONE transform call as extra lines (small levels)
};

// XCD-aware tile order: workgroup b runs on XCD b % 8 (own L2).  A column tile is only tile*8 bytes wide, so the tiles
// that share 128-byte lines are dealt to the SAME XCD back to back.
__device__ __forceinline__ int tile_of_block(int b, int ntiles) {
    const int per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// Generic-engine form of the analysis column pass (lengths the wave engine has no configuration for; the same
// arithmetic as vfi_pyrw_kernels.h: ana_cols_kernel).
// BLU: the transform length goes through Bluestein (chirp factors in the fill / drain).  A compile-time constant: with a
// run-time flag every fill / drain carries conditional chirp loads, and the compiler then waits for ALL outstanding
// memory operations (vmcnt(0): loads and stores share the counter) in front of every element -- also on the smooth
// levels, the large ones, that never load a chirp factor.
template <int NB, bool BLU>
__device__ __forceinline__ void level_cols_body(const LevelColsArgs &a, const int block_x, const int n) {
    using namespace vfi::fft;
    extern __shared__ float2 buf[];
    const int h = a.h, w = a.w, H = a.H, m = a.ph.m, tid = threadIdx.x, C = a.tile;
    const int ntiles = (w + C - 1) / C;
    const int tile = tile_of_block(block_x, ntiles);
    if (tile >= ntiles) return;
    const int pitch = ((padded_length(m) + 31) & ~31) + (C < 32 ? 32 / C : 1);
    const int v0 = tile * C, lines = w - v0 < C ? w - v0 : C;
    const int shift = __ffs(C) - 1, wh = a.W / 2 + 1;
    constexpr bool blu = BLU;
    const size_t hw = (size_t)h * w;
    const float2 *srcn = a.src + (size_t)n * H * wh;                      // (32-bit offsets inside a plane)
    auto load_z = [&](int u, int v) -> float2 {          // the level's window of the expanded half spectrum at (u, v)
        const int fy = signed_freq(u, h), fx = signed_freq(v, w);
        if (fx >= 0) return srcn[mul24(fy < 0 ? fy + H : fy, wh) + fx];
        float2 z = srcn[mul24(fy > 0 ? H - fy : -fy, wh) - fx];
        z.y = -z.y;
        return z;
    };
    float2 *twl = buf + (size_t)a.bands_per_pass * C * pitch;
    load_twiddles(twl, a.ph);
    // this thread's column of the tile (for_tile): validity, a safe column to read, offset of (row u0, that column)
    const int ccm = tid & (C - 1);
    const bool ccok = ccm < lines;
    const int vt = v0 + (ccok ? ccm : 0), tb = mul24(tid >> shift, w) + vt;
    const int BP = a.bands_per_pass;          // bands per transform call: band bb of a pass occupies lines [bb*C, bb*C + C)
#pragma unroll 1
    for (int b0 = 0; b0 < NB; b0 += BP) {
      for (int bb = 0; bb < BP; ++bb) {
        // fill: conj(i * z * Q[b]) (* chirp): the inverse transform runs as a forward one on conjugated data.  The tile
        // of z is re-read per band (L2) rather than kept in 70 registers across the stage calls.
        const int b = b0 + bb;
        float2 *bufb = buf + mul24(bb * C, pitch);
        const float *Pb = a.P + (size_t)b * hw;
        for_tile(h, shift, pitch,
                 [&](int u, int, int) {
                     Slot s;
                     s.z = load_z(u, vt);
                     s.s = Pb[mul24(u, w) + vt];
                     if (blu) s.c = a.ph.chirp[u];
                     return s;
                 },
                 [&](int, int, int, int idx, const Slot &s) {
                     if (ccok)      // * i : (re, im) -> (-im, re)
                         bufb[idx] = load_value<true>(make_float2(-(s.z.y * s.s), s.z.x * s.s), s.c, blu);
                 });
        if (blu) {
            const int totz = (m - h) * C;
            for (int e = tid; e < totz; e += kThreads) {
                const int u = h + (e >> shift), cc = e & (C - 1);
                if (cc < lines) bufb[cc * pitch + phys(u)] = make_float2(0.0f, 0.0f);
            }
        }
      }
        lds_barrier();
        // (a partial tile leaves unused lines between the bands of a pass: they are transformed too and never stored)
        fft_lines(buf, BP > 1 ? BP * C : lines, pitch, a.ph, twl);
      for (int bb = 0; bb < BP; ++bb) {
        const float2 *bufb = buf + mul24(bb * C, pitch);
        const __amdgpu_buffer_rsrc_t Tr = plane_rsrc(a.T + ((size_t)n * NB + b0 + bb) * hw, hw * sizeof(float2));
        const unsigned tlane = ccok ? (unsigned)tb * 8u : 0xffffffffu;
        for_tile(h, shift, pitch,
                 [&](int u, int, int) {
                     Slot s;
                     if (blu) s.c = a.ph.chirp[u];
                     return s;
                 },
                 [&](int, int, int, int idx, const Slot &s) { return store_value<true>(bufb[idx], s.c, blu); },
                 [&](int, int uq, int, const float2 &o) { plane_store(Tr, tlane, mul24(uq, w) * 8, o); });
      }
        lds_barrier();
    }
}

struct RowsPolarArgs {
    Plan1D pw;                    // row transform (length w)
    float2 *T;                    // planes x h x w
    float *phase, *amp;           // caller's planes (PlaneMap)
    PlaneMap pm;
    long long rows;               // planes * h
    int h, lines;
    float inv_hw, phase_scale;
    unsigned *amp_max;            // analysis only, optional: [groups] bit patterns of the largest amplitude per image group
    int groups;
};

template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_level_cols_kernel(const LevelColsArgs a) {
    level_cols_body<NB, BLU>(a, blockIdx.x, blockIdx.y);
}
// Several SMALL levels in one launch: below ~135 x 240 a level is a handful of workgroups and a launch costs 16-30 us
// whatever it holds, and the levels of an analysis do not depend on each other.  blockIdx.x runs over the levels' tiles
// (first[i] = first block of entry i), every entry has its own region of T.
constexpr int kMaxMulti = 12;
struct MultiColsArgs {
    LevelColsArgs lev[kMaxMulti];
    int first[kMaxMulti + 1];
    int count;
};
template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_multi_level_cols_kernel(const MultiColsArgs ma) {
    int i = 0;
    while (i + 1 < ma.count && (int)blockIdx.x >= ma.first[i + 1]) ++i;      // (uniform)
    level_cols_body<NB, BLU>(ma.lev[i], (int)blockIdx.x - ma.first[i], blockIdx.y);
}

// per-line output base (in elements of h*w planes): plane map applied once per row, not per element
__device__ __forceinline__ void line_bases(size_t *base, int lines, long long row0, int h, int w, const PlaneMap &pm, int NB) {
    for (int l = threadIdx.x; l < lines; l += kThreads) {
        const long long g = row0 + l;
        const int plane = (int)(g / h), y = (int)(g - (long long)plane * h), n = plane / NB, b = plane - n * NB;
        base[l] = (size_t)(pm.idx[n] + b * pm.band_stride) * h * w + (size_t)y * w;
    }
}

__device__ __forceinline__ void zero_row_padding(float2 *buf, int lines, int pitch, int w, int m) {
    using namespace vfi::fft;
    const int pad = m - w, totz = lines * pad;
    const float inv_pad = 1.0f / (float)pad;
    for (int e = threadIdx.x; e < totz; e += kThreads) {
        const int l = fast_div(e, inv_pad), j = w + e - l * pad;
        buf[l * pitch + phys(j)] = make_float2(0.0f, 0.0f);
    }
}

// rows of T -> inverse row FFT -> (phase, amplitude) or the complex coefficient (coeff_to_values, src/train/pyramid.py:63-69)
template <int NB, bool BLU>
__device__ __forceinline__ void rows_polar_body(const RowsPolarArgs &a, const int block_x) {
    using namespace vfi::fft;
    extern __shared__ float2 buf[];
    const int w = a.pw.n, m = a.pw.m, pitch = padded_length(m);
    const long long row0 = (long long)block_x * a.lines;
    const int lines = (int)(a.rows - row0 < a.lines ? a.rows - row0 : a.lines);
    float2 *twl = buf + (size_t)a.lines * pitch;
    size_t *base = reinterpret_cast<size_t *>(twl + a.pw.tw_len);
    load_twiddles(twl, a.pw);
    line_bases(base, lines, row0, a.h, w, a.pm, NB);
    const int total = lines * w;
    const float inv_w = 1.0f / (float)w;
    constexpr bool blu = BLU;
    const float2 *Trow = a.T + row0 * w;
    auto fill_load = [&](int l, int j) {
        Slot s;
        s.z = Trow[mul24(l, w) + j];
        if (blu) s.c = a.pw.chirp[j];
        return s;
    };
    auto fill_use = [&](int idx, const Slot &s) { buf[idx] = load_value<true>(s.z, s.c, blu); };
    auto drain_load = [&](int j) {
        Slot s;
        if (blu) s.c = a.pw.chirp[j];
        return s;
    };
    // per-group maximum of the amplitudes this thread writes (groups <= 4: one running value per group)
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool want_max = a.amp_max != nullptr;
    auto drain_use = [&](int l, int j, int idx, const Slot &s) {
        const float2 z = store_value<true>(buf[idx], s.c, blu);
        const float re = z.x * a.inv_hw, im = z.y * a.inv_hw;
        const size_t o = base[l] + j;
        if (a.pm.complex_coeff) {
            reinterpret_cast<float2 *>(a.phase)[o] = make_float2(re, im);
        } else {
            const float am = sqrtf(re * re + im * im);
            a.phase[o] = atan2f(im, re) * a.phase_scale;
            a.amp[o] = am;
            if (want_max) {
                const int g = (int)(((row0 + l) / a.h) / NB) % a.groups;      // (uniform per line)
#pragma unroll
                for (int t = 0; t < 4; ++t) gmax[t] = t == g ? fmaxf(gmax[t], am) : gmax[t];
            }
        }
    };
    const bool wide = w >= kThreads;          // long rows: the structured walk (no per-element division)
    if (wide)
        for_rows(lines, w, pitch, [&](int l, int j, int) { return fill_load(l, j); },
                 [&](int, int, int, int idx, const Slot &s) { fill_use(idx, s); });
    else
        for_slots(total, [&](int e) { const int l = fast_div(e, inv_w); return fill_load(l, e - mul24(l, w)); },
                  [&](int e, const Slot &s) { const int l = fast_div(e, inv_w); fill_use(mul24(l, pitch) + phys(e - mul24(l, w)), s); });
    if (blu) zero_row_padding(buf, lines, pitch, w, m);
    lds_barrier();
    fft_lines(buf, lines, pitch, a.pw, twl);
    if (wide)
        for_rows(lines, w, pitch, [&](int, int j, int) { return drain_load(j); },
                 [&](int l, int j, int, int idx, const Slot &s) { drain_use(l, j, idx, s); });
    else
        for_slots(total, [&](int e) { return drain_load(e - mul24(fast_div(e, inv_w), w)); },
                  [&](int e, const Slot &s) {
                      const int l = fast_div(e, inv_w), j = e - mul24(l, w);
                      drain_use(l, j, mul24(l, pitch) + phys(j), s);
                  });
    if (want_max) {        // 64-lane butterfly, then one atomic per wave and group (amplitudes are >= 0: their bit patterns order like the values)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float m = gmax[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if ((threadIdx.x & 63) == 0 && t < a.groups && m > 0.0f) atomicMax(a.amp_max + t, __float_as_uint(m));
        }
    }
}

template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_rows_polar_kernel(const RowsPolarArgs a) {
    rows_polar_body<NB, BLU>(a, blockIdx.x);
}
struct MultiRowsArgs {
    RowsPolarArgs lev[kMaxMulti];
    int first[kMaxMulti + 1];
    int count;
};
template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_multi_rows_polar_kernel(const MultiRowsArgs ma) {
    int i = 0;
    while (i + 1 < ma.count && (int)blockIdx.x >= ma.first[i + 1]) ++i;      // (uniform)
    rows_polar_body<NB, BLU>(ma.lev[i], (int)blockIdx.x - ma.first[i]);
}

__global__ void pyr_amp_max_finish_kernel(const unsigned *__restrict__ bits, float *__restrict__ out, int count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = __uint_as_float(bits[i]) + eps;
}

// (phase, amplitude) rows -> complex -> forward row FFT -> T (values_to_coeff, src/train/pyramid.py:99-107, + the row half of
// reconstruct's fft2)
template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_rows_from_polar_kernel(const RowsPolarArgs a) {
    using namespace vfi::fft;
    extern __shared__ float2 buf[];
    const int w = a.pw.n, m = a.pw.m, pitch = padded_length(m);
    const long long row0 = (long long)blockIdx.x * a.lines;
    const int lines = (int)(a.rows - row0 < a.lines ? a.rows - row0 : a.lines);
    float2 *twl = buf + (size_t)a.lines * pitch;
    size_t *base = reinterpret_cast<size_t *>(twl + a.pw.tw_len);
    load_twiddles(twl, a.pw);
    line_bases(base, lines, row0, a.h, w, a.pm, NB);
    lds_barrier();
    const int total = lines * w;
    const float inv_w = 1.0f / (float)w;
    constexpr bool blu = BLU;
    auto fill_load = [&](int l, int j) {
        const size_t o = base[l] + j;
        Slot s;
        if (a.pm.complex_coeff) s.z = reinterpret_cast<const float2 *>(a.phase)[o];
        else s.z = make_float2(a.phase[o], a.amp[o]);
        if (blu) s.c = a.pw.chirp[j];
        return s;
    };
    auto fill_use = [&](int idx, const Slot &s) {
        float2 x = s.z;
        if (!a.pm.complex_coeff) {
            float sn, cs;
            sincosf(s.z.x, &sn, &cs);
            x = make_float2(cs * s.z.y, sn * s.z.y);
        }
        buf[idx] = load_value<false>(x, s.c, blu);
    };
    float2 *Trow = a.T + row0 * w;
    auto drain_load = [&](int j) {
        Slot s;
        if (blu) s.c = a.pw.chirp[j];
        return s;
    };
    auto drain_use = [&](int l, int j, int idx, const Slot &s) { Trow[mul24(l, w) + j] = store_value<false>(buf[idx], s.c, blu); };
    const bool wide = w >= kThreads;
    if (wide)
        for_rows(lines, w, pitch, [&](int l, int j, int) { return fill_load(l, j); },
                 [&](int, int, int, int idx, const Slot &s) { fill_use(idx, s); });
    else
        for_slots(total, [&](int e) { const int l = fast_div(e, inv_w); return fill_load(l, e - mul24(l, w)); },
                  [&](int e, const Slot &s) { const int l = fast_div(e, inv_w); fill_use(mul24(l, pitch) + phys(e - mul24(l, w)), s); });
    if (blu) zero_row_padding(buf, lines, pitch, w, m);
    lds_barrier();
    fft_lines(buf, lines, pitch, a.pw, twl);
    if (wide)
        for_rows(lines, w, pitch, [&](int, int j, int) { return drain_load(j); },
                 [&](int l, int j, int, int idx, const Slot &s) { drain_use(l, j, idx, s); });
    else
        for_slots(total, [&](int e) { return drain_load(e - mul24(fast_div(e, inv_w), w)); },
                  [&](int e, const Slot &s) {
                      const int l = fast_div(e, inv_w), j = e - mul24(l, w);
                      drain_use(l, j, mul24(l, pitch) + phys(j), s);
                  });
}

struct CombineColsArgs {
    Plan1D ph;
    const float2 *T;              // N x NB x h x w (row-transformed bands)
    const float2 *res;            // N x h2 x w2 (coarser level's spectrum; may be null)
    float2 *cur;                  // N x h x w
    const float *P, *lomask;
    int h, w, h2, w2, tile;
    int bands_per_pass;
};

// cur = sum_b (-i) * FFTcol(T_b) * P_s[b]  +  embed(res * lomask)     (reconstruct: orientdft + resdft)
template <int NB, bool BLU>
__global__ __launch_bounds__(kThreads, kThreads / 128) void pyr_combine_cols_kernel(const CombineColsArgs a) {
    using namespace vfi::fft;
    extern __shared__ float2 buf[];
    const int h = a.h, w = a.w, m = a.ph.m, tid = threadIdx.x, C = a.tile, n = blockIdx.y;
    const int ntiles = (w + C - 1) / C;
    const int tile = tile_of_block(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int pitch = ((padded_length(m) + 31) & ~31) + (C < 32 ? 32 / C : 1);
    const int v0 = tile * C, lines = w - v0 < C ? w - v0 : C;
    const int shift = __ffs(C) - 1;
    constexpr bool blu = BLU;
    const size_t hw = (size_t)h * w;
    float2 *cur = a.cur + (size_t)n * hw;
    const int BP = a.bands_per_pass;
    float2 *twl = buf + (size_t)BP * C * pitch;
    load_twiddles(twl, a.ph);
    const int ccm = tid & (C - 1);
    const bool ccok = ccm < lines;
    const int vt = v0 + (ccok ? ccm : 0), tb = mul24(tid >> shift, w) + vt;      // (for_tile: this thread's column)
#pragma unroll 1
    for (int b0 = 0; b0 < NB; b0 += BP) {
      for (int bb = 0; bb < BP; ++bb) {
        const float2 *Tb = a.T + ((size_t)n * NB + b0 + bb) * hw;
        float2 *bufb = buf + mul24(bb * C, pitch);
        for_tile(h, shift, pitch,
                 [&](int u, int uq, int) {
                     Slot s;
                     s.z = Tb[tb + mul24(uq, w)];
                     if (blu) s.c = a.ph.chirp[u];
                     return s;
                 },
                 [&](int, int, int, int idx, const Slot &s) {
                     if (ccok) bufb[idx] = load_value<false>(s.z, s.c, blu);
                 });
        if (blu) {
            const int totz = (m - h) * C;
            for (int e = tid; e < totz; e += kThreads) {
                const int u = h + (e >> shift), cc = e & (C - 1);
                if (cc < lines) bufb[cc * pitch + phys(u)] = make_float2(0.0f, 0.0f);
            }
        }
      }
        lds_barrier();
        fft_lines(buf, BP > 1 ? BP * C : lines, pitch, a.ph, twl);
      for (int bb = 0; bb < BP; ++bb) {
        const int b = b0 + bb;
        const float2 *bufb = buf + mul24(bb * C, pitch);
        const float *Pb = a.P + (size_t)b * hw;
        // drain: band 0 starts the sum from the embedded coarser level, bands 1.. add to what this thread wrote for the
        // previous band (its own elements: still in L2)
        for_tile(h, shift, pitch,
                 [&](int u, int uq, int) {
                      const int v = vt, o = tb + mul24(uq, w);
                      Slot s;
                      if (blu) s.c = a.ph.chirp[u];
                      s.s = Pb[o];
                      if (b == 0) {
                          s.z = make_float2(0.0f, 0.0f);
                          if (a.res) {
                              const int h2 = a.h2, w2 = a.w2, fy = signed_freq(u, h), fx = signed_freq(v, w);
                              if (fy >= -(h2 / 2) && fy <= h2 - h2 / 2 - 1 && fx >= -(w2 / 2) && fx <= w2 - w2 / 2 - 1) {
                                  const int u2 = fy < 0 ? fy + h2 : fy, v2 = fx < 0 ? fx + w2 : fx;
                                  const float2 r = a.res[((size_t)n * h2 + u2) * w2 + v2];
                                  const float lom = a.lomask[(size_t)u2 * w2 + v2];
                                  s.z = make_float2(r.x * lom, r.y * lom);
                              }
                          }
                      } else {
                          s.z = cur[o];
                      }
                      return s;
                 },
                 [&](int, int uq, int, int idx, const Slot &s) {
                      if (ccok) {
                          const float2 z = store_value<false>(bufb[idx], s.c, blu);
                          // * (-i) : (re, im) -> (im, -re)
                          cur[tb + mul24(uq, w)] = make_float2(s.z.x + z.y * s.s, s.z.y - z.x * s.s);
                      }
                 });
      }
        lds_barrier();
    }
}

template <auto kernel>
void allow_big_lds() {      // > 64 KiB of dynamic LDS needs the attribute, per KERNEL (not per signature) and device (idempotent)
    static bool done[vfi::kMaxDevices] = {};
    bool &d = done[vfi::current_device()];
    if (!d) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(vfi::fft::kLdsElemsMax * sizeof(float2) + 8192));
        d = true;
    }
}

// Column tile width and bands per transform call of a fused level kernel: large levels take the widest tile that fits
// and one band per call; small levels (latency-bound: few workgroups, each a chain of short phases) put 2 or all 4 bands
// through one call as extra lines, with tiles of >= 8 columns.
void level_tiling(const vfi::fft::Plan1D &ph, int w, int *tile, int *bands) {
    using namespace vfi::fft;
    *tile = cols_per_group(ph, w);
    *bands = 1;
    for (int bp : {4, 2}) {
        int c = *tile;
        while (c > 8 && (bp * c * ph.m > max_elems(ph) || bp * c * col_pitch(ph, c) + ph.tw_len > kLdsElems)) c /= 2;
        if (c >= 8 || c == *tile) {
            if (bp * c * ph.m <= max_elems(ph) && bp * c * col_pitch(ph, c) + ph.tw_len <= kLdsElems) {
                *tile = c;
                *bands = bp;
                return;
            }
        }
    }
}
size_t level_lds_bytes(const vfi::fft::Plan1D &ph, int tile, int bands) {
    return ((size_t)bands * tile * vfi::fft::col_pitch(ph, tile) + ph.tw_len) * sizeof(float2);
}

// ---- 2-D transforms = a row pass and a column pass of the LDS engine (vfi_fft.h / vfi_fft.hip) ---------------------------
int get_fft(vfi_pyr_plan *p, int n, vfi::fft::Plan1D *out) {
    auto it = p->fft1d.find(n);
    if (it == p->fft1d.end()) {
        vfi::fft::Plan1D pl;
        const int rc = vfi::fft::make_plan(n, &pl, [](void *ctx, void *dev) { static_cast<vfi_pyr_plan *>(ctx)->allocs.push_back(dev); }, p);
        if (rc) return rc;
        it = p->fft1d.emplace(n, pl).first;
    }
    *out = it->second;
    return VFI_OK;
}

// ---- wave engine (vfi_wfft.h) selection: engine length of a pass or 0, and its stage twiddles (built once per plan) ----
enum WavePass { kWaveRows = 0, kWaveAnaCols = 1, kWaveSynCols = 2 };
int wave_twiddles(vfi_pyr_plan *p, WavePass kind, int M, const float2 **out) {
    auto &cache = p->wave_tw[kind];
    auto it = cache.find(M);
    if (it == cache.end()) {
        std::vector<float2> tw(4096);
        const int cap = (int)tw.size();
        const int cnt = kind == kWaveRows ? vfi::pyrw::rows_twiddles(M, tw.data(), cap)
                                          : (kind == kWaveAnaCols ? vfi::pyrw::cols_twiddles(M, tw.data(), cap) : vfi::pyrw::syn_twiddles(M, tw.data(), cap));
        if (cnt < 0) return vfi::fail(VFI_ERR_UNSUPPORTED, "pyramid: no wave-engine twiddles for length %d", M);
        tw.resize(cnt > 0 ? cnt : 1);
        float2 *dev = nullptr;
        const int rc = dev_upload(p, tw, &dev);
        if (rc) return rc;
        it = cache.emplace(M, dev).first;
    }
    *out = it->second;
    return VFI_OK;
}
// tables of a pass on the wave engine; tb->M == 0 when the engine has no configuration for this length
int wave_tables(vfi_pyr_plan *p, WavePass kind, const vfi::fft::Plan1D &pl, vfi::pyrw::Tables *tb) {
    // A/B switch: VFI_PYR_WAVE = bit mask of the passes that may run on the wave engine (1 rows, 2 analysis columns and
    // plain column passes, 4 synthesis columns; default all, 0 = the generic LDS engine everywhere)
    static const int allowed = [] { const char *e = getenv("VFI_PYR_WAVE"); return e ? atoi(e) : 7; }();
    *tb = vfi::pyrw::Tables{};
    const bool off = !((allowed >> (int)kind) & 1);
    const int M = off ? 0 : (kind != kWaveRows ? vfi::pyrw::cols_engine_length(pl.n, pl.bluestein ? pl.m : 0) : vfi::pyrw::rows_engine_length(pl.n, pl.bluestein ? pl.m : 0));
    if (!M) return VFI_OK;
    const int rc = wave_twiddles(p, kind, M, &tb->tw);
    if (rc) return rc;
    tb->chirp = pl.chirp; tb->bfilt = pl.bfilt; tb->M = M; tb->n = pl.n; tb->bluestein = pl.bluestein;
    return VFI_OK;
}
inline int round_up16(int x) { return (x + 15) & ~15; }

// debugging aid (VFI_PYR_CHECK=1): wait for the stream and count the NaNs of a device array
void debug_scan(const void *dev, size_t floats, hipStream_t s, const char *what, int level) {
    static const bool on = getenv("VFI_PYR_CHECK") != nullptr;
    if (!on) return;
    (void)hipStreamSynchronize(s);
    std::vector<float> host(floats);
    (void)hipMemcpy(host.data(), dev, floats * sizeof(float), hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < floats; ++i)
        if (host[i] != host[i]) { if (!bad) first = i; ++bad; }
    if (bad) fprintf(stderr, "[vfi_pyr check] %s level %d: %zu NaNs of %zu floats, first at %zu\n", what, level, bad, floats, first);
}

// one row / column pass of a plain 2-D transform: on the wave engine where it has a configuration for the length,
// otherwise on the generic LDS engine
int pass_rows(vfi_pyr_plan *p, const vfi::fft::Plan1D &pw, const void *src, void *dst, long long rows, int src_pitch, int dst_pitch,
              vfi::fft::RowLoad load, vfi::fft::RowStore store, bool inverse, hipStream_t s) {
    using namespace vfi::fft;
    vfi::pyrw::Tables tb;
    int rc = wave_tables(p, kWaveRows, pw, &tb);
    if (rc) return rc;
    if (tb.M && rows < (1LL << 31)) {
        vfi::pyrw::GenRowsArgs a{tb, src, dst, (int)rows, src_pitch, dst_pitch, 1.0f};
        return vfi::pyrw::launch_gen_rows(a, (int)load, (int)store, inverse, s);      // (RowLoad / RowStore == GenRowKind values)
    }
    RowArgs r{pw, src, dst, rows, src_pitch, dst_pitch, rows_per_group(pw, rows), 1.0f};
    return launch_rows(r, load, store, inverse, s);
}
int pass_cols(vfi_pyr_plan *p, const vfi::fft::Plan1D &ph, float2 *data, int planes, int cols, int ld, bool inverse, hipStream_t s) {
    using namespace vfi::fft;
    vfi::pyrw::Tables tb;
    int rc = wave_tables(p, kWaveAnaCols, ph, &tb);      // (the plain column pass runs with the analysis column geometry)
    if (rc) return rc;
    if (tb.M) {
        vfi::pyrw::GenColsArgs a{tb, data, planes, cols, ld, 1.0f};
        return vfi::pyrw::launch_gen_cols(a, inverse, s);
    }
    ColArgs c{ph, data, planes, cols, ld, cols_per_group(ph, cols), 1.0f};
    return launch_cols(c, inverse, s);
}

// in-place complex 2-D transform of `planes` dense h x w arrays (un-normalised)
int fft2d_c2c(vfi_pyr_plan *p, float2 *data, int planes, int h, int w, bool inverse, hipStream_t s) {
    using namespace vfi::fft;
    Plan1D ph, pw;
    int rc;
    if ((rc = get_fft(p, h, &ph)) || (rc = get_fft(p, w, &pw))) return rc;
    if ((rc = pass_cols(p, ph, data, planes, w, w, inverse, s))) return rc;
    return pass_rows(p, pw, data, data, (long long)planes * h, w, w, kLoadComplex, kStoreComplex, inverse, s);
}
// real H x W images -> half spectra N x H x (W/2+1)
int fft2d_r2c(vfi_pyr_plan *p, const float *img, float2 *half, int N, hipStream_t s) {
    using namespace vfi::fft;
    Plan1D ph, pw;
    int rc;
    if ((rc = get_fft(p, p->H, &ph)) || (rc = get_fft(p, p->W, &pw))) return rc;
    const int wh = p->W / 2 + 1;
    if ((rc = pass_rows(p, pw, img, half, (long long)N * p->H, p->W, wh, kLoadReal, kStoreHalf, false, s))) return rc;
    return pass_cols(p, ph, half, N, wh, wh, false, s);
}
// half spectra (destroyed) -> real images, un-normalised inverse
int fft2d_c2r(vfi_pyr_plan *p, float2 *half, float *out, int N, hipStream_t s) {
    using namespace vfi::fft;
    Plan1D ph, pw;
    int rc;
    if ((rc = get_fft(p, p->H, &ph)) || (rc = get_fft(p, p->W, &pw))) return rc;
    const int wh = p->W / 2 + 1;
    if ((rc = pass_cols(p, ph, half, N, wh, wh, true, s))) return rc;
    return pass_rows(p, pw, half, out, (long long)N * p->H, wh, p->W, kLoadHalf, kStoreReal, true, s);
}

PlaneMap make_map(const int *plane_index, int level, int N, int nb, int flags) {
    PlaneMap pm;
    const bool band_major = flags & VFI_PYR_BAND_MAJOR;
    for (int d = 0; d < kMaxImages; ++d)
        pm.idx[d] = d < N ? (plane_index ? plane_index[level * N + d] : (band_major ? d : d * nb)) : 0;
    pm.band_stride = band_major ? N : 1;
    pm.complex_coeff = (flags & VFI_PYR_COMPLEX_COEFF) ? 1 : 0;
    return pm;
}

inline int blocks_1d(long long n) { long long b = (n + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int vfi_pyr_plan_create(int H, int W, int height, int nbands, double scale_factor, int max_images,
                                   vfi_pyr_plan **out) {
    VFI_REQUIRE(out, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_create: null out");
    *out = nullptr;
    VFI_REQUIRE(H >= 4 && W >= 4 && height >= 3 && height - 2 <= kMaxLevels, VFI_ERR_INVALID_ARG,
                "vfi_pyr_plan_create: bad size %dx%d height %d", H, W, height);
    VFI_REQUIRE(nbands == 4, VFI_ERR_UNSUPPORTED, "vfi_pyr_plan_create: nbands=%d (the path uses 4)", nbands);
    VFI_REQUIRE(scale_factor > 1.0 && scale_factor <= 2.0, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_create: scale_factor %g", scale_factor);
    VFI_REQUIRE(max_images >= 1 && max_images <= kMaxImages, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_create: max_images %d", max_images);
    vfi_pyr_plan *p = new (std::nothrow) vfi_pyr_plan();
    VFI_REQUIRE(p, VFI_ERR_NOMEM, "vfi_pyr_plan_create: host allocation");
    p->H = H; p->W = W; p->height = height; p->nbands = nbands; p->nlev = height - 2; p->scale = scale_factor;
    p->max_images = max_images;
    p->lev.resize(p->nlev);
    for (int k = 0; k < p->nlev; ++k) { p->lev[k].h = level_size(H, scale_factor, k); p->lev[k].w = level_size(W, scale_factor, k); }
    p->hl = level_size(H, scale_factor, p->nlev);
    p->wl = level_size(W, scale_factor, p->nlev);
    int rc = VFI_OK;
    if (p->hl < 2 || p->wl < 2) rc = vfi::fail(VFI_ERR_SHAPE, "vfi_pyr_plan_create: height %d too large for %dx%d", height, H, W);
    if (!rc) rc = build_tables(p);
    {   // FFT tables of every length the plan can meet (so that no later call allocates)
        vfi::fft::Plan1D tmp;
        for (int k = 0; k <= p->nlev && !rc; ++k) {
            vfi::pyrw::Tables tb;
            rc = get_fft(p, k < p->nlev ? p->lev[k].h : p->hl, &tmp);
            if (!rc && k < p->nlev) rc = wave_tables(p, kWaveAnaCols, tmp, &tb);
            if (!rc && k < p->nlev) rc = wave_tables(p, kWaveSynCols, tmp, &tb);
            if (!rc) rc = get_fft(p, k < p->nlev ? p->lev[k].w : p->wl, &tmp);
            if (!rc && k < p->nlev) rc = wave_tables(p, kWaveRows, tmp, &tb);
        }
    }
    p->tpitch_max = round_up16(W);
    const size_t N = max_images, HW = (size_t)H * p->tpitch_max, half = (size_t)H * (W / 2 + 1);
    if (!rc) rc = dev_alloc(p, (void **)&p->half0, N * half * sizeof(float2));
    if (!rc) rc = dev_alloc(p, (void **)&p->half_hi, N * half * sizeof(float2));
    if (!rc) rc = dev_alloc(p, (void **)&p->bands, N * nbands * HW * sizeof(float2));
    if (!rc) rc = dev_alloc(p, (void **)&p->lod[0], N * HW * sizeof(float2));
    if (!rc) rc = dev_alloc(p, (void **)&p->lod[1], N * HW * sizeof(float2));
    if (!rc) rc = dev_alloc(p, (void **)&p->amp_bits, kMaxLevels * 4 * sizeof(unsigned));
    if (!rc && getenv("VFI_PYR_POISON")) {      // debugging aid: NaN-fill the workspace, so a read of anything not yet written shows
        (void)hipMemset(p->half0, 0xff, N * half * sizeof(float2));
        (void)hipMemset(p->half_hi, 0xff, N * half * sizeof(float2));
        (void)hipMemset(p->bands, 0xff, N * nbands * HW * sizeof(float2));
        (void)hipMemset(p->lod[0], 0xff, N * HW * sizeof(float2));
        (void)hipMemset(p->lod[1], 0xff, N * HW * sizeof(float2));
    }
    if (rc) {
        if (rc == VFI_ERR_NOMEM) vfi::set_error("vfi_pyr_plan_create: device allocation failed");
        vfi_pyr_plan_destroy(p);
        return rc;
    }
    *out = p;
    return VFI_OK;
}

extern "C" int vfi_pyr_plan_prepare_filter(vfi_pyr_plan *p, unsigned long long level_mask, int keep_high, int keep_low,
                                           int *filter_id) {
    VFI_REQUIRE(p && filter_id, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_prepare_filter: null pointer");
    const int H = p->H, W = p->W, wh = W / 2 + 1;
    const double ls = std::log2(p->scale);
    std::vector<float> g((size_t)H * wh);
    for (int u = 0; u < H; ++u)
        for (int v = 0; v < wh; ++v) {
            const int fy = u <= H - H / 2 - 1 ? u : u - H, fx = v;          // signed frequencies (v < wh: non-negative)
            const double lr = p->log_rad[(size_t)shifted_of(u, H) * W + shifted_of(v, W)];
            const double lo0 = interp(lr, p->xr0, p->yir), hi0 = interp(lr, p->xr0, p->yr);
            double lowchain = 1.0, acc = 0.0;       // prod_{j<k} lomask_j^2 on the running window
            std::vector<double> xr = p->xr0;
            for (int k = 0; k <= p->nlev; ++k) {
                const int h = k < p->nlev ? p->lev[k].h : p->hl, w = k < p->nlev ? p->lev[k].w : p->wl;
                const bool inside = fy >= -(h / 2) && fy <= h - h / 2 - 1 && fx >= -(w / 2) && fx <= w - w / 2 - 1;
                if (!inside) { lowchain = 0.0; break; }
                if (k == p->nlev) break;
                for (auto &x : xr) x -= ls;
                const double hm = interp(lr, xr, p->yr), lm = interp(lr, xr, p->yir);
                if ((level_mask >> k) & 1ull) acc += lowchain * hm * hm;
                lowchain *= lm * lm;
            }
            if (keep_low) acc += lowchain;
            const double gain = (keep_high ? hi0 * hi0 : 0.0) + lo0 * lo0 * acc;
            g[(size_t)u * wh + v] = (float)(gain / ((double)H * W));
        }
    float *dev = nullptr;
    int rc = dev_upload(p, g, &dev);
    if (rc) return vfi::fail(rc, "vfi_pyr_plan_prepare_filter: device allocation / upload failed");
    p->filters.push_back(dev);
    *filter_id = (int)p->filters.size() - 1;
    return VFI_OK;
}

extern "C" int vfi_pyr_apply_filter(vfi_pyr_plan *p, int filter_id, const float *img, int N, float *out, vfi_stream_t stream) {
    VFI_REQUIRE(p && img && out, VFI_ERR_INVALID_ARG, "vfi_pyr_apply_filter: null pointer");
    VFI_REQUIRE(filter_id >= 0 && filter_id < (int)p->filters.size(), VFI_ERR_INVALID_ARG, "vfi_pyr_apply_filter: bad filter id %d", filter_id);
    VFI_REQUIRE(N >= 1 && N <= p->max_images, VFI_ERR_INVALID_ARG, "vfi_pyr_apply_filter: N=%d (plan max %d)", N, p->max_images);
    hipStream_t s = vfi::as_stream(stream);
    int rc;
    if ((rc = fft2d_r2c(p, img, p->half0, N, s))) return rc;
    const long long per = (long long)p->H * (p->W / 2 + 1);
    hipLaunchKernelGGL(pyr_gain_kernel, dim3(blocks_1d(per * N)), dim3(256), 0, s, p->half0, p->filters[filter_id], N, per);
    if ((rc = fft2d_c2r(p, p->half0, out, N, s))) return rc;
    return vfi::check_launch("vfi_pyr_apply_filter");
}

extern "C" int vfi_pyr_apply_filter_pair(vfi_pyr_plan *p, int filter_a, const float *img_a, int filter_b, const float *img_b,
                                         int N, float *out, vfi_stream_t stream) {
    VFI_REQUIRE(p && img_a && img_b && out, VFI_ERR_INVALID_ARG, "vfi_pyr_apply_filter_pair: null pointer");
    const int nf = (int)p->filters.size();
    VFI_REQUIRE(filter_a >= 0 && filter_a < nf && filter_b >= 0 && filter_b < nf, VFI_ERR_INVALID_ARG,
                "vfi_pyr_apply_filter_pair: bad filter ids %d, %d", filter_a, filter_b);
    VFI_REQUIRE(N >= 1 && 2 * N <= p->max_images, VFI_ERR_INVALID_ARG, "vfi_pyr_apply_filter_pair: N=%d (plan max %d images in all)",
                N, p->max_images);
    hipStream_t s = vfi::as_stream(stream);
    int rc;
    if ((rc = fft2d_r2c(p, img_a, p->half0, N, s)) || (rc = fft2d_r2c(p, img_b, p->half_hi, N, s))) return rc;
    const long long per = (long long)p->H * (p->W / 2 + 1);
    hipLaunchKernelGGL(pyr_gain_pair_kernel, dim3(blocks_1d(per * N)), dim3(256), 0, s, p->half0, p->half_hi, p->filters[filter_a],
                       p->filters[filter_b], N, per);
    if ((rc = fft2d_c2r(p, p->half0, out, N, s))) return rc;
    return vfi::check_launch("vfi_pyr_apply_filter_pair");
}

extern "C" int vfi_pyr_plan_destroy(vfi_pyr_plan *p) {
    if (!p) return VFI_OK;
    for (void *d : p->allocs) (void)hipFree(d);
    delete p;
    return VFI_OK;
}

extern "C" int vfi_pyr_plan_level_size(const vfi_pyr_plan *p, int level, int *h, int *w) {
    VFI_REQUIRE(p && h && w, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_level_size: null pointer");
    VFI_REQUIRE(level >= 0 && level <= p->nlev, VFI_ERR_INVALID_ARG, "vfi_pyr_plan_level_size: level %d", level);
    *h = level < p->nlev ? p->lev[level].h : p->hl;
    *w = level < p->nlev ? p->lev[level].w : p->wl;
    return VFI_OK;
}

static int pyr_analyze_impl(vfi_pyr_plan *p, const float *img, int N, float *high, float *const *phase,
                            float *const *amp, const int *plane_index, float *low, float phase_scale,
                            unsigned long long level_mask, int flags, float *amp_max, int groups, float eps, vfi_stream_t stream) {
    VFI_REQUIRE(p && img, VFI_ERR_INVALID_ARG, "vfi_pyr_analyze: null pointer");
    VFI_REQUIRE(!amp_max || (groups >= 1 && groups <= 4 && !(flags & VFI_PYR_COMPLEX_COEFF)), VFI_ERR_INVALID_ARG,
                "vfi_pyr_analyze_max: groups must be 1..4 and the outputs (phase, amplitude)");
    VFI_REQUIRE(N >= 1 && N <= p->max_images, VFI_ERR_INVALID_ARG, "vfi_pyr_analyze: N=%d (plan max %d)", N, p->max_images);
    VFI_REQUIRE((phase && (amp || (flags & VFI_PYR_COMPLEX_COEFF))) || level_mask == 0, VFI_ERR_INVALID_ARG,
                "vfi_pyr_analyze: null phase/amp tables");
    hipStream_t s = vfi::as_stream(stream);
    const int H = p->H, W = p->W, nb = p->nbands;
    int rc;
    if (amp_max && hipMemsetAsync(p->amp_bits, 0, sizeof(unsigned) * p->nlev * groups, s) != hipSuccess)
        return vfi::fail(VFI_ERR_LAUNCH, "vfi_pyr_analyze_max: memset");
    if ((rc = fft2d_r2c(p, img, p->half0, N, s))) return rc;
    debug_scan(p->half0, (size_t)N * H * (W / 2 + 1) * 2, s, "half spectrum", -1);
    // ---- the small levels (<= 40 k coefficients per band: from 135 x 240 down at 1080p) as FOUR launches on the generic LDS
    // engine instead of two per level: column passes of the smooth / Bluestein heights, row passes of the smooth / Bluestein
    // widths; every level has its own region of T.  VFI_PYR_MULTI=0: one launch pair per level (A/B aid)
    unsigned long long multi_mask = 0;
    {
        static const bool multi_on = !(getenv("VFI_PYR_MULTI") && atoi(getenv("VFI_PYR_MULTI")) == 0);
        // (coarsest first, while their T regions fit into the workspace together: it is sized for ONE level, the finest)
        const size_t cap = (size_t)p->max_images * nb * H * p->tpitch_max;
        size_t need = 0;
        int cnt = 0;
        for (int k = p->nlev - 1; k >= 0 && cnt < kMaxMulti; --k) {
            if (!((level_mask >> k) & 1ull)) continue;
            const size_t t = (size_t)N * nb * p->lev[k].h * p->lev[k].w;
            if ((long long)p->lev[k].h * p->lev[k].w > 40000 || need + t > cap) break;
            multi_mask |= 1ull << k; need += t; ++cnt;
        }
        if (!multi_on || cnt < 2) multi_mask = 0;
    }
    if (multi_mask) {
        using namespace vfi::fft;
        MultiColsArgs mc[2];      // [bluestein]
        MultiRowsArgs mr[2];
        size_t clds[2] = {0, 0}, rlds[2] = {0, 0};
        for (int b = 0; b < 2; ++b) { mc[b].count = 0; mc[b].first[0] = 0; mr[b].count = 0; mr[b].first[0] = 0; }
        size_t toff = 0;          // (float2 elements into p->bands)
        for (int k = 0; k < p->nlev; ++k) {
            if (!((multi_mask >> k) & 1ull)) continue;
            const Level &L = p->lev[k];
            VFI_REQUIRE(phase[k] && ((flags & VFI_PYR_COMPLEX_COEFF) || amp[k]), VFI_ERR_INVALID_ARG,
                        "vfi_pyr_analyze: null output for level %d", k);
            Plan1D ph, pw;
            if ((rc = get_fft(p, L.h, &ph)) || (rc = get_fft(p, L.w, &pw))) return rc;
            float2 *T = p->bands + toff;
            toff += (size_t)N * nb * L.h * L.w;
            int tile, bpp;
            level_tiling(ph, L.w, &tile, &bpp);
            MultiColsArgs &c = mc[ph.bluestein ? 1 : 0];
            c.lev[c.count] = LevelColsArgs{ph, p->half0, T, L.P_a, L.h, L.w, H, W, tile, bpp};
            c.first[c.count + 1] = c.first[c.count] + 8 * ceil_div(ceil_div(L.w, tile), 8);
            ++c.count;
            clds[ph.bluestein ? 1 : 0] = std::max(clds[ph.bluestein ? 1 : 0], level_lds_bytes(ph, tile, bpp));
            const long long rows = (long long)N * nb * L.h;
            int lines = rows_per_group(pw, rows);
            if (lines > 256) lines = 256;
            MultiRowsArgs &r = mr[pw.bluestein ? 1 : 0];
            r.lev[r.count] = RowsPolarArgs{pw, T, phase[k], amp ? amp[k] : nullptr, make_map(plane_index, k, N, nb, flags), rows, L.h, lines,
                                           1.0f / ((float)L.h * (float)L.w), phase_scale, amp_max ? p->amp_bits + (size_t)k * groups : nullptr, groups};
            r.first[r.count + 1] = r.first[r.count] + (int)((rows + lines - 1) / lines);
            ++r.count;
            rlds[pw.bluestein ? 1 : 0] = std::max(rlds[pw.bluestein ? 1 : 0], row_lds_bytes(pw, lines, (size_t)lines * sizeof(size_t)));
        }
        if (mc[0].count) {
            allow_big_lds<pyr_multi_level_cols_kernel<4, false>>();
            hipLaunchKernelGGL((pyr_multi_level_cols_kernel<4, false>), dim3(mc[0].first[mc[0].count], N), dim3(kThreads), clds[0], s, mc[0]);
        }
        if (mc[1].count) {
            allow_big_lds<pyr_multi_level_cols_kernel<4, true>>();
            hipLaunchKernelGGL((pyr_multi_level_cols_kernel<4, true>), dim3(mc[1].first[mc[1].count], N), dim3(kThreads), clds[1], s, mc[1]);
        }
        if (mr[0].count) {
            allow_big_lds<pyr_multi_rows_polar_kernel<4, false>>();
            hipLaunchKernelGGL((pyr_multi_rows_polar_kernel<4, false>), dim3(mr[0].first[mr[0].count]), dim3(kThreads), rlds[0], s, mr[0]);
        }
        if (mr[1].count) {
            allow_big_lds<pyr_multi_rows_polar_kernel<4, true>>();
            hipLaunchKernelGGL((pyr_multi_rows_polar_kernel<4, true>), dim3(mr[1].first[mr[1].count]), dim3(kThreads), rlds[1], s, mr[1]);
        }
    }
    for (int k = 0; k < p->nlev; ++k) {
        const Level &L = p->lev[k];
        if (!((level_mask >> k) & 1ull) || ((multi_mask >> k) & 1ull)) continue;      // (the levels read the half spectrum directly: nothing to pass along)
        VFI_REQUIRE(phase[k] && ((flags & VFI_PYR_COMPLEX_COEFF) || amp[k]), VFI_ERR_INVALID_ARG,
                    "vfi_pyr_analyze: null output for level %d", k);
        using namespace vfi::fft;
        Plan1D ph, pw;
        vfi::pyrw::Tables tbh, tbw;
        if ((rc = get_fft(p, L.h, &ph)) || (rc = get_fft(p, L.w, &pw)) || (rc = wave_tables(p, kWaveAnaCols, ph, &tbh)) ||
            (rc = wave_tables(p, kWaveRows, pw, &tbw)))
            return rc;
        const int tpitch = tbh.M && tbw.M ? round_up16(L.w) : L.w;      // (the generic kernels address T densely)
        const PlaneMap pm = make_map(plane_index, k, N, nb, flags);
        unsigned *amax = amp_max ? p->amp_bits + (size_t)k * groups : nullptr;
        if (tbh.M) {
            vfi::pyrw::AnaColsArgs ca{tbh, p->half0, W / 2 + 1, H, L.P_a, p->bands, tpitch, N, L.h, L.w};
            if ((rc = vfi::pyrw::launch_ana_cols(ca, s))) return rc;
            debug_scan(p->bands, (size_t)N * nb * L.h * tpitch * 2, s, "T after the wave column pass", k);
        } else {
            int tile, bpp;
            level_tiling(ph, L.w, &tile, &bpp);
            LevelColsArgs ca{ph, p->half0, p->bands, L.P_a, L.h, L.w, H, W, tile, bpp};
            const dim3 cgrid(8 * ceil_div(ceil_div(L.w, ca.tile), 8), N);
            const size_t clds = level_lds_bytes(ph, tile, bpp);
            if (ph.bluestein) {
                allow_big_lds<pyr_level_cols_kernel<4, true>>();
                hipLaunchKernelGGL((pyr_level_cols_kernel<4, true>), cgrid, dim3(kThreads), clds, s, ca);
            } else {
                allow_big_lds<pyr_level_cols_kernel<4, false>>();
                hipLaunchKernelGGL((pyr_level_cols_kernel<4, false>), cgrid, dim3(kThreads), clds, s, ca);
            }
        }
        if (tbw.M) {
            vfi::pyrw::RowsArgs ra{tbw, p->bands, tpitch, phase[k], amp ? amp[k] : nullptr, pm, N * nb, L.h, L.w,
                                   1.0f / ((float)L.h * (float)L.w), phase_scale, amax, groups};
            if ((rc = vfi::pyrw::launch_rows_polar(ra, s))) return rc;
        } else {
            const long long rows = (long long)N * nb * L.h;
            int lines = rows_per_group(pw, rows);
            if (lines > 256) lines = 256;
            RowsPolarArgs ra{pw, p->bands, phase[k], amp ? amp[k] : nullptr, pm, rows, L.h, lines,
                             1.0f / ((float)L.h * (float)L.w), phase_scale, amax, groups};
            if (pw.bluestein) {
                allow_big_lds<pyr_rows_polar_kernel<4, true>>();
                hipLaunchKernelGGL((pyr_rows_polar_kernel<4, true>), dim3((unsigned)((rows + lines - 1) / lines)), dim3(kThreads),
                                   row_lds_bytes(pw, lines, (size_t)lines * sizeof(size_t)), s, ra);
            } else {
                allow_big_lds<pyr_rows_polar_kernel<4, false>>();
                hipLaunchKernelGGL((pyr_rows_polar_kernel<4, false>), dim3((unsigned)((rows + lines - 1) / lines)), dim3(kThreads),
                                   row_lds_bytes(pw, lines, (size_t)lines * sizeof(size_t)), s, ra);
            }
        }
    }
    if (low) {  // low residual: real(ifft2(window(dft) * low_gain))
        float2 *buf = p->lod[0];
        const int tot1 = p->hl * p->wl;
        hipLaunchKernelGGL(pyr_low_kernel, dim3(ceil_div(tot1, 256)), dim3(256), 0, s, p->half0, buf, p->low_gain, N, H, W, p->hl, p->wl);
        if ((rc = fft2d_c2c(p, buf, N, p->hl, p->wl, true, s))) return rc;
        const long long tot = (long long)N * tot1;
        hipLaunchKernelGGL(complex_real_kernel, dim3(blocks_1d(tot)), dim3(256), 0, s, buf, low, tot,
                           1.0f / ((float)p->hl * (float)p->wl));
    }
    if (high) {  // high residual: C2R of half * hi0 / (H W)
        hipLaunchKernelGGL(pyr_high_kernel, dim3(ceil_div(W / 2 + 1, 256), H), dim3(256), 0, s, p->half0, p->half_hi, p->hi0, N, H, W,
                           1.0f / ((float)H * (float)W));
        if ((rc = fft2d_c2r(p, p->half_hi, high, N, s))) return rc;
    }
    if (amp_max) {
        const int count = p->nlev * groups;
        hipLaunchKernelGGL(pyr_amp_max_finish_kernel, dim3(ceil_div(count, 64)), dim3(64), 0, s, p->amp_bits, amp_max, count, eps);
    }
    return vfi::check_launch("vfi_pyr_analyze");
}

extern "C" int vfi_pyr_analyze(vfi_pyr_plan *p, const float *img, int N, float *high, float *const *phase,
                               float *const *amp, const int *plane_index, float *low, float phase_scale,
                               unsigned long long level_mask, int flags, vfi_stream_t stream) {
    return pyr_analyze_impl(p, img, N, high, phase, amp, plane_index, low, phase_scale, level_mask, flags, nullptr, 1, 0.0f, stream);
}

extern "C" int vfi_pyr_analyze_max(vfi_pyr_plan *p, const float *img, int N, float *high, float *const *phase,
                                   float *const *amp, const int *plane_index, float *low, float phase_scale,
                                   unsigned long long level_mask, int flags, float *amp_max, int groups, float eps,
                                   vfi_stream_t stream) {
    VFI_REQUIRE(amp_max, VFI_ERR_INVALID_ARG, "vfi_pyr_analyze_max: null amp_max");
    return pyr_analyze_impl(p, img, N, high, phase, amp, plane_index, low, phase_scale, level_mask, flags, amp_max, groups, eps, stream);
}

extern "C" int vfi_pyr_synthesize(vfi_pyr_plan *p, const float *high, const float *const *phase, const float *const *amp,
                                  const int *plane_index, const float *low, unsigned long long level_mask, int flags,
                                  float *img, int N, vfi_stream_t stream) {
    VFI_REQUIRE(p && img, VFI_ERR_INVALID_ARG, "vfi_pyr_synthesize: null pointer");
    VFI_REQUIRE(N >= 1 && N <= p->max_images, VFI_ERR_INVALID_ARG, "vfi_pyr_synthesize: N=%d (plan max %d)", N, p->max_images);
    VFI_REQUIRE((phase && (amp || (flags & VFI_PYR_COMPLEX_COEFF))) || level_mask == 0, VFI_ERR_INVALID_ARG,
                "vfi_pyr_synthesize: null phase/amp tables");
    hipStream_t s = vfi::as_stream(stream);
    const int H = p->H, W = p->W, nb = p->nbands;
    int rc;
    // coarsest: res = FFT(low) (zeros when low is NULL)
    float2 *res = p->lod[p->nlev & 1];
    {
        const long long tot = (long long)N * p->hl * p->wl;
        hipLaunchKernelGGL(real_to_complex_kernel, dim3(blocks_1d(tot)), dim3(256), 0, s, low, res, tot);
        if (low && (rc = fft2d_c2c(p, res, N, p->hl, p->wl, false, s))) return rc;
    }
    for (int k = p->nlev - 1; k >= 0; --k) {
        const Level &L = p->lev[k];
        const int h2 = k + 1 < p->nlev ? p->lev[k + 1].h : p->hl, w2 = k + 1 < p->nlev ? p->lev[k + 1].w : p->wl;
        const int hb = (level_mask >> k) & 1ull ? 1 : 0;
        float2 *cur = p->lod[k & 1];
        if (!hb) {      // no bands at this level: embed the coarser spectrum only
            hipLaunchKernelGGL((pyr_combine_kernel<4>), dim3(ceil_div(L.w, 256), L.h), dim3(256), 0, s, p->bands, res, cur, L.P_s,
                               L.lomask, N, L.h, L.w, h2, w2, 0);
            res = cur;
            continue;
        }
        VFI_REQUIRE(phase[k] && ((flags & VFI_PYR_COMPLEX_COEFF) || amp[k]), VFI_ERR_INVALID_ARG,
                    "vfi_pyr_synthesize: null input for level %d", k);
        using namespace vfi::fft;
        Plan1D ph, pw;
        vfi::pyrw::Tables tbh, tbw;
        if ((rc = get_fft(p, L.h, &ph)) || (rc = get_fft(p, L.w, &pw)) || (rc = wave_tables(p, kWaveSynCols, ph, &tbh)) ||
            (rc = wave_tables(p, kWaveRows, pw, &tbw)))
            return rc;
        const int tpitch = tbh.M && tbw.M ? round_up16(L.w) : L.w;
        const PlaneMap pm = make_map(plane_index, k, N, nb, flags);
        if (tbw.M) {
            vfi::pyrw::RowsArgs ra{tbw, p->bands, tpitch, const_cast<float *>(phase[k]), amp ? const_cast<float *>(amp[k]) : nullptr, pm,
                                   N * nb, L.h, L.w, 1.0f, 1.0f, nullptr, 1};
            if ((rc = vfi::pyrw::launch_rows_from_polar(ra, s))) return rc;
        } else {
            const long long rows = (long long)N * nb * L.h;
            int lines = rows_per_group(pw, rows);
            if (lines > 256) lines = 256;
            RowsPolarArgs ra{pw, p->bands, const_cast<float *>(phase[k]), amp ? const_cast<float *>(amp[k]) : nullptr, pm, rows, L.h,
                             lines, 1.0f, 1.0f, nullptr, 1};
            if (pw.bluestein) {
                allow_big_lds<pyr_rows_from_polar_kernel<4, true>>();
                hipLaunchKernelGGL((pyr_rows_from_polar_kernel<4, true>), dim3((unsigned)((rows + lines - 1) / lines)), dim3(kThreads),
                                   row_lds_bytes(pw, lines, (size_t)lines * sizeof(size_t)), s, ra);
            } else {
                allow_big_lds<pyr_rows_from_polar_kernel<4, false>>();
                hipLaunchKernelGGL((pyr_rows_from_polar_kernel<4, false>), dim3((unsigned)((rows + lines - 1) / lines)), dim3(kThreads),
                                   row_lds_bytes(pw, lines, (size_t)lines * sizeof(size_t)), s, ra);
            }
        }
        if (tbh.M) {
            vfi::pyrw::SynColsArgs ca{tbh, p->bands, tpitch, L.P_s, res, L.lomask, cur, N, L.h, L.w, h2, w2};
            if ((rc = vfi::pyrw::launch_syn_cols(ca, s))) return rc;
        } else {
            int tile, bpp;
            level_tiling(ph, L.w, &tile, &bpp);
            CombineColsArgs ca{ph, p->bands, res, cur, L.P_s, L.lomask, L.h, L.w, h2, w2, tile, bpp};
            if (ph.bluestein) {
                allow_big_lds<pyr_combine_cols_kernel<4, true>>();
                hipLaunchKernelGGL((pyr_combine_cols_kernel<4, true>), dim3(8 * ceil_div(ceil_div(L.w, ca.tile), 8), N), dim3(kThreads),
                                   level_lds_bytes(ph, tile, bpp), s, ca);
            } else {
                allow_big_lds<pyr_combine_cols_kernel<4, false>>();
                hipLaunchKernelGGL((pyr_combine_cols_kernel<4, false>), dim3(8 * ceil_div(ceil_div(L.w, ca.tile), 8), N), dim3(kThreads),
                                   level_lds_bytes(ph, tile, bpp), s, ca);
            }
        }
        res = cur;
    }
    const float2 *hi_half = nullptr;
    if (high) {
        if ((rc = fft2d_r2c(p, high, p->half0, N, s))) return rc;
        hi_half = p->half0;
    }
    hipLaunchKernelGGL(pyr_final_kernel, dim3(ceil_div(W, 256), H), dim3(256), 0, s, res, hi_half, p->lo0, p->hi0, N, H, W);
    if ((rc = fft2d_c2c(p, res, N, H, W, true, s))) return rc;
    const long long tot = (long long)N * H * W;
    hipLaunchKernelGGL(complex_real_kernel, dim3(blocks_1d(tot)), dim3(256), 0, s, res, img, tot, 1.0f / ((float)H * (float)W));
    return vfi::check_launch("vfi_pyr_synthesize");
}
