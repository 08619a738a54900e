// Image-space stages of the fused per-frame path that the reference runs on the HOST CPU
// (skimage / scipy round trips, reference src/fusion_net/interpolate_twoframe.py:148-149,190-225):
//   rgb <-> Lab             reference src/train/transform.py:17-25,40-49 (skimage.color, D65 / 2 deg)
//   Gaussian sigma=5         scipy.ndimage.gaussian_filter(x, 5)         (interpolate_twoframe.py:212-213)
//   50x50 median             scipy.ndimage.median_filter(x, size=50)     (interpolate_twoframe.py:221-222)
//   colour mean / |a-b| / scale / clamp glue around them                (interpolate_twoframe.py:207-225)
// Keeping them on the GPU removes five device<->host round trips per frame; the 50x50 median alone
// costs the reference 66 s per 1080p frame on 8 CPU cores.
#include "vfi_common.h"

namespace {

using vfi::ceil_div;
inline int blocks_1d(long long n) { long long b = (n + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

// ---- colour -------------------------------------------------------------------------------------------
// skimage.color.rgb2lab: sRGB -> linear -> XYZ (xyz_from_rgb) -> / D65 white -> f(t) -> Lab; then the
// reference's scaling L/100, (a+128)/255, (b+128)/255 (transform.py:20-22).
__global__ void rgb2lab_kernel(const float *__restrict__ rgb, float *__restrict__ lab, int N, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, n = i / HW;
        const float *s = rgb + (size_t)n * 3 * HW + p;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = s[(size_t)k * HW];
            c[k] = v > 0.04045f ? powf((v + 0.055f) / 1.055f, 2.4f) : v / 12.92f;
        }
        float x = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        float y = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        float z = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
        x /= 0.95047f; z /= 1.08883f;
        auto f = [](float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.0f / 116.0f; };
        const float fx = f(x), fy = f(y), fz = f(z);
        float *d = lab + (size_t)n * 3 * HW + p;
        d[0] = (116.0f * fy - 16.0f) / 100.0f;
        d[(size_t)HW] = (500.0f * (fx - fy) + 128.0f) / 255.0f;
        d[(size_t)2 * HW] = (200.0f * (fy - fz) + 128.0f) / 255.0f;
    }
}

// skimage.color.lab2rgb after the reference's un-scaling (transform.py:43-45): Lab -> XYZ (z < 0 -> 0)
// -> linear rgb (inverse matrix) -> sRGB gamma -> clip [0,1].
__global__ void lab2rgb_kernel(const float *__restrict__ lab, float *__restrict__ rgb, int N, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, n = i / HW;
        const float *s = lab + (size_t)n * 3 * HW + p;
        const float L = s[0] * 100.0f, a = s[(size_t)HW] * 255.0f - 128.0f, b = s[(size_t)2 * HW] * 255.0f - 128.0f;
        const float fy = (L + 16.0f) / 116.0f;
        const float fx = a / 500.0f + fy;
        float fz = fy - b / 200.0f;
        fz = fz < 0.0f ? 0.0f : fz;
        auto g = [](float t) { return t > 0.2068966f ? t * t * t : (t - 16.0f / 116.0f) / 7.787f; };
        const float x = g(fx) * 0.95047f, y = g(fy), z = g(fz) * 1.08883f;
        // rgb_from_xyz = inv(xyz_from_rgb)
        float c[3];
        c[0] = 3.24048134f * x - 1.53715152f * y - 0.49853633f * z;
        c[1] = -0.96925495f * x + 1.87599f * y + 0.04155593f * z;
        c[2] = 0.05564664f * x - 0.20404134f * y + 1.05731107f * z;
        float *d = rgb + (size_t)n * 3 * HW + p;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = c[k] > 0.0031308f ? 1.055f * powf(c[k], 1.0f / 2.4f) - 0.055f : c[k] * 12.92f;
            d[(size_t)k * HW] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

// ---- colour mean / difference glue --------------------------------------------------------------------------
// out[p] = (mean_c a[c][p] (- mean_c b[c][p], abs)) * scale, optionally clamped to [0,1]
__global__ void channel_mean_diff_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out,
                                         int N, int C, int HW, float scale, int clamp01) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = i % HW, n = i / HW;
        float sa = 0.0f, sb = 0.0f;
        for (int c = 0; c < C; ++c) {
            sa += a[((size_t)n * C + c) * HW + p];
            if (b) sb += b[((size_t)n * C + c) * HW + p];
        }
        float v = sa / (float)C;
        if (b) { v = v - sb / (float)C; if (!(clamp01 & 2)) v = fabsf(v); }
        v *= scale;
        out[i] = (clamp01 & 1) ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
    }
}
// out = |x - y| * scale (|x| * scale without y), optionally clamped   (subtract_values: src/train/utils.py:322-346; :223-224)
__global__ void absdiff_kernel(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out,
                               long long total, float scale, int clamp01) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float v = fabsf(y ? x[i] - y[i] : x[i]) * scale;
        out[i] = clamp01 ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
    }
}

// ---- separable Gaussian, scipy 'reflect' boundary (d c b a | a b c d | d c b a) --------------------------------
constexpr int kMaxRadius = 64;
struct GaussTaps { float w[2 * kMaxRadius + 1]; int radius; };

__device__ __forceinline__ int sym_reflect(int i, int n) {  // scipy mode='reflect' (half-sample symmetric)
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

template <bool VERTICAL>
__global__ void gauss_pass_kernel(const float *__restrict__ x, float *__restrict__ y, int N, int H, int W, GaussTaps t) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = i % W, yy = (i / W) % H, n = i / ((long long)W * H);
        const float *p = x + (size_t)n * H * W;
        float acc = 0.0f;
        for (int k = -t.radius; k <= t.radius; ++k) {
            const int sy = VERTICAL ? sym_reflect(yy + k, H) : yy;
            const int sx = VERTICAL ? xx : sym_reflect(xx + k, W);
            acc += t.w[k + t.radius] * p[(size_t)sy * W + sx];
        }
        y[i] = acc;
    }
}

// ---- size x size median, scipy semantics (mode='reflect', origin 0, rank size*size/2) -----------------------------
// Exact selection.  A 32x8 output tile stages its (32+S-1) x (8+S-1) input window in LDS as order-preserving
// 32-bit keys; each thread then bisects the key range [min, max] of ITS window for the smallest key whose
// rank reaches the median: every step is one conflict-free sweep of the window in LDS.
__device__ __forceinline__ unsigned key_of(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int kMedTW = 32, kMedTH = 8;

__global__ __launch_bounds__(256) void median_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int S) {
    extern __shared__ unsigned tile[];
    const int PW = kMedTW + S - 1, PH = kMedTH + S - 1;
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * kMedTW, y0 = blockIdx.y * kMedTH;
    const int lo_off = S / 2;  // window of output i covers [i - S/2, i - S/2 + S - 1]
    const float *p = x + (size_t)n * H * W;
    for (int e = threadIdx.x; e < PW * PH; e += 256) {
        const int r = e / PW, c = e % PW;
        const int sy = sym_reflect(y0 - lo_off + r, H), sx = sym_reflect(x0 - lo_off + c, W);
        tile[e] = key_of(p[(size_t)sy * W + sx]);
    }
    __syncthreads();
    const int tx = threadIdx.x % kMedTW, ty = threadIdx.x / kMedTW;
    const int gx = x0 + tx, gy = y0 + ty;
    const unsigned *base = tile + ty * PW + tx;
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S; ++c) {
            const unsigned k = base[r * PW + c];
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
    const int need = (S * S) / 2 + 1;  // smallest key K with count(key <= K) >= rank+1
    // Invariants: count(key < lo) = c_lo < need  and  count(key <= hi) = c_hi >= need.  Every second probe is
    // an interpolation step on the (locally near-linear) rank function, the others plain bisection, so the
    // search ends in <= 2*32 probes worst case and typically a third of the bisection count.
    unsigned lo = kmin, hi = kmax;
    int c_lo = 0, c_hi = S * S;
    int it = 0;
    while (__any(lo < hi)) {
        const bool active = lo < hi;
        unsigned mid = lo + (hi - lo) / 2;
        if ((it & 1) == 0 && active) {
            const float frac = ((float)(need - c_lo) - 0.5f) / (float)(c_hi - c_lo);
            const unsigned span = hi - lo;
            const unsigned g = lo + (unsigned)fminf((float)span * fminf(fmaxf(frac, 0.0f), 1.0f), (float)(span - 1));
            mid = min(max(g, lo), hi - 1);
        }
        int cnt = 0;
        for (int r = 0; r < S; ++r)
            for (int c = 0; c < S; ++c) cnt += base[r * PW + c] <= mid ? 1 : 0;
        if (active) {
            if (cnt >= need) { hi = mid; c_hi = cnt; } else { lo = mid + 1; c_lo = cnt; }
        }
        ++it;
    }
    if (gx < W && gy < H) y[(size_t)n * H * W + (size_t)gy * W + gx] = float_of(lo);
}

// ---- exact median, rank-transform + sliding bitset (the fast path, used whenever the tile fits) -------------------
// A 16-column x 64-row output tile and its (16+S-1) x (64+S-1) input window (<= 8192 keys):
//   1. the window's keys are sorted ONCE and every position is replaced by its rank, a unique integer < 8192.  The
//      sort is a bitonic network over (key, position) pairs held in REGISTERS, 32 per thread: compare distances below
//      32 stay inside a thread (55 of the 91 stages), distances 32..1024 are lane exchanges inside a wave (33 stages,
//      ds_bpermute), only distances 2048 / 4096 (3 stages) go through LDS and a barrier;
//   2. every output row keeps the set of ranks inside its SxS window as a bitset (four lanes per row share the
//      work); sliding the window one column clears S bits and sets S bits (LDS atomics, no return value) and nudges
//      a (word, popcount-below) cursor to the median rank -- ~2*S LDS operations per output pixel instead of ~20
//      sweeps of S*S keys;
//   3. the median is the sorted key at that rank: an element of the window, bit-exact with scipy.
constexpr int kRkTW = 16, kRkTH = 64, kRkN = 8192, kRkWords = kRkN / 32, kRkPer = kRkN / 256;
constexpr size_t kRkKeyBytes = (size_t)(kRkN + kRkN / 32) * 4;      // sorted keys at index e + e/32 (conflict-free column writes)
constexpr size_t kRkLds = kRkKeyBytes + (size_t)kRkN * 2 + (size_t)kRkTH * kRkWords * 4;
static_assert(kRkLds >= (size_t)kRkN * 8, "the cross-wave exchange buffer aliases the arrays of the later phases");

typedef unsigned long long med_pair;     // key << 32 | position: unique, so the order is total

__device__ __forceinline__ void med_cmpex(med_pair &a, med_pair &b, bool asc) {      // a is the lower index of the pair
    const bool sw = (a > b) == asc;
    const med_pair lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}

__global__ __launch_bounds__(256) void median_rank_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int S) {
    extern __shared__ unsigned smem[];
    unsigned *K = smem;                                                                  // sorted keys
    unsigned short *R = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(smem) + kRkKeyBytes);   // [8192] rank of each position
    unsigned *B = reinterpret_cast<unsigned *>(R + kRkN);                                // [64][256] one bitset per output row
    med_pair *X = reinterpret_cast<med_pair *>(smem);                                   // [32][256] exchange (sort phase only)
    const int PW = kRkTW + S - 1, PH = kRkTH + S - 1, U = PW * PH;
    const int n = blockIdx.z, x0 = blockIdx.x * kRkTW, y0 = blockIdx.y * kRkTH, lo_off = S / 2;
    const float *p = x + (size_t)n * H * W;
    const int tid = threadIdx.x;
    // element r of thread t sits at index i = 32 t + r of the network; which window position starts there is irrelevant
    med_pair e[kRkPer];
#pragma unroll
    for (int r = 0; r < kRkPer; ++r) {
        const int pos = r * 256 + tid;
        unsigned k = 0xffffffffu;                            // padding sorts last: ranks >= U unused
        if (pos < U) {
            const int rr = pos / PW, cc = pos - rr * PW;
            k = key_of(p[(size_t)sym_reflect(y0 - lo_off + rr, H) * W + sym_reflect(x0 - lo_off + cc, W)]);
        }
        e[r] = ((med_pair)k << 32) | (unsigned)pos;
    }
    // k = 2 .. 32: inside the thread
#pragma unroll
    for (int k = 2; k <= kRkPer; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int r = 0; r < kRkPer; ++r)
                if ((r & j) == 0) med_cmpex(e[r], e[r | j], (((tid << 5) | r) & k) == 0);
    // k = 64 .. 8192: partner threads first (distance j = 32 m, thread t ^ m), then the in-thread tail
    for (int k = 2 * kRkPer; k <= kRkN; k <<= 1) {
        const bool asc = ((tid << 5) & k) == 0;
        for (int j = k >> 1; j >= kRkPer; j >>= 1) {
            const int m = j >> 5;
            const bool keep_min = ((tid & m) == 0) == asc;      // the lower index of an ascending pair keeps the minimum
            if (m >= 64) {                                       // partner in another wave
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kRkPer; ++r) X[r * 256 + tid] = e[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kRkPer; ++r) {
                    const med_pair o = X[r * 256 + (tid ^ m)];
                    e[r] = (o < e[r]) == keep_min ? o : e[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < kRkPer; ++r) {
                    const med_pair o = __shfl_xor(e[r], m, 64);
                    e[r] = (o < e[r]) == keep_min ? o : e[r];
                }
            }
        }
#pragma unroll
        for (int j = kRkPer >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int r = 0; r < kRkPer; ++r)
                if ((r & j) == 0) med_cmpex(e[r], e[r | j], asc);
    }
    __syncthreads();                                         // the exchange buffer is dead: its space becomes K, R, B
#pragma unroll
    for (int r = 0; r < kRkPer; ++r) {
        const int i = (tid << 5) | r;
        K[i + tid] = (unsigned)(e[r] >> 32);                 // index i + i/32
        const unsigned pos = (unsigned)e[r];
        if (pos < (unsigned)U) R[pos] = (unsigned short)i;
    }
    for (int w = tid; w < kRkTH * kRkWords; w += 256) B[w] = 0u;
    __syncthreads();
    // Four adjacent lanes share an output row: they split the insertions / removals (the LDS serves one wave's
    // operations in order, so every lane's later reads see all four lanes' earlier atomics) and each keeps the cursor.
    {
        const int row = tid >> 2, q = tid & 3;
        unsigned *bits = B + row * kRkWords;
        const unsigned short *rows = R + row * PW;      // window of output row `row` starts at union row `row`
        const int need = (S * S) / 2 + 1;
        {
            int r = 0, c = q;
            while (c >= S) { c -= S; ++r; }
#pragma unroll 8
            for (int i = q; i < S * S; i += 4) {
                const unsigned rk = rows[r * PW + c];
                atomicOr(&bits[rk >> 5], 1u << (rk & 31));
                c += 4;
                while (c >= S) { c -= S; ++r; }
            }
        }
        int ptr = 0, below = 0;
        const int gy = y0 + row;
        for (int tx = 0; tx < kRkTW; ++tx) {
            if (tx > 0) {
                int delta = 0;
#pragma unroll 4
                for (int r = q; r < S; r += 4) {
                    const unsigned out = rows[r * PW + tx - 1], in = rows[r * PW + tx + S - 1];
                    atomicAnd(&bits[out >> 5], ~(1u << (out & 31)));
                    atomicOr(&bits[in >> 5], 1u << (in & 31));
                    delta += ((int)(in >> 5) < ptr) - ((int)(out >> 5) < ptr);
                }
                delta += __shfl_xor(delta, 1, 64);
                delta += __shfl_xor(delta, 2, 64);
                below += delta;
            }
            while (below >= need) { --ptr; below -= __popc(bits[ptr]); }
            unsigned wv = bits[ptr];
            while (below + __popc(wv) < need) { below += __popc(wv); wv = bits[++ptr]; }
            for (int i = need - below; i > 1; --i) wv &= wv - 1;      // drop the (need-below-1) lowest set bits
            const int rank = ptr * 32 + __ffs(wv) - 1;
            const int gx = x0 + tx;
            if (q == (tx & 3) && gx < W && gy < H) y[(size_t)n * H * W + (size_t)gy * W + gx] = float_of(K[rank + (rank >> 5)]);
        }
    }
}

// ---- quality scoring (reference src/evaluation/evaluate_image.py:7-30) ---------------------------------------------
// Deterministic two-stage reductions: per-block partial sums in double, then one block adds them in fixed order.
constexpr int kRedBlocks = 1024;
__global__ __launch_bounds__(256) void diff_stats_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                 long long n, double *__restrict__ part) {
    double s1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double d = (double)a[i] - (double)b[i];
        s1 += d; s2 += d * d;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[1][0]; }
}
// SSIM map of piq.ssim (Gaussian-filtered moments in, sum of the per-pixel SSIM over the valid interior out):
//   cs = (2 sxy + c2) / (sxx + syy + c2) ; ssim = (2 mx my + c1) / (mx^2 + my^2 + c1) * cs ; s** = E[.] - mu products
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float *__restrict__ mx, const float *__restrict__ my,
                                                           const float *__restrict__ exx, const float *__restrict__ eyy,
                                                           const float *__restrict__ exy, int planes, int H, int W, int border,
                                                           float c1, float c2, double *__restrict__ part) {
    const int Hv = H - 2 * border, Wv = W - 2 * border;
    const long long n = (long long)planes * Hv * Wv;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = i % Wv, y = (i / Wv) % Hv, p = i / ((long long)Wv * Hv);
        const size_t o = ((size_t)p * H + y + border) * W + x + border;
        const float a = mx[o], b = my[o];
        const float sxx = exx[o] - a * a, syy = eyy[o] - b * b, sxy = exy[o] - a * b;
        const float cs = (2.0f * sxy + c2) / (sxx + syy + c2);
        s += (double)((2.0f * a * b + c1) / (a * a + b * b + c1) * cs);
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0]; part[2 * blockIdx.x + 1] = 0.0; }
}
__global__ void sum_partials_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < nblocks; ++i) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
        out[0] = s1; out[1] = s2;
    }
}
__global__ void mul_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i];
}

}  // namespace

extern "C" int vfi_diff_sums(const float *a, const float *b, long long count, double *out2, void *workspace, vfi_stream_t stream) {
    VFI_REQUIRE(a && b && out2 && workspace, VFI_ERR_INVALID_ARG, "vfi_diff_sums: null pointer");
    VFI_REQUIRE(count > 0, VFI_ERR_INVALID_ARG, "vfi_diff_sums: bad size");
    hipStream_t s = vfi::as_stream(stream);
    const int nb = (int)((count + 255) / 256 < kRedBlocks ? (count + 255) / 256 : kRedBlocks);
    hipLaunchKernelGGL(diff_stats_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, count, static_cast<double *>(workspace));
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, static_cast<const double *>(workspace), nb, out2);
    return vfi::check_launch("vfi_diff_sums");
}

extern "C" int vfi_ssim_sum(const float *mu_x, const float *mu_y, const float *e_xx, const float *e_yy, const float *e_xy,
                            int planes, int H, int W, int border, float c1, float c2, double *out2, void *workspace,
                            vfi_stream_t stream) {
    VFI_REQUIRE(mu_x && mu_y && e_xx && e_yy && e_xy && out2 && workspace, VFI_ERR_INVALID_ARG, "vfi_ssim_sum: null pointer");
    VFI_REQUIRE(planes > 0 && H > 2 * border && W > 2 * border && border >= 0, VFI_ERR_INVALID_ARG, "vfi_ssim_sum: bad sizes");
    hipStream_t s = vfi::as_stream(stream);
    const long long n = (long long)planes * (H - 2 * border) * (W - 2 * border);
    const int nb = (int)((n + 255) / 256 < kRedBlocks ? (n + 255) / 256 : kRedBlocks);
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(nb), dim3(256), 0, s, mu_x, mu_y, e_xx, e_yy, e_xy, planes, H, W, border, c1, c2,
                       static_cast<double *>(workspace));
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, static_cast<const double *>(workspace), nb, out2);
    return vfi::check_launch("vfi_ssim_sum");
}

extern "C" int vfi_mul(const float *a, const float *b, float *out, long long count, vfi_stream_t stream) {
    VFI_REQUIRE(a && b && out, VFI_ERR_INVALID_ARG, "vfi_mul: null pointer");
    VFI_REQUIRE(count > 0, VFI_ERR_INVALID_ARG, "vfi_mul: bad size");
    hipLaunchKernelGGL(mul_kernel, dim3(blocks_1d(count)), dim3(256), 0, vfi::as_stream(stream), a, b, out, count);
    return vfi::check_launch("vfi_mul");
}

namespace {
// F.avg_pool2d(kernel_size=f): stride f, floor output size; accumulation in the row-major order of the window
__global__ void avg_pool_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int H, int W, int f) {
    const int Ho = H / f, Wo = W / f;
    const long long total = (long long)planes * Ho * Wo;
    const float inv = 1.0f / (float)(f * f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
        const long long p = i / ((long long)Wo * Ho);
        const float *src = x + (p * H + (long long)yo * f) * W + (long long)xo * f;
        float acc = 0.0f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) acc += src[(long long)dy * W + dx];
        y[i] = acc * inv;
    }
}
}  // namespace

extern "C" int vfi_avg_pool(const float *x, float *y, int planes, int H, int W, int f, vfi_stream_t stream) {
    VFI_REQUIRE(x && y, VFI_ERR_INVALID_ARG, "vfi_avg_pool: null pointer");
    VFI_REQUIRE(planes > 0 && f >= 1 && H >= f && W >= f, VFI_ERR_INVALID_ARG, "vfi_avg_pool: bad sizes");
    const long long total = (long long)planes * (H / f) * (W / f);
    hipLaunchKernelGGL(avg_pool_kernel, dim3(blocks_1d(total)), dim3(256), 0, vfi::as_stream(stream), x, y, planes, H, W, f);
    return vfi::check_launch("vfi_avg_pool");
}

extern "C" int vfi_rgb2lab(const float *rgb, float *lab, int N, int HW, vfi_stream_t stream) {
    VFI_REQUIRE(rgb && lab, VFI_ERR_INVALID_ARG, "vfi_rgb2lab: null pointer");
    VFI_REQUIRE(N > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_rgb2lab: bad sizes");
    hipLaunchKernelGGL(rgb2lab_kernel, dim3(blocks_1d((long long)N * HW)), dim3(256), 0, vfi::as_stream(stream), rgb, lab, N, HW);
    return vfi::check_launch("vfi_rgb2lab");
}

extern "C" int vfi_lab2rgb(const float *lab, float *rgb, int N, int HW, vfi_stream_t stream) {
    VFI_REQUIRE(rgb && lab, VFI_ERR_INVALID_ARG, "vfi_lab2rgb: null pointer");
    VFI_REQUIRE(N > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_lab2rgb: bad sizes");
    hipLaunchKernelGGL(lab2rgb_kernel, dim3(blocks_1d((long long)N * HW)), dim3(256), 0, vfi::as_stream(stream), lab, rgb, N, HW);
    return vfi::check_launch("vfi_lab2rgb");
}

extern "C" int vfi_channel_mean_diff(const float *a, const float *b, float *out, int N, int C, int HW, float scale,
                                     int clamp01, vfi_stream_t stream) {
    VFI_REQUIRE(a && out, VFI_ERR_INVALID_ARG, "vfi_channel_mean_diff: null pointer");
    VFI_REQUIRE(N > 0 && C > 0 && HW > 0, VFI_ERR_INVALID_ARG, "vfi_channel_mean_diff: bad sizes");
    hipLaunchKernelGGL(channel_mean_diff_kernel, dim3(blocks_1d((long long)N * HW)), dim3(256), 0, vfi::as_stream(stream), a, b,
                       out, N, C, HW, scale, clamp01);
    return vfi::check_launch("vfi_channel_mean_diff");
}

extern "C" int vfi_absdiff(const float *x, const float *y, float *out, long long count, float scale, int clamp01,
                           vfi_stream_t stream) {
    VFI_REQUIRE(x && out, VFI_ERR_INVALID_ARG, "vfi_absdiff: null pointer");      // (y == NULL: |x| * scale)
    VFI_REQUIRE(count > 0, VFI_ERR_INVALID_ARG, "vfi_absdiff: bad size");
    hipLaunchKernelGGL(absdiff_kernel, dim3(blocks_1d(count)), dim3(256), 0, vfi::as_stream(stream), x, y, out, count, scale, clamp01);
    return vfi::check_launch("vfi_absdiff");
}

extern "C" int vfi_gaussian_filter(const float *x, float *tmp, float *y, int N, int H, int W, float sigma, float truncate,
                                   vfi_stream_t stream) {
    VFI_REQUIRE(x && tmp && y, VFI_ERR_INVALID_ARG, "vfi_gaussian_filter: null pointer");
    VFI_REQUIRE(N > 0 && H > 0 && W > 0 && sigma > 0 && truncate > 0, VFI_ERR_INVALID_ARG, "vfi_gaussian_filter: bad arguments");
    GaussTaps t;
    t.radius = (int)(truncate * sigma + 0.5f);  // scipy: int(truncate * sd + 0.5)
    VFI_REQUIRE(t.radius <= kMaxRadius, VFI_ERR_UNSUPPORTED, "vfi_gaussian_filter: radius %d > %d", t.radius, kMaxRadius);
    double sum = 0.0, w[2 * kMaxRadius + 1];
    for (int k = -t.radius; k <= t.radius; ++k) { w[k + t.radius] = exp(-0.5 / ((double)sigma * sigma) * k * k); sum += w[k + t.radius]; }
    for (int k = 0; k <= 2 * t.radius; ++k) t.w[k] = (float)(w[k] / sum);
    hipStream_t s = vfi::as_stream(stream);
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(gauss_pass_kernel<true>, dim3(blocks_1d(total)), dim3(256), 0, s, x, tmp, N, H, W, t);   // axis 0 first
    hipLaunchKernelGGL(gauss_pass_kernel<false>, dim3(blocks_1d(total)), dim3(256), 0, s, tmp, y, N, H, W, t);
    return vfi::check_launch("vfi_gaussian_filter");
}

extern "C" int vfi_median_filter(const float *x, float *y, int N, int H, int W, int size, vfi_stream_t stream) {
    VFI_REQUIRE(x && y, VFI_ERR_INVALID_ARG, "vfi_median_filter: null pointer");
    VFI_REQUIRE(N > 0 && H > 0 && W > 0 && size >= 1, VFI_ERR_INVALID_ARG, "vfi_median_filter: bad arguments");
    VFI_REQUIRE(size <= 64, VFI_ERR_UNSUPPORTED, "vfi_median_filter: size %d > 64", size);
    VFI_REQUIRE(N <= 65535, VFI_ERR_UNSUPPORTED, "vfi_median_filter: batch");
    if ((kRkTW + size - 1) * (kRkTH + size - 1) <= kRkN && size >= 2) {
        constexpr size_t lds = kRkLds;   // 113 KiB
        static bool attr_done_dev[vfi::kMaxDevices] = {};  // per device, idempotent
        bool &attr_done = attr_done_dev[vfi::current_device()];
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(median_rank_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return vfi::fail(VFI_ERR_LAUNCH, "vfi_median_filter: set LDS size: %s", hipGetErrorString(e));
            attr_done = true;
        }
        dim3 grid(ceil_div(W, kRkTW), ceil_div(H, kRkTH), N);
        hipLaunchKernelGGL(median_rank_kernel, grid, dim3(256), lds, vfi::as_stream(stream), x, y, H, W, size);
        return vfi::check_launch("vfi_median_filter");
    }
    const size_t lds = (size_t)(kMedTW + size - 1) * (kMedTH + size - 1) * sizeof(unsigned);
    dim3 grid(ceil_div(W, kMedTW), ceil_div(H, kMedTH), N);
    hipLaunchKernelGGL(median_kernel, grid, dim3(256), lds, vfi::as_stream(stream), x, y, H, W, size);
    return vfi::check_launch("vfi_median_filter");
}
