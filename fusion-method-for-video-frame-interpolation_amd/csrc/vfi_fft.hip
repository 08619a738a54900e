// Batched 1-D FFT passes over rows / column tiles of dense 2-D arrays, built on the LDS engine of vfi_fft.h: the
// plain (un-fused) transforms of the steerable pyramid (R2C / C2R of the image and the high residual, the low
// residual, vfi_pyr_apply_filter) and the building blocks the fused level kernels of vfi_pyramid.hip follow.
//
//   row pass    : a workgroup owns `lines` consecutive rows (of any planes: rows are one flat list); coalesced loads
//                 straight into the LDS lines, transform, coalesced stores.  Real input / real output / Hermitian half
//                 rows are load / store variants, so R2C and C2R cost no extra pass.
//   column pass : a workgroup owns `tile` adjacent columns of one plane: rows of tile*8 bytes are read into LDS
//                 transposed (line = column, odd line pitch: conflict-free), transformed, written back in place.
// Roofline: HBM (each pass reads and writes the array once); the engine itself is LDS-bound.
#include "vfi_common.h"
#include "vfi_fft.h"

#include <cmath>
#include <complex>
#include <vector>

namespace vfi {
namespace fft {
namespace {

using cd = std::complex<double>;
constexpr double kPi = 3.14159265358979323846;

void host_fft_pow2(std::vector<cd> &a) {       // in-place radix-2, forward, double precision (table building only)
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w = std::polar(1.0, -2.0 * kPi * (double)k / (double)len);
                const cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

// ... of 2^k or 3 * 2^k points: three interleaved power-of-two transforms and one radix-3 step
void host_fft_bluestein(std::vector<cd> &a) {
    const size_t m = a.size();
    if (m % 3) return host_fft_pow2(a);
    const size_t q = m / 3;
    std::vector<cd> f[3] = {std::vector<cd>(q), std::vector<cd>(q), std::vector<cd>(q)};
    for (size_t j = 0; j < q; ++j)
        for (int r = 0; r < 3; ++r) f[r][j] = a[3 * j + r];
    for (int r = 0; r < 3; ++r) host_fft_pow2(f[r]);
    for (size_t k = 0; k < m; ++k) {
        cd acc(0.0, 0.0);
        for (int r = 0; r < 3; ++r) acc += f[r][k % q] * std::polar(1.0, -2.0 * kPi * (double)((r * k) % m) / (double)m);
        a[k] = acc;
    }
}

template <int LOAD, int STORE, bool INV>
__global__ __launch_bounds__(kThreads, kThreads / 128) void fft_rows_kernel(const RowArgs a) {
    extern __shared__ float2 buf[];
    const int n = a.pl.n, m = a.pl.m, pitch = padded_length(m), tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * a.lines;
    const int lines = (int)(a.rows - row0 < a.lines ? a.rows - row0 : a.lines);
    const float inv_n = 1.0f / (float)n;
    const int wh = n / 2 + 1;
    const bool blu = a.pl.bluestein != 0;
    float2 *twl = buf + (size_t)a.lines * pitch;
    load_twiddles(twl, a.pl);
    auto fill_load = [&](int l, int j) {
        Slot v;
        if (LOAD == kLoadComplex) {
            v.z = (static_cast<const float2 *>(a.src) + (row0 + l) * a.src_pitch)[j];
        } else if (LOAD == kLoadReal) {
            v.z = make_float2((static_cast<const float *>(a.src) + (row0 + l) * a.src_pitch)[j], 0.0f);
        } else {
            const float2 *r = static_cast<const float2 *>(a.src) + (row0 + l) * a.src_pitch;
            v.z = j < wh ? r[j] : cconj(r[n - j]);
        }
        if (blu) v.c = a.pl.chirp[j];
        return v;
    };
    auto fill_use = [&](int idx, const Slot &v) { buf[idx] = load_value<INV>(v.z, v.c, blu); };
    const bool wide = n >= kThreads;
    if (wide)
        for_rows(lines, n, pitch, [&](int l, int j, int) { return fill_load(l, j); },
                 [&](int, int, int, int idx, const Slot &v) { fill_use(idx, v); });
    else
        for_slots(lines * n, [&](int e) { const int l = fast_div(e, inv_n); return fill_load(l, e - mul24(l, n)); },
                  [&](int e, const Slot &v) { const int l = fast_div(e, inv_n); fill_use(mul24(l, pitch) + phys(e - mul24(l, n)), v); });
    if (blu) {                                     // zero padding [n, m)
        const int pad = m - n, totz = lines * pad;
        const float inv_pad = 1.0f / (float)pad;
        for (int e = tid; e < totz; e += kThreads) {
            const int l = fast_div(e, inv_pad), j = n + e - l * pad;
            buf[l * pitch + phys(j)] = make_float2(0.0f, 0.0f);
        }
    }
    lds_barrier();
    fft_lines(buf, lines, pitch, a.pl, twl);
    const int nout = STORE == kStoreHalf ? wh : n;
    const float inv_o = 1.0f / (float)nout;
    auto drain_load = [&](int j) {
        Slot v;
        if (blu) v.c = a.pl.chirp[j];
        return v;
    };
    auto drain_use = [&](int l, int j, int idx, const Slot &v) {
        const float2 z = store_value<INV>(buf[idx], v.c, blu);
        if (STORE == kStoreReal) (static_cast<float *>(a.dst) + (row0 + l) * a.dst_pitch)[j] = z.x * a.scale;
        else (static_cast<float2 *>(a.dst) + (row0 + l) * a.dst_pitch)[j] = make_float2(z.x * a.scale, z.y * a.scale);
    };
    if (nout >= kThreads)
        for_rows(lines, nout, pitch, [&](int, int j, int) { return drain_load(j); },
                 [&](int l, int j, int, int idx, const Slot &v) { drain_use(l, j, idx, v); });
    else
        for_slots(lines * nout, [&](int e) { return drain_load(e - mul24(fast_div(e, inv_o), nout)); },
                  [&](int e, const Slot &v) {
                      const int l = fast_div(e, inv_o), j = e - mul24(l, nout);
                      drain_use(l, j, mul24(l, pitch) + phys(j), v);
                  });
}

template <bool INV>
__global__ __launch_bounds__(kThreads, kThreads / 128) void fft_cols_kernel(const ColArgs a) {
    extern __shared__ float2 buf[];
    const int h = a.pl.n, m = a.pl.m, tid = threadIdx.x, C = a.tile;
    const int pitch = ((padded_length(m) + 31) & ~31) + (C < 32 ? 32 / C : 1);
    const int v0 = blockIdx.x * C;
    const int lines = a.cols - v0 < C ? a.cols - v0 : C;
    float2 *plane = a.data + (size_t)blockIdx.y * h * a.ld;
    const int shift = __ffs(C) - 1;
    const bool blu = a.pl.bluestein != 0;
    float2 *twl = buf + (size_t)C * pitch;
    load_twiddles(twl, a.pl);
    const int ccm = tid & (C - 1);
    const bool ccok = ccm < lines;
    const int tb = mul24(tid >> shift, a.ld) + v0 + (ccok ? ccm : 0);      // (for_tile: this thread's column)
    for_tile(h, shift, pitch,
             [&](int u, int uq, int) {
                 Slot v;
                 v.z = plane[tb + mul24(uq, a.ld)];
                 if (blu) v.c = a.pl.chirp[u];
                 return v;
             },
             [&](int, int, int, int idx, const Slot &v) {
                 if (ccok) buf[idx] = load_value<INV>(v.z, v.c, blu);
             });
    if (blu) {
        const int totz = (m - h) * C;
        for (int e = tid; e < totz; e += kThreads) {
            const int u = h + (e >> shift), cc = e & (C - 1);
            if (cc < lines) buf[cc * pitch + phys(u)] = make_float2(0.0f, 0.0f);
        }
    }
    lds_barrier();
    fft_lines(buf, lines, pitch, a.pl, twl);
    for_tile(h, shift, pitch,
             [&](int u, int, int) {
                 Slot v;
                 if (blu) v.c = a.pl.chirp[u];
                 return v;
             },
             [&](int, int uq, int, int idx, const Slot &v) {
                 if (ccok) {
                     const float2 z = store_value<INV>(buf[idx], v.c, blu);
                     plane[tb + mul24(uq, a.ld)] = make_float2(z.x * a.scale, z.y * a.scale);
                 }
             });
}

template <auto kernel>
void allow_lds() {      // > 64 KiB of dynamic LDS needs the attribute, per KERNEL (not per signature) and device (idempotent)
    static bool done[vfi::kMaxDevices] = {};
    bool &d = done[vfi::current_device()];
    if (!d) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(kLdsElemsMax * sizeof(float2)));
        d = true;
    }
}

template <int LOAD, int STORE>
void launch_rows_dir(const RowArgs &a, bool inverse, dim3 grid, size_t lds, hipStream_t s) {
    allow_lds<fft_rows_kernel<LOAD, STORE, true>>();
    allow_lds<fft_rows_kernel<LOAD, STORE, false>>();
    if (inverse) hipLaunchKernelGGL((fft_rows_kernel<LOAD, STORE, true>), grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((fft_rows_kernel<LOAD, STORE, false>), grid, dim3(kThreads), lds, s, a);
}

}  // namespace

int make_plan(int n, Plan1D *out, void (*own)(void *ctx, void *dev), void *ctx) {
    Plan1D pl{};
    pl.n = n;
    pl.bluestein = factor_smooth(n, pl.radix, &pl.nstages) ? 0 : 1;
    pl.m = pl.bluestein ? bluestein_length(n) : n;
    pl.tw_len = twiddle_entries(pl.m);
    if (pl.m > kMaxElems) return vfi::fail(VFI_ERR_UNSUPPORTED, "FFT length %d needs %d LDS elements (max %d)", n, pl.m, kMaxElems);
    if (pl.bluestein && !factor_bluestein(pl.m, pl.radix, &pl.nstages)) return vfi::fail(VFI_ERR_UNSUPPORTED, "FFT length %d", n);
    if (pl.tw_len < min_twiddle_entries(pl)) pl.tw_len = min_twiddle_entries(pl);
    if (max_lines(pl) < 1)       // one line + its twiddle table must fit a workgroup's LDS budget (non-smooth lengths above 2048)
        return vfi::fail(VFI_ERR_UNSUPPORTED, "FFT length %d (transformed as %d points) exceeds the LDS budget of the engine", n, pl.m);
    auto upload = [&](const std::vector<cd> &v, const float2 **dev) -> int {
        std::vector<float2> f(v.size());
        for (size_t i = 0; i < v.size(); ++i) f[i] = make_float2((float)v[i].real(), (float)v[i].imag());
        void *d = nullptr;
        if (hipMalloc(&d, f.size() * sizeof(float2)) != hipSuccess) return vfi::fail(VFI_ERR_NOMEM, "FFT tables: device allocation");
        own(ctx, d);
        if (hipMemcpy(d, f.data(), f.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
            return vfi::fail(VFI_ERR_LAUNCH, "FFT tables: upload");
        *dev = static_cast<const float2 *>(d);
        return VFI_OK;
    };
    std::vector<cd> tw(pl.m);
    for (int k = 0; k < pl.m; ++k) tw[k] = std::polar(1.0, -2.0 * kPi * (double)k / (double)pl.m);
    int rc = upload(tw, &pl.tw);
    if (rc) return rc;
    if (pl.bluestein) {
        std::vector<cd> w(n), b(pl.m, cd(0.0, 0.0));
        for (int j = 0; j < n; ++j) w[j] = std::polar(1.0, -kPi * (double)(((long long)j * j) % (2LL * n)) / (double)n);
        b[0] = std::conj(w[0]);
        for (int j = 1; j < n; ++j) b[j] = b[pl.m - j] = std::conj(w[j]);
        host_fft_bluestein(b);
        for (auto &v : b) v /= (double)pl.m;
        if ((rc = upload(w, &pl.chirp)) || (rc = upload(b, &pl.bfilt))) return rc;
    }
    *out = pl;
    return VFI_OK;
}

// rows per workgroup: as many as fit, but keep >= ~1024 workgroups in the launch when the array is large enough
int rows_per_group(const Plan1D &pl, long long total_rows) {
    long long l = max_lines(pl), want = (total_rows + 1023) / 1024;
    if (want < 1) want = 1;
    if (l > want) l = want;
    // a thread should still hold a handful of butterflies: at least ~2048 elements per workgroup when rows are short
    const long long min_l = (2048 + pl.m - 1) / pl.m;
    if (l < min_l) l = min_l;
    if (l > max_lines(pl)) l = max_lines(pl);
    if (l > total_rows) l = total_rows;
    if (l >= 4) l -= l % 4;         // a multiple of the wave count: the stages then run wave-private (no workgroup barriers)
    return (int)(l < 1 ? 1 : l);
}
int cols_per_group(const Plan1D &pl, int cols) {
    int c = 1;
    while (c < cols && c * 2 <= 32 && c * 2 * pl.m <= max_elems(pl) && c * 2 * col_pitch(pl, c * 2) + pl.tw_len <= kLdsElems) c *= 2;
    return c;
}

int launch_rows(const RowArgs &a, RowLoad load, RowStore store, bool inverse, hipStream_t s) {
    const dim3 grid((unsigned)((a.rows + a.lines - 1) / a.lines));
    const size_t lds = row_lds_bytes(a.pl, a.lines);
    if (load == kLoadComplex && store == kStoreComplex) launch_rows_dir<kLoadComplex, kStoreComplex>(a, inverse, grid, lds, s);
    else if (load == kLoadReal && store == kStoreComplex) launch_rows_dir<kLoadReal, kStoreComplex>(a, inverse, grid, lds, s);
    else if (load == kLoadReal && store == kStoreHalf) launch_rows_dir<kLoadReal, kStoreHalf>(a, inverse, grid, lds, s);
    else if (load == kLoadComplex && store == kStoreReal) launch_rows_dir<kLoadComplex, kStoreReal>(a, inverse, grid, lds, s);
    else if (load == kLoadHalf && store == kStoreReal) launch_rows_dir<kLoadHalf, kStoreReal>(a, inverse, grid, lds, s);
    else return vfi::fail(VFI_ERR_UNSUPPORTED, "fft row pass: load %d / store %d", (int)load, (int)store);
    return vfi::check_launch("fft row pass");
}

int launch_cols(const ColArgs &a, bool inverse, hipStream_t s) {
    const dim3 grid((unsigned)vfi::ceil_div(a.cols, a.tile), (unsigned)a.planes);
    const size_t lds = col_lds_bytes(a.pl, a.tile);
    allow_lds<fft_cols_kernel<true>>();
    allow_lds<fft_cols_kernel<false>>();
    if (inverse) hipLaunchKernelGGL(fft_cols_kernel<true>, grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL(fft_cols_kernel<false>, grid, dim3(kThreads), lds, s, a);
    return vfi::check_launch("fft column pass");
}

}  // namespace fft
}  // namespace vfi
