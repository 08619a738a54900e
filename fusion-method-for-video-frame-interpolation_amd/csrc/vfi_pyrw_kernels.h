// The steerable pyramid's level kernels on the wave-private FFT engine (vfi_wfft.h): templates over the engine
// configuration, instantiated per engine length in vfi_pyrw_rows.hip / vfi_pyrw_cols.hip.
//
//   analysis  level k :  ana_cols  : (image, band, L adjacent columns) per wave: window_k of the R2C half spectrum S
//                                    (Hermitian half expanded by index arithmetic) * Q_k[b] * i -> inverse column FFT -> T
//                        rows_polar: L rows of T per wave -> inverse row FFT -> 1/(hw) -> atan2 / hypot -> the caller's planes
//                                    (coeff_to_values, src/train/pyramid.py:63-69)
//   synthesis level k :  rows_from_polar: (phase, amplitude) rows -> A cos p, A sin p -> forward row FFT -> T
//                                    (values_to_coeff, src/train/pyramid.py:99-107)
//                        syn_cols  : (image, L adjacent columns) per wave, the four bands one after the other:
//                                    cur = sum_b (-i) * FFTcol(T_b) * P_s[b] + embedded coarser level
// Every wave is on its own: loads straight into the first stage's registers, stores straight from the last stage's
// (vfi_wfft.h); a workgroup only shares the small tables (stage twiddles; Bluestein's chirp and filter) in LDS.
// Roofline: HBM -- per coefficient 8 B of T written and read once plus the 8 B of (phase, amplitude).
#pragma once
#include <mutex>
#include "vfi_pyramid_wave.h"
#include "vfi_wfft.h"
#include "vfi_wfft_configs.h"

namespace vfi {
namespace pyrw {

using namespace vfi::wfft;
constexpr int kMaxThreads = 512;

__device__ __forceinline__ int signed_freq(int u, int h) { return u <= h - h / 2 - 1 ? u : u - h; }

// LDS of a workgroup: [stage twiddles][Bluestein filter M][chirp M/2][16 ints][waves x exchange buffer]
template <class C, bool BLU> struct Lds {
    static constexpr int kTables = C::TW + (BLU ? C::M + C::M / 2 : 0);       // float2 entries
    static size_t bytes(int waves) { return (size_t)kTables * sizeof(float2) + 16 * sizeof(int) + (size_t)waves * C::XBUF * sizeof(float); }
    const float2 *tw, *bf, *ch;
    int *ints;
    float *xb;                 // this wave's exchange buffer
    __device__ __forceinline__ Lds(float2 *lds, const Tables &tb, int wave) {
        float2 *t = lds, *b = lds + C::TW, *c = b + (BLU ? C::M : 0);
        for (int k = threadIdx.x; k < C::TW; k += blockDim.x) t[k] = tb.tw[k];
        if (BLU) {
            for (int k = threadIdx.x; k < C::M; k += blockDim.x) b[k] = tb.bfilt[k];
            for (int k = threadIdx.x; k < tb.n; k += blockDim.x) c[k] = tb.chirp[k];
            // the tail [n, M/2) is read (and multiplied with zero padding) by the lanes whose positions lie past the
            // transform: it must hold finite numbers, not whatever the previous workgroup left in LDS
            for (int k = tb.n + threadIdx.x; k < C::M / 2; k += blockDim.x) c[k] = make_float2(0.0f, 0.0f);
        }
        tw = t; bf = b; ch = c;
        ints = reinterpret_cast<int *>(c + (BLU ? C::M / 2 : 0));
        xb = reinterpret_cast<float *>(ints + 16) + wave * C::XBUF;
    }
};

// ---- global memory through buffer descriptors -------------------------------------------------------------------------
// Every global access of these kernels is `descriptor (SGPRs, one per plane / row batch) + one 32-bit VGPR offset per
// round of butterflies + a compile-time or uniform constant`: no 64-bit per-element addresses (they cost two VGPRs per
// element in flight), and idle lanes pass kOob, which the range check turns into "load 0 / drop the store" without a branch.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOob = 0x80000000u;        // beyond num_records (2^31 - 1): every plane / batch here is far smaller
template <typename T>
__device__ __forceinline__ rsrc_t rsrc_of(const T *p) {      // p must be wave-uniform (made provably so for the compiler)
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float2 ld2(rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 d = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(d.x), __uint_as_float(d.y));
}
__device__ __forceinline__ float ld1(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void st2(rsrc_t r, unsigned voff, unsigned soff, float2 v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 d;
    d.x = __float_as_uint(v.x);
    d.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
}
__device__ __forceinline__ void st1(rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// ---- polar conversion on the hardware's fast paths -----------------------------------------------------------------------
// atan2 of libm costs ~55 instructions and a dozen registers per coefficient -- as much as a whole FFT stage.  Here:
// a = min/max through v_rcp_f32, atan(a) = a * p(a^2) with a degree-8 polynomial (fitted to minimise the absolute error of
// a * p: 9.2e-8 in fp32 evaluation on [0, 1], tools/gen_wfft_configs.py's sibling one-liner in DESIGN.md section 7), then the
// octant fix-ups; <= 3e-7 rad absolute, i.e. one ulp of pi.  atan2(+-0, x < 0) = +-pi and atan2(0, 0) = 0 as in libm.
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y), mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
    const float s = a * a;
    float p = 0.0024566983338445425f;
    p = fmaf(p, s, -0.01440124586224556f);
    p = fmaf(p, s, 0.03978102654218674f);
    p = fmaf(p, s, -0.07234838604927063f);
    p = fmaf(p, s, 0.10498936474323273f);
    p = fmaf(p, s, -0.14161226153373718f);
    p = fmaf(p, s, 0.19985906779766083f);
    p = fmaf(p, s, -0.33332598209381104f);
    p = fmaf(p, s, 0.9999998807907104f);
    float r = a * p;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return copysignf(r, y);
}
// |z| through v_sqrt_f32 (1 ulp; libm's correctly rounded sqrtf adds a Newton step and denormal scaling per coefficient)
__device__ __forceinline__ float fast_hypot(float re, float im) { return __builtin_amdgcn_sqrtf(fmaf(re, re, im * im)); }

template <class C, bool BLU>
__device__ __forceinline__ void transform(float2 (&v)[C::E], int lane, const Lds<C, BLU> &m) {
#ifdef VFI_PYRW_PROBE_NO_FFT      // elimination build (timing only, results are wrong): the passes without their transforms
    return;
#endif
    forward<C>(v, lane, m.xb, m.tw);
    if (BLU) bluestein_middle<C>(v, lane, m.xb, m.tw, m.bf);
}

// =====================================================================================================================
// rows: a batch = L consecutive rows of ONE plane (so every descriptor of a batch is wave-uniform)
// =====================================================================================================================
template <class C, bool BLU>
__global__ __launch_bounds__(C::TEAM > 64 ? C::TEAM : kMaxThreads, 2) void rows_polar_kernel(const RowsArgs a) {
    using I = Io<C>;
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x % C::TEAM, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / C::TEAM)), nw = blockDim.x / C::TEAM;   // (wave = index of this lane's team)
    Lds<C, BLU> m(lds, a.tb, wave);
    if (threadIdx.x < kMaxImages) m.ints[threadIdx.x] = a.pm.idx[threadIdx.x];
    __syncthreads();
    const int n = a.w, h = a.h, ngrp = (h + C::L - 1) / C::L, nbatch = a.planes * ngrp;
    float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool want_max = a.amp_max != nullptr;
    for (int b = blockIdx.x * nw + wave; b < nbatch; b += gridDim.x * nw) {
        const int plane = b / ngrp, y0 = (b - plane * ngrp) * C::L, img = plane / kBands, band = plane - img * kBands;
        const rsrc_t rT = rsrc_of(a.T + ((size_t)plane * h + y0) * a.tpitch);
        float2 v[C::E];
#pragma unroll
        for (int q = 0; q < I::Q0; ++q) {
            int l, i;
            bool ok;
            lane_index<C, 0>(lane, q, l, i, ok);
            ok = ok && y0 + l < h;
            const unsigned vo = ok ? (unsigned)(l * a.tpitch + i) * 8u : kOob;
#pragma unroll
            for (int r = 0; r < I::R0; ++r) {
                if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                const int pos = i + r * I::T0;
                // (Bluestein: positions past the row are zero padding -- kOob reads 0 and never leaves the buffer)
                const float2 x = fft::load_value<true>(ld2(rT, BLU && pos >= n ? kOob : vo, r * I::T0 * 8),
                                                       BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                v[q * I::R0 + r] = (!BLU || pos < n) ? x : make_float2(0.0f, 0.0f);
            }
        }
        transform<C, BLU>(v, lane, m);
        const size_t obase = ((size_t)(m.ints[img] + band * a.pm.band_stride) * h + y0) * n;
        const int grp = img % a.groups;
        if (a.pm.complex_coeff) {
            const rsrc_t rC = rsrc_of(reinterpret_cast<float2 *>(a.phase) + obase);
#pragma unroll
            for (int q = 0; q < I::QL; ++q) {
                int l, k;
                bool ok;
                lane_index<C, I::SL>(lane, q, l, k, ok);
                ok = ok && y0 + l < h;
#pragma unroll
                for (int r = 0; r < I::RL; ++r) {
                    if (BLU && r >= I::RL_BLU) continue;
                    const int pos = k + r * I::PL;
                    const float2 z = fft::store_value<true>(v[q * I::RL + r], BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                    const unsigned vo = ok && (!BLU || pos < n) ? (unsigned)(l * n + k) * 8u : kOob;
                    st2(rC, vo, r * I::PL * 8, make_float2(z.x * a.inv_hw, z.y * a.inv_hw));
                }
            }
        } else {
            const rsrc_t rP = rsrc_of(a.phase + obase), rA = rsrc_of(a.amp + obase);
            float bmax = 0.0f;      // largest amplitude this lane writes in this batch (one plane, hence one image group)
#pragma unroll
            for (int q = 0; q < I::QL; ++q) {
                int l, k;
                bool ok;
                lane_index<C, I::SL>(lane, q, l, k, ok);
                ok = ok && y0 + l < h;
#pragma unroll
                for (int r = 0; r < I::RL; ++r) {
                    if (BLU && r >= I::RL_BLU) continue;
                    const int pos = k + r * I::PL;
                    const float2 z = fft::store_value<true>(v[q * I::RL + r], BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                    const float re = z.x * a.inv_hw, im = z.y * a.inv_hw;
                    const bool valid = ok && (!BLU || pos < n);
                    const unsigned vo = valid ? (unsigned)(l * n + k) * 4u : kOob;
                    const float am = fast_hypot(re, im);
                    st1(rP, vo, r * I::PL * 4, fast_atan2(im, re) * a.phase_scale);
                    st1(rA, vo, r * I::PL * 4, am);
                    if (want_max) bmax = valid ? fmaxf(bmax, am) : bmax;
                }
            }
            if (want_max) {
#pragma unroll
                for (int t = 0; t < 4; ++t) gmax[t] = t == grp ? fmaxf(gmax[t], bmax) : gmax[t];
            }
        }
    }
    if (want_max) {        // 64-lane butterfly, then one atomic per wave and group (amplitudes are >= 0: their bit patterns order like the values)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float mx = gmax[t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if ((threadIdx.x & 63) == 0 && t < a.groups && mx > 0.0f) atomicMax(a.amp_max + t, __float_as_uint(mx));
        }
    }
}

template <class C, bool BLU>
__global__ __launch_bounds__(C::TEAM > 64 ? C::TEAM : kMaxThreads, 2) void rows_from_polar_kernel(const RowsArgs a) {
    using I = Io<C>;
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x % C::TEAM, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / C::TEAM)), nw = blockDim.x / C::TEAM;   // (wave = index of this lane's team)
    Lds<C, BLU> m(lds, a.tb, wave);
    if (threadIdx.x < kMaxImages) m.ints[threadIdx.x] = a.pm.idx[threadIdx.x];
    __syncthreads();
    const int n = a.w, h = a.h, ngrp = (h + C::L - 1) / C::L, nbatch = a.planes * ngrp;
    for (int b = blockIdx.x * nw + wave; b < nbatch; b += gridDim.x * nw) {
        const int plane = b / ngrp, y0 = (b - plane * ngrp) * C::L, img = plane / kBands, band = plane - img * kBands;
        const size_t ibase = ((size_t)(m.ints[img] + band * a.pm.band_stride) * h + y0) * n;
        float2 v[C::E];
        if (a.pm.complex_coeff) {
            const rsrc_t rC = rsrc_of(reinterpret_cast<const float2 *>(a.phase) + ibase);
#pragma unroll
            for (int q = 0; q < I::Q0; ++q) {
                int l, i;
                bool ok;
                lane_index<C, 0>(lane, q, l, i, ok);
                ok = ok && y0 + l < h;
                const unsigned vo = ok ? (unsigned)(l * n + i) * 8u : kOob;
#pragma unroll
                for (int r = 0; r < I::R0; ++r) {
                    if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                    const int pos = i + r * I::T0;
                    const float2 x = fft::load_value<false>(ld2(rC, BLU && pos >= n ? kOob : vo, r * I::T0 * 8),
                                                            BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                    v[q * I::R0 + r] = (!BLU || pos < n) ? x : make_float2(0.0f, 0.0f);
                }
            }
        } else {
            const rsrc_t rP = rsrc_of(a.phase + ibase), rA = rsrc_of(a.amp + ibase);
#pragma unroll
            for (int q = 0; q < I::Q0; ++q) {
                int l, i;
                bool ok;
                lane_index<C, 0>(lane, q, l, i, ok);
                ok = ok && y0 + l < h;
                const unsigned vo = ok ? (unsigned)(l * n + i) * 4u : kOob;
#pragma unroll
                for (int r = 0; r < I::R0; ++r) {
                    if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                    const int pos = i + r * I::T0;
                    const unsigned ve = BLU && pos >= n ? kOob : vo;          // (positions past the row read 0: amplitude 0)
                    // sin / cos on the hardware units (v_sin_f32 / v_cos_f32 take REVOLUTIONS and are only valid up to 256 of
                    // them, with an absolute error that grows with the argument: <= 4e-7 for |p| <= pi).  The phase is a
                    // caller's value (vfi_pyr_synthesize), so it is reduced first: r - rint(r) is exact in fp32 and leaves
                    // |p| <= pi untouched (same bits as without it); libm's sincosf would carry a Payne-Hanek path and ~40
                    // registers into the load phase.
                    const float ph = ld1(rP, ve, r * I::T0 * 4);
                    float rev = ph * 0.15915494309189535f;
                    rev -= rintf(rev);
                    const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
                    const float am = ld1(rA, ve, r * I::T0 * 4);
                    const float2 x = fft::load_value<false>(make_float2(cs * am, sn * am), BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                    v[q * I::R0 + r] = (!BLU || pos < n) ? x : make_float2(0.0f, 0.0f);
                }
            }
        }
        transform<C, BLU>(v, lane, m);
        const rsrc_t rT = rsrc_of(a.T + ((size_t)plane * h + y0) * a.tpitch);
#pragma unroll
        for (int q = 0; q < I::QL; ++q) {
            int l, k;
            bool ok;
            lane_index<C, I::SL>(lane, q, l, k, ok);
            ok = ok && y0 + l < h;
#pragma unroll
            for (int r = 0; r < I::RL; ++r) {
                if (BLU && r >= I::RL_BLU) continue;
                const int pos = k + r * I::PL;
                const float2 z = fft::store_value<false>(v[q * I::RL + r], BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                st2(rT, ok && (!BLU || pos < n) ? (unsigned)(l * a.tpitch + k) * 8u : kOob, r * I::PL * 8, z);
            }
        }
    }
}

// =====================================================================================================================
// columns: a batch = L adjacent columns of one plane
// =====================================================================================================================
// XCD-aware item order: workgroup b runs on XCD b % 8 (own L2): the items an XCD works on at one time are neighbours
// (the bands of one column tile, then the next tile), so the 128-byte lines they share stay in one L2.
__device__ __forceinline__ int xcd_item(int s, int per) { return (s & 7) * per + (s >> 3); }

template <class C, bool BLU>
__global__ __launch_bounds__(C::TEAM > 64 ? C::TEAM : kMaxThreads, 2) void ana_cols_kernel(const AnaColsArgs a) {
    using I = Io<C>;
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x % C::TEAM, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / C::TEAM)), nw = blockDim.x / C::TEAM;   // (wave = index of this lane's team)
    Lds<C, BLU> m(lds, a.tb, wave);
    __syncthreads();
    const int h = a.h, w = a.w, H = a.H, tilew = nw * C::L, ntile = (w + tilew - 1) / tilew;
    const int nitem = a.N * ntile * kBands, per = (nitem + 7) >> 3;
    const int hpos = h - h / 2;                                   // rows [0, hpos) hold fy >= 0
    constexpr int RX = BLU ? I::R0_BLU : I::R0;                   // first-stage inputs that are loaded (the others are zero padding)
    for (int s = blockIdx.x; s < 8 * per; s += gridDim.x) {
        const int item = xcd_item(s, per);
        if (item >= nitem) continue;
        // item order: band fastest, then image, then column tile -- the 4 N items of one tile follow each other on one XCD, so the
        // spectrum columns of an image (shared by its four bands) and the gain columns of a band (shared by the N images) can
        // stay in that XCD's L2 (time-neutral against the image-major order: the pass is bound by its CU <-> L2 traffic,
        // 1.0 GB per call at 1080p of which 0.5 GB is unique, not by HBM)
        const int band = item & (kBands - 1), img = (item >> 2) % a.N, tile = (item >> 2) / a.N;
        const int col0 = tile * tilew + wave * C::L;
        if (col0 >= w) continue;                                  // (wave-uniform: this wave's columns lie outside)
        const int col = col0 + lane % C::L;
        const bool colok = col < w;
        const int fx = signed_freq(colok ? col : 0, w);
        const bool neg = fx < 0;
        const rsrc_t rS = rsrc_of(a.S + (size_t)img * H * a.spitch);
        const rsrc_t rQ = rsrc_of(a.Q + (size_t)band * h * w);
        const rsrc_t rT = rsrc_of(a.T + (size_t)(img * kBands + band) * h * a.tpitch);
        const int sV = neg ? -fx : fx;
        float2 v[C::E];
#pragma unroll
        for (int q = 0; q < I::Q0; ++q) {
            int l, i;
            bool ok;
            lane_index<C, 0>(lane, q, l, i, ok);
            ok = ok && colok;
            const unsigned vq = ok ? (unsigned)(i * w + col) * 4u : kOob;
#pragma unroll
            for (int r = 0; r < I::R0; ++r) {
                if (r >= RX) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                const int u = i + r * I::T0;                      // row of the level's window (unshifted order)
                // row of the full-size half spectrum: fy = u (u < hpos) or u - h; mirrored (-fy) for the columns with fx < 0
                const int U = neg ? (u < hpos ? (u == 0 ? 0 : H - u) : h - u) : (u < hpos ? u : u + (H - h));
                const bool valid = ok && (!BLU || u < h);
                float2 z = ld2(rS, valid ? (unsigned)(U * a.spitch + sV) * 8u : kOob, 0);
                const float g = ld1(rQ, valid ? vq : kOob, (unsigned)(r * I::T0) * (unsigned)w * 4u);
                if (neg) z.y = -z.y;
                // * i : (re, im) -> (-im, re); inverse transform: conjugate in.  (Out-of-range loads returned 0.)
                const float2 x = fft::load_value<true>(make_float2(-(z.y * g), z.x * g), BLU ? m.ch[u] : make_float2(0.0f, 0.0f), BLU);
                v[q * I::R0 + r] = (!BLU || u < h) ? x : make_float2(0.0f, 0.0f);
            }
        }
        transform<C, BLU>(v, lane, m);
#pragma unroll
        for (int q = 0; q < I::QL; ++q) {
            int l, k;
            bool ok;
            lane_index<C, I::SL>(lane, q, l, k, ok);
            ok = ok && colok;
#pragma unroll
            for (int r = 0; r < I::RL; ++r) {
                if (BLU && r >= I::RL_BLU) continue;
                const int pos = k + r * I::PL;
                const float2 z = fft::store_value<true>(v[q * I::RL + r], BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                st2(rT, ok && (!BLU || pos < h) ? (unsigned)(k * a.tpitch + col) * 8u : kOob, (unsigned)(r * I::PL) * (unsigned)a.tpitch * 8u, z);
            }
        }
    }
}

// Workgroup = 4 waves = the four bands of the same L columns: every wave transforms its band's columns, multiplies by
// its mask, and the four results are summed through LDS in a fixed order (band 0 + 1 + 2 + 3: deterministic); each wave
// then adds the embedded coarser level to a quarter of the sums and stores them.
template <class C, bool BLU>
__global__ __launch_bounds__(256, 2) void syn_cols_kernel(const SynColsArgs a) {
    using I = Io<C>;
    static_assert(C::E * wfft::kWave <= C::XBUF, "the exchange buffer doubles as E x 64 scratch words");
    static_assert(C::TEAM == wfft::kWave, "the synthesis workgroup is four independent waves, one per band");
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x & 63, band = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Lds<C, BLU> m(lds, a.tb, band);
    const float *xall = m.xb - band * C::XBUF;                    // the four waves' buffers
    __syncthreads();
    const int h = a.h, w = a.w, h2 = a.h2, w2 = a.w2, ntile = (w + C::L - 1) / C::L;
    const int nitem = a.N * ntile, per = (nitem + 7) >> 3;
    constexpr int EQ = (C::E + kBands - 1) / kBands;              // sums per wave
    for (int s = blockIdx.x; s < 8 * per; s += gridDim.x) {
        const int item = xcd_item(s, per);
        if (item >= nitem) continue;                              // (workgroup-uniform)
        const int img = item % a.N, tile = item / a.N;            // (image fastest: the mask columns of a tile stay in the XCD's L2 for all N images)
        const int col = tile * C::L + lane % C::L;
        const bool colok = col < w;
        const rsrc_t rT = rsrc_of(a.T + (size_t)(img * kBands + band) * h * a.tpitch);
        const rsrc_t rP = rsrc_of(a.P + (size_t)band * h * w);
        float2 v[C::E];
#pragma unroll
        for (int q = 0; q < I::Q0; ++q) {
            int l, i;
            bool ok;
            lane_index<C, 0>(lane, q, l, i, ok);
            ok = ok && colok;
            const unsigned vo = ok ? (unsigned)(i * a.tpitch + col) * 8u : kOob;
#pragma unroll
            for (int r = 0; r < I::R0; ++r) {
                if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                const int u = i + r * I::T0;
                const float2 x = fft::load_value<false>(ld2(rT, BLU && u >= h ? kOob : vo, (unsigned)(r * I::T0) * (unsigned)a.tpitch * 8u),
                                                        BLU ? m.ch[u] : make_float2(0.0f, 0.0f), BLU);
                v[q * I::R0 + r] = (!BLU || u < h) ? x : make_float2(0.0f, 0.0f);
            }
        }
        transform<C, BLU>(v, lane, m);
        // (-i) * FFTcol(T_b) * P_s[b], in place (0 wherever nothing is to be stored)
#pragma unroll
        for (int q = 0; q < I::QL; ++q) {
            int l, k;
            bool ok;
            lane_index<C, I::SL>(lane, q, l, k, ok);
            ok = ok && colok;
            const unsigned vp = ok ? (unsigned)(k * w + col) * 4u : kOob;
#pragma unroll
            for (int r = 0; r < I::RL; ++r) {
                const int u = k + r * I::PL;
                const bool valid = ok && (!BLU || (r < I::RL_BLU && u < h));
                const float2 z = fft::store_value<false>(v[q * I::RL + r], BLU ? m.ch[valid ? u : 0] : make_float2(0.0f, 0.0f), BLU);
                const float ps = ld1(rP, valid ? vp : kOob, (unsigned)(r * I::PL) * (unsigned)w * 4u);
                // * (-i) : (re, im) -> (im, -re)
                v[q * I::RL + r] = valid ? make_float2(z.y * ps, -(z.x * ps)) : make_float2(0.0f, 0.0f);
            }
        }
        // sum over the bands: element e = q * RL + r of lane `lane` sits at word e * 64 + lane of each wave's buffer
        float2 sum[EQ];
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
#pragma unroll
            for (int e = 0; e < C::E; ++e) m.xb[e * wfft::kWave + lane] = comp ? v[e].y : v[e].x;
            fft::lds_barrier();
#pragma unroll
            for (int j = 0; j < EQ; ++j) {
                const float *x = xall + (kBands * j + band) * wfft::kWave + lane;      // (e = 4 j + band; e >= E reads scratch that is never stored)
                const float t = ((x[0] + x[C::XBUF]) + x[2 * C::XBUF]) + x[3 * C::XBUF];
                if (comp) sum[j].y = t; else sum[j].x = t;
            }
            fft::lds_barrier();
        }
        // + the coarser level's window, embedded (reconstruct's `resdft`), and out
        const int fx = signed_freq(colok ? col : 0, w);
        const bool xin = a.res != nullptr && fx >= -(w2 / 2) && fx <= w2 - w2 / 2 - 1;
        const int v2 = fx < 0 ? fx + w2 : fx;
        const rsrc_t rCur = rsrc_of(a.cur + (size_t)img * h * w);
        const rsrc_t rRes = rsrc_of(a.res ? a.res + (size_t)img * h2 * w2 : a.cur), rLo = rsrc_of(a.res ? a.lomask : a.P);
#pragma unroll
        for (int j = 0; j < EQ; ++j) {
            const int e = kBands * j + band, q = e / I::RL, r = e - q * I::RL;      // (uniform)
            const int id = lane + wfft::kWave * q, k = id / C::L, u = k + r * I::PL;     // (column mode: line = id % L = this lane's column)
            const bool valid = e < C::E && id < C::NB(I::SL) && colok && u < h;
            const int fy = signed_freq(valid ? u : 0, h);
            const bool in = valid && xin && fy >= -(h2 / 2) && fy <= h2 - h2 / 2 - 1;
            const unsigned o2 = (unsigned)((fy < 0 ? fy + h2 : fy) * w2 + v2);
            const float2 rz = ld2(rRes, in ? o2 * 8u : kOob, 0);
            const float lom = ld1(rLo, in ? o2 * 4u : kOob, 0);
            st2(rCur, valid ? (unsigned)(u * w + col) * 8u : kOob, 0, make_float2(sum[j].x + rz.x * lom, sum[j].y + rz.y * lom));
        }
    }
}

// =====================================================================================================================
// launch helpers
// =====================================================================================================================
struct Occupancy { int blocks = 0, cus = 0; };

// The kernel is a template ARGUMENT of every helper below, so each kernel instantiation has its own per-device cache
// (keyed on the function-pointer type, kernels that share a signature would share -- and skip -- the LDS attribute).
template <auto kernel>
inline Occupancy occupancy_of(int threads, size_t lds) {
    Occupancy o;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        o.blocks = -1;          // (the launch helper reports it)
        return o;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o.blocks, kernel, threads, lds) != hipSuccess || o.blocks < 1) o.blocks = 1;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) o.cus = prop.multiProcessorCount;
    if (o.cus < 1) o.cus = 256;
    return o;
}

struct Pick { int threads = 0, blocks = 0, cus = 0; };
// (one lock for the per-kernel, per-device geometry caches below: plans on different host threads launch the same kernels)
inline std::mutex &pick_mutex() {
    static std::mutex mu;
    return mu;
}

// row passes: waves per workgroup = the choice (4 or 8) that keeps more waves resident per CU (tables are per workgroup)
template <class C, bool BLU, auto kernel, typename A>
inline int launch_rows(const A &a, int nbatch, hipStream_t s) {
    static Pick cache[kMaxDevices];
    Pick p;
    {
        std::lock_guard<std::mutex> lock(pick_mutex());
        Pick &c = cache[current_device()];
        if (!c.threads) {
            if (C::TEAM > 64) {      // a team of waves is a workgroup of its own
                const Occupancy o = occupancy_of<kernel>(C::TEAM, Lds<C, BLU>::bytes(1));
                if (o.blocks < 0) return vfi::fail(VFI_ERR_LAUNCH, "pyramid wave row pass (M = %d): the kernel's LDS attribute was rejected", C::M);
                c.blocks = o.blocks; c.cus = o.cus; c.threads = C::TEAM;
            } else {
                const Occupancy o4 = occupancy_of<kernel>(256, Lds<C, BLU>::bytes(4)), o8 = occupancy_of<kernel>(512, Lds<C, BLU>::bytes(8));
                if (o4.blocks < 0 || o8.blocks < 0) return vfi::fail(VFI_ERR_LAUNCH, "pyramid wave row pass (M = %d): the kernel's LDS attribute was rejected", C::M);
                if (o8.blocks * 8 > o4.blocks * 4) { c.blocks = o8.blocks; c.cus = o4.cus; c.threads = 512; } else { c.blocks = o4.blocks; c.cus = o4.cus; c.threads = 256; }
            }
        }
        p = c;
    }
    const int nw = p.threads / C::TEAM;
    int grid = (nbatch + nw - 1) / nw;
    if (grid > p.blocks * p.cus) grid = p.blocks * p.cus;
    const size_t lds = Lds<C, BLU>::bytes(nw);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(p.threads), lds, s, a);
    return vfi::check_launch("pyramid wave row pass");
}

// column passes: 16 adjacent columns (one 128-byte line) per workgroup where the lines per wave allow it
template <class C, bool BLU, auto kernel, typename A>
inline int launch_cols(const A &a, int w, int items_per_tile, hipStream_t s) {
    static Pick cache[kMaxDevices];
    Pick p;
    {
        std::lock_guard<std::mutex> lock(pick_mutex());
        Pick &c = cache[current_device()];
        if (!c.threads) {
            const int threads = C::TEAM > 64 ? C::TEAM : (C::L * 4 >= 16 ? 256 : 512);
            const Occupancy o = occupancy_of<kernel>(threads, Lds<C, BLU>::bytes(threads / C::TEAM));
            if (o.blocks < 0) return vfi::fail(VFI_ERR_LAUNCH, "pyramid wave column pass (M = %d): the kernel's LDS attribute was rejected", C::M);
            c.blocks = o.blocks; c.cus = o.cus; c.threads = threads;
        }
        p = c;
    }
    const int nw = p.threads / C::TEAM, tilew = nw * C::L, ntile = (w + tilew - 1) / tilew;
    const int nitem = items_per_tile * ntile, per = (nitem + 7) / 8;
    int grid = 8 * per;
    const int cap = (p.blocks * p.cus) / 8 * 8;
    if (grid > cap) grid = cap;
    const size_t lds = Lds<C, BLU>::bytes(nw);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(p.threads), lds, s, a);
    return vfi::check_launch("pyramid wave column pass");
}

// synthesis columns: a workgroup is the four bands of L columns
template <class C, bool BLU, auto kernel>
inline int launch_syn(const SynColsArgs &a, hipStream_t s) {
    static Pick cache[kMaxDevices];
    Pick p;
    const size_t lds = Lds<C, BLU>::bytes(kBands);
    {
        std::lock_guard<std::mutex> lock(pick_mutex());
        Pick &c = cache[current_device()];
        if (!c.blocks) {
            const Occupancy o = occupancy_of<kernel>(256, lds);
            if (o.blocks < 0) return vfi::fail(VFI_ERR_LAUNCH, "pyramid wave synthesis columns (M = %d): the kernel's LDS attribute was rejected", C::M);
            c.blocks = o.blocks; c.cus = o.cus;
        }
        p = c;
    }
    const int ntile = (a.w + C::L - 1) / C::L, nitem = a.N * ntile, per = (nitem + 7) / 8;
    int grid = 8 * per;
    const int cap = (p.blocks * p.cus) / 8 * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, s, a);
    return vfi::check_launch("pyramid wave synthesis columns");
}

constexpr bool blu_capable(int m) {
    while (m % 2 == 0) m /= 2;
    return m == 1 || m == 3;
}

}  // namespace pyrw
}  // namespace vfi
