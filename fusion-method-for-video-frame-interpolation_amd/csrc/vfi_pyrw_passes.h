// Plain batched 1-D passes on the wave-private FFT engine (vfi_wfft.h): the un-fused transforms of the steerable pyramid
// (R2C of the images, C2R of the high residual and of the radial-filter shortcuts, the final inverse of the synthesis).
//   rows : L consecutive rows per wave; real / complex / Hermitian-half loads and complex / half / real stores are
//          variants of the same kernel, so R2C and C2R cost no extra pass;
//   cols : L adjacent columns per wave, in place.
// Same structure as the level kernels of vfi_pyrw_kernels.h.  Roofline: HBM (each pass reads and writes the array once).
#pragma once
#include "vfi_pyrw_kernels.h"

namespace vfi {
namespace pyrw {

template <class C, bool BLU, int LOAD, int STORE, bool INV>
__global__ __launch_bounds__(C::TEAM > 64 ? C::TEAM : kMaxThreads, 2) void gen_rows_kernel(const GenRowsArgs a) {
    using I = Io<C>;
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x % C::TEAM, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / C::TEAM)), nw = blockDim.x / C::TEAM;   // (wave = index of this lane's team)
    Lds<C, BLU> m(lds, a.tb, wave);
    __syncthreads();
    const int n = a.tb.n, wh = n / 2 + 1, nbatch = (a.rows + C::L - 1) / C::L;
    constexpr int IN_BYTES = LOAD == kGenReal ? 4 : 8, OUT_BYTES = STORE == kGenReal ? 4 : 8;
    for (int b = blockIdx.x * nw + wave; b < nbatch; b += gridDim.x * nw) {
        const int row0 = b * C::L;
        const rsrc_t rI = rsrc_of(static_cast<const char *>(a.src) + (size_t)row0 * a.src_pitch * IN_BYTES);
        const rsrc_t rO = rsrc_of(static_cast<char *>(a.dst) + (size_t)row0 * a.dst_pitch * OUT_BYTES);
        float2 v[C::E];
#pragma unroll
        for (int q = 0; q < I::Q0; ++q) {
            int l, i;
            bool ok;
            lane_index<C, 0>(lane, q, l, i, ok);
            ok = ok && row0 + l < a.rows;
            const unsigned vo = ok ? (unsigned)(l * a.src_pitch + i) * IN_BYTES : kOob;
#pragma unroll
            for (int r = 0; r < I::R0; ++r) {
                if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                const int pos = i + r * I::T0;
                const bool in = !BLU || pos < n;
                float2 x;
                if (LOAD == kGenReal) {
                    x = make_float2(ld1(rI, in ? vo : kOob, r * I::T0 * 4), 0.0f);
                } else if (LOAD == kGenComplex) {
                    x = ld2(rI, in ? vo : kOob, r * I::T0 * 8);
                } else {      // Hermitian half: entries [0, n/2] are stored, the others are conj(x[n - pos])
                    const bool lo = pos < wh;
                    x = ld2(rI, ok && in ? (unsigned)(l * a.src_pitch + (lo ? pos : n - pos)) * 8u : kOob, 0);
                    if (!lo) x.y = -x.y;
                }
                x = fft::load_value<INV>(x, BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                v[q * I::R0 + r] = in ? x : make_float2(0.0f, 0.0f);
            }
        }
        transform<C, BLU>(v, lane, m);
#pragma unroll
        for (int q = 0; q < I::QL; ++q) {
            int l, k;
            bool ok;
            lane_index<C, I::SL>(lane, q, l, k, ok);
            ok = ok && row0 + l < a.rows;
            const unsigned vo = ok ? (unsigned)(l * a.dst_pitch + k) * OUT_BYTES : kOob;
#pragma unroll
            for (int r = 0; r < I::RL; ++r) {
                if (BLU && r >= I::RL_BLU) continue;
                const int pos = k + r * I::PL;
                const float2 z = fft::store_value<INV>(v[q * I::RL + r], BLU ? m.ch[pos] : make_float2(0.0f, 0.0f), BLU);
                const bool keep = pos < (STORE == kGenHalf ? wh : n);
                if (STORE == kGenReal) st1(rO, keep ? vo : kOob, r * I::PL * 4, z.x * a.scale);
                else st2(rO, keep ? vo : kOob, r * I::PL * 8, make_float2(z.x * a.scale, z.y * a.scale));
            }
        }
    }
}

template <class C, bool BLU, bool INV>
__global__ __launch_bounds__(C::TEAM > 64 ? C::TEAM : kMaxThreads, 2) void gen_cols_kernel(const GenColsArgs a) {
    using I = Io<C>;
    extern __shared__ float2 lds[];
    const int lane = threadIdx.x % C::TEAM, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / C::TEAM)), nw = blockDim.x / C::TEAM;   // (wave = index of this lane's team)
    Lds<C, BLU> m(lds, a.tb, wave);
    __syncthreads();
    const int h = a.tb.n, tilew = nw * C::L, ntile = (a.cols + tilew - 1) / tilew;
    const int nitem = a.planes * ntile, per = (nitem + 7) >> 3;
    for (int s = blockIdx.x; s < 8 * per; s += gridDim.x) {
        const int item = xcd_item(s, per);
        if (item >= nitem) continue;
        const int tile = item % ntile, plane = item / ntile;
        const int col0 = tile * tilew + wave * C::L;
        if (col0 >= a.cols) continue;
        const int col = col0 + lane % C::L;
        const bool colok = col < a.cols;
        const rsrc_t rD = rsrc_of(a.data + (size_t)plane * h * a.ld);
        float2 v[C::E];
#pragma unroll
        for (int q = 0; q < I::Q0; ++q) {
            int l, i;
            bool ok;
            lane_index<C, 0>(lane, q, l, i, ok);
            const unsigned vo = ok && colok ? (unsigned)(i * a.ld + col) * 8u : kOob;
#pragma unroll
            for (int r = 0; r < I::R0; ++r) {
                if (BLU && r >= I::R0_BLU) { v[q * I::R0 + r] = make_float2(0.0f, 0.0f); continue; }
                const int u = i + r * I::T0;
                const float2 x = fft::load_value<INV>(ld2(rD, BLU && u >= h ? kOob : vo, (unsigned)(r * I::T0) * (unsigned)a.ld * 8u),
                                                      BLU ? m.ch[u] : make_float2(0.0f, 0.0f), BLU);
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
                const int u = k + r * I::PL;
                const float2 z = fft::store_value<INV>(v[q * I::RL + r], BLU ? m.ch[u] : make_float2(0.0f, 0.0f), BLU);
                st2(rD, ok && (!BLU || u < h) ? (unsigned)(k * a.ld + col) * 8u : kOob, (unsigned)(r * I::PL) * (unsigned)a.ld * 8u,
                    make_float2(z.x * a.scale, z.y * a.scale));
            }
        }
    }
}

}  // namespace pyrw
}  // namespace vfi
