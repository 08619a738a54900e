// Internal interface between vfi_pyramid.hip (plans, level loop) and the pyramid's level kernels on the wave-private FFT
// engine (vfi_wfft.h; kernels in vfi_pyrw_kernels.h, instantiated per engine length in vfi_pyrw_rows.hip /
// vfi_pyrw_cols.hip).  Reference call sites: src/train/pyramid.py:35-46 (build / reconstruct), adapters :48-112.
#pragma once
#include "vfi_common.h"

namespace vfi {
namespace pyrw {

constexpr int kMaxImages = 16;
constexpr int kBands = 4;

struct PlaneMap {          // where image d's band-0 plane goes, in planes of h*w elements from the level base
    int idx[kMaxImages];
    int band_stride;       // planes between consecutive bands of one image
    int complex_coeff;     // 1: `phase` holds interleaved (re, im) coefficients, `amp` is unused
};

struct Tables {            // device tables of one (engine length, transform length) pair
    const float2 *tw;      // stage twiddles of the engine length (wfft::for_twiddles order)
    const float2 *chirp;   // n entries exp(-i pi j^2 / n)                      (Bluestein only)
    const float2 *bfilt;   // M entries FFT_M(conj(chirp) wrapped) / M          (Bluestein only)
    int M, n, bluestein;
};

// rows of T (planes x h x tpitch complex, row length w) <-> (phase, amplitude) planes of the caller
struct RowsArgs {
    Tables tb;                    // tb.n == w
    float2 *T;
    int tpitch;
    float *phase, *amp;
    PlaneMap pm;
    int planes, h, w;
    float inv_hw, phase_scale;    // analysis: 1 / (h w), phase scale; synthesis: unused
    unsigned *amp_max;            // analysis only, optional: [groups] bit patterns of the largest amplitude per image group
    int groups;
};

// analysis columns: T[n][b] = IFFT_cols( i * window_k(S[n]) * Q[b] )
struct AnaColsArgs {
    Tables tb;                    // tb.n == h
    const float2 *S;              // N x H x spitch half spectra (R2C output, un-normalised)
    int spitch, H;
    const float *Q;               // [bands][h][w]: lo0 * prod_{j<k} lomask_j * himask_k * anglemask_b on level k's window (unshifted order)
    float2 *T;                    // N x bands x h x tpitch
    int tpitch, N, h, w;
};

// synthesis columns: cur[n] = sum_b (-i) * FFT_cols(T[n][b]) * P[b]  +  embed(res[n] * lomask)
struct SynColsArgs {
    Tables tb;
    const float2 *T;
    int tpitch;
    const float *P;               // [bands][h][w]
    const float2 *res;            // N x h2 x w2 (coarser level's spectrum; may be null)
    const float *lomask;          // [h2][w2]
    float2 *cur;                  // N x h x w
    int N, h, w, h2, w2;
};

// plain passes (vfi_pyrw_passes.h): how a row pass reads / writes its rows
enum GenRowKind { kGenComplex = 0, kGenReal = 1, kGenHalf = 2 };      // Half: the first n/2+1 entries of a Hermitian row
struct GenRowsArgs {
    Tables tb;                    // tb.n = row length
    const void *src;
    void *dst;
    int rows, src_pitch, dst_pitch;   // rows of all planes together; pitches in elements
    float scale;
};
struct GenColsArgs {
    Tables tb;                    // tb.n = column length (rows of a plane); tables of the ANALYSIS column configurations
    float2 *data;                 // [planes][n][ld], transformed in place
    int planes, cols, ld;
    float scale;
};

// engine length that serves a transform of length n (n itself, or Bluestein's 2^k / 3*2^k length), 0 when the engine has
// no configuration for it (the caller then uses the generic LDS engine of vfi_fft.h for that pass)
int rows_engine_length(int n, int bluestein_m);
int cols_engine_length(int n, int bluestein_m);
// stage-twiddle table of an engine length (host side, double precision): entries as (cos, sin) floats
int rows_twiddles(int M, float2 *out, int cap);     // -> number of entries, or -1
int cols_twiddles(int M, float2 *out, int cap);     // analysis column pass
int syn_twiddles(int M, float2 *out, int cap);      // synthesis column pass (same lengths as the analysis one)

int launch_rows_polar(const RowsArgs &a, hipStream_t s);        // analysis rows  (coeff_to_values, src/train/pyramid.py:63-69)
int launch_rows_from_polar(const RowsArgs &a, hipStream_t s);   // synthesis rows (values_to_coeff, src/train/pyramid.py:99-107)
int launch_ana_cols(const AnaColsArgs &a, hipStream_t s);
int launch_syn_cols(const SynColsArgs &a, hipStream_t s);
// supported (load, store, direction): (real, half, forward) = R2C rows, (half, real, inverse) = C2R rows, (complex, complex, *)
int launch_gen_rows(const GenRowsArgs &a, int load, int store, bool inverse, hipStream_t s);
int launch_gen_cols(const GenColsArgs &a, bool inverse, hipStream_t s);

}  // namespace pyrw
}  // namespace vfi
