// Host-side check of the LDS FFT engine (csrc/vfi_fft.h): the very same gather / scatter / Bluestein index arithmetic the
// device runs, executed on the CPU by looping over the 256 "threads" between the synchronisation points, against a
// double-precision O(n^2) DFT.  Built and run by tests/test_fft_host.py (hipcc --cuda-host-only: no GPU involved).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vfi_fft.h"

using namespace vfi::fft;
using cd = std::complex<double>;

struct HostPlan {
    Plan1D pl;
    std::vector<float2> tw, chirp, bfilt;
};

static void build(HostPlan &hp, int n) {
    Plan1D &pl = hp.pl;
    pl.n = n;
    pl.bluestein = !factor_smooth(n, pl.radix, &pl.nstages);
    pl.m = pl.bluestein ? bluestein_length(n) : n;
    pl.tw_len = twiddle_entries(pl.m);
    if (pl.bluestein && !factor_bluestein(pl.m, pl.radix, &pl.nstages)) { std::printf("plan failed for %d\n", n); std::exit(2); }
    if (pl.tw_len < min_twiddle_entries(pl)) pl.tw_len = min_twiddle_entries(pl);
    hp.tw.resize(pl.m);
    for (int k = 0; k < pl.m; ++k) hp.tw[k] = make_float2((float)std::cos(-2.0 * M_PI * k / pl.m), (float)std::sin(-2.0 * M_PI * k / pl.m));
    // the device keeps only the first tw_len entries (in LDS): poison the rest so that a read beyond them shows
    for (int k = pl.tw_len; k < pl.m; ++k) hp.tw[k] = make_float2(NAN, NAN);
    pl.tw = hp.tw.data();
    if (pl.bluestein) {
        std::vector<cd> w(n), b(pl.m, cd(0, 0)), B(pl.m);
        for (int j = 0; j < n; ++j) {
            const long long e = ((long long)j * j) % (2LL * n);
            w[j] = std::polar(1.0, -M_PI * (double)e / n);
        }
        b[0] = std::conj(w[0]);
        for (int j = 1; j < n; ++j) b[j] = b[pl.m - j] = std::conj(w[j]);
        for (int k = 0; k < pl.m; ++k) {           // O(M^2) is fine for a test
            cd s(0, 0);
            for (int j = 0; j < pl.m; ++j) s += b[j] * std::polar(1.0, -2.0 * M_PI * (double)((long long)j * k % pl.m) / pl.m);
            B[k] = s / (double)pl.m;
        }
        hp.chirp.resize(n); hp.bfilt.resize(pl.m);
        for (int j = 0; j < n; ++j) hp.chirp[j] = make_float2((float)w[j].real(), (float)w[j].imag());
        for (int k = 0; k < pl.m; ++k) hp.bfilt[k] = make_float2((float)B[k].real(), (float)B[k].imag());
        pl.chirp = hp.chirp.data(); pl.bfilt = hp.bfilt.data();
    }
}

// wave_private: the four waves run one after the other, each through ALL stages of its own lines before the next
// starts -- only correct if the lines of different waves are independent, which is what the mode relies on
template <int R, bool MULB>
static void host_stage(float2 *buf, int lines, int pitch, int m, int p, const float2 *tw, int tw_len, const float2 *bfilt, int wave) {
    std::vector<StageRegs<R>> regs(kThreads);
    const int t0 = wave < 0 ? 0 : 64 * wave, t1 = wave < 0 ? kThreads : 64 * wave + 64;   // (kThreads / 64 waves)
    for (int t = t0; t < t1; ++t) stage_gather<R, MULB>(regs[t], wave < 0 ? team_all(t, lines) : team_wave(t, lines), buf, pitch, m, p, tw, tw_len, bfilt);
    for (int t = t0; t < t1; ++t) stage_scatter<R>(regs[t], wave < 0 ? team_all(t, lines) : team_wave(t, lines), buf, pitch, m, p);
}
template <bool MULB>
static void host_stage_any(int R, float2 *buf, int lines, int pitch, int m, int p, const float2 *tw, int tw_len, const float2 *bfilt, int wave) {
    switch (R) {
        case 16: host_stage<16, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 15: host_stage<15, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 12: host_stage<12, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 9: host_stage<9, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 8: host_stage<8, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 4: host_stage<4, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 2: host_stage<2, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        case 3: host_stage<3, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
        default: host_stage<5, MULB>(buf, lines, pitch, m, p, tw, tw_len, bfilt, wave); break;
    }
}
// mirrors fft_lines of vfi_fft.h
static void host_fft_team(float2 *buf, int lines, int pitch, const Plan1D &pl, int wave) {
    int p = 1;
    for (int s = 0; s < pl.nstages; ++s) { host_stage_any<false>(pl.radix[s], buf, lines, pitch, pl.m, p, pl.tw, pl.tw_len, nullptr, wave); p *= pl.radix[s]; }
    if (!pl.bluestein) return;
    host_stage_any<true>(pl.radix[0], buf, lines, pitch, pl.m, 1, pl.tw, pl.tw_len, pl.bfilt, wave);
    p = pl.radix[0];
    for (int s = 1; s < pl.nstages; ++s) { host_stage_any<false>(pl.radix[s], buf, lines, pitch, pl.m, p, pl.tw, pl.tw_len, nullptr, wave); p *= pl.radix[s]; }
}
static void host_fft(float2 *buf, int lines, int pitch, const Plan1D &pl) {
    if (lines % (kThreads / 64) == 0) { for (int w = 0; w < kThreads / 64; ++w) host_fft_team(buf, lines, pitch, pl, w); }
    else host_fft_team(buf, lines, pitch, pl, -1);
}

static double check(int n, int lines, bool inv) {
    HostPlan hp;
    build(hp, n);
    if (lines > max_lines(hp.pl)) lines = max_lines(hp.pl);
    const int pitch = lines > 2 ? col_pitch(hp.pl, 4) : row_pitch(hp.pl);      // both pitch rules get exercised
    std::vector<float2> buf((size_t)lines * pitch, make_float2(7.0f, 7.0f));
    std::vector<cd> x((size_t)lines * n);
    unsigned s = 12345u + n;
    for (auto &v : x) {
        s = s * 1664525u + 1013904223u; const double a = (double)(s >> 8) / (1 << 24) - 0.5;
        s = s * 1664525u + 1013904223u; const double b = (double)(s >> 8) / (1 << 24) - 0.5;
        v = cd(a, b);
    }
    // fill as the kernels do: x[j] * chirp_factor(j) for Bluestein plans, zero padding up to m
    for (int l = 0; l < lines; ++l)
        for (int j = 0; j < hp.pl.m; ++j) {
            float2 z = make_float2(0.0f, 0.0f);
            if (j < n) {
                z = make_float2((float)x[(size_t)l * n + j].real(), (float)x[(size_t)l * n + j].imag());
                const float2 ch = hp.pl.bluestein ? hp.pl.chirp[j] : make_float2(1.0f, 0.0f);
                z = inv ? load_value<true>(z, ch, hp.pl.bluestein != 0) : load_value<false>(z, ch, hp.pl.bluestein != 0);
            }
            buf[(size_t)l * pitch + phys(j)] = z;
        }
    host_fft(buf.data(), lines, pitch, hp.pl);
    double worst = 0, scale = 0;
    for (int l = 0; l < lines; ++l)
        for (int k = 0; k < n; ++k) {
            cd acc(0, 0);
            for (int j = 0; j < n; ++j) {
                const cd xv((float)x[(size_t)l * n + j].real(), (float)x[(size_t)l * n + j].imag());
                acc += xv * std::polar(1.0, (inv ? 2.0 : -2.0) * M_PI * (double)((long long)j * k % n) / n);
            }
            float2 g = buf[(size_t)l * pitch + phys(k)];
            const float2 ch = hp.pl.bluestein ? hp.pl.chirp[k] : make_float2(1.0f, 0.0f);
            g = inv ? store_value<true>(g, ch, hp.pl.bluestein != 0) : store_value<false>(g, ch, hp.pl.bluestein != 0);
            worst = std::fmax(worst, std::abs(acc - cd(g.x, g.y)));
            scale = std::fmax(scale, std::abs(acc));
        }
    return worst / scale;
}

int main() {
    const int sizes[] = {2, 3, 4, 5, 6, 8, 9, 11, 12, 15, 16, 17, 22, 24, 30, 32, 43, 48, 60, 64, 65, 68, 77, 85, 90, 96, 120, 128,
                         135, 170, 182, 191, 240, 256, 270, 340, 382, 480, 512, 540, 679, 720, 764, 905, 960, 1024, 1080, 1280,
                         1358, 1920, 2048, 2716, 4096, 8192};
    int bad = 0;
    for (int n : sizes)
        for (int inv = 0; inv < 2; ++inv) {
            const double e = check(n, n % 3 == 0 ? 8 : (n % 2 ? 3 : 4), inv);
            const bool ok = e < 2e-6;
            if (!ok) ++bad;
            std::printf("n=%d %s rel.err %.2e %s\n", n, inv ? "inv" : "fwd", e, ok ? "" : "FAIL");
        }
    return bad ? 1 : 0;
}
