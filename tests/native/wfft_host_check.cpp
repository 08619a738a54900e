// Host-side check of the wave-private FFT engine (csrc/vfi_wfft.h): the very same per-lane stage / exchange code the
// device runs, executed on the CPU by looping over the lanes of a team between the points where its LDS traffic is
// ordered, for EVERY configuration of csrc/vfi_wfft_configs.h, against a double-precision O(n^2) DFT -- plain
// transforms and Bluestein's form (a non-smooth length n on each engine length of the form 2^k / 3*2^k).
// Also checks that no exchange touches a dword outside the wave's buffer.
// Built and run by tests/test_fft_host.py (hipcc --cuda-host-only: no GPU involved).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "vfi_wfft.h"
#include "vfi_wfft_configs.h"

using namespace vfi::wfft;
using cd = std::complex<double>;

static int g_fail = 0;

static std::vector<cd> dft(const std::vector<cd> &x) {
    const size_t n = x.size();
    std::vector<cd> w(n), y(n);
    for (size_t k = 0; k < n; ++k) w[k] = std::polar(1.0, -2.0 * M_PI * (double)k / (double)n);
    for (size_t k = 0; k < n; ++k) {
        cd s(0, 0);
        for (size_t j = 0; j < n; ++j) s += x[j] * w[(j * k) % n];
        y[k] = s;
    }
    return y;
}

template <class C> struct Sim {
    std::vector<float2> tw;
    std::vector<float> xb;
    float2 v[C::TEAM][C::E];
    Sim() : tw(C::TW > 0 ? C::TW : 1), xb(C::XBUF + 64, NAN) {
        for_twiddles<C>([&](int idx, int e) {
            tw[idx] = make_float2((float)std::cos(-2.0 * M_PI * e / C::M), (float)std::sin(-2.0 * M_PI * e / C::M));
        });
        for (auto &lane : v) for (auto &z : lane) z = make_float2(NAN, NAN);
    }
    template <int S> void run_stage() { for (int lane = 0; lane < C::TEAM; ++lane) stage<C, S>(v[lane], lane, tw.data()); }
    template <int SW, int SR> void run_exchange() {
        for (int i = 0; i < 64; ++i) xb[C::XBUF + i] = 12345.0f;                                    // guard words
        for (int lane = 0; lane < C::TEAM; ++lane) exchange_write<C, SW, 0>(v[lane], lane, xb.data());
        for (int lane = 0; lane < C::TEAM; ++lane) exchange_read<C, SW, SR, 0>(v[lane], lane, xb.data());
        for (int lane = 0; lane < C::TEAM; ++lane) exchange_write<C, SW, 1>(v[lane], lane, xb.data());
        for (int lane = 0; lane < C::TEAM; ++lane) exchange_read<C, SW, SR, 1>(v[lane], lane, xb.data());
        for (int i = 0; i < 64; ++i) if (xb[C::XBUF + i] != 12345.0f) { std::printf("FAIL buffer overrun M=%d\n", C::M); ++g_fail; break; }
    }
    void forward() {
        run_stage<0>();
        if constexpr (C::NS > 1) { run_exchange<0, 1>(); run_stage<1>(); }
        if constexpr (C::NS > 2) { run_exchange<1, 2>(); run_stage<2>(); }
        if constexpr (C::NS > 3) { run_exchange<2, 3>(); run_stage<3>(); }
    }
    // line l, position pos of the first-stage input / last-stage output distribution
    void put_inputs(const std::vector<std::vector<cd>> &x) {
        using I = Io<C>;
        for (int lane = 0; lane < C::TEAM; ++lane)
            for (int q = 0; q < I::Q0; ++q) {
                int l, i; bool ok;
                lane_index<C, 0>(lane, q, l, i, ok);
                for (int r = 0; r < I::R0; ++r) {
                    const cd z = ok ? x[l][i + r * I::T0] : cd(NAN, NAN);
                    v[lane][q * I::R0 + r] = make_float2((float)z.real(), (float)z.imag());
                }
            }
    }
    std::vector<std::vector<cd>> get_outputs() {
        using I = Io<C>;
        std::vector<std::vector<cd>> y(C::L, std::vector<cd>(C::M, cd(NAN, NAN)));
        for (int lane = 0; lane < C::TEAM; ++lane)
            for (int q = 0; q < I::QL; ++q) {
                int l, k; bool ok;
                lane_index<C, I::SL>(lane, q, l, k, ok);
                if (!ok) continue;
                for (int r = 0; r < I::RL; ++r) y[l][k + r * I::PL] = cd(v[lane][q * I::RL + r].x, v[lane][q * I::RL + r].y);
            }
        return y;
    }
};

static double rel_err(const std::vector<cd> &got, const std::vector<cd> &ref, size_t n) {
    double e = 0, m = 0;
    for (size_t k = 0; k < n; ++k) { e = std::max(e, std::abs(got[k] - ref[k])); m = std::max(m, std::abs(ref[k])); }
    return e / m;
}

template <class C> static void check_plain(const char *mode) {
    auto sim = std::make_unique<Sim<C>>();
    std::vector<std::vector<cd>> x(C::L, std::vector<cd>(C::M));
    for (auto &line : x) for (auto &z : line) z = cd(drand48() - 0.5, drand48() - 0.5);
    sim->put_inputs(x);
    sim->forward();
    const auto y = sim->get_outputs();
    double worst = 0;
    for (int l = 0; l < C::L; l += (C::L > 4 ? C::L / 4 : 1)) worst = std::max(worst, rel_err(y[l], dft(x[l]), C::M));
    const bool ok = worst < 2e-6;
    if (!ok) ++g_fail;
    std::printf("%s %s M=%4d L=%2d team %3d E=%2d stages %d: rel.err %.2e\n", ok ? "ok  " : "FAIL", mode, C::M, C::L, C::TEAM, C::E, C::NS, worst);
}

// Bluestein: DFT of length n (any n with 2n-1 <= M) through two engine transforms, as the device kernels do it
template <class C> static void check_bluestein(const char *mode, int n) {
    using I = Io<C>;
    constexpr int M = C::M;
    std::vector<cd> w(n), b(M, cd(0, 0));
    for (int j = 0; j < n; ++j) w[j] = std::polar(1.0, -M_PI * (double)(((long long)j * j) % (2LL * n)) / (double)n);
    b[0] = std::conj(w[0]);
    for (int j = 1; j < n; ++j) b[j] = b[M - j] = std::conj(w[j]);
    auto B = dft(b);
    std::vector<float2> bf(M);
    for (int k = 0; k < M; ++k) bf[k] = make_float2((float)(B[k].real() / M), (float)(B[k].imag() / M));
    auto sim = std::make_unique<Sim<C>>();
    std::vector<std::vector<cd>> x(C::L, std::vector<cd>(n)), a(C::L, std::vector<cd>(M, cd(0, 0)));
    for (int l = 0; l < C::L; ++l)
        for (int j = 0; j < n; ++j) {
            x[l][j] = cd(drand48() - 0.5, drand48() - 0.5);
            const cd c((float)w[j].real(), (float)w[j].imag());
            a[l][j] = x[l][j] * c;
        }
    sim->put_inputs(a);
    sim->forward();
    for (int lane = 0; lane < C::TEAM; ++lane)
        for (int q = 0; q < I::QL; ++q) {
            int l, k; bool ok;
            lane_index<C, I::SL>(lane, q, l, k, ok);
            if (!ok) k = 0;
            for (int r = 0; r < I::RL; ++r) sim->v[lane][q * I::RL + r] = cconj(cmul(sim->v[lane][q * I::RL + r], bf[k + r * I::PL]));
        }
    if constexpr (I::R0 != I::RL) sim->template run_exchange<I::SL, 0>();
    sim->forward();
    auto y = sim->get_outputs();
    double worst = 0;
    for (int l = 0; l < C::L; l += (C::L > 4 ? C::L / 4 : 1)) {
        std::vector<cd> got(n);
        for (int k = 0; k < n; ++k) got[k] = cd((float)w[k].real(), (float)w[k].imag()) * std::conj(y[l][k]);
        worst = std::max(worst, rel_err(got, dft(x[l]), n));
    }
    const bool ok = worst < 4e-6;
    if (!ok) ++g_fail;
    std::printf("%s %s M=%4d L=%2d bluestein n=%4d (first radix %d, last %d): rel.err %.2e\n", ok ? "ok  " : "FAIL", mode, M, C::L, n,
                I::R0, I::RL, worst);
}

static bool blu_length(int m) { while (m % 2 == 0) m /= 2; return m == 1 || m == 3; }

template <class C> static void check(const char *mode) {
    check_plain<C>(mode);
    if (blu_length(C::M)) {
        check_bluestein<C>(mode, (C::M + 1) / 2);             // the longest length this M serves
        check_bluestein<C>(mode, (C::M + 1) / 2 - C::M / 7);  // an arbitrary shorter one
    }
}

int main() {
    srand48(1);
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) check<Cfg<M, L, TEAM, false, PITCH, P0, P1, P2, R0, R1, R2, R3>>("rows");
    VFI_WFFT_ROW_CONFIGS(X)
#undef X
#define X(M, L, TEAM, PITCH, P0, P1, P2, R0, R1, R2, R3) check<Cfg<M, L, TEAM, true, PITCH, P0, P1, P2, R0, R1, R2, R3>>("cols");
    VFI_WFFT_COL_CONFIGS(X)
    VFI_WFFT_SYN_CONFIGS(X)
#undef X
    std::printf(g_fail ? "FAIL: %d check(s)\n" : "all wave-engine checks passed (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
