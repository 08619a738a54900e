// Host-side check of the exact division by reciprocal that the Winograd kernel decodes its work items with
// (csrc/vfi_conv_common.h: make_fastdiv / fast_div), and of how the F(4x4) kernel deals its weight requests to the waves.  Built and run by tests/test_conv_host.py (hipcc --cuda-host-only).
#include <cstdio>
#include <initializer_list>

#include "vfi_conv_common.h"

using namespace vfi::conv;

int main() {
    long bad = 0, checked = 0;
    // every divisor a layer can produce (tiles, channel blocks, runs, splits ...), dividends on a stride
    for (unsigned d = 1; d < 6000; ++d) {
        const FastDiv f = make_fastdiv(d);
        for (unsigned n = 0; n < 4000000u; n += d % 13 + 1, ++checked)
            if ((unsigned)fast_div((int)n, f) != n / d) ++bad;
    }
    // large divisors and the ends of the 31-bit range (items are non-negative ints)
    for (unsigned d : {8100u, 24304u, 48960u, 340256u, 1000003u, 16777259u, 0x7fffffffu}) {
        const FastDiv f = make_fastdiv(d);
        for (unsigned n : {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7ffffffeu, 0x7fffffffu}) {
            ++checked;
            if ((unsigned)fast_div((int)n, f) != n / d) ++bad;
        }
    }
    std::printf("checked %ld divisions, %ld wrong\n", checked, bad);
    // F(4x4) kernel: every one of the 18 weight pieces requested exactly once, and per wave (input pieces w, w+8, w+16 < 20
    // plus its weight pieces) the 5 | 4 | 5 requests per chunk that the kernel's counted waits are written for
    int seen[18] = {0}, wrong = 0;
    for (int wave = 0; wave < 8; ++wave) {
        int requests = 0;
        for (int k = 0; k < 3; ++k) requests += wave + 8 * k < 20;
        for (int t = 0; t < 3; ++t) {
            const int wp = weight_piece(wave, t);
            if (wp >= 18) ++wrong;
            if (wp >= 0 && wp < 18) { ++seen[wp]; ++requests; }
        }
        const int expect = (wave < 2 || wave >= 4) ? 5 : 4;
        if (requests != expect) ++wrong;
    }
    for (int p = 0; p < 18; ++p) wrong += seen[p] != 1;
    std::printf("weight pieces: %d inconsistencies\n", wrong);
    return (bad || wrong) ? 1 : 0;
}
