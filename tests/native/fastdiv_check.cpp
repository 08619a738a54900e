// Host-side check of the exact division by reciprocal that the Winograd kernel decodes its work items with
// (csrc/vfi_conv_common.h: make_fastdiv / fast_div).  Built and run by tests/test_conv_host.py (hipcc --cuda-host-only).
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
    return bad ? 1 : 0;
}
