// Host-side check of cf::make_idiv (co_fusion_amd/csrc/cf_kernels.h): n / d == umulhi(n, M) >> s for every 0 <= n < 2^31 the kernels can pass
// (image widths 2 .. 8200: boundaries of every multiple, the top of the range, random values).  Built and run by tests/test_cpu_abi.py.
#include "cf_kernels.h"
#include <cstdio>
#include <cstdint>
int main()
{
    uint64_t rng = 88172645463325252ull; long bad = 0, checked = 0;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int d = 2; d <= 8200; d++) {
        const cf::IDiv D = cf::make_idiv(d);
        auto chk = [&](uint32_t n) { if (n < 0x80000000u) { checked++; if ((uint32_t)(((uint64_t)n * D.M) >> 32) >> D.s != n / (uint32_t)d) bad++; } };
        for (uint32_t n : {0u, 1u, (uint32_t)d - 1, (uint32_t)d, (uint32_t)d + 1, 0x7fffffffu, 0x7ffffffeu, 0x7fffffffu - (uint32_t)d}) chk(n);
        for (int k = 1; k < 64; k++) { chk((uint32_t)k * d - 1); chk((uint32_t)k * d); chk((0x7fffffffu / d - k) * d); chk((0x7fffffffu / d - k) * d - 1); }
        for (int k = 0; k < 400; k++) chk((uint32_t)(next() >> 33));
    }
    printf("checked %ld bad %ld\n", checked, bad);
    return bad != 0;
}
