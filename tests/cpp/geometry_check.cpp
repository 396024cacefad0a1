// Host-side check of the tile geometry macros (tamcmc_dev.h): for every (units, big, small) the tiles cover the
// sub-blocks [0, units) exactly once, in order, with 1 <= size <= big.  Built and run by tests/test_capi_host.py.
#include "tamcmc_dev.h"
#include <cstdio>
int main() {
    long checked = 0;
    for (int units = 1; units <= 400; units++)
        for (int big = 1; big <= 12; big++)
            for (int small = 1; small <= big; small++) {
                const int T = tm_tile_count(units, big, small);
                int next = 0;
                for (int t = 0; t < T; t++) {
                    const int u0 = TM_TILE_U0(t, big, small), S = TM_TILE_S(t, big, small, units);
                    if (u0 != next || S < 1 || S > big) { printf("bad: units %d big %d small %d tile %d u0 %d S %d\n", units, big, small, t, u0, S); return 1; }
                    next = u0 + S;
                }
                if (next != units) { printf("bad cover: units %d big %d small %d -> %d\n", units, big, small, next); return 1; }
                checked++;
            }
    printf("ok %ld geometries\n", checked);
    return 0;
}
