// Host-side check of the tile geometry (tamcmc_dev.h): tile counts keep tiles * TM_TILE_MAXU > units, and the
// equal-cost boundaries of tm_tile_bound -- the same function the setup kernel runs -- cover [0, units) in order with
// tiles of at most TM_TILE_MAXU units, for adversarial and random unit costs.  Built and run by tests/test_capi_host.py.
#include "tamcmc_dev.h"
#include <cstdio>
#include <vector>

static unsigned long long rng = 88172645463325252ull;
static unsigned next_u32() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 32); }

int main() {
    long checked = 0;
    for (int units = 1; units <= TM_EQ_MAXU; units = units < 700 ? units + 1 : units + 97) {
        for (int grad = 0; grad < 2; grad++) {
            const int T = tm_tiles(units, grad);
            if (T < 1 || (long long)T * TM_TILE_MAXU < units) { printf("bad tile count: units %d -> %d\n", units, T); return 1; }
            if (T == 1 || (long long)T * TM_TILE_MAXU <= units) continue;      // the balancer needs slack; such grids get equal-length tiles
            for (int pattern = 0; pattern < 6; pattern++) {
                std::vector<int> cost(units), pre(units);
                for (int u = 0; u < units; u++) {
                    switch (pattern) {
                    case 0: cost[u] = 60; break;                                            // flat
                    case 1: cost[u] = 60 + ((u * 7 < units) ? 12800 : 0); break;            // a dense head, a cheap tail
                    case 2: cost[u] = 60 + ((u * 7 > units * 6) ? 12800 : 0); break;        // a cheap head, a dense tail
                    case 3: cost[u] = 60 + (int)(next_u32() % 3000); break;                 // random
                    case 4: cost[u] = 110 + ((u % 23) < 3 ? 9000 : 0); break;               // spikes
                    default: cost[u] = 1 + (int)(next_u32() % 2) * 50000; break;            // extreme contrast
                    }
                }
                long long C = 0, cmin = cost[0];
                for (int u = 0; u < units; u++) { C += cost[u]; pre[u] = (int)C; if (cost[u] < cmin) cmin = cost[u]; }
                int prev = 0;
                long long cmax_tile = 0;
                for (int t = 1; t <= T; t++) {
                    const int b = (t == T) ? units : tm_tile_bound(t, T, units, pre.data(), C, cmin);
                    if (b < prev || b - prev > TM_TILE_MAXU || b > units) { printf("bad boundary: units %d T %d pattern %d tile %d [%d, %d)\n", units, T, pattern, t - 1, prev, b); return 1; }
                    const long long ct = (b > 0 ? pre[b - 1] : 0) - (prev > 0 ? pre[prev - 1] : 0);
                    if (ct > cmax_tile) cmax_tile = ct;
                    prev = b;
                }
                if (prev != units) { printf("bad cover: units %d T %d pattern %d -> %d\n", units, T, pattern, prev); return 1; }
                // balance: with flat costs no tile may exceed the mean by more than one unit's cost
                if (pattern == 0 && cmax_tile > C / T + 2 * 60) { printf("unbalanced flat split: units %d T %d max %lld mean %lld\n", units, T, cmax_tile, C / T); return 1; }
                checked++;
            }
        }
    }
    printf("ok %ld geometries\n", checked);
    return 0;
}
