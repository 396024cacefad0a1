// tamcmc_setup_body.h -- device code of the per-chain prologue (params row -> multiplet table, noise record, tile
// descriptors, active lists, launch ranks), shared by tamcmc_setup_kernel (tamcmc_setup.hip) and the fused small-grid
// kernel (tamcmc_fused.hip).  Everything here is compiled WITHOUT floating-point contraction, whatever the flags of
// the including file: window indices must come out of the same non-fused fp64 operations as the oracle's (a one-ulp
// difference can move a window edge by one bin), see tamcmc_derive.h.
#ifndef TAMCMC_SETUP_BODY_H
#define TAMCMC_SETUP_BODY_H
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#pragma clang fp contract(off)
#include "tamcmc_derive.h"

// One workgroup of NT >= 192 threads (whole waves) per chain.  p: this chain's params row in LDS (L.Nparams doubles, filled here).
// wave 0: multiplets; wave 1: noise record + tile polynomials; wave 2: m-ratios -- side by side.
#ifdef TM_SU_TRACE   // timing-only build: cycle stamps of the phases leave through the chain's series table (tools/su_trace.py)
#define SU_TS(i, who) do { if (tid == (who)) s_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SU_TS(i, who) do { } while (0)
#endif
template <int NT>
__device__ __forceinline__ void tm_setup_body(const TmLayout &L, const int chain, const double *__restrict__ params,
                                              const double *__restrict__ Tcoefs, double *__restrict__ wt,
                                              const double *__restrict__ lx, int units, int cells, int tiles, int equal_cost,
                                              const TmCostModel cm /* cm.pad: units per tile at most */, TmMult *__restrict__ mult, TmNoise *__restrict__ noise,
                                              TmCellRec *__restrict__ cell, TmTileHdr *__restrict__ thdr, TmActive *__restrict__ tidx,
                                              TmChain *__restrict__ chain_rec, TmMultFull *__restrict__ aux,
                                              double *__restrict__ hser, int32_t *__restrict__ order, double *p,
                                              int *s_pre /* LDS, [units] when tiles > 1 and units <= TM_EQ_MAXU, else unused */)
{
    __shared__ int s_win[TM_MAXMULT][3];   // truncation windows and component counts, for the per-tile active lists
    // gradient launch with the asymmetry among the variables: asymmetric code path also where asym == 0 (TmLayout::asym_var)
    const bool asym_forced = (L.asym_var != 0) && (chain_rec != nullptr);
    __shared__ __attribute__((aligned(16))) int s_cost[TM_ORDER_MAX + 4];
    const int tid = threadIdx.x;
#ifdef TM_SU_TRACE
    __shared__ unsigned long long s_ts[20];
    if (tid < 20) s_ts[tid] = 0;
    __syncthreads();
#endif
    SU_TS(0, 0);
    for (int e = tid; e < L.Nparams; e += NT) p[e] = params[(size_t)chain * L.Nparams + e];
    // the caller's temperature array may live in host memory: read it once, now (latency hidden behind the work below)
    const double Tc = (tid == 64) ? Tcoefs[chain] : 1.0;
    __shared__ TmChain C;
    __shared__ int s_status;
    __syncthreads();
    SU_TS(1, 0);
#if defined(TM_SETUP_STOP) && TM_SETUP_STOP == 1
    return;   // timing-only build
#endif

    if (tid == 0) s_status = 0;
    // The chain's scalars: lane 0 of wave 0, which goes on to the multiplets -- the only consumers -- without a workgroup
    // barrier (LDS operations of one wave complete in order).  Inclination and m-ratio tables: wave 2, from the params
    // row directly.  Wave 1: the noise record and the cell polynomials.  All three side by side.
    if (L.family != TM_FAM_GAUSS && tid == 0) {
#if !(defined(TM_SETUP_SKIP) && (TM_SETUP_SKIP & 4))   // timing-only build: no chain-level derivation
        tm_derive_chain_scalars(L, p, C);
#endif
    }
    if (tid < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    SU_TS(2, 0);
#if defined(TM_SETUP_STOP) && TM_SETUP_STOP == 3
    return;   // timing-only build
#endif
    if (L.family != TM_FAM_GAUSS && tid >= 128 && tid < 192) {
#if !(defined(TM_SETUP_SKIP) && (TM_SETUP_SKIP & 4))
        tm_derive_chain_tables(L, p, C, tid - 128);
#endif
    }

    __shared__ TmNoise s_N;           // N.H[N.nh] is dynamically indexed: LDS, not scratch
    if (tid == 64) {
        TmNoise &N = s_N;
        for (int k = 0; k < TM_MAXH; k++) { N.H[k] = 0.0; N.lt[k] = 0.0; N.p[k] = 0.0; }
        N.N0 = 0.0; N.gA = 0.0; N.gnu0 = 0.0; N.gs2 = 1.0; N.nh = 0; N.has_gauss = 0; N.pad = 0;
        int z = L.z, Nnoise = L.Nnoise, nharvey = L.nharvey;
        bool take_abs = true;
        if (L.model_case == 0) {
            // model_Test_Gaussian, models.cpp:2021-2034 (no abs anywhere)
            N.has_gauss = 1; N.gA = p[0]; N.gnu0 = p[2]; N.gs2 = p[1] * p[1];
            z = 3; Nnoise = 1; nharvey = 0; take_abs = false;
        } else if (L.model_case == 1) {
            // model_Harvey_Gaussian, models.cpp:1968-1992
            N.has_gauss = 1; N.gA = fabs(p[0]); N.gnu0 = p[2]; N.gs2 = fabs(p[1]) * fabs(p[1]);
            z = 3; Nnoise = 4; nharvey = 1;
        }
        double extra = 0.0;
        for (int k = 0; k < nharvey; k++) {
            const double H = fabs(p[z + 3 * k]), tau = fabs(p[z + 3 * k + 1]), pw = fabs(p[z + 3 * k + 2]);
            if (tau != 0) {                           // noise_models.cpp:31
                if (pw == 0) { extra = extra + H * 0.5; continue; } // (..)^0 = 1 for every bin
                N.H[N.nh] = H; N.lt[N.nh] = log((1e-3) * tau); N.p[N.nh] = pw; N.nh++;
            }
        }
        const double n0 = p[z + Nnoise - 1];
        N.N0 = (take_abs ? fabs(n0) : n0) + extra;
    }
    if (tid >= 64 && tid < 128) {
        // Wave 1, concurrently with wave 0's multiplet derivation: the Harvey background of every cell as Taylor
        // polynomials in z = p (log x - log x_c).  u(z) = 1/(1 + t0 e^z) is analytic for |z| < pi (nearest pole at
        // ln(1/t0) + i pi), so for |z| <= 0.04 the series truncated at degree 8 is exact to (0.04/pi)^9 ~ 1e-17.
        // Coefficients by the power-series reciprocal of D(z) = 1 + t0 sum z^j/j!.
        // s_N was written by lane 0 of THIS wave: LDS operations of one wave complete in order.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#if defined(TM_SETUP_SKIP) && (TM_SETUP_SKIP & 1)   // timing-only build: no cell polynomials
        for (int ce = cells; ce < cells; ce += 64) {
#else
        for (int ce = tid - 64; ce < cells; ce += 64) {
#endif
            const int base = ce * (TM_CELL_UNITS * TM_UNIT_BINS);
            int TB = TM_CELL_UNITS * TM_UNIT_BINS; if (base + TB > L.Nx) TB = L.Nx - base;
            TmCellRec R;
            R.pad = 0;
            int ic = base + TB / 2; if (ic > L.Nx - 1) ic = L.Nx - 1;
            int i1 = base + TB - 1; if (i1 > L.Nx - 1) i1 = L.Nx - 1;
            const double lx_c = lx[ic], lx_0 = lx[base], lx_1 = lx[i1];
            const double span = fmax(fabs(lx_0 - lx_c), fabs(lx_1 - lx_c));
            bool ok = (span == span) && (lx_c - lx_c == 0.0) && (L.bg_exact == 0);
            R.lxc = lx_c;
#pragma unroll
            for (int j = 0; j <= TM_PDEG; j++) R.bg[j] = 0.0;
            R.bg[0] = s_N.N0;
#pragma unroll
            for (int h = 0; h < TM_MAXH; h++) {
                R.t0[h] = 0.0;
                if (h < s_N.nh) {
                    const double ph = s_N.p[h], Hh = s_N.H[h];
                    ok = ok && (ph * span <= 0.04);
                    const double t0 = exp(ph * (s_N.lt[h] + lx_c));
                    ok = ok && (t0 < 1e290);
                    R.t0[h] = t0;
                    // one degree more than the model polynomial uses: the gradient path differentiates the series
                    const double ifac[TM_HSER] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880};
                    double d[TM_HSER], u[TM_HSER];
                    d[0] = 1.0 + t0;
#pragma unroll
                    for (int j = 1; j < TM_HSER; j++) d[j] = t0 * ifac[j];
                    const double id0 = 1.0 / d[0];
                    u[0] = id0;
#pragma unroll
                    for (int j = 1; j < TM_HSER; j++) {
                        double acc = 0.0;
#pragma unroll
                        for (int i = 1; i <= j; i++) acc = acc + d[i] * u[j - i];
                        u[j] = -id0 * acc;
                    }
                    double pj = 1.0;
                    double *hs = (hser != nullptr) ? hser + (((size_t)chain * cells + ce) * TM_MAXH + h) * TM_HSER : nullptr;
#pragma unroll
                    for (int j = 0; j < TM_HSER; j++) {
                        const double cj = u[j] * pj;                 // coefficient of (log x - lxc)^j in u_h
                        if (j <= TM_PDEG) R.bg[j] = R.bg[j] + Hh * cj;
                        if (hs != nullptr) hs[j] = cj;
                        pj = pj * ph;
                    }
                }
            }
            R.npoly = ok ? 1 : 0;
            cell[(size_t)chain * cells + ce] = R;
        }
    }

    // ---------------- tiles: per-chain boundaries, active multiplet lists, launch ranks ----------------
    __shared__ int s_b[TM_ORDER_MAX + 2];      // tile boundaries in units (tiles <= TM_ORDER_MAX; else computed on the fly)
    __shared__ int s_tot[2];                   // total cost, cheapest unit
    const int nm = (L.family != TM_FAM_GAUSS) ? L.n_mult : 0;
    const bool eq = tm_setup_balances(units, tiles, equal_cost, cm.pad) != 0 && s_pre != nullptr;
    const int su = (units + tiles - 1) / tiles;       // uniform tiles: su units each (su <= TM_TILE_MAXU by the tile count)
    // The multiplets whose window meets the tile, in table order (this fixes the summation order of the eval kernel).
    // One GROUP of lanes per tile, one lane per multiplet: the windows come out of LDS side by side, a ballot compacts the
    // active ones in table order, and the tile's cost comes from ballots per component count -- scalar popcounts, no
    // exchange between lanes (a lane walking a tile's multiplets one after the other took 3 us of dependent LDS round
    // trips; with three waves instead of eight the pass took 2 us longer).  The waves w_first .. w_first + NW - 1 share the
    // tiles (all their lanes must make the call).  A group is a whole wave, or a half / quarter
    // of one when the chain has at most 32 / 16 multiplets: every wave then lists two / four tiles at a time.
    auto tile_pass = [&](const int w_first, const int NW) __attribute__((always_inline)) {
        const int lane = tid & 63, wv = (tid >> 6) - w_first;
        const int gshift = (nm <= 16) ? 4 : (nm <= 32) ? 5 : 6;            // log2(lanes per group)
        const int G = 1 << gshift, gpw = 64 >> gshift;                     // lanes per group, groups per wave
        const int g = lane >> gshift, gl = lane & (G - 1);
        const unsigned long long gmask = (G == 64) ? ~0ULL : ((1ULL << G) - 1ULL);
        const int asym_bit = (C.asym != 0 || asym_forced) ? 256 : 0;
        for (int t0 = wv * gpw; t0 < tiles; t0 += NW * gpw) {              // wave-uniform trip count: the ballots below need every lane
            const int tile = t0 + g;
            const bool live = tile < tiles;
            int u0 = 0, u1 = 0;
            if (live) {
                if (eq) { u0 = s_b[tile]; u1 = s_b[tile + 1]; }
                else { u0 = tm_tile_first_unit(cm, su, tile); u1 = u0 + tm_tile_units(cm, su, tile); if (u0 > units) u0 = units; if (u1 > units) u1 = units; }
                if (u1 - u0 > cm.pad) { u1 = u0 + cm.pad; if (gl == 0) atomicMax(&s_status, 3); }   // cannot happen (see above); never overrun the LDS of the eval kernel
            }
            const int base = u0 << TM_UNIT_SHIFT, end = u1 << TM_UNIT_SHIFT;
            TmActive *ti = tidx + ((size_t)chain * tiles + (live ? tile : 0)) * (nm > 0 ? nm : 1);
            int nact = 0, cost = 0;
            for (int j0 = 0; j0 < nm; j0 += G) {
                const int j = j0 + gl;
                int wmin = 0, wmax = 0, nc = 0;
                if (j < nm) { wmin = s_win[j][0]; wmax = s_win[j][1]; nc = s_win[j][2]; }
                const bool act = live && (u1 > u0) && (j < nm) && (wmin < end) && (wmax > base);
                const int sh = g << gshift;
                const unsigned long long m = (__builtin_amdgcn_ballot_w64(act) >> sh) & gmask;
                if (act) {
                    TmActive A;
                    A.idx = j; A.imin = wmin; A.imax = wmax; A.shape = nc | asym_bit;
                    ti[nact + __builtin_popcountll(m & ((1ULL << gl) - 1ULL))] = A;
                }
                // cost ~ instructions per bin of the tile's multiplets (a * ncomp + b each), from ballots per component count
                const int n3 = __builtin_popcountll((__builtin_amdgcn_ballot_w64(act && nc == 3) >> sh) & gmask);
                const int n5 = __builtin_popcountll((__builtin_amdgcn_ballot_w64(act && nc == 5) >> sh) & gmask);
                const int n7 = __builtin_popcountll((__builtin_amdgcn_ballot_w64(act && nc == 7) >> sh) & gmask);
                const int na = __builtin_popcountll(m);
                nact += na;
                cost += cm.a * ((na - n3 - n5 - n7) + 3 * n3 + 5 * n5 + 7 * n7) + cm.b * na;
            }
            cost = (cost + cm.c0) * 2 * (u1 - u0);
            if (live && gl == 0) {
                TmTileHdr H;
                H.u0 = u0; H.u1 = u1; H.nact = nact; H.cost = cost;
                thdr[(size_t)chain * tiles + tile] = H;
                // unique key: cost first, lower tile index first among equals
                if (tile < TM_ORDER_MAX) s_cost[tile] = ((cost < (1 << 20) ? cost : (1 << 20)) << 10) + (TM_ORDER_MAX - 1 - tile);
            }
        }
    };

    // wave 0: lane j derives multiplet j up to the ratio products, waits for the ratios, finishes it; chains with more
    // than 64 multiplets do the rest afterwards in one go
    TmMultFull M;        // in registers: every array index inside is a compile-time constant
#if defined(TM_SETUP_SKIP) && (TM_SETUP_SKIP & 2)   // timing-only build: no multiplet derivation
    const int n_mult = 0;
#else
    const int n_mult = (L.family != TM_FAM_GAUSS) ? L.n_mult : 0;
#endif
    const bool first = tid < 64 && tid < n_mult;
    // The windows do not depend on the m-ratios: when one wave holds every multiplet and the tiles have equal length,
    // they are published before the barrier below, and the other waves list the tiles while wave 0 completes and stores
    // the records (3 400 of the kernel's 23 000 cycles side by side instead of one after the other).
    const bool early_tiles = (n_mult <= 64) && !eq && (NT >= 128);
    if (first) {
        tm_derive_mult_pre(L, C, p, tid, M);
        if (early_tiles) { s_win[tid][0] = M.imin; s_win[tid][1] = M.imax; s_win[tid][2] = M.ncomp; }
    }
    SU_TS(3, 0);
    SU_TS(4, 64);
    SU_TS(5, 128);
    __syncthreads();     // ratios (wave 2) and the chain record are complete
    SU_TS(6, 0);
    if (L.family != TM_FAM_GAUSS && chain_rec != nullptr)   // keep the chain record for the backward kernel (gradient path)
        for (int e = tid; e < (int)(sizeof(TmChain) / sizeof(double)); e += NT)
            reinterpret_cast<double *>(chain_rec + chain)[e] = reinterpret_cast<const double *>(&C)[e];
    if (tid < 64) {
        for (int j = tid; j < n_mult; j += 64) {
            if (j == tid) { if (first) tm_mult_apply_ratios(L, C, M); }
            else tm_derive_mult(L, C, p, j, M);
            TmMult out;
            const double g2 = M.W * M.W;
            out.g2 = g2;
            if (C.asym == 0 && !asym_forced) {
                out.aA = 0.0; out.aB = 1.0; out.c2 = 0.0; out.has_asym = 0;
            } else {
                // A(x) = (1 + asym (x/f - 1))^2 + (0.5 Gamma asym / f)^2, build_lorentzian.cpp:96
                const double cc = 0.5 * M.W * C.asym / M.f;
                out.aA = C.asym / M.f; out.aB = 1.0 - C.asym; out.c2 = cc * cc; out.has_asym = 1;
            }
#pragma unroll
            for (int k = 0; k < TM_MAXM; k++) {
                if (k < M.ncomp) { out.nu2[k] = 2.0 * M.nu[k]; out.hq[k] = M.h[k] * g2; }
                else             { out.nu2[k] = 0.0;           out.hq[k] = 0.0; }
            }
            out.imin = M.imin; out.imax = M.imax; out.ncomp = M.ncomp;
            if (!early_tiles) { s_win[j][0] = M.imin; s_win[j][1] = M.imax; s_win[j][2] = M.ncomp; }
            if (M.status != 0) atomicMax(&s_status, M.status);
            mult[(size_t)chain * L.n_mult + j] = out;
            if (aux != nullptr) aux[(size_t)chain * L.n_mult + j] = M;
        }
    }
    SU_TS(7, 0);
    if (tid == 64) {       // the noise record is complete (this wave built it): out it goes; its status follows at the very end
        s_N.status = 0;
        wt[2 * chain] = Tc;                  // device copy for the eval / backward kernels
        wt[2 * chain + 1] = (L.likelihood_case == 0) ? L.like_p / Tc : 2.0 / Tc;
        noise[chain] = s_N;
    }
    if (early_tiles && tid >= 64) tile_pass(1, NT / 64 - 1);
    __syncthreads();
    SU_TS(8, 0);
#if defined(TM_SETUP_STOP) && TM_SETUP_STOP == 2
    return;   // timing-only build
#endif

    // ---------------- tiles: per-chain boundaries (balanced mode), active multiplet lists unless done, launch ranks ----------------
    if (eq) {
        // Equal-cost tiles.  (1) cost of every unit from the windows; (2) inclusive prefix sum (one wave: each lane a
        // contiguous chunk, then a wave scan); (3) every boundary by a binary search in the prefix (tm_tile_bound,
        // tamcmc_dev.h: integer arithmetic, tiles of at most TM_TILE_MAXU units guaranteed).
        for (int u = tid; u < units; u += NT) {
            const int lo = u << TM_UNIT_SHIFT, hi = lo + TM_UNIT_BINS;
            int c = cm.c0;
            for (int j = 0; j < nm; j++)
                if (s_win[j][0] < hi && s_win[j][1] > lo) c += cm.a * s_win[j][2] + cm.b;
            s_pre[u] = c;
        }
        __syncthreads();
        if (tid < 64) {
            const int ch = (units + 63) >> 6;
            const int lo = tid * ch, hi = (lo + ch < units) ? lo + ch : units;
            int sum = 0, mn = 0x7fffffff;
            for (int u = lo; u < hi; u++) { const int c = s_pre[u]; sum += c; mn = c < mn ? c : mn; }
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (tid >= off) incl += t; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const int t = __shfl_xor(mn, off, 64); mn = t < mn ? t : mn; }
            int run = incl - sum;
            for (int u = lo; u < hi; u++) { run += s_pre[u]; s_pre[u] = run; }
            if (tid == 63) { s_tot[0] = incl; s_tot[1] = mn; }
        }
        __syncthreads();
        const long long C = s_tot[0], cmin = s_tot[1];
        for (int t = tid; t <= tiles; t += NT) {
            const int b = (t == 0) ? 0 : (t == tiles) ? units : tm_tile_bound(t, tiles, units, s_pre, C, cmin);
            s_b[t] = b;
        }
        __syncthreads();
    }
    if (!early_tiles) tile_pass(0, NT / 64);
    SU_TS(9, 0);
#if defined(TM_SETUP_STOP) && TM_SETUP_STOP == 4
    return;   // timing-only build
#endif
    // launch order of this chain's tiles: costliest first (rank = number of tiles with a larger key)
    if (order != nullptr) {
        if (tiles <= TM_ORDER_MAX) {
            if (tid < 4) s_cost[tiles + tid] = -1;            // padding for the 4-wide reads below: never "costs more"
            __syncthreads();
            for (int tile = tid; tile < tiles; tile += NT) {
                const int c = s_cost[tile];
                int r = 0;
#pragma unroll 4
                for (int t2 = 0; t2 < tiles; t2 += 4) {
                    const int4 c4 = *reinterpret_cast<const int4 *>(&s_cost[t2]);
                    r += (c4.x > c) ? 1 : 0;
                    r += (c4.y > c) ? 1 : 0;
                    r += (c4.z > c) ? 1 : 0;
                    r += (c4.w > c) ? 1 : 0;
                }
                order[(size_t)chain * tiles + r] = tile;
            }
        } else {
            for (int tile = tid; tile < tiles; tile += NT) order[(size_t)chain * tiles + tile] = tile;
        }
    }
    // a tile that had to be clamped (never expected) is reported through the chain status
    SU_TS(10, 0);
    __syncthreads();
    if (tid == 64 && s_status != 0) noise[chain].status = s_status;
#ifdef TM_SU_TRACE
    SU_TS(11, 0);
    __syncthreads();
    if (tid < 20 && hser != nullptr) hser[(size_t)chain * cells * TM_MAXH * TM_HSER + tid] = (double)(long long)(s_ts[tid] - s_ts[0]);
#ifdef TM_SU_TRACE_FINE
    if (tid < 8 && hser != nullptr) hser[(size_t)chain * cells * TM_MAXH * TM_HSER + 12 + tid] = (double)(long long)(g_fine[tid] - s_ts[0]);
#endif
#endif
}
#pragma clang fp contract(fast)

#endif
