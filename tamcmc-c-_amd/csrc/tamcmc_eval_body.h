// tamcmc_eval_body.h -- device code of the eval kernel (helpers + the per-workgroup body), shared by the two launch
// shapes: tamcmc_eval_kernel (one workgroup per (chain, tile), records prepared by tamcmc_setup_kernel) and
// tamcmc_fused_kernel (short grids with one tile per chain: setup and evaluation in one launch).
// See tamcmc_eval.hip for the description of the algorithm.
#ifndef TAMCMC_EVAL_BODY_H
#define TAMCMC_EVAL_BODY_H
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"

#define TM_WAVES (TM_THREADS / 64)

// A multiplet record addressed through the CONSTANT address space: with a wave-uniform address the compiler
// fetches it with scalar loads (s_load_dwordx8/x16) into SGPRs, which VALU instructions take directly as one
// operand -- no LDS read, no VGPRs for the record.  The table was written by the previous kernel (setup), so
// it is invariant for this launch.
typedef const __attribute__((address_space(4))) TmMult *TmMultK;
typedef const __attribute__((address_space(4))) TmTileRec *TmTileRecK;
typedef const __attribute__((address_space(4))) TmNoise *TmNoiseK;
typedef const __attribute__((address_space(4))) int32_t *TmIdxK;

// 1/x for finite, normal, positive x: v_rcp_f64 + two Newton steps (no scaling / fix-up needed
// because every denominator here is bounded away from the subnormal and overflow ranges).
__device__ __forceinline__ double tm_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// ---- lane exchanges without LDS traffic -----------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (two per double, through the LDS crossbar); the reductions of the gradient
// pass issue ~50 of them per multiplet and wave and were bound by that path (profiles/README.md).  gfx950 has
// v_permlane32_swap / v_permlane16_swap for the two wide steps, and DPP covers masks 1..8 (semantics checked on the
// hardware with tools/xor_exchange_probe.hip).
template <int MASK>
__device__ __forceinline__ unsigned tm_xor_dw(unsigned v, int lane)
{
    if constexpr (MASK == 1) return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) {
        const unsigned r = __builtin_amdgcn_update_dpp(0u, v, 0x104, 0xf, 0x5, false);                // row_shl:4 -> banks 0, 2
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);                             // row_shr:4 -> banks 1, 3
    } else if constexpr (MASK == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false); // row_ror:8
    else if constexpr (MASK == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

template <int MASK>
__device__ __forceinline__ double tm_xor(double v, int lane)   // the value lane ^ MASK holds
{
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = tm_xor_dw<MASK>((unsigned)b, lane), hi = tm_xor_dw<MASK>((unsigned)(b >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// One butterfly step on a PAIR of values with the two-register swaps: returns, in lanes with the MASK bit clear,
// a[lane] + a[lane ^ MASK], and in lanes with the bit set, b[lane ^ MASK] + b[lane]  (MASK = 16 or 32 only).
template <int MASK>
__device__ __forceinline__ double tm_swap_add(double a, double b)
{
    static_assert(MASK == 16 || MASK == 32, "two-register swaps exist for 16 and 32 lanes");
    const unsigned long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    unsigned r0l, r1l, r0h, r1h;
    if constexpr (MASK == 32) {
        const auto l = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        r0l = l[0]; r1l = l[1]; r0h = h[0]; r1h = h[1];
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
        r0l = l[0]; r1l = l[1]; r0h = h[0]; r1h = h[1];
    }
    const double r0 = __longlong_as_double(((unsigned long long)r0h << 32) | r0l);
    const double r1 = __longlong_as_double(((unsigned long long)r1h << 32) | r1l);
    return r0 + r1;
}

__device__ __forceinline__ double tm_wave_sum(double v)   // lane 0 ends with the same tree as the shfl_down form
{
    const int lane = threadIdx.x & 63;
    v += tm_xor<32>(v, lane);
    v += tm_xor<16>(v, lane);
    v += tm_xor<8>(v, lane);
    v += tm_xor<4>(v, lane);
    v += tm_xor<2>(v, lane);
    v += tm_xor<1>(v, lane);
    return v;
}

// exp(z) for |z| <= 0.04: Taylor degree 8, truncation error < 0.04^9/9! = 7e-19
__device__ __forceinline__ double tm_exp_small(double z)
{
    double p = 1.0 / 40320.0;
    p = __builtin_fma(p, z, 1.0 / 5040.0);
    p = __builtin_fma(p, z, 1.0 / 720.0);
    p = __builtin_fma(p, z, 1.0 / 120.0);
    p = __builtin_fma(p, z, 1.0 / 24.0);
    p = __builtin_fma(p, z, 1.0 / 6.0);
    p = __builtin_fma(p, z, 0.5);
    p = __builtin_fma(p, z, 1.0);
    p = __builtin_fma(p, z, 1.0);
    return p;
}

// Horner evaluation of the tile's background polynomial; the coefficients are SGPR operands (scalar loads)
__device__ __forceinline__ double tm_poly(TmTileRecK tr, double z)
{
    double p = tr->bg[TM_PDEG];
#pragma unroll
    for (int j = TM_PDEG - 1; j >= 0; j--) p = __builtin_fma(p, z, tr->bg[j]);
    return p;
}

// ---- transposing butterfly: sums N values per lane over the 64 lanes of a wave -------------------
// One step with lane mask MASK halves the number of live values: lanes with the bit set keep the upper
// half.  After all steps lane L holds the total of slot tm_bfly_slot<N>(L) (replicated over the lane
// bits that were reduced plainly).  Cost: ~N exchanges instead of 6N.
template <int N, int MASK>
struct TmBfly {
    static constexpr int NE = (N + 1) & ~1;   // padded to even
    static constexpr int H = NE / 2;
    __device__ static __forceinline__ void run(double (&v)[TM_GSLOTS], int lane)
    {
        if constexpr (N == 1) {
            v[0] += tm_xor<MASK>(v[0], lane);
        } else if constexpr (MASK >= 16) {
#pragma unroll
            for (int i = 0; i < H; i++) v[i] = tm_swap_add<MASK>(v[i], (i + H < N) ? v[i + H] : 0.0);
        } else {
            const bool hi = (lane & MASK) != 0;
#pragma unroll
            for (int i = 0; i < H; i++) {
                const double up = (i + H < N) ? v[i + H] : 0.0;
                const double send = hi ? v[i] : up;
                const double keep = hi ? up : v[i];
                v[i] = keep + tm_xor<MASK>(send, lane);
            }
        }
        if constexpr (MASK > 1) TmBfly<(N == 1 ? 1 : H), MASK / 2>::run(v, lane);
    }
    // index (at this level) of the value lane `lane` ends up holding; valid = false if it is padding
    __device__ static __forceinline__ int slot_of(int lane, bool &valid)
    {
        if constexpr (N == 1) {
            return 0;
        } else {
            int sub = 0;
            if constexpr (MASK > 1) sub = TmBfly<H, MASK / 2>::slot_of(lane, valid);
            const int idx = sub + (((lane & MASK) != 0) ? H : 0);
            if (idx >= N) valid = false;
            return idx;
        }
    }
};

// Sum over the m-components of one multiplet at one bin, also returning d_m and every 1/E_m.
template <int NC>
__device__ __forceinline__ double tm_mult_value(double x2, const double (&nu2)[NC], const double (&hq)[NC], double g2,
                                                double (&d)[NC], double (&r)[NC])
{
    double E[NC], P[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) {
        d[m] = x2 - nu2[m];
        E[m] = __builtin_fma(d[m], d[m], g2);
    }
    P[0] = E[0];
#pragma unroll
    for (int m = 1; m < NC; m++) P[m] = P[m - 1] * E[m];
    double inv = tm_rcp(P[NC - 1]);
    double s = 0.0;
#pragma unroll
    for (int m = NC - 1; m >= 1; m--) {
        r[m] = inv * P[m - 1];
        inv = inv * E[m];
    }
    r[0] = inv;
#pragma unroll
    for (int m = 0; m < NC; m++) s = __builtin_fma(hq[m], r[m], s);
    return s;
}

// Forward only: the same sum as one rational N/Q built term by term (N <- N E_m + h_m Q, Q <- Q E_m: 3 ops per
// component instead of the 4 of batch inversion, whose individual 1/E_m only the gradient needs).
template <int NC>
__device__ __forceinline__ double tm_mult_sum(double x2, const double (&nu2)[NC], const double (&hq)[NC], double g2)
{
    double d = x2 - nu2[0];
    double Q = __builtin_fma(d, d, g2);
    double N = hq[0];
#pragma unroll
    for (int m = 1; m < NC; m++) {
        d = x2 - nu2[m];
        const double E = __builtin_fma(d, d, g2);
        N = __builtin_fma(hq[m], Q, N * E);
        Q = Q * E;
    }
    return N * tm_rcp(Q);
}

// Forward: add multiplet `sm` (LDS) to acc[] for KU bins.
template <int NC, int KU, bool ASYM>
__device__ __forceinline__ void tm_accum_mult(TmMultK sm, const double (&x2)[KU], const int (&bi)[KU], double (&acc)[KU])
{
    double nu2[NC], hq[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) { nu2[m] = sm->nu2[m]; hq[m] = sm->hq[m]; }
    const double g2 = sm->g2;
    const int imin = sm->imin, imax = sm->imax;
    const double aAh = ASYM ? 0.5 * sm->aA : 0.0, aB = ASYM ? sm->aB : 1.0, c2 = ASYM ? sm->c2 : 0.0;
#pragma unroll
    for (int k = 0; k < KU; k++) {
        double s = tm_mult_sum<NC>(x2[k], nu2, hq, g2);
        if (ASYM) {
            const double a = __builtin_fma(x2[k], aAh, aB);
            s = s * __builtin_fma(a, a, c2);
        }
        const bool inside = (bi[k] >= imin) && (bi[k] < imax);
        acc[k] += inside ? s : 0.0;
    }
}

// Backward: accumulate this thread's partial sums for multiplet `sm` over all S sub-blocks, reduce over
// the wave and leave the wave totals in s_red_row[slot].
//   g[3m+0] = sum wA r_m, g[3m+1] = sum wA d_m r_m^2, g[3m+2] = sum wA r_m^2   (d = 2x - 2nu, r = 1/E)
//   g[3NC..3NC+2] = sum w S, sum w S a, sum w S a x   (S = un-asymmetrised multiplet sum; zero if asym == 0)
template <int NC, int KU, bool ASYM>
__device__ __forceinline__ void tm_grad_mult(TmMultK sm, const double *__restrict__ gx, const double *s_w,
                                             int base, int S, int Nx, int tid, int lane, double *s_red_row)
{
    constexpr int V = ASYM ? 3 * NC + 3 : 3 * NC;
    double g[TM_GSLOTS];
    double nu2[NC], hq[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) { nu2[m] = sm->nu2[m]; hq[m] = sm->hq[m]; }
    const double g2 = sm->g2;
    const int imin = sm->imin, imax = sm->imax;
    const double aAh = ASYM ? 0.5 * sm->aA : 0.0, aB = ASYM ? sm->aB : 1.0, c2 = ASYM ? sm->c2 : 0.0;
#pragma unroll
    for (int s = 0; s < TM_GSLOTS; s++) g[s] = 0.0;
#pragma unroll 1
    for (int u = 0; u < S; u++) {
        // skip sub-blocks that do not meet the window (wave-uniform: bounds of the whole sub-block)
        const int lo = base + u * KU * TM_THREADS, hi = lo + KU * TM_THREADS;
        if (hi <= imin || lo >= imax) continue;
#pragma unroll
        for (int k = 0; k < KU; k++) {
            const int i = lo + k * TM_THREADS + tid;
            const bool inside = (i >= imin) && (i < imax) && (i < Nx);
            const double x2 = 2.0 * gx[i < Nx ? i : Nx - 1];
            const double wk = inside ? s_w[(u * KU + k) * TM_THREADS + tid] : 0.0;
            double d[NC], r[NC];
            const double Sv = tm_mult_value<NC>(x2, nu2, hq, g2, d, r);
            double wA = wk;
            if (ASYM) {
                const double a = __builtin_fma(x2, aAh, aB);
                wA = wk * __builtin_fma(a, a, c2);
                const double ws = wk * Sv;
                g[3 * NC + 0] += ws;
                g[3 * NC + 1] = __builtin_fma(ws, a, g[3 * NC + 1]);
                g[3 * NC + 2] = __builtin_fma(ws * a, 0.5 * x2, g[3 * NC + 2]);
            }
#pragma unroll
            for (int m = 0; m < NC; m++) {
                const double t1 = wA * r[m];
                const double t2 = t1 * r[m];
                g[3 * m + 0] += t1;
                g[3 * m + 1] = __builtin_fma(t2, d[m], g[3 * m + 1]);
                g[3 * m + 2] += t2;
            }
        }
    }
#if defined(TM_ABLATE) && (TM_ABLATE & 1)   // timing-only build: no wave reduction of the partials
    if (lane == 0) s_red_row[0] = g[0] + g[1] + g[2];
    return;
#endif
    TmBfly<V, 32>::run(g, lane);
    // lanes whose plainly-reduced low bits are zero publish; padded slots are skipped
    constexpr int NSPLIT = (V > 16) ? 5 : (V > 8) ? 4 : (V > 4) ? 3 : (V > 2) ? 2 : 1;   // butterfly levels that split
    const int lowmask = (64 >> NSPLIT) - 1;
    bool valid = true;
    const int slot = TmBfly<V, 32>::slot_of(lane, valid);
    if ((lane & lowmask) == 0 && valid) {
        const int dst = (slot < 3 * NC) ? slot : 21 + (slot - 3 * NC);
        s_red_row[dst] = g[0];
    }
}

#ifndef TM_UG_FWD
#define TM_UG_FWD 1     // sub-blocks per multiplet-loop group, likelihood kernel (KU = 4: already 4 chains per record fetch)
#endif
#ifndef TM_UG_GRAD
#define TM_UG_GRAD 2    // gradient kernel (KU = 2): 4 bins per record fetch
#endif
#ifndef TM_LB_FWD
#define TM_LB_FWD 4    // resident waves per SIMD the register allocator must allow (likelihood-only kernel)
#endif
#ifndef TM_LB_GRAD
#define TM_LB_GRAD 3   // same for the gradient kernel: 155 VGPRs at KU=2 once the multiplet records live in SGPRs
#endif
#ifdef TM_TRACE   // developer build: per-workgroup time stamps (tools/block_trace.py)
__device__ unsigned long long *g_tm_trace = nullptr;
#define TM_STAMP(slot) do { if (threadIdx.x == 0 && g_tm_trace) g_tm_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define TM_STAMP(slot) do { } while (0)
#endif
// The workgroup's work for (chain, tile): pass 1 (model, likelihood partials, weights), in-launch finalize or pass 2.
// s_w: dynamic LDS for the weights, [TM_THREADS * KU * S] doubles (GRAD only).  Called by tamcmc_eval_kernel
// (tamcmc_eval.hip) and by the fused small-grid kernel (tamcmc_fused.hip).
template <int KU, bool GRAD>
__device__ __forceinline__ void tm_eval_body(const TmEvalArgs &a, const int chain, const int tile, double *s_w)
{
    const int tid = threadIdx.x;
    TmNoiseK sn = (TmNoiseK)(a.noise + chain);
    const int u_first = TM_TILE_U0(tile, a.tile_big, a.tile_small);
    const int S = TM_TILE_S(tile, a.tile_big, a.tile_small, a.units);   // sub-blocks of this tile
    constexpr int KU2 = (KU > 2) ? 2 : KU;   // pass 2 keeps 3 accumulators per component: fewer bins in flight
    const int Sp2 = S * (KU / KU2);          // sub-blocks of pass 2
    const int lane = tid & 63, wave = tid >> 6;
    const int base = u_first * (TM_THREADS * KU);

#if defined(TM_ABLATE) && (TM_ABLATE & 16)   // timing-only build: empty workgroups (dispatch cost only)
    if (a.Nx > 0) return;
#endif
    __shared__ double s_red[2][TM_WAVES][TM_GSLOTS];   // double-buffered: one barrier per multiplet in pass 2

    // ---------------- prologue: nothing but scalar loads ----------------
    // Everything a tile needs -- its list of active multiplets (table order), the background polynomial, the
    // chain's noise record, the multiplet records themselves -- was prepared by the setup kernel and is
    // wave-uniform: it is fetched with s_load into SGPRs.  No LDS staging, no barrier before the first bin.
    // the chain's spectrum: contexts holding several spectra on one grid (ensembles, tamcmc_ctx_set_spectra) keep them as
    // consecutive blocks of Nx
    // (read through the constant address space: a scalar load, so that the two base pointers stay in SGPRs)
    const size_t spec_off = (a.spec != nullptr) ? (size_t)((TmIdxK)a.spec)[chain] * (size_t)a.Nx : 0;
    const double *__restrict__ yp = a.y + spec_off;
    const double *__restrict__ isp = a.isig2 + spec_off;      // dereferenced only when the likelihood is chi_square (then non-NULL)
    const TmMult *__restrict__ gm = a.mult + (size_t)chain * a.n_mult;
    TmTileRecK tr = (TmTileRecK)(a.trec + (size_t)chain * a.tiles + tile);
    TmIdxK tix = (TmIdxK)(a.tidx + ((size_t)chain * a.tiles + tile) * (a.n_mult > 0 ? a.n_mult : 1));
    const int nh = sn->nh;
    const int nact = tr->nact;
    const bool npoly = tr->npoly != 0;
    const double lxc = tr->lxc;
    const bool has_gauss = sn->has_gauss != 0;
    const double N0 = sn->N0;
    const int row = (a.row_of_chain != nullptr) ? a.row_of_chain[chain] : -1;

#if defined(TM_ABLATE) && (TM_ABLATE & 32)   // timing-only build: prologue (scalar loads) only
    if (nact >= 0 && lxc == lxc && N0 == N0) { if (tid == 0 && !GRAD) a.part[((size_t)chain * a.tiles + tile) * 2] = (double)(nh + row + (npoly ? 1 : 0) + (has_gauss ? 1 : 0)); return; }
#endif
    // ---------------- pass 1: model spectrum and likelihood partial sums ----------------
    double S1 = 0.0;
    double P = 1.0;
    int esum = 0;
    double Mmin = 1.0;
    double gn_[GRAD ? TM_GSLOTS : 1];   // GRAD: noise partial sums (3 per Harvey, sum w at slot 3*TM_MAXH, Gaussian at 13..15)
#pragma unroll
    for (int s_ = 0; s_ < (GRAD ? TM_GSLOTS : 1); s_++) gn_[s_] = 0.0;
    const double wscale = GRAD ? a.wt[2 * chain + 1] : 0.0;

    // The multiplet loop runs on UG sub-blocks at a time (UG * KU bins per thread): a multiplet's record is fetched
    // (scalar loads) once per group, and UG * KU independent rational chains are in flight; background, likelihood
    // and the gradient weights are then done sub-block by sub-block as before.
    constexpr int UG = GRAD ? TM_UG_GRAD : TM_UG_FWD, KG = UG * KU;
#pragma unroll 1
    for (int ug = 0; ug < S; ug += UG) {
        double x2g[KG], accg[KG];
        int big[KG];
#pragma unroll
        for (int k = 0; k < KG; k++) {
            const int i = base + (ug * KU + k) * TM_THREADS + tid;
            const bool valid = (i < a.Nx) && (ug + k / KU < S);
            big[k] = valid ? i : -1;
            x2g[k] = 2.0 * a.x[(i < a.Nx) ? i : a.Nx - 1];
            accg[k] = 0.0;
        }
        {
            const int lo = base + ug * KU * TM_THREADS;
            const int hi = lo + ((ug + UG <= S) ? KG : (S - ug) * KU) * TM_THREADS;
            for (int jj = 0; jj < nact; jj++) {
                TmMultK sm = (TmMultK)(gm + tix[jj]);
                if (hi <= sm->imin || lo >= sm->imax) continue;   // wave-uniform skip
                if (sm->has_asym) {
                    switch (sm->ncomp) {
                    case 1: tm_accum_mult<1, KG, true>(sm, x2g, big, accg); break;
                    case 3: tm_accum_mult<3, KG, true>(sm, x2g, big, accg); break;
                    case 5: tm_accum_mult<5, KG, true>(sm, x2g, big, accg); break;
                    default: tm_accum_mult<7, KG, true>(sm, x2g, big, accg); break;
                    }
                } else {
                    switch (sm->ncomp) {
                    case 1: tm_accum_mult<1, KG, false>(sm, x2g, big, accg); break;
                    case 3: tm_accum_mult<3, KG, false>(sm, x2g, big, accg); break;
                    case 5: tm_accum_mult<5, KG, false>(sm, x2g, big, accg); break;
                    default: tm_accum_mult<7, KG, false>(sm, x2g, big, accg); break;
                    }
                }
            }
        }
#pragma unroll
      for (int uu = 0; uu < UG; uu++) {
        const int u = ug + uu;
        if (u >= S) break;
        double x2[KU], acc[KU], wreg[GRAD ? KU : 1];
        int bi[KU];
#pragma unroll
        for (int k = 0; k < KU; k++) { x2[k] = x2g[uu * KU + k]; acc[k] = accg[uu * KU + k]; bi[k] = big[uu * KU + k]; }
        double hu[TM_MAXH][GRAD ? KU : 1], harg[GRAD ? KU : 1];   // GRAD: u = 1/(1+t) per Harvey and bin (t u = 1 - u), log x per bin
        bool n0_done = false;
        if (nh > 0) {
            double dl[KU];
#pragma unroll
#if defined(TM_ABLATE) && (TM_ABLATE & 128)   // timing-only build: no log x load (wrong values)
            for (int k = 0; k < KU; k++) { dl[k] = lxc + 1e-9 * x2[k]; if constexpr (GRAD) harg[k] = dl[k]; }
#else
            for (int k = 0; k < KU; k++) { dl[k] = a.lx[bi[k] >= 0 ? bi[k] : a.Nx - 1]; if constexpr (GRAD) harg[k] = dl[k]; }
#endif
            if (npoly) {
#pragma unroll
                for (int k = 0; k < KU; k++) dl[k] -= lxc;
                // whole background (all profiles + white noise) as ONE polynomial: 8 FMAs per bin.  The gradient path
                // accumulates the moments sum w dl^j instead of per-profile sums (below); the backward kernel turns
                // them into the per-profile sums with each profile's own series coefficients.
#pragma unroll
                for (int k = 0; k < KU; k++) { acc[k] += tm_poly(tr, dl[k]); if constexpr (GRAD) harg[k] = dl[k]; }
                n0_done = true;
            } else {
#pragma unroll
                for (int h = 0; h < TM_MAXH; h++) {
                    if (h < nh) {
                        const double Hh = sn->H[h], ph = sn->p[h], lth = sn->lt[h];
#pragma unroll
                        for (int k = 0; k < KU; k++) {
                            const double t = exp(ph * (lth + dl[k]));
                            const double uu = 1.0 / (t + 1.0);
                            acc[k] += Hh * uu;
                            if constexpr (GRAD) hu[h][k] = uu;
                        }
                    }
                }
            }
        }
        if (has_gauss) {
            const double gA = sn->gA, gnu0 = sn->gnu0, gs2 = sn->gs2;
#pragma unroll
            for (int k = 0; k < KU; k++) {
                const double dd = 0.5 * x2[k] - gnu0;
                acc[k] = gA * exp((-0.5 * (dd * dd)) / gs2) + acc[k];
            }
        }
        if (!n0_done) {
#pragma unroll
            for (int k = 0; k < KU; k++) acc[k] += N0;
        }

        if (row >= 0) {
#pragma unroll
            for (int k = 0; k < KU; k++)
                if (bi[k] >= 0) a.model_out[(size_t)row * a.Nx + bi[k]] = acc[k];
        }

        if (a.likelihood_case == 0) {
            // -p * (sum y/M + sum log M), likelihoods.cpp:23-25
#pragma unroll
            for (int k = 0; k < KU; k++) {
                double wv = 0.0;
                if (bi[k] >= 0) {
                    // A model value that is NaN, +-inf or 0 turns rM (and so S1) into NaN by itself (v_rcp_f64 + the
                    // Newton steps); a negative one is caught by the running minimum.  Either way logL becomes NaN,
                    // which is what log() of such a value gives the reference.
                    const double M = acc[k];
                    const double yv = yp[bi[k]];
                    Mmin = __builtin_fmin(Mmin, M);
                    const double rM = tm_rcp(M);
                    S1 = __builtin_fma(yv, rM, S1);
                    int e;
                    P *= frexp(M, &e);
                    esum += e;
                    wv = wscale * (yv * rM * rM - rM);   // d(logL/T)/dM_i
                }
                if constexpr (GRAD) { s_w[(u * KU + k) * TM_THREADS + tid] = wv; wreg[k] = wv; }
            }
            {
                int e;
                P = frexp(P, &e);   // keep the running mantissa product in [0.5, 1)
                esum += e;
            }
        } else {
            // -sum (y-M)^2 / sigma^2, likelihoods.cpp:36
#pragma unroll
            for (int k = 0; k < KU; k++) {
                double wv = 0.0;
                if (bi[k] >= 0) {
                    const double dd = yp[bi[k]] - acc[k];
                    const double is2 = isp[bi[k]];
                    S1 = __builtin_fma(dd * dd, is2, S1);
                    wv = wscale * dd * is2;
                }
                if constexpr (GRAD) { s_w[(u * KU + k) * TM_THREADS + tid] = wv; wreg[k] = wv; }
            }
        }
        if constexpr (GRAD) {
            // noise partial sums, reusing u and t*u of this sub-block (no second pass over the bins)
#pragma unroll
            for (int k = 0; k < KU; k++) {
                const double wk = wreg[k];
                gn_[3 * TM_MAXH] += wk;
                if (nh > 0 && npoly) {
                    // moments of the weights about the tile centre: slot j-1 <- sum w dl^j, j = 1..TM_HSER-1
                    double q = wk;
#pragma unroll
                    for (int j = 0; j < TM_HSER - 1; j++) { q *= harg[k]; gn_[j] += q; }
                } else if (nh > 0) {
#pragma unroll
                    for (int h = 0; h < TM_MAXH; h++) {
                        if (h < nh) {
                            const double wu = wk * hu[h][k];
                            const double tu2 = wu * (1.0 - hu[h][k]);   // t u = 1 - u
                            gn_[3 * h] += wu;
                            gn_[3 * h + 1] += tu2;
                            gn_[3 * h + 2] = __builtin_fma(tu2, sn->lt[h] + harg[k], gn_[3 * h + 2]);
                        }
                    }
                }
                if (has_gauss) {
                    // Gaussian term: sum w e, sum w e d, sum w e d^2 with e = exp(-0.5 d^2/s2), d = x - nu0
                    const double dd = 0.5 * x2[k] - sn->gnu0;
                    const double we = wk * exp((-0.5 * (dd * dd)) / sn->gs2);
                    gn_[13] += we;
                    gn_[14] = __builtin_fma(we, dd, gn_[14]);
                    gn_[15] = __builtin_fma(we * dd, dd, gn_[15]);
                }
            }
        }
      }   // sub-blocks of the group
    }
    TM_STAMP(1);
    double S2 = 0.0;
#if defined(TM_ABLATE) && (TM_ABLATE & 8)    // timing-only build: no log in the epilogue
    S2 = P + (double)esum + Mmin;
#else
    if (a.likelihood_case == 0) {
        S2 = log(P) + (double)esum * 0.693147180559945309417232;
        if (!(Mmin > 0.0)) S2 = __builtin_nan("");
    }
#endif
    S1 = tm_wave_sum(S1);
    S2 = tm_wave_sum(S2);
    if (lane == 0) { s_red[0][wave][0] = S1; s_red[0][wave][1] = S2; }
    __syncthreads();
    if constexpr (GRAD) {
        if (tid == 0) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int wv = 0; wv < TM_WAVES; wv++) { t1 += s_red[0][wv][0]; t2 += s_red[0][wv][1]; }
            double *out = a.part + ((size_t)chain * a.tiles + tile) * 2;   // summed by the backward kernel
            out[0] = t1;
            out[1] = t2;
        }
    } else if (wave == 0) {
        // Finalize inside this launch: the workgroup that publishes the LAST tile partial of a chain sums all of them
        // in a fixed order (lane-strided, then a wave tree), so the result does not depend on which workgroup that is.
        // Hand-off without cache-wide fences: the partials are written and read with agent-scope relaxed atomics
        // (8-byte write-through stores / L1-bypassing loads), the publisher drains its stores (vmcnt(0)) before it
        // takes its ticket, and the last arriver reads only after its ticket returned -- placement-independent.
        int last = 0;
        double *pp = a.part + (size_t)chain * a.tiles * 2;
        if (lane == 0) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int wv = 0; wv < TM_WAVES; wv++) { t1 += s_red[0][wv][0]; t2 += s_red[0][wv][1]; }
#if defined(TM_ABLATE) && (TM_ABLATE & 4)    // timing-only build: plain stores, no ticket, no finalize
            pp[2 * tile] = t1; pp[2 * tile + 1] = t2;
#else
            if (a.tiles == 1) {
                // the chain's only tile: nothing to hand over.  Same value as the general path below, which would add
                // zeros to t1 and t2 (the other lanes' empty strides), without its store-drain-ticket-reload round trip
                pp[0] = t1; pp[1] = t2;
                double f = (a.likelihood_case == 0) ? -a.like_p * (t1 + t2) : -t1;
                f = f / a.wt[2 * chain];
                int st = a.noise[chain].status;
                if (st != 0) f = __builtin_nan("");
                else if (!(f == f)) st = 1;
                a.logL[chain] = f;
                if (a.status) a.status[chain] = st;
            } else {
            __hip_atomic_store(pp + 2 * tile, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 2 * tile + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int prev = __hip_atomic_fetch_add(a.ticket + chain, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (prev == a.tiles - 1) ? 1 : 0;
            }
#endif
        }
        last = __shfl(last, 0, 64);
        if (last) {
            double s1 = 0.0, s2 = 0.0;
            for (int t = lane; t < a.tiles; t += 64) {
                s1 += __hip_atomic_load(pp + 2 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(pp + 2 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s1 = tm_wave_sum(s1);
            s2 = tm_wave_sum(s2);
            if (lane == 0) {
                double f = (a.likelihood_case == 0) ? -a.like_p * (s1 + s2) : -s1;
                f = f / a.wt[2 * chain];
                int st = a.noise[chain].status;
                if (st != 0) f = __builtin_nan("");
                else if (!(f == f)) st = 1;
                a.logL[chain] = f;
                if (a.status) a.status[chain] = st;
                __hip_atomic_store(a.ticket + chain, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
        }
    }

    TM_STAMP(2);
#ifdef TM_TRACE
    if (threadIdx.x == 0 && g_tm_trace) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_tm_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // ---------------- pass 2: gradient partial sums ----------------
    if constexpr (GRAD) {
        __syncthreads();   // s_red[0] (likelihood) consumed; s_w complete
        // noise partial sums were accumulated in pass 1 (gn_): reduce and publish them
        {
            static_assert(TM_NSLOTS == 16, "butterfly below assumes 16 noise slots");
            TmBfly<TM_NSLOTS, 32>::run(gn_, lane);
            bool valid = true;
            const int slot = TmBfly<TM_NSLOTS, 32>::slot_of(lane, valid);
            double *red = s_red[1][wave];   // the buffer the last multiplet did not use
            if ((lane & 3) == 0) red[slot] = gn_[0];
        }
        __syncthreads();
        if (tid < TM_NSLOTS) {
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[1][wv][tid];
            a.gnoise[((size_t)chain * a.tiles + tile) * TM_NSLOTS + tid] = t;
        }
        __syncthreads();   // s_red[1] (noise) consumed before multiplet 0 reuses it
#if defined(TM_ABLATE) && (TM_ABLATE & 2)   // timing-only build: no multiplet pass 2 at all
        for (int jj = 0; jj < 0; jj++) {
#else
        for (int jj = 0; jj < nact; jj++) {
#endif
            TmMultK sm = (TmMultK)(gm + tix[jj]);
            const int nc = sm->ncomp;
            double *red = s_red[(jj + 1) & 1][wave];
            if (sm->has_asym) {
                switch (nc) {
                case 1: tm_grad_mult<1, KU2, true>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                case 3: tm_grad_mult<3, KU2, true>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                case 5: tm_grad_mult<5, KU2, true>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                default: tm_grad_mult<7, KU2, true>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                }
            } else {
                if (lane == 0) { red[21] = 0.0; red[22] = 0.0; red[23] = 0.0; }
                switch (nc) {
                case 1: tm_grad_mult<1, KU2, false>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                case 3: tm_grad_mult<3, KU2, false>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                case 5: tm_grad_mult<5, KU2, false>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                default: tm_grad_mult<7, KU2, false>(sm, a.x, s_w, base, Sp2, a.Nx, tid, lane, red); break;
                }
            }
#if defined(TM_ABLATE) && (TM_ABLATE & 64)   // timing-only build: no workgroup barrier per multiplet (wave 0 publishes its own partial)
            if (tid < TM_GSLOTS) a.gmult[(((size_t)chain * a.tiles + tile) * a.n_mult + tix[jj]) * TM_GSLOTS + tid] = red[tid];
#else
            __syncthreads();
            if (tid < TM_GSLOTS) {
                double t = 0.0;
                if (tid < 3 * nc || tid >= 21) {
#pragma unroll
                    for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[(jj + 1) & 1][wv][tid];
                }
                a.gmult[(((size_t)chain * a.tiles + tile) * a.n_mult + tix[jj]) * TM_GSLOTS + tid] = t;
            }
#endif
        }
        TM_STAMP(2);   // gradient kernel: slot 2 = end of pass 2
    }
}

#endif
