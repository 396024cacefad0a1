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
typedef const __attribute__((address_space(4))) TmCellRec *TmCellRecK;
typedef const __attribute__((address_space(4))) TmTileHdr *TmTileHdrK;
typedef const __attribute__((address_space(4))) TmNoise *TmNoiseK;
typedef const __attribute__((address_space(4))) int32_t *TmIdxK;
typedef const __attribute__((address_space(4))) TmActive *TmActiveK;

// One entry of a tile's active list as ONE 16-byte scalar load (s_load_dwordx4): {idx, imin, imax, shape}.
typedef int TmEntry __attribute__((ext_vector_type(4)));
__device__ __forceinline__ TmEntry tm_load_entry(TmActiveK p)
{
    return *reinterpret_cast<const __attribute__((address_space(4))) TmEntry *>(p);
}

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

// The same with one Newton step: 2.2e-15 relative (tools/rcp_accuracy.hip).  Used by pass 2 of the gradient kernel only
// -- the gradient is new functionality checked against finite differences (2e-5), and its sums carry cancellation far
// above that level; everything that enters logL keeps the two-step form.
__device__ __forceinline__ double tm_rcp1(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
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

__device__ __forceinline__ double tm_wave_prod(double v)   // product over the 64 lanes (every lane ends with it)
{
    const int lane = threadIdx.x & 63;
    v *= tm_xor<32>(v, lane);
    v *= tm_xor<16>(v, lane);
    v *= tm_xor<8>(v, lane);
    v *= tm_xor<4>(v, lane);
    v *= tm_xor<2>(v, lane);
    v *= tm_xor<1>(v, lane);
    return v;
}

// sum of log M over a tile, kept as (mantissa product, exponent sum): sum log M = log(mant) + e ln 2.  The product of the
// 64 lanes' running mantissas (each in [0.5, 1): at least 2^-64, no underflow) and then of the 4 waves' replaces a log per
// thread; the logs are taken once per tile by whoever sums the tiles (tm_tile_logsum).
// (tm_tile_logsum: tamcmc_dev.h, shared with the backward kernel)

// Horner evaluation of the cell's background polynomial; the coefficients are SGPR operands (scalar loads)
__device__ __forceinline__ double tm_poly(TmCellRecK tr, double z)
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
    double inv = tm_rcp1(P[NC - 1]);
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

// Forward: add multiplet `sm` to acc[] for KG bins.  edge (wave-uniform) = false: the whole group lies inside the
// multiplet's window [imin, imax) and inside the grid, so no bin needs a test; edge = true: bins outside the window
// (bi[k] = -1 marks a bin outside the grid) get nothing -- the reference adds the multiplet inside its window only
// (build_lorentzian.cpp:423-427).  One copy of the arithmetic; the tests sit behind a scalar branch.
template <int NC, int KG, bool ASYM>
__device__ __forceinline__ void tm_accum_mult(TmMultK sm, const double (&x2)[KG], const int (&bi)[KG], double (&acc)[KG], const bool edge)
{
    // (the window itself is read from the record only on the edge path: rare, and the record is in flight anyway)
    double nu2[NC], hq[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) { nu2[m] = sm->nu2[m]; hq[m] = sm->hq[m]; }
    const double g2 = sm->g2;
    const double aAh = ASYM ? 0.5 * sm->aA : 0.0, aB = ASYM ? sm->aB : 1.0, c2 = ASYM ? sm->c2 : 0.0;
    double s[KG];
#pragma unroll
    for (int k = 0; k < KG; k++) {
        s[k] = tm_mult_sum<NC>(x2[k], nu2, hq, g2);
        if (ASYM) {
            const double a = __builtin_fma(x2[k], aAh, aB);
            s[k] = s[k] * __builtin_fma(a, a, c2);
        }
    }
    if (edge) {
        const int imin = sm->imin, imax = sm->imax;
#pragma unroll
        for (int k = 0; k < KG; k++) {
            const bool inside = (bi[k] >= imin) && (bi[k] < imax);
            s[k] = inside ? s[k] : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < KG; k++) acc[k] += s[k];
}

template <int KG>
__device__ __forceinline__ void tm_accum_dispatch(TmMultK sm, const int shape, const double (&x2)[KG], const int (&bi)[KG], double (&acc)[KG], const bool edge)
{
    // shape = ncomp | has_asym << 8, from the active list (known before the record arrives)
    switch (shape) {
    case 1: tm_accum_mult<1, KG, false>(sm, x2, bi, acc, edge); break;
    case 3: tm_accum_mult<3, KG, false>(sm, x2, bi, acc, edge); break;
    case 5: tm_accum_mult<5, KG, false>(sm, x2, bi, acc, edge); break;
    case 7: tm_accum_mult<7, KG, false>(sm, x2, bi, acc, edge); break;
    case 256 + 1: tm_accum_mult<1, KG, true>(sm, x2, bi, acc, edge); break;
    case 256 + 3: tm_accum_mult<3, KG, true>(sm, x2, bi, acc, edge); break;
    case 256 + 5: tm_accum_mult<5, KG, true>(sm, x2, bi, acc, edge); break;
    default: tm_accum_mult<7, KG, true>(sm, x2, bi, acc, edge); break;
    }
}

// Backward, one unit (2 bins per thread) of one multiplet: add this thread's terms to g[].  With d = 2x - 2nu, r = 1/E,
// wA the weight times the asymmetry factor, L = (NC - 1) / 2, the chain rule needs (tamcmc_backward.hip):
//   g[k]           = sum wA d_k r_k^2                      k = 0 .. NC-1   (d/d nu_k: every component has its own)
//   g[NC + am]     = sum wA (r_{L-am} + r_{L+am})          am = 0 .. L     (d/d height: the components +-m share theirs)
//   g[NC + L + 1]  = sum wA sum_k hq_k r_k^2                               (d/d Gamma^2: one per multiplet)
//   g[V-3 .. V-1]  = sum w S, sum w S a, sum w S a x   (S = un-asymmetrised multiplet sum; only if asym != 0)
// -- NC + L + 2 sums instead of 3 NC: fewer values to reduce over the workgroup per multiplet and tile.
// edge (wave-uniform) = false: the unit lies inside the window and the grid (no test per bin).  Only the fetch of x and
// of the weight differs between the two cases; the arithmetic exists once (two copies updating g[] behind a branch cost
// ~40 registers in copies at the join).
template <int NC> struct TmGradSlots {
    static constexpr int L = (NC - 1) / 2;
    static constexpr int A0 = NC, C = NC + L + 1, ASYM0 = NC + L + 2;
    __device__ static constexpr int count(bool asym) { return asym ? ASYM0 + 3 : ASYM0; }
};
// where the compact slot v of a multiplet with NC components is kept in a gmult row (TM_GSLOTS doubles):
// B_k at k, A_am at 7 + am, C at 11, the asymmetry sums at 21 .. 23
template <int NC> __device__ __forceinline__ int tm_grad_store_slot(int v)
{
    using SL = TmGradSlots<NC>;
    return (v < NC) ? v : (v < SL::C) ? 7 + (v - NC) : (v == SL::C) ? 11 : 21 + (v - SL::ASYM0);
}

template <int NC, bool ASYM>
__device__ __forceinline__ void tm_grad_unit(const double (&nu2)[NC], const double (&hq)[NC], double g2, double aAh, double aB, double c2,
                                             int imin, int imax, const double *__restrict__ gx, const double *wq, int i0, int Nx,
                                             const bool edge, double (&g)[TM_GSLOTS])
{
    using SL = TmGradSlots<NC>;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        double x2, wk;
        if (edge) {
            const int i = i0 + k * TM_THREADS;
            const bool inside = (i >= imin) && (i < imax) && (i < Nx);
            x2 = gx[i < Nx ? i : Nx - 1];
            wk = inside ? wq[k * TM_THREADS] : 0.0;
        } else {
            x2 = gx[i0 + k * TM_THREADS];
            wk = wq[k * TM_THREADS];
        }
        double d[NC], r[NC];
        const double Sv = tm_mult_value<NC>(x2, nu2, hq, g2, d, r);
        double wA = wk;
        if (ASYM) {
            const double a = __builtin_fma(x2, aAh, aB);
            wA = wk * __builtin_fma(a, a, c2);
            const double ws = wk * Sv;
            g[SL::ASYM0 + 0] += ws;
            g[SL::ASYM0 + 1] = __builtin_fma(ws, a, g[SL::ASYM0 + 1]);
            g[SL::ASYM0 + 2] = __builtin_fma(ws * a, 0.5 * x2, g[SL::ASYM0 + 2]);
        }
#pragma unroll
        for (int m = 0; m < NC; m++) {
            const double t1 = wA * r[m];
            const double t2 = t1 * r[m];
            const int am = (m >= SL::L) ? m - SL::L : SL::L - m;
            g[SL::A0 + am] += t1;
            g[m] = __builtin_fma(t2, d[m], g[m]);
            g[SL::C] = __builtin_fma(hq[m], t2, g[SL::C]);
        }
    }
}

// Backward: accumulate this thread's partial sums for multiplet `sm` over the tile's units [u0, u1), reduce over the wave
// and leave the wave totals in s_red_row[slot].
template <int NC, bool ASYM>
__device__ __forceinline__ void tm_grad_mult(TmMultK sm, const double *__restrict__ gx, const double *s_w,
                                             int u0, int u1, int Nx, int tid, int lane, double *s_red_row)
{
    constexpr int V = TmGradSlots<NC>::count(ASYM);
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
    for (int u = u0; u < u1; u++) {
        // units that do not meet the window are skipped, units inside it need no test per bin (wave-uniform)
        const int lo = u << TM_UNIT_SHIFT, hi = lo + TM_UNIT_BINS;
        if (hi <= imin || lo >= imax) continue;
        const double *wq = s_w + (u - u0) * TM_UNIT_BINS + tid;
        tm_grad_unit<NC, ASYM>(nu2, hq, g2, aAh, aB, c2, imin, imax, gx, wq, lo + tid, Nx, !(lo >= imin && hi <= imax && hi <= Nx), g);
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
        s_red_row[tm_grad_store_slot<NC>(slot)] = g[0];
    }
}

#ifndef TM_LB_FWD
#define TM_LB_FWD 4    // resident waves per SIMD the register allocator must allow (likelihood-only kernel)
#endif
#ifndef TM_LB_GRAD
#define TM_LB_GRAD 4   // same for the gradient kernel: 128 VGPRs (the pair loop is written to fit: see tm_eval_body)
#endif
#ifdef TM_TRACE   // developer build: per-workgroup time stamps (tools/block_trace.py)
__device__ unsigned long long *g_tm_trace = nullptr;
#define TM_STAMP(slot) do { if (threadIdx.x == 0 && g_tm_trace) g_tm_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define TM_STAMP(slot) do { } while (0)
#endif

template <int N> struct TmInt { static constexpr int value = N; };
template <bool B> struct TmBool { static constexpr bool value = B; };

// The workgroup's work for (chain, tile): pass 1 (model, likelihood partials, weights), in-launch finalize or pass 2.
// s_w: dynamic LDS for the weights, [TM_TILE_MAXU * TM_UNIT_BINS] doubles (GRAD only).  Called by tamcmc_eval_kernel
// (tamcmc_eval.hip) and by the fused small-grid kernel (tamcmc_fused.hip).
//
// The tile's units [u0, u1) (TmTileHdr, chosen per chain by the setup kernel) are walked in GROUPS of two units (four
// bins per thread share one fetch of a multiplet's record and one window test; four independent rational chains are in
// flight) -- a single unit where the tile ends on an odd unit or a group would straddle two cells.  A group that lies
// inside the grid takes the FULL path: no index clamps, no per-bin validity tests, loads at constant offsets from one
// per-thread pointer; within it a multiplet whose window covers the whole group takes the no-test form of
// tm_accum_mult.  Only the grid's last, partial unit takes the guarded path.
// GEN = false: the common case compiled on its own -- chi(2,2p) likelihood, no Gaussian term (ids 0, 1), no model rows
// requested; the launcher picks it from the context (TmEvalArgs::generic).  Same arithmetic, fewer live scalars.
template <bool GRAD, bool GEN = true>
__device__ __forceinline__ void tm_eval_body(const TmEvalArgs &a, const int chain, const int tile, double *s_w)
{
    const int like = GEN ? a.likelihood_case : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    TmNoiseK sn = (TmNoiseK)(a.noise + chain);

#if defined(TM_ABLATE) && (TM_ABLATE & 16)   // timing-only build: empty workgroups (dispatch cost only)
    if (a.Nx > 0) return;
#endif
    __shared__ double s_red[2][TM_WAVES][TM_GSLOTS];   // double-buffered: one barrier per multiplet in pass 2

    // ---------------- prologue: nothing but scalar loads ----------------
    // Everything a tile needs -- its bounds, its list of active multiplets (table order), the cells' background
    // polynomials, the chain's noise record, the multiplet records themselves -- was prepared by the setup kernel and is
    // wave-uniform: it is fetched with s_load into SGPRs.  No LDS staging, no barrier before the first bin.
    // The chain's spectrum: contexts holding several spectra on one grid (ensembles, tamcmc_ctx_set_spectra) keep them
    // as consecutive blocks of Nx.
    const size_t spec_off = (a.spec != nullptr) ? (size_t)((TmIdxK)a.spec)[chain] * (size_t)a.Nx : 0;
    const double *__restrict__ yp = a.y + spec_off;
    const double *__restrict__ isp = a.isig2 + spec_off;      // dereferenced only when the likelihood is chi_square (then non-NULL)
    const TmMult *__restrict__ gm = a.mult + (size_t)chain * a.n_mult;
    TmTileHdrK th = (TmTileHdrK)(a.thdr + (size_t)chain * a.tiles + tile);
    const int u0 = th->u0;
    int u1 = th->u1;
    if (u1 - u0 > (GRAD ? TM_TILE_MAXU : TM_TILE_MAXU_L)) u1 = u0 + (GRAD ? TM_TILE_MAXU : TM_TILE_MAXU_L);      // the setup kernel never builds such a tile (and flags it if it did)
    const int nact = th->nact;
    TmCellRecK cells = (TmCellRecK)(a.cell + (size_t)chain * a.cells);
    TmActiveK tix = (TmActiveK)(a.tidx + ((size_t)chain * a.tiles + tile) * (a.n_mult > 0 ? a.n_mult : 1));
    const int nh = sn->nh;
    const bool has_gauss = GEN && sn->has_gauss != 0;
    const double N0 = sn->N0;
    const int row = (GEN && a.row_of_chain != nullptr) ? a.row_of_chain[chain] : -1;
    // FULL groups address the grid as (wave-uniform base pointer) + (32-bit per-thread bin index): global loads with a
    // scalar base and one VGPR of offset, instead of one 64-bit pointer per array and thread (Nx < 2^28 is checked at create)
    const double *__restrict__ xb = a.x2;
    const double *__restrict__ lb = a.lx;

#if defined(TM_ABLATE) && (TM_ABLATE & 32)   // timing-only build: prologue (scalar loads) only
    if (nact >= 0 && N0 == N0) { if (tid == 0 && !GRAD) a.part[((size_t)chain * a.tiles + tile) * 4] = (double)(nh + row + u0 + (has_gauss ? 1 : 0)); return; }
#endif
    // ---------------- pass 1: model spectrum and likelihood partial sums ----------------
    double S1 = 0.0;
    double P = 1.0;
    int esum = 0;
    int sgn = 0;                        // OR of the model values' high words: the sign bit tells whether any was negative
    double gn_[GRAD ? TM_GSLOTS : 1];   // GRAD: noise partial sums (3 per Harvey, sum w at slot 3*TM_MAXH, Gaussian at 13..15)
#pragma unroll
    for (int s_ = 0; s_ < (GRAD ? TM_GSLOTS : 1); s_++) gn_[s_] = 0.0;
    const double wscale = GRAD ? a.wt[2 * chain + 1] : 0.0;

    // one group of KG / 2 units starting at unit u, inside cell `tr`
    auto group = [&](auto kg_c, auto full_c, const int u, TmCellRecK tr) __attribute__((always_inline)) {
        constexpr int KG = decltype(kg_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        const int lo = u << TM_UNIT_SHIFT, hi = lo + KG * TM_THREADS;
        double x2[KG], acc[KG];
        int bi[KG];
#pragma unroll
        for (int k = 0; k < KG; k++) {
            const int i = lo + k * TM_THREADS + tid;
            if (FULL) {
                bi[k] = i;
                x2[k] = xb[(unsigned)i];
            } else {
                const bool valid = i < a.Nx;
                bi[k] = valid ? i : -1;
                x2[k] = a.x2[valid ? i : a.Nx - 1];
            }
            acc[k] = 0.0;
        }
        // likelihood-only kernel (registers to spare): log x and y are requested now, so that their latency passes
        // behind the multiplet loop instead of in front of the background / likelihood arithmetic
        constexpr bool EARLY = FULL && !GRAD;
        double lxe[EARLY ? KG : 1], ye[EARLY ? KG : 1];
        if constexpr (EARLY) {
#pragma unroll
            for (int k = 0; k < KG; k++) { lxe[k] = (nh > 0) ? lb[(unsigned)bi[k]] : 0.0; ye[k] = yp[(unsigned)bi[k]]; }
        }
        if constexpr (GRAD) {
            for (int jj = 0; jj < nact; jj++) {
                const int idx = tix[jj].idx, imin = tix[jj].imin, imax = tix[jj].imax, shape = tix[jj].shape;
                if (hi <= imin || lo >= imax) continue;   // wave-uniform skip
                tm_accum_dispatch<KG>((TmMultK)(gm + idx), shape, x2, bi, acc, !(FULL && lo >= imin && hi <= imax));
            }
        } else if (nact > 0) {
            // Likelihood kernel (SGPRs to spare): the list entry of the NEXT multiplet is requested before the current one
            // is evaluated (one 16-byte scalar load, in flight beside the record's), so a wave waits once per multiplet
            // instead of three times in a row (window; index and shape; record) -- waits that a wave with few
            // neighbours on its SIMD cannot hide (the tail of a launch).
            TmEntry e = tm_load_entry(tix);
            asm volatile("" :: "s"(e.x));             // a use: the first entry is waited for here, not inside the loop
            for (int jj = 0; jj < nact; jj++) {
                const TmEntry en = tm_load_entry(tix + (jj + 1 < nact ? jj + 1 : jj));
                const int idx = e.x, imin = e.y, imax = e.z, shape = e.w;
                if (!(hi <= imin || lo >= imax))   // wave-uniform skip
                    tm_accum_dispatch<KG>((TmMultK)(gm + idx), shape, x2, bi, acc, !(FULL && lo >= imin && hi <= imax));
                e = en;
            }
        }

        // background, likelihood and (GRAD) weights + noise partial sums, KH bins at a time: the gradient kernel keeps
        // three accumulators per component in pass 2 and sixteen noise sums here, so it takes the four bins of a group
        // in two halves (fewer values live at once: 4 waves per SIMD instead of 3)
        constexpr int KH = GRAD ? 2 : KG;
        const bool npoly = tr->npoly != 0;
#pragma unroll
        for (int h0 = 0; h0 < KG; h0 += KH) {
        double hu[TM_MAXH][GRAD ? KH : 1], harg[GRAD ? KH : 1];   // GRAD: u = 1/(1+t) per Harvey and bin (t u = 1 - u), log x per bin
        bool n0_done = false;
        if (nh > 0) {
            double dl[KH];
#pragma unroll
            for (int k = 0; k < KH; k++) {
#if defined(TM_ABLATE) && (TM_ABLATE & 128)   // timing-only build: no log x load (wrong values)
                dl[k] = tr->lxc + 1e-9 * x2[h0 + k];
#else
                dl[k] = EARLY ? lxe[h0 + k] : FULL ? lb[(unsigned)bi[h0 + k]] : a.lx[bi[h0 + k] >= 0 ? bi[h0 + k] : a.Nx - 1];
#endif
                if constexpr (GRAD) harg[k] = dl[k];
            }
            if (npoly) {
                const double lxc = tr->lxc;
                // whole background (all profiles + white noise) as ONE polynomial: 8 FMAs per bin.  The gradient path
                // accumulates the moments sum w dl^j instead of per-profile sums (below); the backward kernel turns
                // them into the per-profile sums with each profile's own series coefficients.
#pragma unroll
                for (int k = 0; k < KH; k++) { dl[k] -= lxc; acc[h0 + k] += tm_poly(tr, dl[k]); if constexpr (GRAD) harg[k] = dl[k]; }
                n0_done = true;
            } else {
#pragma unroll
                for (int h = 0; h < TM_MAXH; h++) {
                    if (h < nh) {
                        const double Hh = sn->H[h], ph = sn->p[h], lth = sn->lt[h];
#pragma unroll
                        for (int k = 0; k < KH; k++) {
                            const double t = exp(ph * (lth + dl[k]));
                            const double uu = 1.0 / (t + 1.0);
                            acc[h0 + k] += Hh * uu;
                            if constexpr (GRAD) hu[h][k] = uu;
                        }
                    }
                }
            }
        }
        if (has_gauss) {
            const double gA = sn->gA, gnu0 = sn->gnu0, gs2 = sn->gs2;
#pragma unroll
            for (int k = 0; k < KH; k++) {
                const double dd = 0.5 * x2[h0 + k] - gnu0;
                acc[h0 + k] = gA * exp((-0.5 * (dd * dd)) / gs2) + acc[h0 + k];
            }
        }
        if (!n0_done) {
#pragma unroll
            for (int k = 0; k < KH; k++) acc[h0 + k] += N0;
        }
        if (row >= 0) {
#pragma unroll
            for (int k = 0; k < KH; k++)
                if (FULL || bi[h0 + k] >= 0) a.model_out[(size_t)row * a.Nx + (size_t)(lo + (h0 + k) * TM_THREADS + tid)] = acc[h0 + k];
        }

        double wreg[GRAD ? KH : 1];
        if (like == 0) {
            // -p * (sum y/M + sum log M), likelihoods.cpp:23-25
#pragma unroll
            for (int k = 0; k < KH; k++) {
                double wv = 0.0;
                if (FULL || bi[h0 + k] >= 0) {
                    // A model value that is NaN, +-inf or 0 turns rM (and so S1) into NaN by itself (v_rcp_f64 + the
                    // Newton steps); a negative one is caught by the sign bits collected in sgn.  Either way logL
                    // becomes NaN, which is what log() of such a value gives the reference.
                    const double M = acc[h0 + k];
                    const double yv = EARLY ? ye[h0 + k] : FULL ? yp[(unsigned)bi[h0 + k]] : yp[bi[h0 + k]];
                    sgn |= __double2hiint(M);
                    const double rM = tm_rcp(M);
                    S1 = __builtin_fma(yv, rM, S1);
                    int e;
                    P *= frexp(M, &e);
                    esum += e;
                    wv = wscale * (yv * rM * rM - rM);   // d(logL/T)/dM_i
                }
                if constexpr (GRAD) wreg[k] = wv;
            }
            {
                int e;
                P = frexp(P, &e);   // keep the running mantissa product in [0.5, 1)
                esum += e;
            }
        } else {
            // -sum (y-M)^2 / sigma^2, likelihoods.cpp:36
#pragma unroll
            for (int k = 0; k < KH; k++) {
                double wv = 0.0;
                if (FULL || bi[h0 + k] >= 0) {
                    const int i = lo + (h0 + k) * TM_THREADS + tid;
                    const double dd = yp[i] - acc[h0 + k];
                    const double is2 = isp[i];
                    S1 = __builtin_fma(dd * dd, is2, S1);
                    wv = wscale * dd * is2;
                }
                if constexpr (GRAD) wreg[k] = wv;
            }
        }
        if constexpr (GRAD) {
            // weights for pass 2 (each thread re-reads only its own entries) and the noise partial sums, reusing u and
            // t*u of these bins (no second pass over the bins)
#pragma unroll
            for (int k = 0; k < KH; k++) {
                const double wk = wreg[k];
                s_w[((u - u0) * 2 + h0 + k) * TM_THREADS + tid] = wk;
                gn_[3 * TM_MAXH] += wk;
#if defined(TM_ABLATE) && (TM_ABLATE & 512)   // timing-only build: no weight moments
                if (false) {
#else
                if (nh > 0 && npoly) {
#endif
                    // moments of the weights about the cell centre: slot j-1 <- sum w dl^j, j = 1..TM_HSER-1
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
                    const double dd = 0.5 * x2[h0 + k] - sn->gnu0;
                    const double we = wk * exp((-0.5 * (dd * dd)) / sn->gs2);
                    gn_[13] += we;
                    gn_[14] = __builtin_fma(we, dd, gn_[14]);
                    gn_[15] = __builtin_fma(we * dd, dd, gn_[15]);
                }
            }
        }
        if constexpr (GRAD) __builtin_amdgcn_sched_barrier(0);   // keep the halves apart: interleaved they need 40 more registers
        }   // halves of the group
    };

    // GRAD: the noise partial sums of the cell just finished -> gnoise[chain][tile][part]
    auto flush_noise = [&](const int part) __attribute__((always_inline)) {
        if constexpr (GRAD) {
            static_assert(TM_NSLOTS == 16, "butterfly below assumes 16 noise slots");
            TmBfly<TM_NSLOTS, 32>::run(gn_, lane);
            bool valid = true;
            const int slot = TmBfly<TM_NSLOTS, 32>::slot_of(lane, valid);
            double *red = s_red[1][wave];
            if ((lane & 3) == 0) red[slot] = gn_[0];
            __syncthreads();
            if (tid < TM_NSLOTS) {
                double t = 0.0;
#pragma unroll
                for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[1][wv][tid];
                a.gnoise[(((size_t)chain * a.tiles + tile) * 2 + part) * TM_NSLOTS + tid] = t;
            }
            __syncthreads();
#pragma unroll
            for (int s_ = 0; s_ < TM_GSLOTS; s_++) gn_[s_] = 0.0;
        }
    };

    {
        // The walk over the tile's units.  The hot loop is the run of PAIRS (two units: four bins per thread), aligned
        // to even units so that a pair never straddles two cells; it holds the only copy of the pair code, and nothing
        // else updates the accumulators inside it (a second variant behind a branch in the same loop costs ~40
        // registers in copies where the two meet).  An odd first unit is taken alone before the loop; what is left after
        // it -- an odd last unit, and at the end of the grid the partial unit -- alone after it.
        const int cell0 = u0 >> TM_CELL_SHIFT;
        int cur_cell = cell0;
        auto cell_of = [&](const int uu) __attribute__((always_inline)) {
            const int ce = uu >> TM_CELL_SHIFT;
            if (GRAD && ce != cur_cell) { flush_noise(0); cur_cell = ce; }   // at most once per tile (TM_TILE_MAXU <= TM_CELL_UNITS)
            return cells + ce;
        };
        int u = u0;
#if defined(TM_ABLATE) && (TM_ABLATE & 256)   // timing-only build: no pass 1
        u = u1;
#endif
        if ((u & 1) && u < u1 && ((u + 1) << TM_UNIT_SHIFT) <= a.Nx) { TmCellRecK tr = cell_of(u); group(TmInt<2>(), TmBool<true>(), u, tr); u += 1; }
#pragma unroll 1
        for (; u + 2 <= u1 && ((u + 2) << TM_UNIT_SHIFT) <= a.Nx; u += 2) { TmCellRecK tr = cell_of(u); group(TmInt<4>(), TmBool<true>(), u, tr); }
#pragma unroll 1
        for (; u < u1; u += 1) {
            TmCellRecK tr = cell_of(u);
            if (((u + 1) << TM_UNIT_SHIFT) <= a.Nx) group(TmInt<2>(), TmBool<true>(), u, tr);
            else                                    group(TmInt<2>(), TmBool<false>(), u, tr);
        }
        if (GRAD) flush_noise(cur_cell - cell0);
    }
    TM_STAMP(1);
    // tile partials: S1 = sum y/M (or the chi-square sum), and sum log M as (mantissa product, exponent sum)
    double Pm = 1.0, Pe = 0.0;
    if (like == 0) {
        int e;
        Pm = frexp(tm_wave_prod(P), &e);
        Pe = tm_wave_sum((double)esum) + (double)e;
        if (__builtin_amdgcn_ballot_w64(sgn < 0) != 0) Pm = __builtin_nan("");     // a negative model value anywhere: log -> NaN
    }
    S1 = tm_wave_sum(S1);
    if (lane == 0) { s_red[0][wave][0] = S1; s_red[0][wave][1] = Pm; s_red[0][wave][2] = Pe; }
    __syncthreads();
    if constexpr (GRAD) {
        if (tid == 0) {
            double t1 = 0.0, tm = 1.0, te = 0.0;
#pragma unroll
            for (int wv = 0; wv < TM_WAVES; wv++) { t1 += s_red[0][wv][0]; tm *= s_red[0][wv][1]; te += s_red[0][wv][2]; }
            double *out = a.part + ((size_t)chain * a.tiles + tile) * 4;   // summed by the backward kernel
            out[0] = t1;
            out[1] = tm;
            out[2] = te;
        }
    } else if (wave == 0) {
        // Finalize inside this launch: the workgroup that publishes the LAST tile partial of a chain sums all of them
        // in a fixed order (lane-strided, then a wave tree), so the result does not depend on which workgroup that is.
        // Hand-off without cache-wide fences: the partials are written and read with agent-scope relaxed atomics, which
        // on gfx942 / gfx950 compile to global_store / global_load with sc1 (write-through to, and read from, the
        // device-coherent level past the per-XCD L2s); the publisher drains its stores (s_waitcnt vmcnt(0)) before it
        // takes its ticket (an agent-scope atomic add, executed at that same level), and the last arriver reads only
        // after its ticket returned.  This is NOT the HIP memory model's release / acquire (no fence instructions: those
        // cost 18 us per launch here); it relies on these gfx9 properties, and tests/test_parity_gpu.py
        // (test_launch_order_does_not_change_results, run-to-run bitwise equality) is its guard.
        int last = 0;
        double *pp = a.part + (size_t)chain * a.tiles * 4;
        if (lane == 0) {
            double t1 = 0.0, tm = 1.0, te = 0.0;
#pragma unroll
            for (int wv = 0; wv < TM_WAVES; wv++) { t1 += s_red[0][wv][0]; tm *= s_red[0][wv][1]; te += s_red[0][wv][2]; }
#if defined(TM_ABLATE) && (TM_ABLATE & 4)    // timing-only build: plain stores, no ticket, no finalize
            pp[4 * tile] = t1; pp[4 * tile + 1] = tm; pp[4 * tile + 2] = te;
#else
            if (a.tiles == 1) {
                // the chain's only tile: nothing to hand over.  Same value as the general path below, without its
                // store-drain-ticket-reload round trip
                pp[0] = t1; pp[1] = tm; pp[2] = te;
                const double t2 = (like == 0) ? tm_tile_logsum(tm, te) : 0.0;
                double f = (like == 0) ? -a.like_p * (t1 + t2) : -t1;
                f = f / a.wt[2 * chain];
                int st = a.noise[chain].status;
                if (st != 0) f = __builtin_nan("");
                else if (!(f == f)) st = 1;
                a.logL[chain] = f;
                if (a.status) a.status[chain] = st;
            } else {
            __hip_atomic_store(pp + 4 * tile, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 4 * tile + 1, tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 4 * tile + 2, te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                s1 += __hip_atomic_load(pp + 4 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (like == 0)
                    s2 += tm_tile_logsum(__hip_atomic_load(pp + 4 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                         __hip_atomic_load(pp + 4 * t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            s1 = tm_wave_sum(s1);
            s2 = tm_wave_sum(s2);
            if (lane == 0) {
                double f = (like == 0) ? -a.like_p * (s1 + s2) : -s1;
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
        __syncthreads();   // s_red[0] (likelihood) consumed; s_w complete; s_red[1] (noise) consumed before multiplet 0 reuses it
#if defined(TM_ABLATE) && (TM_ABLATE & 2)   // timing-only build: no multiplet pass 2 at all
        for (int jj = 0; jj < 0; jj++) {
#else
        for (int jj = 0; jj < nact; jj++) {
#endif
            const int idx = tix[jj].idx, shape = tix[jj].shape;
            TmMultK sm = (TmMultK)(gm + idx);
            const int nc = shape & 255;
            double *red = s_red[(jj + 1) & 1][wave];
            if (lane == 0) {      // slots this multiplet does not write (A_am beyond its L; the asymmetry sums)
                const int Lm = (nc - 1) >> 1;
                for (int am = Lm + 1; am <= 3; am++) red[7 + am] = 0.0;
                if (shape < 256) { red[21] = 0.0; red[22] = 0.0; red[23] = 0.0; }
            }
            switch (shape) {
            case 1: tm_grad_mult<1, false>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 3: tm_grad_mult<3, false>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 5: tm_grad_mult<5, false>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 7: tm_grad_mult<7, false>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 256 + 1: tm_grad_mult<1, true>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 256 + 3: tm_grad_mult<3, true>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            case 256 + 5: tm_grad_mult<5, true>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            default: tm_grad_mult<7, true>(sm, a.x2, s_w, u0, u1, a.Nx, tid, lane, red); break;
            }
#if defined(TM_ABLATE) && (TM_ABLATE & 64)   // timing-only build: no workgroup barrier per multiplet (wave 0 publishes its own partial)
            if (tid < TM_GSLOTS) a.gmult[(((size_t)chain * a.tiles + tile) * a.n_mult + idx) * TM_GSLOTS + tid] = red[tid];
#else
            __syncthreads();
            if (tid < TM_GSLOTS) {
                double t = 0.0;
                if (tid < nc || (tid >= 7 && tid < 12) || tid >= 21) {      // B_k, A_am, C, asymmetry sums (tm_grad_store_slot)
#pragma unroll
                    for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[(jj + 1) & 1][wv][tid];
                }
                a.gmult[(((size_t)chain * a.tiles + tile) * a.n_mult + idx) * TM_GSLOTS + tid] = t;
            }
#endif
        }
        TM_STAMP(2);   // gradient kernel: slot 2 = end of pass 2
    }
}

#endif
