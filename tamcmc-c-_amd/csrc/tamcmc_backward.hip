// tamcmc_backward.hip -- chain rule from the eval kernel's per-multiplet partial sums to
// d(logL/T)/d vars.  NOT in the reference (its sampler never had a gradient: MALA.cpp:18,317-333;
// SURVEY.md F2 / App. D): this is new functionality, validated entry by entry against an analytic gradient
// written independently into the CPU oracle (itself pinned against long-double finite differences of its
// log-likelihood; no reference output exists: "parity unpinned").  The truncation window [imin,imax) moves
// with the parameters; like any analytic gradient of a truncated model this ignores the motion.
//
// One workgroup (512 threads) per chain.  The chain record (TmChain) and the per-multiplet records
// (TmMultFull) written by the setup kernel are read back instead of being re-derived.  The kernel is a chain of
// single-wave stages separated by trips to memory (~1 us each: its inputs were written by other XCDs), so its
// shape follows the time line measured with cycle stamps inside it (TM_BW_TRACE, tools/bw_trace.py):
// Staging: six waves issue every load of the LDS copies (params row, records, tile starts, chain record, inverse
//   of index_to_relax) at once, clear the pair tables while those are in flight, then store.
// Phase 1a: every (multiplet, slot) pair sums its partials over the tiles the window touches IN TILE ORDER.
// Phase 1b: lane j runs multiplet j's chain rule (tm_bw_mult) and emits (param index, value) pairs plus
//   chain-level adjoints into LDS; beside it wave 1 sums the noise partials and wave 3 the likelihood partials
//   (the finalize step: logL, status) -- each one trip to memory of its own.
// Phase 2: the chain-level adjoints are summed over multiplets (four lanes per slot, multiplet order) and two
//   lanes on two waves turn them (splitting, inclination, visibilities, asymmetry, numax | noise) into more pairs.
// Phase 3: every variable gathers, in pair order, what is addressed to it: per 64-pair chunk one ballot per
//   variable tells lane k which pairs are its own; the waves' partial rows are added in wave order.
// No atomics: bitwise reproducible.  Compiled with -ffp-contract=off like the setup TU.
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#include "tamcmc_derive.h"

#ifdef TM_BW_TRACE   // timing-only build: cycle stamps of chain 0's phases leave through its gradient row (tools/bw_trace.py)
#define BW_TS(i, who) do { if (tid == (who)) s_ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BW_TS(i, who) do { } while (0)
#endif
#ifdef TM_BW_TRACE
__shared__ unsigned long long g_bfine[8];
#define BFINE(i) do { if (threadIdx.x == 0) g_bfine[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BFINE(i) do { } while (0)
#endif
#define NWV_ROWS (TM_BW_THREADS / 64)
#define TM_GSTRIDE (TM_GSLOTS + 1)   // row stride of the tile-summed partials in LDS: 25 doubles, so that the lanes of phase 1b
                                     // (one multiplet each, reading the same slot of their rows) fall on different banks
#define TM_NPAIR 12   // pairs a multiplet can emit
#define TM_NSHARED 20 // chain-level adjoint slots per multiplet
#define TM_NCPAIR 64  // base number of chain-level pairs (plus numax pairs for id 9)

// chain-level slots
#define SL_A1 0
#define SL_FS1 1
#define SL_FS2 2
#define SL_ETA 3
#define SL_A3 4
#define SL_ASYM 5
#define SL_V 6      // +l-1, l=1..3
#define SL_RATIO 9  // l=1: 9,10  l=2: 11,12,13  l=3: 14..17
#define SL_NUMAX 18

__device__ __forceinline__ int tm_ratio_slot(int l, int am) { return SL_RATIO + (l == 1 ? 0 : l == 2 ? 2 : 5) + am; }
__device__ __forceinline__ double tm_sign(double v) { return (v < 0.0) ? -1.0 : 1.0; }

// Chain rule of one multiplet: from its tile-summed partials G to (parameter, value) pairs and to its share of the
// chain-level adjoints.  One lane runs this alone, so what it costs is LDS round trips: every operand is declared free
// of aliases (the tables are disjoint parts of the workgroup's LDS), which lets the loads go out ahead of the stores
// instead of one by one behind them, and the heights' adjoints stay in registers.
__device__ __forceinline__ void tm_bw_mult(const TmLayout &L, const TmChain *__restrict__ Cp, const TmMultFull *__restrict__ Mp,
                                           const double *__restrict__ G, const double *__restrict__ p, double *__restrict__ sh,
                                           int *__restrict__ pi, double *__restrict__ pv)
{
    const double PI = 3.141592653589793238462643383279502884;
    const TmChain &C = *Cp;
    const TmMultFull &M = *Mp;
    int np = 0;
    BFINE(0);
    if (M.status != 0) return;
    const int l = M.l;
    const double W = M.W, g2 = W * W, f = M.f;
    double adj_f = 0.0, adj_W = 0.0, adj_fs = 0.0;
    // Tile-summed partials of this multiplet (tamcmc_eval_body.h, tm_grad_unit): G[k] = sum wA d_k r_k^2 per component,
    // G[7 + am] = sum wA (r_{l-am} + r_{l+am}) per |m| (the components +-m share their height), G[11] = sum wA
    // sum_k hq_k r_k^2.
    double s_eta = 0.0, s_a3 = 0.0;
    const int ncomp = M.ncomp;
    // Every operand is fetched unconditionally (all indices are inside the records whatever l is) and the per-component
    // tests become selects: behind a branch per component the compiler issues that component's LDS loads only once the
    // branch is taken, one round trip after the other (2 100 cycles for this block; the sums are the same).
    double Gk[TM_MAXM], hk[TM_MAXM], Qk[TM_MAXM], ck[TM_MAXM], GA[4], hA[4];
#pragma unroll
    for (int k = 0; k < TM_MAXM; k++) { Gk[k] = G[k]; hk[k] = M.h[k]; Qk[k] = M.Q[k]; ck[k] = M.c[k]; }
#pragma unroll
    for (int am = 0; am <= 3; am++) { GA[am] = G[7 + am]; hA[am] = M.h[(l + am < TM_MAXM) ? l + am : TM_MAXM - 1]; }
    const double eta = C.eta;
    double adj_g2 = -G[11];
    double ah[4];                       // d/d(h) of the pair of components +-am (kept at the +m component), am = 0..l
#pragma unroll
    for (int am = 0; am <= 3; am++) {
        const bool on = am <= l;
        const double A = GA[am];                        // d/d(hq) summed over the components l-am and l+am
        adj_g2 = on ? adj_g2 + A * hA[am] : adj_g2;
        ah[am] = on ? A * g2 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < TM_MAXM; k++) {
        const bool on = k < ncomp;
        const double hq = hk[k] * g2;
        const double adj_nu = 4.0 * hq * Gk[k];
        const int m = k - l;
        if (l != 0) {
            adj_f = on ? adj_f + adj_nu * (1. + eta * Qk[k]) : adj_f;
            s_eta = on ? s_eta + adj_nu * f * Qk[k] : s_eta;
            adj_fs = on ? adj_fs + adj_nu * (double)m : adj_fs;
            s_a3 = on ? s_a3 + adj_nu * ck[k] : s_a3;
        } else {
            adj_f = on ? adj_f + adj_nu : adj_f;
        }
    }
    BFINE(1);
    if (l != 0) { sh[SL_ETA] = s_eta; sh[SL_A3] = s_a3; }      // the slots were zero
    adj_W += 2.0 * W * adj_g2;
    if (C.asym != 0 || L.asym_var != 0) {   // same predicate as the setup kernel's asymmetric path (the sums exist)
        const double al = C.asym;
        const double cc = 0.5 * W * al / f, c2 = cc * cc;
        const double C0 = G[21], C1 = G[22], C2 = G[23];
        // dA/d(asym) = 2 a (x/f - 1) + asym W^2 / (2 f^2): regular at asym = 0, where A = 1 but the derivative is not 0
        sh[SL_ASYM] += 2.0 * (C2 / f - C1) + (0.5 * W * W * al / (f * f)) * C0;
        adj_W += (2.0 * c2 / W) * C0;
        adj_f += -2.0 * al / (f * f) * C2 - (2.0 * c2 / f) * C0;
    }
    // splitting
    if (L.variant == 1) {
        double a1s = 0.0, a2s = 0.0;
        if (l == 1) a1s = adj_fs;
        if (l == 2) a2s = adj_fs;
        if (l == 3) { a1s = 0.5 * adj_fs; a2s = 0.5 * adj_fs; }
        if (L.model_case == 6) { sh[SL_FS1] += a1s; sh[SL_FS2] += a2s; }
        if (L.model_case == 7) { pi[np] = L.s + 6 + M.n; pv[np] = tm_sign(p[L.s + 6 + M.n]) * (a1s + a2s); np++; }
        if (L.model_case == 8) {
            pi[np] = L.s + 6 + M.n; pv[np] = tm_sign(p[L.s + 6 + M.n]) * a1s; np++;
            pi[np] = L.s + 6 + L.Nmax + M.n; pv[np] = tm_sign(p[L.s + 6 + L.Nmax + M.n]) * a2s; np++;
        }
    } else {
        sh[SL_A1] += adj_fs;
    }
    BFINE(2);
    // heights
    const double piW = PI * W;
    if (L.variant != 2) {
        // components below l carry no adjoint of their own: the sum over the components reduces to the +m ones
        double adj_H = 0.0, rt[4];
#pragma unroll
        for (int am = 0; am <= 3; am++) rt[am] = C.ratios[l][(l + am < TM_MAXM) ? l + am : TM_MAXM - 1];
#pragma unroll
        for (int am = 0; am <= 3; am++) adj_H = (am <= l) ? adj_H + ah[am] * rt[am] : adj_H;
        // ratio slot |m| collects the pair's adjoint; the slots were zero
#pragma unroll
        for (int am = 0; am <= 3; am++)
            if (am <= l && l > 0) sh[tm_ratio_slot(l, am)] = ah[am] * M.H;
        const double pn = p[M.idx_h];
        const bool plain = (l == 0 || L.family == TM_FAM_LOCAL);
        const double Vl = plain ? 1.0 : C.Vl[l];
        const double scale = C.do_amp ? 1.0 / piW : 1.0;
        pi[np] = M.idx_h; pv[np] = tm_sign(pn) * scale * Vl * adj_H; np++;
        if (!plain) sh[SL_V + l - 1] += fabs(pn) * scale * adj_H;
        if (C.do_amp) adj_W += -M.H / W * adj_H;
    } else {
        const double scale = C.do_amp ? 1.0 / piW : 1.0;
#pragma unroll
        for (int am = 0; am <= 3; am++) {
            if (am <= l) {
                const double a = ah[am];
                pi[np] = M.idx_h + am; pv[np] = tm_sign(p[M.idx_h + am]) * scale * a; np++;
                if (C.do_amp) adj_W += -M.h[l + am] / W * a;
            }
        }
    }
    BFINE(3);
    // width
    if (M.width_kind == 0) {
        pi[np] = M.idx_w0; pv[np] = tm_sign(M.Wraw) * adj_W; np++;
    } else if (M.width_kind == 1) {
        const double adj_v = tm_sign(M.Wraw) * adj_W;
        const double F0 = p[M.idx_F0], F1 = p[M.idx_F1];
        const double t = (f - F0) / (F1 - F0), a = M.slope;
        adj_f += a * adj_v;
        pi[np] = M.idx_w0; pv[np] = (1.0 - t) * adj_v; np++;
        pi[np] = M.idx_w1; pv[np] = t * adj_v; np++;
        pi[np] = M.idx_F0; pv[np] = a * (t - 1.0) * adj_v; np++;
        pi[np] = M.idx_F1; pv[np] = -a * t * adj_v; np++;
    } else {
        const int w = L.w;
        const double adj_ln = W * adj_W;
        if (M.width_kind == 2) {
            const double numax = C.numax;
            const double N = log(f / p[w + 0]), D = log(p[w + 3] / numax), A = log(p[w + 4]);
            const double e = 2. * N / D, q1 = 1. + e * e;
            const double dLde = 2.0 * A * e / (q1 * q1);
            pi[np] = w + 1; pv[np] = adj_ln * log(f / numax); np++;
            pi[np] = w + 2; pv[np] = adj_ln / p[w + 2]; np++;
            pi[np] = w + 4; pv[np] = -adj_ln / (p[w + 4] * q1); np++;
            pi[np] = w + 0; pv[np] = adj_ln * dLde * (-2.0 / (p[w + 0] * D)); np++;
            pi[np] = w + 3; pv[np] = adj_ln * dLde * (-2.0 * N / (D * D * p[w + 3])); np++;
            adj_f += adj_ln * (p[w + 1] / f + dLde * 2.0 / (f * D));
            sh[SL_NUMAX] += adj_ln * (-p[w + 1] / numax + dLde * 2.0 * N / (D * D * numax));
        } else {
            const double N = log(f / p[w + 1]), D = log(p[w + 4] / p[w + 0]), A = log(p[w + 5]);
            const double e = 2. * N / D, q1 = 1. + e * e;
            const double dLde = 2.0 * A * e / (q1 * q1);
            pi[np] = w + 2; pv[np] = adj_ln * log(f / p[w + 0]); np++;
            pi[np] = w + 3; pv[np] = adj_ln / p[w + 3]; np++;
            pi[np] = w + 5; pv[np] = -adj_ln / (p[w + 5] * q1); np++;
            pi[np] = w + 1; pv[np] = adj_ln * dLde * (-2.0 / (p[w + 1] * D)); np++;
            pi[np] = w + 4; pv[np] = adj_ln * dLde * (-2.0 * N / (D * D * p[w + 4])); np++;
            pi[np] = w + 0; pv[np] = adj_ln * (-p[w + 2] / p[w + 0] + dLde * 2.0 * N / (D * D * p[w + 0])); np++;
            adj_f += adj_ln * (p[w + 2] / f + dLde * 2.0 / (f * D));
        }
    }
    pi[np] = M.idx_f; pv[np] = adj_f; np++;
    BFINE(4);
}

#ifndef TM_BW_THREADS
#define TM_BW_THREADS 512
#endif
__global__ __launch_bounds__(TM_BW_THREADS) void tamcmc_backward_kernel(TmLayout L, int tiles, int cells, int uniform_su, TmCostModel geom,
                                                             const double *__restrict__ params,
                                                             const double *__restrict__ Tcoefs,
                                                             const TmChain *__restrict__ chain_rec,
                                                             const TmMultFull *__restrict__ aux,
                                                             const TmNoise *__restrict__ noise,
                                                             const double *__restrict__ part,
                                                             const double *__restrict__ gmult,
                                                             const double *__restrict__ gnoise,
                                                             const TmCellRec *__restrict__ cell, const TmTileHdr *__restrict__ thdr,
                                                             const double *__restrict__ hser,
                                                             int Nvars, const int32_t *__restrict__ relax,
                                                             double *__restrict__ grad,
                                                             double *__restrict__ logL, int32_t *__restrict__ status,
                                                             int aux_in_lds)
{
    const double PI = 3.141592653589793238462643383279502884;
    const int chain = blockIdx.x, tid = threadIdx.x;
#ifdef TM_BW_TRACE
    __shared__ unsigned long long s_ts[20];
    if (tid < 20) s_ts[tid] = 0;
    __syncthreads();
#endif
    BW_TS(0, 0);
#ifdef TM_BW_TRACE
    asm volatile("" :: "s"(tiles), "s"(Nvars));      // the first kernel arguments have arrived
    BW_TS(16, 0);
    { const double probe = params[(size_t)chain * L.Nparams]; asm volatile("" :: "v"(probe)); }   // first trip to memory
    BW_TS(17, 0);
    { const double probe = params[(size_t)chain * L.Nparams + 1]; asm volatile("" :: "v"(probe)); }   // same line again
    BW_TS(18, 0);
    { const double probe = gmult[(size_t)chain * tiles * L.n_mult * TM_GSLOTS]; asm volatile("" :: "v"(probe)); }   // another buffer
    BW_TS(19, 0);
#endif
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    __shared__ TmChain C;
    __shared__ double s_gn[TM_NSLOTS];
    __shared__ double s_gt[TM_NSLOTS][64 + 1];   // per-lane noise partials of wave 1 (padded: conflict-free column reads)
    __shared__ double s_S[TM_NSHARED];
    const int nm = L.n_mult;
    // dynamic LDS carve-up: the chain's params row first (every later access is an LDS read)
    double *p = s_dyn;                                         // [Nparams]
    double *s_row = s_dyn + L.Nparams;                         // [Nvars] this chain's gradient row, written out coalesced
    double *pair_val = s_row + Nvars;                          // [nm*TM_NPAIR + ncp]
    const int ncp = TM_NCPAIR + (L.model_case == 9 ? (L.Nmax * (L.lmax + 2) + L.lmax) : 0);
    const int npairs_max = nm * TM_NPAIR + ncp;
    double *shared_adj = pair_val + npairs_max;                // [nm*TM_NSHARED]
    double *s_G = shared_adj + (size_t)nm * TM_NSHARED;        // [nm][TM_GSTRIDE] tile-summed partials
    const int npairs_pad = (npairs_max + 7) & ~7;              // the gather reads the pairs eight at a time
    int *pair_idx = reinterpret_cast<int *>(s_G + (size_t)nm * TM_GSTRIDE);  // [npairs_pad], moved up to a 16-byte boundary:
    pair_idx += (4 - (((unsigned)(uintptr_t)pair_idx >> 2) & 3)) & 3;
    // per-multiplet records: staged in LDS when they fit (coalesced copy), else read in place from global
    constexpr int AUXD = (int)(sizeof(TmMultFull) / sizeof(double));
    constexpr int CD = (int)(sizeof(TmChain) / sizeof(double));
    int *s_u = pair_idx + npairs_pad;                          // [tiles + 1] first unit of every tile, then the end of the last
    int *s_inv = s_u + ((tiles + 2) & ~1);                     // [Nparams] variable that a parameter is, or -1
    double *s_part = reinterpret_cast<double *>(s_inv + ((L.Nparams + 1) & ~1));   // [waves][Nvars] phase 3: per-wave partial rows
    double *s_aux = s_part + (size_t)(TM_BW_THREADS / 64) * Nvars;
    const TmMultFull *auxp = aux + (size_t)chain * nm;
    const double *aux_src = reinterpret_cast<const double *>(auxp);
    if (aux_in_lds) auxp = reinterpret_cast<const TmMultFull *>(s_aux);

    // Staging.  A trip to memory costs ~1 us here (what this kernel reads was written by other XCDs), and a wave's loads
    // complete in order: a wave that copied the params row, then the records, then the headers paid three trips before
    // its real work.  So waves 1 and 3 go straight to the noise partials and the likelihood (below), and the other six
    // first ISSUE every load of the staging copies, clear the LDS tables while those are in flight, then store.
    const int wv = tid >> 6;
    int rv = -1, rk = 0;                                       // staging thread rk < Nvars: the parameter that variable rk is
    if (wv != 1 && wv != 3) {
        constexpr int NS = TM_BW_THREADS - 128;                // staging threads
        const int sid = tid - 64 * ((wv > 1 ? 1 : 0) + (wv > 3 ? 1 : 0));
        constexpr int KP = 2, KA = 4;
        double rp[KP], ra[KA], rc = 0.0;
        TmTileHdr rh;
        rh.u0 = 0; rh.u1 = 0; rh.nact = 0; rh.cost = 0;
#pragma unroll
        for (int k = 0; k < KP; k++) { const int e = sid + k * NS; rp[k] = (e < L.Nparams) ? params[(size_t)chain * L.Nparams + e] : 0.0; }
#pragma unroll
        for (int k = 0; k < KA; k++) { const int e = sid + k * NS; ra[k] = (aux_in_lds && e < nm * AUXD) ? aux_src[e] : 0.0; }
        if (sid < tiles) rh = thdr[(size_t)chain * tiles + sid];
        if (L.family != TM_FAM_GAUSS && sid < CD) rc = reinterpret_cast<const double *>(chain_rec + chain)[sid];
        rk = sid;
        rv = (sid < Nvars) ? relax[sid] : -1;
        for (int e = sid; e < L.Nparams; e += NS) s_inv[e] = -1;
        for (int e = sid; e < npairs_pad; e += NS) { pair_idx[e] = -1; if (e < npairs_max) pair_val[e] = 0.0; }   // (index padding: the ballot gather reads whole 64-pair chunks; the pad entries match no variable)
        for (int e = sid; e < nm * TM_NSHARED; e += NS) shared_adj[e] = 0.0;
#pragma unroll
        for (int k = 0; k < KP; k++) { const int e = sid + k * NS; if (e < L.Nparams) p[e] = rp[k]; }
        for (int e = sid + KP * NS; e < L.Nparams; e += NS) p[e] = params[(size_t)chain * L.Nparams + e];
        if (aux_in_lds) {
#pragma unroll
            for (int k = 0; k < KA; k++) { const int e = sid + k * NS; if (e < nm * AUXD) s_aux[e] = ra[k]; }
            for (int e = sid + KA * NS; e < nm * AUXD; e += NS) s_aux[e] = aux_src[e];
        }
        if (sid < tiles) { s_u[sid] = rh.u0; if (sid == tiles - 1) s_u[tiles] = rh.u1; }
        for (int t = sid + NS; t < tiles; t += NS) {
            const TmTileHdr H = thdr[(size_t)chain * tiles + t];
            s_u[t] = H.u0;
            if (t == tiles - 1) s_u[tiles] = H.u1;
        }
        if (L.family != TM_FAM_GAUSS) {
            if (sid < CD) reinterpret_cast<double *>(&C)[sid] = rc;
            for (int e = sid + NS; e < CD; e += NS) reinterpret_cast<double *>(&C)[e] = reinterpret_cast<const double *>(chain_rec + chain)[e];
        }
    }
    BW_TS(12, 0);
    BW_TS(15, 0);
    __syncthreads();
    if (rv >= 0 && rv < L.Nparams) s_inv[rv] = rk;             // inverse of index_to_relax (the clears are behind the barrier)
    for (int k = TM_BW_THREADS - 128 + tid; k < Nvars; k += TM_BW_THREADS) {   // more variables than staging threads
        const int r = relax[k];
        if (r >= 0 && r < L.Nparams) s_inv[r] = k;
    }
    BW_TS(1, 0);
#if defined(TM_BW_STOP) && TM_BW_STOP == 1
    return;   // timing-only build
#endif

    // ---------------- phase 1: per multiplet ----------------
    // phase 1a: every (multiplet, slot) pair sums its tile partials in tile order -- all threads, LDS result
    for (int item = tid; item < nm * TM_GSLOTS; item += TM_BW_THREADS) {
        const int j = item / TM_GSLOTS, sl = item - j * TM_GSLOTS;
        const TmMultFull &M = auxp[j];
        double acc = 0.0;
        if (M.status == 0 && (sl < M.ncomp || (sl >= 7 && sl < 12) || sl >= 21)) {      // B_k, A_am, C, asymmetry sums (tamcmc_eval_body.h)
            // tiles whose units meet the window's units [ua, ub] (the multiplet is on their active lists).  Tiles of equal
            // length (the default) are found by division; per-chain boundaries (equal-cost tiles) by a search over the
            // tile starts: the first one is the tile holding unit ua -- the last tile that starts at or before it (empty
            // tiles share a start with their successor and are passed over by taking the last).
            const int ua = M.imin >> TM_UNIT_SHIFT, ub = (M.imax - 1) >> TM_UNIT_SHIFT;
            if (uniform_su > 0) {
                const int su = uniform_su, tA = tm_tile_of_unit(geom, su, ua), tBq = tm_tile_of_unit(geom, su, ub), tB = (tBq < tiles - 1) ? tBq : tiles - 1;
                const double *G = gmult + (((size_t)chain * tiles + tA) * nm + j) * TM_GSLOTS + sl;
                const size_t stride = (size_t)nm * TM_GSLOTS;
#pragma unroll 4
                for (int t = tA; t <= tB; t++) acc += G[(size_t)(t - tA) * stride];
            } else {
                int lo = 0, hi = tiles - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_u[mid] <= ua) lo = mid; else hi = mid - 1;
                }
                for (int t = lo; t < tiles && s_u[t] <= ub; t++)
                    if (s_u[t + 1] > s_u[t] && s_u[t + 1] > ua)
                        acc += gmult[(((size_t)chain * tiles + t) * nm + j) * TM_GSLOTS + sl];
            }
        }
        s_G[j * TM_GSTRIDE + sl] = acc;
    }
    __syncthreads();
    BW_TS(2, 0);
#if defined(TM_BW_STOP) && TM_BW_STOP == 2
    return;   // timing-only build
#endif
    // Beside phase 1b (one wave, 21 lanes of it here, busy for ~2 us): the noise partials on wave 1 and the likelihood on
    // wave 3 -- each one trip to memory and some arithmetic, consumed only after the next barrier.  (Ahead of the staging
    // barrier they held the whole workgroup back by 0.7 us.)
    // noise partials: wave 1, one lane per (tile, cell part) (stride 64).  A tile meets at most two cells and leaves one
    // set of partials per cell.  On a cell whose background was evaluated as a polynomial the eval kernel left the
    // moments m_j = sum w dl^j (slot 9: j = 0, slot j-1: j = 1..9); with profile h's series u_h = sum c_j dl^j (setup
    // kernel) and u(1-u) = -(1/p) du/d(dl) = sum d_j dl^j, d_j = -(j+1) c_{j+1}/p:
    //   sum w u = sum c_j m_j,  sum w u(1-u) = sum d_j m_j,  sum w u(1-u)(lt + log x) = (lt + lxc) sum d_j m_j + sum d_j m_{j+1}.
    // Other cells hold those three sums already.  Lanes then meet in LDS and are summed in a fixed order.
    if (tid >= 64 && tid < 128) {
        const int lane = tid - 64;
        double acc[TM_NSLOTS];
#pragma unroll
        for (int sl = 0; sl < TM_NSLOTS; sl++) acc[sl] = 0.0;
        // Everything this lane needs is fetched in ONE round trip to memory (the partials come from other XCDs, so each
        // trip is ~1 us): with tiles of equal length the tile's units follow from its number, so neither the cell
        // record nor the series wait for the tile header, and the series are fetched whether or not the cell used them.
        const TmNoise *nz = noise + chain;
        const int nh = nz->nh;
        double nzp[TM_MAXH], nzlt[TM_MAXH];
#pragma unroll
        for (int h = 0; h < TM_MAXH; h++) { nzp[h] = nz->p[h]; nzlt[h] = nz->lt[h]; }
        const int units = (int)((L.Nx + TM_UNIT_BINS - 1) >> TM_UNIT_SHIFT);   // tm_units()
        for (int it = lane; it < 2 * tiles; it += 64) {
            const int t = it >> 1, part = it & 1;
            int u0, u1;
            if (uniform_su > 0) {      // what the setup kernel wrote into the header (tamcmc_setup_body.h)
                u0 = tm_tile_first_unit(geom, uniform_su, t); u1 = u0 + tm_tile_units(geom, uniform_su, t);
                if (u0 > units) u0 = units;
                if (u1 > units) u1 = units;
            } else {
                const TmTileHdr H = thdr[(size_t)chain * tiles + t];
                u0 = H.u0; u1 = H.u1;
            }
            if (u1 <= u0) continue;
            const int ce = (u0 >> TM_CELL_SHIFT) + part;
            if (ce > ((u1 - 1) >> TM_CELL_SHIFT)) continue;          // the tile lies in one cell: no second part
            const double *G = gnoise + (((size_t)chain * tiles + t) * 2 + part) * TM_NSLOTS;
            double g[TM_NSLOTS];
#pragma unroll
            for (int sl = 0; sl < TM_NSLOTS; sl++) g[sl] = G[sl];
            const TmCellRec *R = cell + (size_t)chain * cells + ce;
            const int npoly = R->npoly;
            const double lxc = R->lxc;
            double cs[TM_MAXH][TM_HSER];
#pragma unroll
            for (int h = 0; h < TM_MAXH; h++)
#pragma unroll
                for (int j = 0; j < TM_HSER; j++) cs[h][j] = hser[(((size_t)chain * cells + ce) * TM_MAXH + h) * TM_HSER + j];
            if (nh > 0 && npoly != 0) {
                double m[TM_HSER];
                m[0] = g[3 * TM_MAXH];
#pragma unroll
                for (int j = 1; j < TM_HSER; j++) m[j] = g[j - 1];
#pragma unroll
                for (int sl = 0; sl < 3 * TM_MAXH; sl++) g[sl] = 0.0;
#pragma unroll
                for (int h = 0; h < TM_MAXH; h++) {
                    if (h < nh) {
                        const double *c = cs[h];
                        const double ip = -1.0 / nzp[h];
                        double k0 = 0.0, k1 = 0.0, k2 = 0.0;
#pragma unroll
                        for (int j = TM_PDEG; j >= 0; j--) {
                            const double dj = (double)(j + 1) * c[j + 1] * ip;
                            k0 = __builtin_fma(c[j], m[j], k0);
                            k1 = __builtin_fma(dj, m[j], k1);
                            k2 = __builtin_fma(dj, m[j + 1], k2);
                        }
                        g[3 * h] = k0;
                        g[3 * h + 1] = k1;
                        g[3 * h + 2] = __builtin_fma(nzlt[h] + lxc, k1, k2);
                    }
                }
            }
#pragma unroll
            for (int sl = 0; sl < TM_NSLOTS; sl++) acc[sl] += g[sl];
        }
        // lane l -> LDS; then lane l sums slot l/4 over lanes (l%4)*16 .. +15, and the four parts are added in a tree
#pragma unroll
        for (int sl = 0; sl < TM_NSLOTS; sl++) s_gt[sl][lane] = acc[sl];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int sl = lane >> 2, part = lane & 3;
        double tsum = 0.0;
#pragma unroll
        for (int e = 0; e < 16; e++) tsum += s_gt[sl][part * 16 + e];
        tsum += __shfl_xor(tsum, 1, 64);
        tsum += __shfl_xor(tsum, 2, 64);
        if (part == 0) s_gn[sl] = tsum;
    }
    BW_TS(13, 64);
    // finalize (same arithmetic and order as the likelihood-only path's in-launch finalize): wave 3
    if (wv == 3) {
        const int lane = tid - 192;
        const double *pp = part + (size_t)chain * tiles * 4;      // per tile: {S1, mantissa product, exponent sum, -}
        const double Tc = Tcoefs[2 * chain];                      // {T, wscale} pairs written by the setup kernel
        const int st0 = noise[chain].status;                      // (both fetched together with the partials)
        double s1 = 0.0, s2 = 0.0;
        for (int t = lane; t < tiles; t += 64) {
            s1 += pp[4 * t];
            if (L.likelihood_case == 0) s2 += tm_tile_logsum(pp[4 * t + 1], pp[4 * t + 2]);
        }
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
        if (lane == 0) {
            double f = (L.likelihood_case == 0) ? -L.like_p * (s1 + s2) : -s1;
            f = f / Tc;
            int st = st0;
            if (st != 0) f = __builtin_nan("");
            else if (!(f == f)) st = 1;
            logL[chain] = f;
            if (status) status[chain] = st;
        }
    }
    BW_TS(14, 192);
    // phase 1b: chain rule, one thread per multiplet (tm_bw_mult above)
    for (int j = tid; j < nm; j += TM_BW_THREADS)
        tm_bw_mult(L, &C, auxp + j, s_G + (size_t)j * TM_GSTRIDE, p, shared_adj + (size_t)j * TM_NSHARED, pair_idx + j * TM_NPAIR,
                   pair_val + j * TM_NPAIR);
    __syncthreads();
    BW_TS(3, 0);
#if defined(TM_BW_STOP) && TM_BW_STOP == 3
    return;   // timing-only build
#endif

    // ---------------- phase 2: chain-level adjoints and noise ----------------
    if (tid < 4 * TM_NSHARED) {
        // four lanes per slot, each a contiguous quarter of the multiplets in order; the quarters are then added in a
        // fixed order (a single lane per slot is one LDS round trip per multiplet, one after the other)
        const int slot = tid >> 2, q = tid & 3, per = (nm + 3) >> 2;
        const int j_lo = q * per, j_hi = (j_lo + per < nm) ? j_lo + per : nm;
        double acc = 0.0;
        for (int j = j_lo; j < j_hi; j++) acc += shared_adj[(size_t)j * TM_NSHARED + slot];
        const double a1 = __shfl_down(acc, 1, 4), a2 = __shfl_down(acc, 2, 4), a3 = __shfl_down(acc, 3, 4);
        if (q == 0) s_S[slot] = ((acc + a1) + a2) + a3;
    }
    __syncthreads();
    BW_TS(4, 0);
#if defined(TM_BW_STOP) && TM_BW_STOP == 4
    return;   // timing-only build
#endif
    BW_TS(5, 0);
    BW_TS(10, 64);
    if (tid == 0) {
        int *pi = pair_idx + nm * TM_NPAIR;
        double *pv = pair_val + nm * TM_NPAIR;
        int np = 0;
        if (L.family != TM_FAM_GAUSS) {
            const int id = L.model_case, s0 = L.s, q = L.q;
            // every LDS read first (independent, pipelined), then arithmetic, then the pair stores: this thread
            // runs alone, so a load issued after a possibly-aliasing store would cost a full LDS round trip each
            double S[TM_NSHARED], dr[9], ps[8], pqv[9], pVv[3];
#pragma unroll
            for (int i = 0; i < TM_NSHARED; i++) S[i] = s_S[i];
#pragma unroll
            for (int i = 0; i < 8; i++) ps[i] = (s0 + i < L.Nparams) ? p[s0 + i] : 0.0;
#pragma unroll
            for (int i = 0; i < 9; i++) pqv[i] = (q + i < L.Nparams) ? p[q + i] : 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) pVv[i] = (L.Nmax + i < L.Nparams) ? p[L.Nmax + i] : 0.0;
            dr[0] = C.dratios[1][1]; dr[1] = C.dratios[1][2];
            dr[2] = C.dratios[2][2]; dr[3] = C.dratios[2][3]; dr[4] = C.dratios[2][4];
            dr[5] = C.dratios[3][3]; dr[6] = C.dratios[3][4]; dr[7] = C.dratios[3][5]; dr[8] = C.dratios[3][6];
            pi[np] = s0 + 1; pv[np] = S[SL_ETA]; np++;
            pi[np] = s0 + 2; pv[np] = S[SL_A3]; np++;
            pi[np] = s0 + 5; pv[np] = S[SL_ASYM]; np++;
            if (id == 6) {
                pi[np] = s0; pv[np] = tm_sign(ps[0]) * S[SL_FS1]; np++;
                pi[np] = s0 + 6; pv[np] = tm_sign(ps[6]) * S[SL_FS2]; np++;
            }
            // inclination adjoint from the ratio adjoints (slots SL_RATIO.. are in (l, |m|) order like dr[])
            double adj_inc = 0.0;
#pragma unroll
            for (int t = 0; t < 9; t++) adj_inc += S[SL_RATIO + t] * dr[t];
            if (id == 12) {
                const int nr = (L.lmax >= 3) ? 9 : (L.lmax == 2) ? 5 : (L.lmax == 1) ? 2 : 0;
#pragma unroll
                for (int t = 0; t < 9; t++)
                    if (t < nr) { pi[np] = q + t; pv[np] = tm_sign(pqv[t]) * S[SL_RATIO + t]; np++; }
            }
            if (id == 2 || id == 9 || id == 10 || id == 11) {
                const double pa = ps[3], pb = ps[4], r2 = pa * pa + pb * pb;
                pi[np] = s0 + 3; pv[np] = 2.0 * pa * S[SL_A1] + adj_inc * (180. / PI) * (-pb / r2); np++;
                pi[np] = s0 + 4; pv[np] = 2.0 * pb * S[SL_A1] + adj_inc * (180. / PI) * (pa / r2); np++;
            } else {
                if (id == 3 || id == 12 || id == 13 || id == 14) { pi[np] = s0; pv[np] = tm_sign(ps[0]) * S[SL_A1]; np++; }
                if (id == 3 || id == 6 || id == 7 || id == 8) { pi[np] = q; pv[np] = adj_inc; np++; }
            }
            double adj_V[4] = {0.0, S[SL_V], S[SL_V + 1], S[SL_V + 2]};
            if (id == 9) {
                // numax = sum_n p_n (F0_n + sum_l V_l F_l,n) / sum_n p_n (1 + sum_l V_l), models.cpp:1372-1390
                const double an = S[SL_NUMAX], numax = C.numax, Htot = C.Htot;
                double vsum = 1.0;
                for (int l = 1; l <= L.lmax; l++) vsum += C.Vl[l];
                for (int n = 0; n < L.Nmax; n++) {
                    double fsum = p[L.off_f[0] + n];
                    for (int l = 1; l <= L.lmax; l++) fsum += C.Vl[l] * p[L.off_f[l] + n];
                    pi[np] = n; pv[np] = an * (fsum - numax * vsum) / Htot; np++;
                    pi[np] = L.off_f[0] + n; pv[np] = an * p[n] / Htot; np++;
                    for (int l = 1; l <= L.lmax; l++) {
                        pi[np] = L.off_f[l] + n; pv[np] = an * p[n] * C.Vl[l] / Htot; np++;
                        adj_V[l] += an * p[n] * (p[L.off_f[l] + n] - numax) / Htot;
                    }
                }
            }
            if (L.family == TM_FAM_GLOBAL && id != 13)
                for (int l = 1; l <= L.lmax; l++) { pi[np] = L.Nmax + l - 1; pv[np] = tm_sign(pVv[l - 1]) * adj_V[l]; np++; }
        }
    }
    BW_TS(6, 0);
    if (tid == 64) {   // noise terms on another wave, concurrently with the chain-level work of thread 0
        int *pi = pair_idx + nm * TM_NPAIR + (ncp - 16);
        double *pv = pair_val + nm * TM_NPAIR + (ncp - 16);
        int np = 0;
        // noise terms: sum the per-tile partials in tile order
        const double *Gn = s_gn;
        int z = L.z, Nnoise = L.Nnoise, nharvey = L.nharvey;
        bool take_abs = true;
        if (L.model_case == 0) { z = 3; Nnoise = 1; nharvey = 0; take_abs = false; }
        if (L.model_case == 1) { z = 3; Nnoise = 4; nharvey = 1; }
        const double sw = Gn[3 * TM_MAXH];
        int slot = 0;
        for (int k = 0; k < nharvey; k++) {
            const double Hs = p[z + 3 * k], ts = p[z + 3 * k + 1], ps = p[z + 3 * k + 2];
            const double H = fabs(Hs), tau = fabs(ts), pw = fabs(ps);
            if (tau != 0) {
                if (pw == 0) { pi[np] = z + 3 * k; pv[np] = tm_sign(Hs) * 0.5 * sw; np++; continue; }
                const double B1 = Gn[3 * slot], B2 = Gn[3 * slot + 1], B3 = Gn[3 * slot + 2];
                pi[np] = z + 3 * k;     pv[np] = tm_sign(Hs) * B1; np++;
                pi[np] = z + 3 * k + 1; pv[np] = tm_sign(ts) * (-H * pw / tau * B2); np++;
                pi[np] = z + 3 * k + 2; pv[np] = tm_sign(ps) * (-H * B3); np++;
                slot++;
            }
        }
        pi[np] = z + Nnoise - 1; pv[np] = (take_abs ? tm_sign(p[z + Nnoise - 1]) : 1.0) * sw; np++;
        if (L.family == TM_FAM_GAUSS) {
            const double E0 = Gn[13], E1 = Gn[14], E2 = Gn[15];
            const double gA = (L.model_case == 0) ? p[0] : fabs(p[0]);
            const double s2 = p[1] * p[1];
            pi[np] = 0; pv[np] = ((L.model_case == 0) ? 1.0 : tm_sign(p[0])) * E0; np++;
            pi[np] = 2; pv[np] = gA * E1 / s2; np++;
            pi[np] = 1; pv[np] = 0.5 * gA * E2 / (s2 * s2) * 2.0 * p[1]; np++;
        }
    }
    BW_TS(11, 64);
    __syncthreads();
    BW_TS(7, 0);
#if defined(TM_BW_STOP) && TM_BW_STOP == 5
    return;   // timing-only build
#endif

    // ---------------- phase 3: gather per variable, in pair order ----------------
    // Every wave takes a contiguous run of 64-pair chunks; a lane stands for a pair when it reads (the variable its
    // parameter is, through the inverse map) and for a variable when it sums: one ballot per variable tells lane k which
    // of the chunk's pairs are its own, and it adds those values in pair order.  The waves' partial rows are then added
    // in wave order, so a variable's pairs are summed in pair order whatever their number.  (Comparing every variable
    // with every pair -- 44 x 316 here -- kept the LDS pipe busy for 2.7 us; a value read only on a match inside such
    // a scan is a round trip in a divergent branch.)
    {
        constexpr int NWV = TM_BW_THREADS / 64;
        const int lane = tid & 63;
        const int nchunks = (npairs_max + 63) >> 6;
        const int cpw = (nchunks + NWV - 1) / NWV;
        const int c_lo = wv * cpw, c_hi = (c_lo + cpw < nchunks) ? c_lo + cpw : nchunks;
        for (int k0 = 0; k0 < Nvars; k0 += 64) {
            const int nk = (Nvars - k0 < 64) ? Nvars - k0 : 64;
            double acc = 0.0;
            for (int c = c_lo; c < c_hi; c++) {
                const int e = (c << 6) + lane;
                const int idx = (e < npairs_max) ? pair_idx[e] : -1;
                const int ke = (idx >= 0) ? s_inv[idx] - k0 : -1;          // this pair's variable, relative to the block of 64
                unsigned lo = 0, hi = 0;
                for (int kk = 0; kk < nk; kk += 4) {       // four at a time (past nk: no pair matches, the lanes get empty sets)
                    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(ke == kk), m1 = __builtin_amdgcn_ballot_w64(ke == kk + 1);
                    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(ke == kk + 2), m3 = __builtin_amdgcn_ballot_w64(ke == kk + 3);
                    if (lane == kk) { lo = (unsigned)m0; hi = (unsigned)(m0 >> 32); }
                    if (lane == kk + 1) { lo = (unsigned)m1; hi = (unsigned)(m1 >> 32); }
                    if (lane == kk + 2) { lo = (unsigned)m2; hi = (unsigned)(m2 >> 32); }
                    if (lane == kk + 3) { lo = (unsigned)m3; hi = (unsigned)(m3 >> 32); }
                }
                unsigned long long mine = ((unsigned long long)hi << 32) | lo;
                const double *pvc = pair_val + (c << 6);
                while (mine != 0) {
                    acc += pvc[__builtin_ctzll(mine)];
                    mine &= mine - 1;
                }
            }
            if (lane < nk) s_part[(size_t)wv * Nvars + k0 + lane] = acc;
        }
    }
    BW_TS(8, 0);
    __syncthreads();
    for (int k = tid; k < Nvars; k += TM_BW_THREADS) {
        double v[NWV_ROWS];
#pragma unroll
        for (int w = 0; w < NWV_ROWS; w++) v[w] = s_part[(size_t)w * Nvars + k];
        double tot = v[0];
#pragma unroll
        for (int w = 1; w < NWV_ROWS; w++) tot += v[w];
        s_row[k] = tot;
    }
    // the row leaves as consecutive 8-byte stores of consecutive lanes: the caller's buffer may be host memory mapped
    // over PCIe, where a scattered store is a transaction of its own (each thread stores what it has just summed)
    BW_TS(9, 0);
    for (int k = tid; k < Nvars; k += TM_BW_THREADS) grad[(size_t)chain * Nvars + k] = s_row[k];
#ifdef TM_SU_TRACE   // the setup kernel's stamps (tamcmc_setup_body.h), handed on
    __syncthreads();
    if (tid < 20 && tid < Nvars) grad[(size_t)chain * Nvars + tid] = hser[(size_t)chain * cells * TM_MAXH * TM_HSER + tid];
#endif
#ifdef TM_BW_TRACE
    __syncthreads();
    if (tid < 20 && tid < Nvars) grad[(size_t)chain * Nvars + tid] = (double)(long long)(s_ts[tid] - s_ts[0]);
    if (tid < 5 && 20 + tid < Nvars) grad[(size_t)chain * Nvars + 20 + tid] = (double)(long long)(g_bfine[tid] - s_ts[0]);
#endif
}

int tm_launch_backward(const TmLayout &L, int Nchains, int units, int cells, int tiles, int equal_cost, TmCostModel geom, const double *d_params,
                       const double *d_Tcoefs, const void *d_chain_rec, const void *d_aux, const TmNoise *d_noise,
                       const double *d_part, const double *d_gmult, const double *d_gnoise, const TmCellRec *d_cell,
                       const TmTileHdr *d_thdr, const double *d_hser, int Nvars, const int32_t *d_index_to_relax, double *d_grad,
                       double *d_logL, int32_t *d_status, void *stream)
{
    const int nm = L.n_mult;
    const int ncp = TM_NCPAIR + (L.model_case == 9 ? (L.Nmax * (L.lmax + 2) + L.lmax) : 0);
    const int npairs_max = nm * TM_NPAIR + ncp;
    if (units < 1 || cells < 1 || tiles < 1) return (int)hipErrorInvalidValue;
    size_t lds = ((size_t)L.Nparams + (size_t)Nvars) * sizeof(double) + (size_t)npairs_max * sizeof(double) +
                 (size_t)nm * (TM_NSHARED + TM_GSLOTS + 1) * sizeof(double) + 16 + (size_t)((npairs_max + 7) & ~7) * sizeof(int) +
                 (size_t)((tiles + 2) & ~1) * sizeof(int) + (size_t)((L.Nparams + 1) & ~1) * sizeof(int) +
                 (size_t)(TM_BW_THREADS / 64) * Nvars * sizeof(double);
    const size_t aux_bytes = (size_t)nm * sizeof(TmMultFull);
    const int aux_in_lds = (lds + aux_bytes <= 100 * 1024) ? 1 : 0;
    if (aux_in_lds) lds += aux_bytes;
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tamcmc_backward_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // tiles of equal length (su units each): the tiles a window meets follow by division; else (per-chain boundaries) by search
    const int uniform_su = equal_cost ? 0 : (units + tiles - 1) / tiles;
    hipLaunchKernelGGL(tamcmc_backward_kernel, dim3(Nchains), dim3(TM_BW_THREADS), lds, (hipStream_t)stream, L, tiles, cells, uniform_su, geom,
                       d_params, d_Tcoefs, static_cast<const TmChain *>(d_chain_rec), static_cast<const TmMultFull *>(d_aux), d_noise,
                       d_part, d_gmult, d_gnoise, d_cell, d_thdr, d_hser, Nvars, d_index_to_relax, d_grad, d_logL, d_status, aux_in_lds);
    return (int)hipGetLastError();
}
