// tamcmc_eval.hip -- the hot kernel: for one (chain, tile of frequency bins) build the model
// spectrum M(x) from the chain's multiplet table, fold it into the chi(2,2p) / chi_square
// log-likelihood partial sums and (GRAD) into the per-multiplet gradient partial sums.
//
// Reference functions fused here:
//   build_l_mode_a1etaa3 & friends   build_lorentzian.cpp:15-205   sum over m of (asymmetric) Lorentzians
//   optimum_lorentzian_calc_*        build_lorentzian.cpp:258-769  add the multiplet inside [imin,imax) only
//   harvey_like / harvey1985         noise_models.cpp:17-69
//   model_Test_Gaussian / _Harvey_Gaussian   models.cpp:1968-2034 (Gaussian term)
//   likelihood_chi22p / _chi_square  likelihoods.cpp:17-39
//
// Mapping to gfx950 (wave64):
//   grid = (tiles, chains); workgroup = 256 threads = 4 waves; a tile is 256*K consecutive bins and
//   thread t owns bins base + k*256 + t (k < K), so every global load is a coalesced 8-byte-per-lane
//   stream (x, y, log x: 24 B per bin; the 2.4 MB working set of a 1e5-bin star lives in L2).
//   The chain's multiplets whose window meets the tile are compacted (ballot, in table order, so the
//   summation order is fixed) and staged in LDS in chunks of TM_CHUNK; every thread then walks the
//   staged list with wave-uniform control flow.
//   Arithmetic per Lorentzian component: d = 2x - 2nu; E = d*d + Gamma^2 (2 fp64 ops); one
//   reciprocal per MULTIPLET via batch inversion (prefix products of the E's), not per component:
//   sum_m h_m Gamma^2 / E_m.  This is algebraically the reference's H V_m / (1 + 4 (x-nu_m)^2/Gamma^2)
//   and differs from it only in rounding (<~1e-15 relative per bin).
//   Sum of log M: mantissas are multiplied and exponents added per bin (v_frexp_*), one log per
//   thread and tile instead of one per bin.
//   Reductions: wave shuffles -> one LDS slot per wave -> one partial per (chain, tile); the final
//   sum over tiles runs in a fixed order in tamcmc_finalize_kernel (no atomics: bitwise reproducible).
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"

#define TM_WAVES (TM_THREADS / 64)

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

__device__ __forceinline__ double tm_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the m-components of one multiplet at one bin, optionally returning every 1/E_m.
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

// Forward: add multiplet `sm` (LDS) to acc[] for this thread's K bins.
template <int NC, int K>
__device__ __forceinline__ void tm_accum_mult(const TmMult *sm, const double (&x2)[K], const int (&bi)[K], double (&acc)[K])
{
    double nu2[NC], hq[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) { nu2[m] = sm->nu2[m]; hq[m] = sm->hq[m]; }
    const double g2 = sm->g2;
    const int imin = sm->imin, imax = sm->imax;
    const bool has_asym = sm->has_asym != 0;
    const double aAh = 0.5 * sm->aA, aB = sm->aB, c2 = sm->c2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double d[NC], r[NC];
        double s = tm_mult_value<NC>(x2[k], nu2, hq, g2, d, r);
        if (has_asym) {
            const double a = __builtin_fma(x2[k], aAh, aB);
            s = s * __builtin_fma(a, a, c2);
        }
        const bool inside = (bi[k] >= imin) && (bi[k] < imax);
        acc[k] += inside ? s : 0.0;
    }
}

// Backward: per-component partial sums of this thread for multiplet `sm`.
//   g[3m+0] = sum wA r_m, g[3m+1] = sum wA d_m r_m^2, g[3m+2] = sum wA r_m^2   (d = 2x - 2nu, r = 1/E)
//   g[21..23] = sum w S, sum w S a, sum w S a x    (S = un-asymmetrised multiplet sum; only if asym != 0)
template <int NC, int K>
__device__ __forceinline__ void tm_grad_mult(const TmMult *sm, const double (&x2)[K], const int (&bi)[K],
                                             const double (&w)[K], double *s_red_row, int lane)
{
    double g[TM_GSLOTS];
    double nu2[NC], hq[NC];
#pragma unroll
    for (int m = 0; m < NC; m++) { nu2[m] = sm->nu2[m]; hq[m] = sm->hq[m]; }
    const double g2 = sm->g2;
    const int imin = sm->imin, imax = sm->imax;
    const bool has_asym = sm->has_asym != 0;
    const double aAh = 0.5 * sm->aA, aB = sm->aB, c2 = sm->c2;
#pragma unroll
    for (int s = 0; s < TM_GSLOTS; s++) g[s] = 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        double d[NC], r[NC];
        const double S = tm_mult_value<NC>(x2[k], nu2, hq, g2, d, r);
        const bool inside = (bi[k] >= imin) && (bi[k] < imax);
        const double wk = inside ? w[k] : 0.0;
        double wA = wk;
        if (has_asym) {
            const double a = __builtin_fma(x2[k], aAh, aB);
            wA = wk * __builtin_fma(a, a, c2);
            const double ws = wk * S;
            g[21] += ws;
            g[22] = __builtin_fma(ws, a, g[22]);
            g[23] = __builtin_fma(ws * a, 0.5 * x2[k], g[23]);
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
    // wave-level reduction with compile-time slot indices (a runtime-indexed g[] would live in scratch)
#pragma unroll
    for (int s = 0; s < 3 * NC; s++) {
        const double v = tm_wave_sum(g[s]);
        if (lane == 0) s_red_row[s] = v;
    }
#pragma unroll
    for (int s = 21; s < 24; s++) {
        const double v = has_asym ? tm_wave_sum(g[s]) : 0.0;
        if (lane == 0) s_red_row[s] = v;
    }
}

template <int K, bool GRAD>
__global__ __launch_bounds__(TM_THREADS) void tamcmc_eval_kernel(TmEvalArgs a)
{
    constexpr int TB = TM_THREADS * K;
    const int tile = blockIdx.x, chain = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = tile * TB;

    __shared__ double s_mult[TM_CHUNK * TM_MULT_DOUBLES];
    __shared__ int s_idx[TM_CHUNK];
    __shared__ int s_nact;
    __shared__ double s_red[TM_WAVES][TM_GSLOTS];

    const TmMult *gm = a.mult + (size_t)chain * a.n_mult;
    const TmNoise *gn = a.noise + chain;

    double x2[K], acc[K];
    int bi[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = base + k * TM_THREADS + tid;
        const bool valid = i < a.Nx;
        bi[k] = valid ? i : -1;
        x2[k] = 2.0 * a.x[valid ? i : a.Nx - 1];
        acc[k] = 0.0;
    }

    // ---------------- pass 1: model spectrum ----------------
    for (int c0 = 0; c0 < a.n_mult; c0 += TM_CHUNK) {
        __syncthreads();
        if (wave == 0) {
            const int j = c0 + lane;
            bool act = false;
            if (lane < TM_CHUNK && j < a.n_mult) {
                const int imin = gm[j].imin, imax = gm[j].imax;
                act = (imin < base + TB) && (imax > base);
            }
            const unsigned long long mask = __ballot(act);
            if (act) s_idx[__popcll(mask & ((1ull << lane) - 1ull))] = j;
            if (lane == 0) s_nact = __popcll(mask);
        }
        __syncthreads();
        const int nact = s_nact;
        for (int e = tid; e < nact * TM_MULT_DOUBLES; e += TM_THREADS) {
            const int slot = e / TM_MULT_DOUBLES, f = e - slot * TM_MULT_DOUBLES;
            s_mult[e] = reinterpret_cast<const double *>(gm + s_idx[slot])[f];
        }
        __syncthreads();
        for (int jj = 0; jj < nact; jj++) {
            const TmMult *sm = reinterpret_cast<const TmMult *>(s_mult) + jj;
            switch (sm->ncomp) {
            case 1: tm_accum_mult<1, K>(sm, x2, bi, acc); break;
            case 3: tm_accum_mult<3, K>(sm, x2, bi, acc); break;
            case 5: tm_accum_mult<5, K>(sm, x2, bi, acc); break;
            default: tm_accum_mult<7, K>(sm, x2, bi, acc); break;
            }
        }
    }

    // ---------------- noise background + Gaussian term ----------------
    const int nh = gn->nh;
    double lxv[K];
    if (nh > 0) {
#pragma unroll
        for (int k = 0; k < K; k++) lxv[k] = a.lx[bi[k] >= 0 ? bi[k] : a.Nx - 1];
        for (int h = 0; h < nh; h++) {
            const double H = gn->H[h], lt = gn->lt[h], p = gn->p[h];
#pragma unroll
            for (int k = 0; k < K; k++) {
                const double t = exp(p * (lt + lxv[k]));
                acc[k] += H * (1.0 / (t + 1.0));
            }
        }
    }
    if (gn->has_gauss) {
        const double gA = gn->gA, gnu0 = gn->gnu0, gs2 = gn->gs2;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double dd = 0.5 * x2[k] - gnu0;
            acc[k] = gA * exp((-0.5 * (dd * dd)) / gs2) + acc[k];
        }
    }
    {
        const double N0 = gn->N0;
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] += N0;
    }

    if (a.row_of_chain != nullptr) {
        const int row = a.row_of_chain[chain];
        if (row >= 0) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (bi[k] >= 0) a.model_out[(size_t)row * a.Nx + bi[k]] = acc[k];
        }
    }

    // ---------------- likelihood partial sums ----------------
    double S1 = 0.0, S2 = 0.0;
    double w[K];
    if (a.likelihood_case == 0) {
        // -p * (sum y/M + sum log M), likelihoods.cpp:23-25
        double P = 1.0;
        int esum = 0;
        bool bad = false;
        const double scale = GRAD ? a.like_p / a.Tcoefs[chain] : 0.0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (bi[k] >= 0) {
                const double M = acc[k];
                const double yv = a.y[bi[k]];
                const double rM = 1.0 / M;
                S1 = __builtin_fma(yv, rM, S1);
                int e;
                P *= frexp(M, &e);
                esum += e;
                bad = bad || !(M > 0.0) || (M > 1.7e308);
                if (GRAD) w[k] = scale * (yv * rM * rM - rM);  // d(logL/T)/dM_i
            } else if (GRAD) {
                w[k] = 0.0;
            }
        }
        S2 = log(P) + (double)esum * 0.693147180559945309417232;
        if (bad) S2 = __builtin_nan("");
    } else {
        // -sum (y-M)^2 / sigma^2, likelihoods.cpp:36
        const double scale = GRAD ? 1.0 / a.Tcoefs[chain] : 0.0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (bi[k] >= 0) {
                const double dd = a.y[bi[k]] - acc[k];
                const double is2 = a.isig2[bi[k]];
                S1 = __builtin_fma(dd * dd, is2, S1);
                if (GRAD) w[k] = scale * 2.0 * dd * is2;
            } else if (GRAD) {
                w[k] = 0.0;
            }
        }
    }
    S1 = tm_wave_sum(S1);
    S2 = tm_wave_sum(S2);
    __syncthreads();
    if (lane == 0) { s_red[wave][0] = S1; s_red[wave][1] = S2; }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int wv = 0; wv < TM_WAVES; wv++) { t1 += s_red[wv][0]; t2 += s_red[wv][1]; }
        double *out = a.part + ((size_t)chain * a.tiles + tile) * 2;
        out[0] = t1;
        out[1] = t2;
    }

    // ---------------- pass 2: gradient partial sums ----------------
    if (GRAD) {
        for (int c0 = 0; c0 < a.n_mult; c0 += TM_CHUNK) {
            __syncthreads();
            if (wave == 0) {
                const int j = c0 + lane;
                bool act = false;
                if (lane < TM_CHUNK && j < a.n_mult) {
                    const int imin = gm[j].imin, imax = gm[j].imax;
                    act = (imin < base + TB) && (imax > base);
                }
                const unsigned long long mask = __ballot(act);
                if (act) s_idx[__popcll(mask & ((1ull << lane) - 1ull))] = j;
                if (lane == 0) s_nact = __popcll(mask);
            }
            __syncthreads();
            const int nact = s_nact;
            for (int e = tid; e < nact * TM_MULT_DOUBLES; e += TM_THREADS) {
                const int slot = e / TM_MULT_DOUBLES, f = e - slot * TM_MULT_DOUBLES;
                s_mult[e] = reinterpret_cast<const double *>(gm + s_idx[slot])[f];
            }
            __syncthreads();
            for (int jj = 0; jj < nact; jj++) {
                const TmMult *sm = reinterpret_cast<const TmMult *>(s_mult) + jj;
                const int nc = sm->ncomp;
                switch (nc) {
                case 1: tm_grad_mult<1, K>(sm, x2, bi, w, s_red[wave], lane); break;
                case 3: tm_grad_mult<3, K>(sm, x2, bi, w, s_red[wave], lane); break;
                case 5: tm_grad_mult<5, K>(sm, x2, bi, w, s_red[wave], lane); break;
                default: tm_grad_mult<7, K>(sm, x2, bi, w, s_red[wave], lane); break;
                }
                const int nslots = 3 * nc;
                __syncthreads();
                if (tid < TM_GSLOTS) {
                    double t = 0.0;
                    if (tid < nslots || tid >= 21) {
#pragma unroll
                        for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[wv][tid];
                    }
                    a.gmult[(((size_t)chain * a.tiles + tile) * a.n_mult + s_idx[jj]) * TM_GSLOTS + tid] = t;
                }
                __syncthreads();
            }
        }
        // noise terms: per Harvey k: sum w u, sum w t u^2, sum w t u^2 (lt + lx); then sum w  (u = 1/(1+t))
        {
            double gn_[TM_NSLOTS];
#pragma unroll
            for (int s = 0; s < TM_NSLOTS; s++) gn_[s] = 0.0;
            for (int h = 0; h < nh; h++) {
                const double lt = gn->lt[h], p = gn->p[h];
                double b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const double arg = lt + lxv[k];
                    const double t = exp(p * arg);
                    const double u = 1.0 / (t + 1.0);
                    const double wu = w[k] * u;
                    b1 += wu;
                    // t*u^2 -> 0 as t -> inf; written so that inf never multiplies 0
                    const double tu2 = (t < 1.7e308) ? wu * (t * u) : 0.0;
                    b2 += tu2;
                    b3 = __builtin_fma(tu2, arg, b3);
                }
                // compile-time slot indices (h is a runtime, wave-uniform value)
#pragma unroll
                for (int hh = 0; hh < TM_MAXH; hh++)
                    if (hh == h) { gn_[3 * hh] = b1; gn_[3 * hh + 1] = b2; gn_[3 * hh + 2] = b3; }
            }
            double sw = 0.0;
#pragma unroll
            for (int k = 0; k < K; k++) sw += w[k];
            gn_[3 * TM_MAXH] = sw;
            if (gn->has_gauss) {
                // Gaussian term: sum w e, sum w e d, sum w e d^2 with e = exp(-0.5 d^2/s2), d = x - nu0
                const double gnu0 = gn->gnu0, gs2 = gn->gs2;
                double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const double dd = 0.5 * x2[k] - gnu0;
                    const double we = w[k] * exp((-0.5 * (dd * dd)) / gs2);
                    e0 += we;
                    e1 = __builtin_fma(we, dd, e1);
                    e2 = __builtin_fma(we * dd, dd, e2);
                }
                gn_[13] = e0; gn_[14] = e1; gn_[15] = e2;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < TM_NSLOTS; s++) {
                const double v = tm_wave_sum(gn_[s]);
                if (lane == 0) s_red[wave][s] = v;
            }
            __syncthreads();
            if (tid < TM_NSLOTS) {
                double t = 0.0;
#pragma unroll
                for (int wv = 0; wv < TM_WAVES; wv++) t += s_red[wv][tid];
                a.gnoise[((size_t)chain * a.tiles + tile) * TM_NSLOTS + tid] = t;
            }
        }
    }
}

// Sum the per-tile partials of each chain in a fixed order and apply -p (...) / T.
// model_def.cpp:300-302 (logL / Tcoefs[m]); NaN -> status 1; empty window -> NaN, status 2.
__global__ __launch_bounds__(64) void tamcmc_finalize_kernel(int tiles, int likelihood_case, double like_p,
                                                             const double *__restrict__ part,
                                                             const TmNoise *__restrict__ noise,
                                                             const double *__restrict__ Tcoefs,
                                                             double *__restrict__ logL, int32_t *__restrict__ status)
{
    const int chain = blockIdx.x, lane = threadIdx.x;
    const double *p = part + (size_t)chain * tiles * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < tiles; t += 64) { s1 += p[2 * t]; s2 += p[2 * t + 1]; }
    s1 = tm_wave_sum(s1);
    s2 = tm_wave_sum(s2);
    if (lane == 0) {
        double f;
        if (likelihood_case == 0) f = -like_p * (s1 + s2);
        else                      f = -s1;
        f = f / Tcoefs[chain];
        int st = noise[chain].status;
        if (st != 0) f = __builtin_nan("");
        else if (!(f == f)) st = 1;
        logL[chain] = f;
        if (status) status[chain] = st;
    }
}

template <int K>
static int tm_launch_eval_k(const TmEvalArgs &a, int Nchains, bool grad, hipStream_t stream)
{
    dim3 grid(a.tiles, Nchains), block(TM_THREADS);
    if (grad) hipLaunchKernelGGL((tamcmc_eval_kernel<K, true>), grid, block, 0, stream, a);
    else      hipLaunchKernelGGL((tamcmc_eval_kernel<K, false>), grid, block, 0, stream, a);
    return (int)hipGetLastError();
}

int tm_launch_eval(const TmEvalArgs &a, int Nchains, int K, bool grad, void *stream)
{
    switch (K) {
    case 1: return tm_launch_eval_k<1>(a, Nchains, grad, (hipStream_t)stream);
    case 2: return tm_launch_eval_k<2>(a, Nchains, grad, (hipStream_t)stream);
    case 4: return tm_launch_eval_k<4>(a, Nchains, grad, (hipStream_t)stream);
    case 8: return tm_launch_eval_k<8>(a, Nchains, grad, (hipStream_t)stream);
    default: return (int)hipErrorInvalidValue;
    }
}

int tm_launch_finalize(const TmLayout &L, int Nchains, int tiles, const double *d_part, const TmNoise *d_noise,
                       const double *d_Tcoefs, double *d_logL, int32_t *d_status, void *stream)
{
    hipLaunchKernelGGL(tamcmc_finalize_kernel, dim3(Nchains), dim3(64), 0, (hipStream_t)stream, tiles,
                       L.likelihood_case, L.like_p, d_part, d_noise, d_Tcoefs, d_logL, d_status);
    return (int)hipGetLastError();
}
