// tamcmc_derive.h -- device functions that unpack one chain's params row into the quantities the
// eval kernel consumes: per chain (splitting, inclination -> m-height ratios, visibilities) and per
// multiplet (frequency, width, heights, component frequencies, truncation window).
//
// Shared by the setup kernel (forward) and the backward kernel (chain rule of the gradient), which
// re-derives the same intermediates instead of storing them.  Every TU including this file is
// compiled with -ffp-contract=off so that the window arithmetic (floor/ceil of fp64 expressions,
// build_lorentzian.cpp:377-427) rounds exactly like the reference's non-fused CPU code: a one-ulp
// difference there can move a window edge by one bin.
//
// What each block restates (reference file:line):
//   tm_dmm / tm_ratios      function_rot.cpp:20-106  (amplitude_ratio = squared column l of d^l(beta))
//   tm_lin_interpol         interpol.cpp:13-55
//   tm_window               build_lorentzian.cpp:377-427 (and the three textual copies)
//   tm_derive_chain         models.cpp:528-560 (id 2), :846-873 (3), :62-91 (6), :215-244 (7), :370-400 (8),
//                           :1344-1397 (9), :1548-1579 (10), :995-1038 (12), :1163-1171 (13),
//                           :1726-1754 (11), :1877-1880 (14)
//   tm_derive_mult          the `for n` loops of the same functions
#pragma once
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"

struct TmChain {
    double a1;                      // splitting for variants 0 and 2
    double inc;                     // inclination in degrees (used for the ratios)
    double eta, a3, asym, trunc_c;
    double Vl[4];                   // visibilities, Vl[0] = 1
    double ratios[4][TM_MAXM];      // ratios[l][m+l]
    double dratios[4][TM_MAXM];     // d ratios / d inc (per degree)
    double numax, Htot;             // AppWidth v1 (id 9)
    int32_t do_amp;
    int32_t use_ratios;             // 1: heights = H*ratios (variants 0, 1); 0: heights per m from params
};

static_assert(sizeof(TmChain) % sizeof(double) == 0, "TmChain is copied as doubles");

struct TmMultFull {
    int32_t n, l, ncomp;
    int32_t idx_f;                  // params index of the mode frequency
    int32_t idx_h;                  // params index of the height (variants 0,1) / base index of m-heights (variant 2, l>0)
    int32_t width_kind;             // 0: |params[idx_w0]|   1: |lin_interpol|   2: AppWidth v1   3: AppWidth v2
    int32_t idx_w0, idx_w1;         // width nodes (kind 0: idx_w0 only)
    int32_t idx_F0, idx_F1;         // l=0 frequency nodes bracketing f (kind 1)
    double f, W, Wraw;              // W = |Wraw|
    double slope;                   // kind 1: dW/df
    double H;                       // H_l (variants 0,1)
    double f_s1, f_s2, f_s, f_s_win;
    double Q[TM_MAXM], c[TM_MAXM];  // nu_m = f (1 + eta Q_m) + m f_s + c_m a3
    double nu[TM_MAXM], h[TM_MAXM];
    int32_t imin, imax, status;
};

static_assert(sizeof(TmMultFull) % sizeof(double) == 0, "TmMultFull is copied as doubles");

#ifdef TM_SU_TRACE_FINE
__shared__ unsigned long long g_fine[8];
#define FINE_TS(i) do { if (threadIdx.x == 0) g_fine[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FINE_TS(i) do { } while (0)
#endif
__device__ __forceinline__ double tm_ipow(double b, int e)
{
    double r = 1.0;
    for (int i = 0; i < e; i++) r = r * b;
    return r;
}

__device__ __forceinline__ int tm_factorial(int n)
{
    // n <= 6 on this path (l <= 3); a table avoids the serial multiply loops of function_rot.cpp:99-106
    const int t = (n <= 1) ? 1 : (n == 2) ? 2 : (n == 3) ? 6 : (n == 4) ? 24 : (n == 5) ? 120 : 720;
    return t;
}

// combi(n, r) = n! / (n-r)! / r! of function_rot.cpp:95-97 for n <= 6, without integer divisions by variables
__device__ __forceinline__ int tm_combi(int n, int r)
{
    if (r > n - r) r = n - r;
    if (r <= 0) return (r == 0) ? 1 : 0;
    if (r == 1) return n;
    if (r == 2) return n * (n - 1) / 2;
    return n * (n - 1) * (n - 2) / 6;
}

// d^l_{m1,m2}(beta) and its derivative with respect to beta (radians); function_rot.cpp:81-93
__device__ inline double tm_dmm(int l, int m1, int m2, double beta, double *dbeta)
{
    const double cb = cos(beta / 2.), sb = sin(beta / 2.);
    double sum = 0.0, dsum = 0.0;
    for (int s = 0; s <= l - m1; s++) {
        const int ec = 2 * s + m1 + m2, es = 2 * l - 2 * s - m1 - m2;
        double coef = (double)tm_combi(l + m2, l - m1 - s) * (double)tm_combi(l - m2, s);   // function_rot.cpp:85
        if ((l - m1 - s) & 1) coef = -coef;
        sum = sum + coef * tm_ipow(cb, ec) * tm_ipow(sb, es);
        // d/dbeta [c^ec s^es] = 0.5 * (es c^(ec+1) s^(es-1) - ec c^(ec-1) s^(es+1))
        double d = 0.0;
        if (es > 0) d += 0.5 * es * tm_ipow(cb, ec + 1) * tm_ipow(sb, es - 1);
        if (ec > 0) d -= 0.5 * ec * tm_ipow(cb, ec - 1) * tm_ipow(sb, es + 1);
        dsum = dsum + coef * d;
    }
    const double nrm = sqrt((double)(tm_factorial(l + m1) * tm_factorial(l - m1))) /
                       sqrt((double)(tm_factorial(l + m2) * tm_factorial(l - m2)));
    if (dbeta) *dbeta = dsum * nrm;
    return sum * nrm;
}

// amplitude_ratio(l, beta_deg): out[m+l] = d^l_{|m|,0}(beta)^2; dout = d out / d beta_deg
__device__ inline void tm_ratios(int l, double beta_deg, double *out, double *dout)
{
    const double PI = 3.141592653589793238462643;
    const double angle = PI * beta_deg / 180.;
    for (int am = 0; am <= l; am++) {
        double dv;
        const double v = tm_dmm(l, am, 0, angle, &dv);
        const double r = v * v, dr = 2.0 * v * dv * (PI / 180.);
        out[l + am] = r; out[l - am] = r;
        dout[l + am] = dr; dout[l - am] = dr;
    }
}

// interpol.cpp:13-55; also returns the node indices used and the slope
__device__ inline double tm_lin_interpol(const double *x, const double *y, int Nx, double x_int,
                                         int *i0, int *i1, double *slope)
{
    int i = 0;
    double a = 0, b = 0;
    int n0 = 0, n1 = 1;
    if (x_int >= x[0] && x_int <= x[Nx - 1]) {
        // The reference walks i = 0, 1, ... while (x_int < x[i] || x_int > x[i+1]) && i < Nx-2: the first bracket that
        // holds x_int, or Nx-2.  Same result from nine nodes fetched at a time (a lane runs this alone: one LDS round
        // trip per step of the walk was 40 % of the multiplet's derivation).
        i = Nx - 2;
        bool found = false;
        for (int base = 0; base < Nx - 1 && !found; base += 8) {
            double xv[9];
#pragma unroll
            for (int q = 0; q < 9; q++) xv[q] = x[(base + q < Nx - 1) ? base + q : Nx - 1];
#pragma unroll
            for (int q = 7; q >= 0; q--)
                if (base + q <= Nx - 2 && !(x_int < xv[q] || x_int > xv[q + 1])) { i = base + q; found = true; }
        }
        a = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        b = y[i] - a * x[i];
        n0 = i; n1 = i + 1;
    }
    if (x_int < x[0]) {
        a = (y[1] - y[0]) / (x[1] - x[0]);
        b = y[0] - a * x[0];
        n0 = 0; n1 = 1;
    }
    if (x_int > x[Nx - 1]) {
        a = (y[Nx - 1] - y[Nx - 2]) / (x[Nx - 1] - x[Nx - 2]);
        b = y[Nx - 2] - a * x[Nx - 2];
        n0 = Nx - 2; n1 = Nx - 1;
    }
    *i0 = n0; *i1 = n1; *slope = a;
    return a * x_int + b;
}

// build_lorentzian.cpp:377-427.  Returns 0, or 2 where the reference exits (empty window).
__device__ inline int tm_window(const TmLayout &L, double fc_l, double f_s, double gamma_l, int l, double c,
                                int32_t *imin_out, int32_t *imax_out)
{
    double pmin = __builtin_nan(""), pmax = __builtin_nan("");
    const double step = L.step;
    if (gamma_l >= 1 && f_s >= 1) {
        if (l != 0) { pmin = fc_l - c * (l * f_s + gamma_l); pmax = fc_l + c * (l * f_s + gamma_l); }
        else        { pmin = fc_l - c * gamma_l * 2.2;       pmax = fc_l + c * gamma_l * 2.2; }
    }
    if (gamma_l <= 1 && f_s >= 1) {
        if (l != 0) { pmin = fc_l - c * (l * f_s + 1); pmax = fc_l + c * (l * f_s + 1); }
        else        { pmin = fc_l - c * 2.2;           pmax = fc_l + c * 2.2; }
    }
    if (gamma_l >= 1 && f_s <= 1) {
        if (l != 0) { pmin = fc_l - c * (l + gamma_l);   pmax = fc_l + c * (l + gamma_l); }
        else        { pmin = fc_l - c * 2.2 * gamma_l;   pmax = fc_l + c * 2.2 * gamma_l; }
    }
    if (gamma_l <= 1 && f_s <= 1) {
        if (l != 0) { pmin = fc_l - c * (l + 1); pmax = fc_l + c * (l + 1); }
        else        { pmin = fc_l - c * 2.2;     pmax = fc_l + c * 2.2; }
    }
    if ((pmax - step) < L.x0) pmax = L.x0 + c;
    if ((pmin + step) >= L.xlast) pmin = L.xlast - c;
    double fmin_ = floor((pmin - L.x0) / step);
    double fmax_ = ceil((pmax - L.x0) / step);
    if (!(fmin_ == fmin_) || !(fmax_ == fmax_)) { *imin_out = 0; *imax_out = 0; return 2; }
    const double hi = (double)L.Nx;
    if (fmin_ < 0.0) fmin_ = 0.0;
    if (fmin_ > hi) fmin_ = hi;
    if (fmax_ < 0.0) fmax_ = 0.0;   // (imax < 0 gives imax-imin <= 0 just the same)
    if (fmax_ > hi) fmax_ = hi;
    const int32_t imin = (int32_t)fmin_, imax = (int32_t)fmax_;
    if (imax - imin <= 0) { *imin_out = 0; *imax_out = 0; return 2; }
    *imin_out = imin; *imax_out = imax;
    return 0;
}

// Chain-level scalars the multiplets need: everything but the inclination and the m-ratio tables (those are derived on
// another wave meanwhile, tm_derive_chain_tables).  p = this chain's params row.
__device__ inline void tm_derive_chain_scalars(const TmLayout &L, const double *p, TmChain &C)
{
    const int id = L.model_case;
    const int s = L.s, q = L.q;
    C.a1 = 0.0; C.numax = 0.0; C.Htot = 0.0;
    C.Vl[0] = 1.0; C.Vl[1] = 0.0; C.Vl[2] = 0.0; C.Vl[3] = 0.0;
    C.trunc_c = p[q + L.Ninc];
    C.do_amp = (p[q + L.Ninc + 1] != 0.0) ? 1 : 0;
    C.eta = p[s + 1];
    C.a3 = p[s + 2];
    C.asym = p[s + 5];
    C.use_ratios = (L.variant != 2) ? 1 : 0;

    if (id == 2 || id == 9 || id == 10 || id == 11) C.a1 = p[s + 3] * p[s + 3] + p[s + 4] * p[s + 4];
    if (id == 3 || id == 12 || id == 13 || id == 14) C.a1 = fabs(p[s]);

    if (L.family == TM_FAM_GLOBAL && id != 13) {
        for (int l = 1; l <= L.lmax; l++) C.Vl[l] = fabs(p[L.Nmax + l - 1]);
    }
    if (id == 9) {
        // models.cpp:1372-1390
        double numax = 0., Htot = 0.;
        for (int n = 0; n < L.Nmax; n++) {
            numax = numax + p[n] * p[L.Nmax + L.lmax + n];
            Htot = Htot + p[n];
            for (int l = 1; l <= L.lmax; l++) {
                numax = numax + p[n] * C.Vl[l] * p[L.off_f[l] + n];
                Htot = Htot + p[n] * C.Vl[l];
            }
        }
        C.Htot = Htot;
        C.numax = numax / Htot;
    }
}

// The stellar inclination in degrees (0 where the model has none)
__device__ inline double tm_chain_inc(const TmLayout &L, const double *p)
{
    const double PI_L = 3.141592653589793238462643383279502884; // the reference's pi is long double; fp64 on the device
    const int id = L.model_case;
    if (id == 2 || id == 9 || id == 10 || id == 11) {
        const double inc = atan(p[L.s + 4] / p[L.s + 3]);
        return (inc * 180.) / PI_L;
    }
    if (id == 3 || id == 6 || id == 7 || id == 8) return p[L.q];
    return 0.0;
}

// Inclination and m-height ratio tables, by ONE WAVE (all 64 lanes call this; contains wave-level synchronization only):
// tables cleared, then either read from the inclination block (id 12, models.cpp:1010-1030) or derived from the
// inclination, nine (l, |m|) entries on lanes 0..8 in parallel.
__device__ inline void tm_derive_chain_tables(const TmLayout &L, const double *p, TmChain &C, int lane)
{
    const double inc = tm_chain_inc(L, p);            // every lane the same value: no exchange needed
#ifdef TM_SU_TRACE_FINE
    if (threadIdx.x == 128) g_fine[6] = __builtin_amdgcn_s_memtime();
#endif
    if (lane < 4 * TM_MAXM) { (&C.ratios[0][0])[lane] = 0.0; (&C.dratios[0][0])[lane] = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const int q = L.q;
        C.inc = inc;
        C.ratios[0][0] = 1.0;
        if (L.model_case == 12) {
            C.ratios[1][0] = fabs(p[q + 1]); C.ratios[1][1] = fabs(p[q + 0]); C.ratios[1][2] = fabs(p[q + 1]);
            C.ratios[2][0] = fabs(p[q + 4]); C.ratios[2][1] = fabs(p[q + 3]); C.ratios[2][2] = fabs(p[q + 2]);
            C.ratios[2][3] = fabs(p[q + 3]); C.ratios[2][4] = fabs(p[q + 4]);
            C.ratios[3][0] = fabs(p[q + 8]); C.ratios[3][1] = fabs(p[q + 7]); C.ratios[3][2] = fabs(p[q + 6]);
            C.ratios[3][3] = fabs(p[q + 5]); C.ratios[3][4] = fabs(p[q + 6]); C.ratios[3][5] = fabs(p[q + 7]);
            C.ratios[3][6] = fabs(p[q + 8]);
        }
    }
    if (L.model_case == 12 || L.variant == 2) return;      // (wave-uniform: every lane leaves or none)
    // d^l_{|m|,0}(beta) of function_rot.cpp:81-93 is a sum of up to four terms: four lanes per (l, |m|) entry compute one
    // term each -- the same expressions as tm_dmm -- and the entry's first lane adds them in the order of the
    // reference's loop (a lane alone took 7 600 cycles for the sum, the longest stage of the setup kernel).
    const int item = lane >> 2, s_ = lane & 3;             // items 0..8: (l, |m|) = (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) .. (3,3)
    const int l = (item < 2) ? 1 : (item < 5) ? 2 : 3;
    const int am = (item < 2) ? item : (item < 5) ? item - 2 : (item < 9) ? item - 5 : 0;
    const bool need = (item < 9) && ((L.family == TM_FAM_GLOBAL) ? (L.lmax >= l) : (L.Nfl[l] >= 1));
    const double PI = 3.141592653589793238462643;
    const double angle = PI * inc / 180.;
#ifdef TM_SU_TRACE_FINE
    if (threadIdx.x == 128) g_fine[7] = __builtin_amdgcn_s_memtime();
#endif
    const double cb = cos(angle / 2.), sb = sin(angle / 2.);
    const bool live = s_ <= l - am;                        // this lane's term exists
    const int ec = 2 * s_ + am, es = 2 * l - 2 * s_ - am;
    double coef = (double)tm_combi(l, l - am - s_) * (double)tm_combi(l, s_);   // function_rot.cpp:85 with m2 = 0
    if ((l - am - s_) & 1) coef = -coef;
    double term = coef * tm_ipow(cb, ec) * tm_ipow(sb, es);
    // d/dbeta [c^ec s^es] = 0.5 * (es c^(ec+1) s^(es-1) - ec c^(ec-1) s^(es+1))
    double d = 0.0;
    if (es > 0) d += 0.5 * es * tm_ipow(cb, ec + 1) * tm_ipow(sb, es - 1);
    if (ec > 0) d -= 0.5 * ec * tm_ipow(cb, ec - 1) * tm_ipow(sb, es + 1);
    double dterm = coef * d;
    if (!live) { term = 0.0; dterm = 0.0; }
    const double t1 = __shfl_down(term, 1, 4), t2 = __shfl_down(term, 2, 4), t3 = __shfl_down(term, 3, 4);
    const double d1 = __shfl_down(dterm, 1, 4), d2 = __shfl_down(dterm, 2, 4), d3 = __shfl_down(dterm, 3, 4);
    if (!need || s_ != 0) return;
    double sum = 0.0 + term, dsum = 0.0 + dterm;           // the loop's running sums, s = 0 .. l - |m|
    if (l - am >= 1) { sum = sum + t1; dsum = dsum + d1; }
    if (l - am >= 2) { sum = sum + t2; dsum = dsum + d2; }
    if (l - am >= 3) { sum = sum + t3; dsum = dsum + d3; }
    const double nrm = sqrt((double)(tm_factorial(l + am) * tm_factorial(l - am))) /
                       sqrt((double)(tm_factorial(l) * tm_factorial(l)));
    const double v = sum * nrm, dv = dsum * nrm;
    const double r = v * v, dr = 2.0 * v * dv * (PI / 180.);
    C.ratios[l][l + am] = r; C.ratios[l][l - am] = r;
    C.dratios[l][l + am] = dr; C.dratios[l][l - am] = dr;
}

// (n, l) of multiplet j and the offset of degree l's block in local layouts
__device__ __forceinline__ void tm_mult_nl(const TmLayout &L, int j, int *n, int *l, int *off)
{
    if (L.family == TM_FAM_GLOBAL) {
        *n = j / (L.lmax + 1);
        *l = j % (L.lmax + 1);
        *off = 0;
    } else {
        int o = 0, ll = 0, jj = j;
        while (ll < 3 && jj >= L.Nfl[ll]) { jj -= L.Nfl[ll]; o += L.Nfl[ll]; ll++; }
        *n = jj; *l = ll; *off = o;
    }
}

// Everything about multiplet j of the chain whose params row is p, EXCEPT the heights of variants 0 and 1, which
// need the m-ratios (C.ratios): tm_mult_apply_ratios completes the record.  Needs C's scalars only, so a workgroup
// can derive the ratios on another wave meanwhile.
__device__ inline void tm_derive_mult_pre(const TmLayout &L, const TmChain &C, const double *p, int j, TmMultFull &M)
{
    const double PI_L = 3.141592653589793238462643383279502884; // long double in the reference; fp64 on the device
    const int id = L.model_case;
    int n, l, off;
    tm_mult_nl(L, j, &n, &l, &off);
    M.n = n; M.l = l; M.ncomp = 2 * l + 1;
    M.idx_w0 = M.idx_w1 = M.idx_F0 = M.idx_F1 = -1;
    M.slope = 0.0;
    M.status = 0;

    FINE_TS(0);
    // ---- frequency ----
    M.idx_f = L.off_f[l] + n;
    M.f = p[M.idx_f];

    // ---- splitting ----
    M.f_s1 = C.a1; M.f_s2 = C.a1;
    if (id == 6) { M.f_s1 = fabs(p[L.s]); M.f_s2 = fabs(p[L.s + 6]); }
    if (id == 7) { M.f_s1 = fabs(p[L.s + 6 + n]); M.f_s2 = M.f_s1; }
    if (id == 8) { M.f_s1 = fabs(p[L.s + 6 + n]); M.f_s2 = fabs(p[L.s + 6 + L.Nmax + n]); }
    if (L.variant == 1) {
        M.f_s = (l == 1) ? M.f_s1 : (l == 2) ? M.f_s2 : (M.f_s1 + M.f_s2) / 2.;
        M.f_s_win = (M.f_s1 + M.f_s2) / 2.;   // switch without break, build_lorentzian.cpp:269-278
    } else {
        M.f_s = M.f_s1;
        M.f_s_win = M.f_s1;
    }

    FINE_TS(1);
    // ---- width ----
    if (L.family == TM_FAM_LOCAL) {
        M.width_kind = 0;
        M.idx_w0 = L.w + off + n;
        M.Wraw = p[M.idx_w0];
    } else if (id == 9) {
        M.width_kind = 2;
        const int w = L.w;
        double lnGamma0 = p[w + 1] * log(M.f / C.numax) + log(p[w + 2]);
        double e = 2. * log(M.f / p[w + 0]) / log(p[w + 3] / C.numax);
        double lnLorentz = -log(p[w + 4]) / (1. + e * e);
        M.Wraw = exp(lnGamma0 + lnLorentz);
    } else if (id == 10) {
        M.width_kind = 3;
        const int w = L.w;
        double lnGamma0 = p[w + 2] * log(M.f / p[w + 0]) + log(p[w + 3]);
        double e = 2. * log(M.f / p[w + 1]) / log(p[w + 4] / p[w + 0]);
        double lnLorentz = -log(p[w + 5]) / (1. + e * e);
        M.Wraw = exp(lnGamma0 + lnLorentz);
    } else if (l == 0) {
        M.width_kind = 0;
        M.idx_w0 = L.w + n;
        M.Wraw = p[M.idx_w0];
    } else {
        M.width_kind = 1;
        int i0, i1;
        M.Wraw = tm_lin_interpol(p + L.Nmax + L.lmax, p + L.w, L.Nfl[0], M.f, &i0, &i1, &M.slope);
        M.idx_w0 = L.w + i0; M.idx_w1 = L.w + i1;
        M.idx_F0 = L.Nmax + L.lmax + i0; M.idx_F1 = L.Nmax + L.lmax + i1;
    }
    M.W = (M.width_kind == 2 || M.width_kind == 3) ? M.Wraw : fabs(M.Wraw);

    FINE_TS(2);
    // ---- heights ----
    M.H = 0.0;
#pragma unroll
    for (int k = 0; k < TM_MAXM; k++) M.h[k] = 0.0;
    if (L.variant != 2) {
        // variants 0 and 1: H_l times the m-ratios
        M.idx_h = (L.family == TM_FAM_LOCAL) ? (off + n) : n;
        const double pn = p[M.idx_h];
        if (l == 0 || L.family == TM_FAM_LOCAL) {
            M.H = C.do_amp ? fabs(pn / (PI_L * M.W)) : fabs(pn);
        } else {
            M.H = C.do_amp ? fabs(pn / (PI_L * M.W)) * C.Vl[l] : fabs(pn * C.Vl[l]);
        }
        // (the products with the m-ratios are applied by tm_mult_apply_ratios, once the ratios exist)
    } else {
        // variant 2: heights per |m| straight from params (ids 13, 14)
        if (l == 0) {
            M.idx_h = n;
            M.h[0] = C.do_amp ? fabs(p[n] / (PI_L * M.W)) : fabs(p[n]);
        } else {
            M.idx_h = ((L.family == TM_FAM_LOCAL) ? off : L.q) + (l + 1) * n;
            const double den = PI_L * M.W;
#pragma unroll
            for (int k = 0; k < TM_MAXM; k++) {
                if (k < M.ncomp) {
                    int am = k - l; if (am < 0) am = -am;
                    double v = p[M.idx_h + am];
                    if (C.do_amp) v = v / den;
                    M.h[k] = fabs(v);
                }
            }
        }
    }

    FINE_TS(3);
    // ---- component frequencies ----  build_lorentzian.cpp:74-91 / :28-48
    // (fixed trip count + predicate: the record stays in registers, the seven components are independent chains)
#pragma unroll
    for (int k = 0; k < TM_MAXM; k++) {
        const int m = k - l;
        double Qlm = 0.0, clm = 0.0, nu = 0.0;
        if (k < M.ncomp) {
            if (l != 0) {
                // Q_lm = (l(l+1) - 3 m^2) / ((2l-1)(2l+3)) and, for l = 2, c_lm = (5 m^3 - 17 m) / 3: quotients of small
                // integers, the same correctly rounded doubles whether divided at run time or folded by the compiler (k is a
                // constant in each copy of this unrolled loop; only l is not)
                const int k1 = k - 1, k2 = k - 2, k3 = k - 3;         // m for l = 1, 2, 3
                const double q1 = (double)(2 - 3 * k1 * k1) / 5.0, q2 = (double)(6 - 3 * k2 * k2) / 21.0, q3 = (double)(12 - 3 * k3 * k3) / 45.0;
                const double c2 = (5. * (double)(k2 * k2 * k2) - 17. * k2) / 3.;
                Qlm = (l == 1) ? q1 : (l == 2) ? q2 : q3;
                if (l == 1) clm = (L.variant == 1) ? 0.0 : (double)m;
                if (l == 2) clm = c2;
                if (l == 3) clm = 0.0;
                nu = M.f * (1. + C.eta * Qlm) + m * M.f_s + clm * C.a3;
            } else {
                nu = M.f;
            }
        }
        M.nu[k] = nu; M.Q[k] = Qlm; M.c[k] = clm;
    }

    FINE_TS(4);
    // ---- truncation window ----
    M.status = tm_window(L, M.f, M.f_s_win, M.W, l, C.trunc_c, &M.imin, &M.imax);
    FINE_TS(5);
}

// h_m = H_l times the m-ratio (variants 0 and 1; variant 2 read its heights from params already)
__device__ __forceinline__ void tm_mult_apply_ratios(const TmLayout &L, const TmChain &C, TmMultFull &M)
{
    if (L.variant != 2) {
#pragma unroll
        for (int k = 0; k < TM_MAXM; k++) M.h[k] = (k < M.ncomp) ? M.H * C.ratios[M.l][k] : 0.0;
    }
}

__device__ inline void tm_derive_mult(const TmLayout &L, const TmChain &C, const double *p, int j, TmMultFull &M)
{
    tm_derive_mult_pre(L, C, p, j, M);
    tm_mult_apply_ratios(L, C, M);
}
