/*
 * tamcmc_oracle.c -- CPU restatement of the TAMCMC hot path (model -> chi(2,2p) log-likelihood).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported CPU baseline.  The shipped path is the HIP library (tamcmc-c-_amd/csrc).
 *
 * PARITY PIN STATUS: "parity unpinned" for the spectrum model and the likelihood.
 *   The reference (OthmanB/TAMCMC-C, /root/reference) holds no golden vector, known-answer test or
 *   fixture for its model/likelihood functions, and it cannot be built in this
 *   image (every TU needs <Eigen/Dense>; the program needs Boost + gnuplot; none is installed, and
 *   stand-in headers are not allowed).  This file therefore follows the reference source as text,
 *   function by function, each block citing the file:line it restates.  What IS pinned:
 *     - orc_amplitude_ratio(): against the closed-form Wigner d^l_{m0}(beta)^2 expressions and the
 *       three reference values recorded in SURVEY.md App. D (tests/test_oracle.py).
 *   Everything else is checked through properties (window truncation limits, linearity in heights,
 *   agreement of independent code paths) -- see tests/.
 *   The gradient (orc_grad_analytic, end of this file) has no reference counterpart at all (MALA.cpp:18,317-333):
 *   it restates SURVEY.md App. D and is pinned, entry by entry, against finite differences of THIS file's
 *   log-likelihood (tests/test_oracle_grad.py) -- "parity unpinned" with respect to the reference by construction.
 *
 * Arithmetic notes (all fp64 like the reference):
 *   - Build with -ffp-contract=off: the reference is built with plain -O3 (CMakeLists.txt:13-36),
 *     no FMA contraction, so every a*b+c below rounds twice like the Eigen expressions do.
 *   - Eigen reductions (likelihoods.cpp:23) have an unspecified (vectorised, pairwise) order;
 *     here they accumulate in long double, which is at least as accurate as any fp64 order
 *     (differences <~1e-13 relative on 1e5 bins).
 *   - "VectorXd / long double" promotes the scalar to double first in Eigen; mirrored where the
 *     reference does it (models.cpp:1201,1217,1235,1907,1922,1939).
 *   - Where the reference calls exit() from inside the path, the functions here return a status.
 *
 * Status codes: 0 ok; 2 empty truncation window (reference: exit(EXIT_FAILURE),
 * build_lorentzian.cpp:428-443); 3 model id disabled in the reference (ids 4, 5: models.cpp:692-724,
 * model_def.cpp:231-237); 4 unknown id (model_def.cpp:266-285); 5 bad layout for this restatement.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_EMPTY_WINDOW 2
#define ORC_MODEL_DISABLED 3
#define ORC_UNKNOWN_MODEL 4
#define ORC_BAD_LAYOUT 5

static const long double ORC_PI_L = 3.141592653589793238462643383279502884L; /* models.cpp:490 */

/* ------------------------------------------------------------------------------------------------
 * function_rot.cpp
 * ---------------------------------------------------------------------------------------------- */

/* function_rot.cpp:99-106 -- long product returned as int */
static int orc_factorial(int n)
{
    long f = 1;
    for (long i = 1; i <= n; i++) f = f * i;
    return (int)f;
}

/* function_rot.cpp:95-97 -- integer division, result converted to double */
static double orc_combi(int n, int r)
{
    return (double)(orc_factorial(n) / orc_factorial(n - r) / orc_factorial(r));
}

/* function_rot.cpp:81-93 */
static double orc_dmm(int l, int m1, int m2, double beta)
{
    double sum = 0, var = 0;
    for (long s = 0; s <= l - m1; s++) {
        var = orc_combi(l + m2, (int)(l - m1 - s)) * orc_combi(l - m2, (int)s) * pow(-1, (double)(l - m1 - s));
        var = var * pow(cos(beta / 2.), (double)(2 * s + m1 + m2)) * pow(sin(beta / 2.), (double)(2 * l - 2 * s - m1 - m2));
        sum = sum + var;
    }
    sum = sum * sqrt((double)(orc_factorial(l + m1) * orc_factorial(l - m1)));
    sum = sum / sqrt((double)(orc_factorial(l + m2) * orc_factorial(l - m2)));
    return sum;
}

/* function_rot.cpp:49-79 -- full (2l+1)x(2l+1) matrix, filled in the reference's four passes */
static void orc_function_rot(int l, double beta, double *mat /* dim*dim, row-major mat[i*dim+j] */)
{
    const int dim = 2 * l + 1;
    for (int i = 0; i < dim * dim; i++) mat[i] = 0.0;
    for (int i = 0; i <= l; i++)
        for (int j = -i; j <= i; j++)
            mat[(i + l) * dim + (j + l)] = orc_dmm(l, i, j, beta);
    for (int i = -l; i <= 0; i++)
        for (int j = i; j <= -i; j++)
            mat[(i + l) * dim + (j + l)] = mat[(-i + l) * dim + (-j + l)] * pow(-1, (double)(i - j));
    for (int j = 0; j <= l; j++)
        for (int i = -j; i <= j; i++)
            mat[(i + l) * dim + (j + l)] = orc_dmm(l, j, i, -beta);
    for (int j = -l; j <= 0; j++)
        for (int i = j; i <= -j; i++)
            mat[(i + l) * dim + (j + l)] = mat[(-i + l) * dim + (-j + l)] * pow(-1, (double)(i - j));
}

/* function_rot.cpp:20-47 -- squared column l of the rotation matrix; beta in degrees; out[2l+1] */
void orc_amplitude_ratio(int l, double beta, double *out)
{
    const int dim = 2 * l + 1;
    const double PI = 3.141592653589793238462643;
    double angle = PI * beta / 180.;
    double mat[49];
    orc_function_rot(l, angle, mat);
    for (int i = 0; i < dim; i++) {
        double v = mat[i * dim + l];
        out[i] = v * v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * interpol.cpp:13-55
 * ---------------------------------------------------------------------------------------------- */
double orc_lin_interpol(const double *x, const double *y, long Nx, double x_int)
{
    long i = 0;
    double a = 0, b = 0;
    if (x_int >= x[0] && x_int <= x[Nx - 1]) {
        while ((x_int < x[i] || x_int > x[i + 1])) i = i + 1;
        if (i == 0 && (x_int < x[i] || x_int > x[i + 1])) i = i + 1;
        a = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        b = y[i] - a * x[i];
    }
    if (x_int < x[0]) {
        a = (y[1] - y[0]) / (x[1] - x[0]);
        b = y[0] - a * x[0];
    }
    if (x_int > x[Nx - 1]) {
        a = (y[Nx - 1] - y[Nx - 2]) / (x[Nx - 1] - x[Nx - 2]);
        b = y[Nx - 2] - a * x[Nx - 2];
    }
    return a * x_int + b;
}

/* ------------------------------------------------------------------------------------------------
 * build_lorentzian.cpp
 * ---------------------------------------------------------------------------------------------- */

/* One multiplet over the window x_l[0..Nxl): result[] is zeroed then accumulates m=-l..l in order.
 * variant 0: build_l_mode_a1etaa3      (build_lorentzian.cpp:61-102)   heights = H_l*V[m+l], one f_s
 * variant 1: build_l_mode_a1l_etaa3    (build_lorentzian.cpp:15-59)    f_s per degree, clm(l=1)=0
 * variant 2: build_l_mode_a1etaa3_v2   (build_lorentzian.cpp:154-205)  heights = H_lm[m+l] (H_l unused) */
static void orc_build_l_mode(int variant, const double *x_l, long Nxl, double H_l, double fc_l,
                             double f_s1, double f_s2, double eta, double a3, double asym,
                             double gamma_l, int l, const double *V, double *result)
{
    double Qlm = 0, clm = 0, f_s = f_s1;
    const double g2 = pow(gamma_l, 2);
    for (long i = 0; i < Nxl; i++) result[i] = 0.0;
    for (int m = -l; m <= l; m++) {
        double nu;
        if (l != 0) {
            Qlm = (l * (l + 1) - 3 * pow((double)m, 2)) / ((2 * l - 1) * (2 * l + 3));
            if (variant == 1) {
                if (l == 1) { clm = 0; f_s = f_s1; }
                if (l == 2) { clm = (5 * pow((double)m, 3) - 17 * m) / 3.; f_s = f_s2; }
                if (l == 3) { clm = 0; f_s = (f_s1 + f_s2) / 2.; }
            } else {
                if (l == 1) clm = m;
                if (l == 2) clm = (5 * pow((double)m, 3) - 17 * m) / 3.;
                if (l == 3) clm = 0;
            }
            nu = fc_l * (1. + eta * Qlm) + m * f_s + clm * a3;
        } else {
            nu = fc_l;
        }
        const double hv = (variant == 2) ? V[m + l] : H_l * V[m + l];
        if (asym == 0) {
            for (long i = 0; i < Nxl; i++) {
                double d = x_l[i] - nu;
                double profile = d * d;
                profile = 4 * profile / g2;
                result[i] = result[i] + hv * (1.0 / (1.0 + profile));
            }
        } else {
            const double c = 0.5 * gamma_l * asym / fc_l;
            const double c2 = c * c;
            for (long i = 0; i < Nxl; i++) {
                double d = x_l[i] - nu;
                double profile = d * d;
                profile = 4 * profile / g2;
                double a = 1.0 + asym * (x_l[i] / fc_l - 1.0);
                double asymetry = a * a + c2;
                result[i] = result[i] + hv * (asymetry * (1.0 / (1.0 + profile)));
            }
        }
    }
}

/* Truncation window.  build_lorentzian.cpp:377-427 (identical text in :279-329, :474-524, :686-736).
 * Returns ORC_EMPTY_WINDOW where the reference prints and exit(EXIT_FAILURE)s. */
static int orc_window(const double *x, long Nx, double fc_l, double f_s, double gamma_l, int l,
                      double step, double c, long *imin_out, long *imax_out)
{
    double pmin = NAN, pmax = NAN; /* uninitialised in the reference when no branch fires (NaN inputs) */
    long imin, imax;
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
    if ((pmax - step) < x[0]) pmax = x[0] + c;
    if ((pmin + step) >= x[Nx - 1]) pmin = x[Nx - 1] - c;
    {
        double fmin = floor((pmin - x[0]) / step);
        double fmax = ceil((pmax - x[0]) / step);
        /* the reference converts to long without a range check; clamp first so the conversion is defined */
        if (!(fmin == fmin) || !(fmax == fmax)) return ORC_EMPTY_WINDOW;
        if (fmin < -4.0e18) fmin = -4.0e18;
        if (fmin > 4.0e18) fmin = 4.0e18;
        if (fmax < -4.0e18) fmax = -4.0e18;
        if (fmax > 4.0e18) fmax = 4.0e18;
        imin = (long)fmin;
        imax = (long)fmax;
    }
    if (imin < 0) imin = 0;
    if (imax > Nx) imax = Nx;
    if (imax - imin <= 0) return ORC_EMPTY_WINDOW;
    *imin_out = imin;
    *imax_out = imax;
    return ORC_OK;
}

/* optimum_lorentzian_calc_*: window, multiplet on the window, y.segment += m0.
 * variant 0: _a1etaa3 (build_lorentzian.cpp:365-460); 2: _a1etaa3_v2 (:462-557);
 * variant 1: _a1l_etaa3 (:258-362) -- its switch(l) has no break, so the WINDOW always uses
 *            f_s=(f_s1+f_s2)/2 whatever l is (:269-278), while the profile uses the per-degree f_s. */
static int orc_optimum_lorentzian_calc(int variant, const double *x, double *y, long Nx, double H_l,
                                       double fc_l, double f_s1, double f_s2, double eta, double a3,
                                       double asym, double gamma_l, int l, const double *V,
                                       double step, double c, double *scratch)
{
    long imin, imax;
    double f_s_win = (variant == 1) ? (f_s1 + f_s2) / 2. : f_s1;
    int st = orc_window(x, Nx, fc_l, f_s_win, gamma_l, l, step, c, &imin, &imax);
    if (st != ORC_OK) return st;
    orc_build_l_mode(variant, x + imin, imax - imin, H_l, fc_l, f_s1, f_s2, eta, a3, asym, gamma_l, l, V, scratch);
    for (long i = imin; i < imax; i++) y[i] = y[i] + scratch[i - imin];
    return ORC_OK;
}

/* exported for tests: window of one multiplet */
int orc_truncation_window(const double *x, long Nx, double fc_l, double f_s, double gamma_l, int l,
                          double c, long *imin, long *imax)
{
    return orc_window(x, Nx, fc_l, f_s, gamma_l, l, x[1] - x[0], c, imin, imax);
}

/* ------------------------------------------------------------------------------------------------
 * noise_models.cpp
 * ---------------------------------------------------------------------------------------------- */

/* noise_models.cpp:17-41 */
static void orc_harvey_like(const double *noise_params, long Nnoise, const double *x, double *y, long Nx, int Nharvey)
{
    int cpt = 0;
    const double white = noise_params[Nnoise - 1];
    for (long k = 0; k < Nharvey; k++) {
        if (noise_params[cpt + 1] != 0) {
            const double s = (1e-3) * noise_params[cpt + 1];
            const double p = noise_params[cpt + 2];
            const double H = noise_params[cpt];
            for (long i = 0; i < Nx; i++) {
                double tmp = pow(s * x[i], p);
                tmp = H * (1.0 / (tmp + 1.0));
                y[i] = y[i] + tmp;
            }
        }
        cpt = cpt + 3;
    }
    for (long i = 0; i < Nx; i++) y[i] = y[i] + white;
}

/* noise_models.cpp:43-69 -- only reachable from the disabled model id 4 and from
 * model_Harvey1985_Gaussian (models.cpp:1994, not wired into model_def.cpp). Kept for completeness. */
static void orc_harvey1985(const double *noise_params, long Nnoise, const double *x, double *y, long Nx, int Nharvey)
{
    const long double pi = 3.141592653589793238L;
    int cpt = 0;
    const double white = noise_params[Nnoise - 1];
    for (long k = 0; k < Nharvey; k++) {
        if (noise_params[cpt + 1] != 0) {
            const double s = (double)((1e-3) * 2 * pi * noise_params[cpt + 1]);
            const double p = noise_params[cpt + 2];
            const double HT = noise_params[cpt] * noise_params[cpt + 1];
            for (long i = 0; i < Nx; i++) {
                double tmp = pow(s * x[i], p);
                tmp = HT * (1.0 / (tmp + 1.0));
                y[i] = y[i] + tmp;
            }
        }
        cpt = cpt + 3;
    }
    for (long i = 0; i < Nx; i++) y[i] = y[i] + white;
}

void orc_noise_harvey1985(const double *noise_params_abs, long Nnoise, const double *x, double *y, long Nx, int Nharvey)
{
    orc_harvey1985(noise_params_abs, Nnoise, x, y, Nx, Nharvey);
}

/* ------------------------------------------------------------------------------------------------
 * models.cpp
 * ---------------------------------------------------------------------------------------------- */

/* H for the amplitude form, l=0: std::abs(params[n]/(pi*W)) with long double pi (models.cpp:575) */
static double orc_h_amp(double p, double W)
{
    return (double)fabsl((long double)p / (ORC_PI_L * (long double)W));
}
/* l>0: std::abs(params[n]/(pi*W))*V evaluated in long double, rounded on assignment (models.cpp:587) */
static double orc_h_amp_v(double p, double W, double V)
{
    return (double)(fabsl((long double)p / (ORC_PI_L * (long double)W)) * (long double)V);
}

/* Global models sharing one body.  flavour:
 *   2  model_MS_Global_a1etaa3_HarveyLike            models.cpp:480-637
 *   3  ..._HarveyLike_Classic                        models.cpp:803-946
 *   6  model_MS_Global_a1l_etaa3_HarveyLike          models.cpp:15-167
 *   7  model_MS_Global_a1n_etaa3_HarveyLike          models.cpp:169-322
 *   8  model_MS_Global_a1nl_etaa3_HarveyLike         models.cpp:325-478
 *   9  ..._AppWidth_HarveyLike_v1                    models.cpp:1295-1493
 *   10 ..._AppWidth_HarveyLike_v2                    models.cpp:1496-1674
 *   12 ..._HarveyLike_Classic_v2                     models.cpp:952-1113
 *   13 ..._HarveyLike_Classic_v3                     models.cpp:1121-1290 */
static int orc_model_global(int flavour, const double *params, const int *plength, const double *x, long Nx, double *model_final)
{
    const double step = x[1] - x[0];
    const int Nmax = plength[0], lmax = plength[1];
    const int Nfl0 = plength[2], Nfl1 = plength[3], Nfl2 = plength[4], Nfl3 = plength[5];
    const int Nsplit = plength[6], Nwidth = plength[7], Nnoise = plength[8], Ninc = plength[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int s = Nmax + lmax + Nf;          /* splitting block */
    const int w = s + Nsplit;                /* widths */
    const int z = w + Nwidth;                /* noise */
    const int q = z + Nnoise;                /* inclination block */
    const double trunc_c = params[q + Ninc];
    const int do_amp = (params[q + Ninc + 1] != 0.0); /* const bool do_amp = params[...] */
    double ratios_l0[1] = {1.0}, ratios_l1[3], ratios_l2[5], ratios_l3[7];
    double Vl1 = 0, Vl2 = 0, Vl3 = 0, a1 = 0, eta, a3, asym, inclination = 0, numax = 0;
    const double *fl0_all = params + Nmax + lmax;
    const double *Wl0_all = params + w;
    const int a1l_family = (flavour == 6 || flavour == 7 || flavour == 8);
    const int variant = a1l_family ? 1 : (flavour == 13 ? 2 : 0);
    int st = ORC_OK;

    if (lmax < 0 || lmax > 3) return ORC_BAD_LAYOUT;

    if (flavour == 2 || flavour == 9 || flavour == 10) {
        /* models.cpp:528-531 (id 2 computes a1 first; 9/10 compute inclination first: same values) */
        a1 = pow(params[s + 3], 2) + pow(params[s + 4], 2);
        inclination = atan(params[s + 4] / params[s + 3]);
        inclination = (double)((long double)(inclination * 180.) / ORC_PI_L);
    } else if (flavour == 3 || a1l_family) {
        inclination = params[z + Nnoise];     /* models.cpp:848, :64, :216, :371 */
    }
    if (flavour == 3 || flavour == 12 || flavour == 13) a1 = fabs(params[s]); /* models.cpp:870,1035,1168 */

    if (flavour != 13) {
        if (lmax >= 1) Vl1 = fabs(params[Nmax]);
        if (lmax >= 2) Vl2 = fabs(params[Nmax + 1]);
        if (lmax >= 3) Vl3 = fabs(params[Nmax + 2]);
    }
    if (flavour == 12) {
        /* models.cpp:1010-1030 -- m-heights given directly, symmetric in m */
        ratios_l1[0] = fabs(params[z + Nnoise + 1]); ratios_l1[1] = fabs(params[z + Nnoise]); ratios_l1[2] = fabs(params[z + Nnoise + 1]);
        ratios_l2[0] = fabs(params[z + Nnoise + 4]); ratios_l2[1] = fabs(params[z + Nnoise + 3]); ratios_l2[2] = fabs(params[z + Nnoise + 2]);
        ratios_l2[3] = fabs(params[z + Nnoise + 3]); ratios_l2[4] = fabs(params[z + Nnoise + 4]);
        ratios_l3[0] = fabs(params[z + Nnoise + 8]); ratios_l3[1] = fabs(params[z + Nnoise + 7]); ratios_l3[2] = fabs(params[z + Nnoise + 6]);
        ratios_l3[3] = fabs(params[z + Nnoise + 5]); ratios_l3[4] = fabs(params[z + Nnoise + 6]); ratios_l3[5] = fabs(params[z + Nnoise + 7]);
        ratios_l3[6] = fabs(params[z + Nnoise + 8]);
    } else if (flavour != 13) {
        if (lmax >= 1) orc_amplitude_ratio(1, inclination, ratios_l1);
        if (lmax >= 2) orc_amplitude_ratio(2, inclination, ratios_l2);
        if (lmax >= 3) orc_amplitude_ratio(3, inclination, ratios_l3);
    }

    eta = params[s + 1];
    a3 = params[s + 2];
    asym = params[s + 5];

    if (flavour == 9) {
        /* models.cpp:1372-1390 -- numax = height-weighted mean frequency (the reference also prints it) */
        double Htot = 0.;
        numax = 0.;
        for (long n = 0; n < Nmax; n++) {
            numax = numax + params[n] * params[Nmax + lmax + n];
            Htot = Htot + params[n];
            if (lmax >= 1) { numax = numax + params[n] * Vl1 * params[Nmax + lmax + Nfl0 + n]; Htot = Htot + params[n] * Vl1; }
            if (lmax >= 2) { numax = numax + params[n] * Vl2 * params[Nmax + lmax + Nfl0 + Nfl1 + n]; Htot = Htot + params[n] * Vl2; }
            if (lmax >= 3) { numax = numax + params[n] * Vl3 * params[Nmax + lmax + Nfl0 + Nfl1 + Nfl2 + n]; Htot = Htot + params[n] * Vl3; }
        }
        numax = numax / Htot;
    }

    for (long i = 0; i < Nx; i++) model_final[i] = 0.0;
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(Nx > 0 ? Nx : 1));
    if (!scratch) return ORC_BAD_LAYOUT;

    for (long n = 0; n < Nmax && st == ORC_OK; n++) {
        double f_s1 = a1, f_s2 = a1;
        if (flavour == 6) { f_s1 = fabs(params[s]); f_s2 = fabs(params[s + 6]); }                         /* models.cpp:87-88 */
        if (flavour == 7) { f_s1 = fabs(params[s + 6 + n]); f_s2 = f_s1; }                                /* models.cpp:240-241,266-267 */
        if (flavour == 8) { f_s1 = fabs(params[s + 6 + n]); f_s2 = fabs(params[s + 6 + Nmax + n]); }      /* models.cpp:395-396,422-423 */

        for (int l = 0; l <= lmax && st == ORC_OK; l++) {
            double fl, Wl, Hl = 0, Hlm[7];
            const double *V;
            int off_f = Nmax + lmax;
            if (l >= 1) off_f += Nfl0;
            if (l >= 2) off_f += Nfl1;
            if (l >= 3) off_f += Nfl2;
            fl = (l == 0) ? fl0_all[n] : params[off_f + n];

            if (flavour == 9) {
                /* models.cpp:1408-1411 (and :1425-1428, :1441-1444, :1457-1460) */
                double lnGamma0 = params[w + 1] * log(fl / numax) + log(params[w + 2]);
                double e = 2. * log(fl / params[w + 0]) / log(params[w + 3] / numax);
                double lnLorentz = -log(params[w + 4]) / (1. + pow(e, 2));
                Wl = exp(lnGamma0 + lnLorentz);
            } else if (flavour == 10) {
                /* models.cpp:1590-1594 */
                double lnGamma0 = params[w + 2] * log(fl / params[w + 0]) + log(params[w + 3]);
                double e = 2. * log(fl / params[w + 1]) / log(params[w + 4] / params[w + 0]);
                double lnLorentz = -log(params[w + 5]) / (1. + pow(e, 2));
                Wl = exp(lnGamma0 + lnLorentz);
            } else if (l == 0) {
                Wl = fabs(Wl0_all[n]);                                   /* models.cpp:572 */
            } else {
                Wl = fabs(orc_lin_interpol(fl0_all, Wl0_all, Nfl0, fl)); /* models.cpp:584; x.size() there is Nfl0 (:554) */
            }

            if (flavour == 13) {
                /* models.cpp:1184-1238 -- heights per (n,l,|m|) read from the inclination block */
                if (l == 0) {
                    Hlm[0] = do_amp ? orc_h_amp(params[n], Wl) : fabs(params[n]);
                } else {
                    const long pos0 = (long)(l + 1) * n;
                    const int dim = 2 * l + 1;
                    for (int k = 0; k < dim; k++) {
                        int am = k - l; if (am < 0) am = -am;
                        Hlm[k] = params[z + Nnoise + pos0 + am];
                    }
                    if (do_amp) {
                        const double den = (double)(ORC_PI_L * (long double)Wl); /* scalar promoted to double by Eigen */
                        for (int k = 0; k < dim; k++) Hlm[k] = Hlm[k] / den;
                    }
                    for (int k = 0; k < dim; k++) Hlm[k] = fabs(Hlm[k]);
                }
                V = Hlm;
            } else {
                double Vl = (l == 1) ? Vl1 : (l == 2) ? Vl2 : Vl3;
                if (l == 0) Hl = do_amp ? orc_h_amp(params[n], Wl) : fabs(params[n]);            /* models.cpp:574-578 */
                else        Hl = do_amp ? orc_h_amp_v(params[n], Wl, Vl) : fabs(params[n] * Vl); /* models.cpp:585-590 */
                V = (l == 0) ? ratios_l0 : (l == 1) ? ratios_l1 : (l == 2) ? ratios_l2 : ratios_l3;
            }
            st = orc_optimum_lorentzian_calc(variant, x, model_final, Nx, Hl, fl, f_s1, f_s2, eta, a3, asym, Wl, l, V, step, trunc_c, scratch);
        }
    }
    free(scratch);
    if (st != ORC_OK) return st;

    {
        /* models.cpp:623-631 */
        double noise_abs[64];
        if (Nnoise < 1 || Nnoise > 64) return ORC_BAD_LAYOUT;
        for (int k = 0; k < Nnoise; k++) noise_abs[k] = fabs(params[z + k]);
        orc_harvey_like(noise_abs, Nnoise, x, model_final, Nx, (Nnoise - 1) / 3);
    }
    return ORC_OK;
}

/* Local models.  flavour 11: model_MS_local_basic (models.cpp:1683-1829);
 *                flavour 14: model_MS_local_Hnlm  (models.cpp:1832-1963). */
static int orc_model_local(int flavour, const double *params, const int *plength, const double *x, long Nx, double *model_final)
{
    const double step = x[1] - x[0];
    const int Nmax = plength[0], Nvis = plength[1];
    const int Nfl[4] = {plength[2], plength[3], plength[4], plength[5]};
    const int Nsplit = plength[6], Nwidth = plength[7], Nnoise = plength[8], Ninc = plength[9];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    const int s = Nmax + Nvis + Nf;
    const int w = s + Nsplit;
    const int z = w + Nwidth;
    const double trunc_c = params[z + Nnoise + Ninc];
    const int do_amp = (params[z + Nnoise + Ninc + 1] != 0.0);
    double ratios[4][7] = {{1.0}};
    double a1, eta, a3, asym, inclination;
    int st = ORC_OK;

    if (flavour == 11) {
        /* models.cpp:1726-1728 */
        inclination = atan(params[s + 4] / params[s + 3]);
        inclination = (double)((long double)(inclination * 180.) / ORC_PI_L);
        a1 = pow(params[s + 3], 2) + pow(params[s + 4], 2);
        for (int l = 1; l <= 3; l++)
            if (Nfl[l] >= 1) orc_amplitude_ratio(l, inclination, ratios[l]);
    } else {
        a1 = fabs(params[s]); /* models.cpp:1877 */
    }
    eta = params[s + 1];
    a3 = params[s + 2];
    asym = params[s + 5];

    for (long i = 0; i < Nx; i++) model_final[i] = 0.0;
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(Nx > 0 ? Nx : 1));
    if (!scratch) return ORC_BAD_LAYOUT;

    int off = 0;   /* running offset of this degree's modes inside the f / W blocks; the reference uses the
                      SAME offset (Nfl0, Nfl0+Nfl1, ...) for the heights block, also for local_Hnlm where a
                      degree-l mode owns l+1 heights (models.cpp:1903,1916,1931) -- kept literally */
    for (int l = 0; l <= 3 && st == ORC_OK; l++) {
        for (long n = 0; n < Nfl[l] && st == ORC_OK; n++) {
            const double fl = params[Nmax + Nvis + off + n];
            const double Wl = fabs(params[s + Nsplit + off + n]);
            if (flavour == 11) {
                const double p = params[off + n];
                const double Hl = do_amp ? orc_h_amp(p, Wl) : fabs(p); /* models.cpp:1765-1769 etc. */
                st = orc_optimum_lorentzian_calc(0, x, model_final, Nx, Hl, fl, a1, a1, eta, a3, asym, Wl, l, ratios[l], step, trunc_c, scratch);
            } else {
                double Hlm[7];
                const int dim = 2 * l + 1;
                if (l == 0) {
                    Hlm[0] = do_amp ? orc_h_amp(params[n], Wl) : fabs(params[n]); /* models.cpp:1891-1895 */
                } else {
                    const long pos0 = (long)(l + 1) * n;                          /* models.cpp:1902,1915,1930 */
                    for (int k = 0; k < dim; k++) {
                        int am = k - l; if (am < 0) am = -am;
                        Hlm[k] = params[off + pos0 + am];
                    }
                    if (do_amp) {
                        const double den = (double)(ORC_PI_L * (long double)Wl);
                        for (int k = 0; k < dim; k++) Hlm[k] = Hlm[k] / den;
                    }
                    for (int k = 0; k < dim; k++) Hlm[k] = fabs(Hlm[k]);
                }
                st = orc_optimum_lorentzian_calc(2, x, model_final, Nx, 0.0, fl, a1, a1, eta, a3, asym, Wl, l, Hlm, step, trunc_c, scratch);
            }
        }
        off += Nfl[l];
    }
    free(scratch);
    if (st != ORC_OK) return st;
    {
        /* models.cpp:1817-1824 -- Nharvey forced to 0: white noise only (last noise element) */
        double noise_abs[64];
        if (Nnoise < 1 || Nnoise > 64) return ORC_BAD_LAYOUT;
        for (int k = 0; k < Nnoise; k++) noise_abs[k] = fabs(params[z + k]);
        orc_harvey_like(noise_abs, Nnoise, x, model_final, Nx, 0);
    }
    return ORC_OK;
}

/* models.cpp:2021-2034 */
static void orc_model_Test_Gaussian(const double *params, const double *x, long Nx, double *m)
{
    const double s2 = pow(params[1], 2);
    for (long i = 0; i < Nx; i++) {
        double d = x[i] - params[2];
        double v = -0.5 * (d * d) / s2;
        v = params[0] * exp(v);
        m[i] = v + params[3];
    }
}

/* models.cpp:1968-1992 */
static void orc_model_Harvey_Gaussian(const double *params, const double *x, long Nx, double *m)
{
    const double s2 = pow(fabs(params[1]), 2);
    double noise_abs[4];
    for (long i = 0; i < Nx; i++) {
        double d = x[i] - params[2];
        double v = -0.5 * (d * d) / s2;
        m[i] = fabs(params[0]) * exp(v);
    }
    for (int k = 0; k < 4; k++) noise_abs[k] = fabs(params[3 + k]);
    orc_harvey_like(noise_abs, 4, x, m, Nx, 1);
}

/* Model_def::call_model, model_def.cpp:210-289 */
int orc_model(int model_case, const double *params, const int *plength, const double *x, long Nx, double *out)
{
    switch (model_case) {
    case 0: orc_model_Test_Gaussian(params, x, Nx, out); return ORC_OK;
    case 1: orc_model_Harvey_Gaussian(params, x, Nx, out); return ORC_OK;
    case 2: case 3: case 6: case 7: case 8: case 9: case 10: case 12: case 13:
        return orc_model_global(model_case, params, plength, x, Nx, out);
    case 4: case 5:
        return ORC_MODEL_DISABLED;
    case 11: case 14:
        return orc_model_local(model_case, params, plength, x, Nx, out);
    default:
        return ORC_UNKNOWN_MODEL;
    }
}

/* ------------------------------------------------------------------------------------------------
 * likelihoods.cpp
 * ---------------------------------------------------------------------------------------------- */

/* likelihoods.cpp:17-28; p arrives as `long` (truncation of likelihood_params, model_def.cpp:300-301) */
long double orc_likelihood_chi22p_l(const double *y, const double *model, long Nx, long p)
{
    long double s1 = 0.0L, s2 = 0.0L, f;
    for (long i = 0; i < Nx; i++) s1 += (long double)(y[i] * (1.0 / model[i]));
    for (long i = 0; i < Nx; i++) s2 += (long double)log(model[i]);
    f = (long double)((double)s1 + (double)s2); /* the Eigen expression is a double */
    f = -p * f;
    return f;
}

double orc_likelihood_chi22p(const double *y, const double *model, long Nx, double p)
{
    return (double)orc_likelihood_chi22p_l(y, model, Nx, (long)p);
}

/* likelihoods.cpp:31-39 */
long double orc_likelihood_chi_square_l(const double *y, const double *model, const double *sigma, long Nx)
{
    long double s = 0.0L;
    for (long i = 0; i < Nx; i++) {
        double d = y[i] - model[i];
        s += (long double)((d * d) * (1.0 / (sigma[i] * sigma[i])));
    }
    return -(long double)(double)s;
}

double orc_likelihood_chi_square(const double *y, const double *model, const double *sigma, long Nx)
{
    return (double)orc_likelihood_chi_square_l(y, model, sigma, Nx);
}

/* ------------------------------------------------------------------------------------------------
 * model_def.cpp:291-320, 358-367 (model -> logL/T for one chain), batched over chains like the
 * OpenMP loop of MALA.cpp:632-655.
 * ---------------------------------------------------------------------------------------------- */
static int orc_chain_logL(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                          long Nx, const double *x, const double *y, const double *sigma_y,
                          const double *params, double Tcoef, double *model_buf, double *logL)
{
    int st = orc_model(model_case, params, plength, x, Nx, model_buf);
    long double L;
    if (st != ORC_OK) { *logL = NAN; return st; }
    if (likelihood_case == 0) L = orc_likelihood_chi22p_l(y, model_buf, Nx, (long)likelihood_p);
    else                      L = orc_likelihood_chi_square_l(y, model_buf, sigma_y, Nx);
    *logL = (double)(L / (long double)Tcoef);
    return ORC_OK;
}

/* logL[m] = tempered log-likelihood of chain m; status[m] as above, 1 added for NaN logL.
 * model_out (may be NULL): Nchains x Nx row-major models. nthreads<=0: OpenMP default. */
int orc_generate_batch(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                       long Nx, const double *x, const double *y, const double *sigma_y,
                       int Nchains, int Nparams, const double *params, const double *Tcoefs,
                       double *logL, int *status, double *model_out, int nthreads)
{
    int rc = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int m = 0; m < Nchains; m++) {
        double *buf = model_out ? model_out + (size_t)m * (size_t)Nx : (double *)malloc(sizeof(double) * (size_t)Nx);
        int st = orc_chain_logL(model_case, likelihood_case, likelihood_p, plength, Nx, x, y, sigma_y,
                                params + (size_t)m * (size_t)Nparams, Tcoefs[m], buf, &logL[m]);
        if (st == ORC_OK && !(logL[m] == logL[m])) st = 1;
        if (status) status[m] = st;
        if (!model_out) free(buf);
    }
    (void)rc;
    return 0;
}

/* Central finite differences of the tempered logL with respect to params[index_to_relax[k]].
 * NOT in the reference (its "MALA" has no gradient: MALA.cpp:18,317-333) -- this is the only
 * reference-derived check available for the gradient the HIP path adds.
 * h_k = rel_step * max(|theta_k|, 1).  Uses Richardson extrapolation of (h, h/2) central
 * differences, so the truncation error is O(h^4). */
int orc_grad_fd(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                long Nx, const double *x, const double *y, const double *sigma_y,
                int Nparams, const double *params, double Tcoef,
                int Nvars, const int *index_to_relax, double rel_step, double *grad)
{
    double *p = (double *)malloc(sizeof(double) * (size_t)Nparams);
    double *buf = (double *)malloc(sizeof(double) * (size_t)Nx);
    int rc = ORC_OK;
    if (!p || !buf) { free(p); free(buf); return ORC_BAD_LAYOUT; }
    for (int k = 0; k < Nvars; k++) {
        const int j = index_to_relax[k];
        const double th = params[j];
        const double h = rel_step * (fabs(th) > 1.0 ? fabs(th) : 1.0);
        double L[4];
        const double offs[4] = {+1.0, -1.0, +0.5, -0.5};
        for (int e = 0; e < 4; e++) {
            memcpy(p, params, sizeof(double) * (size_t)Nparams);
            p[j] = th + offs[e] * h;
            int st = orc_chain_logL(model_case, likelihood_case, likelihood_p, plength, Nx, x, y, sigma_y, p, Tcoef, buf, &L[e]);
            if (st != ORC_OK) rc = st;
        }
        {
            const double d1 = (L[0] - L[1]) / (2.0 * h);
            const double d2 = (L[2] - L[3]) / h;
            grad[k] = (4.0 * d2 - d1) / 3.0;
        }
    }
    free(p);
    free(buf);
    return rc;
}

/* orc_grad_fd with the likelihood sums kept in long double end to end (the reference's likelihood rounds
 * its two sums to double, 1e-16 |logL| ~ 1e-11 absolute, which a step of 1e-6 turns into 1e-5 of gradient
 * noise; here the noise is that of the per-bin fp64 model values only, ~1e-14 absolute).  Test helper for
 * pinning orc_grad_analytic entry by entry; same Richardson scheme. */
static long double orc_chain_logL_wide(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                                       long Nx, const double *x, const double *y, const double *sigma_y,
                                       const double *params, double Tcoef, double *model_buf, int *st_out)
{
    long double s = 0.0L;
    int st = orc_model(model_case, params, plength, x, Nx, model_buf);
    *st_out = st;
    if (st != ORC_OK) return (long double)NAN;
    if (likelihood_case == 0) {
        for (long i = 0; i < Nx; i++) s += (long double)y[i] / (long double)model_buf[i] + logl((long double)model_buf[i]);
        s = -(long double)(long)likelihood_p * s;
    } else {
        for (long i = 0; i < Nx; i++) {
            const long double d = (long double)y[i] - model_buf[i];
            s -= d * d / ((long double)sigma_y[i] * sigma_y[i]);
        }
    }
    return s / (long double)Tcoef;
}

int orc_grad_fd_wide(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                     long Nx, const double *x, const double *y, const double *sigma_y,
                     int Nparams, const double *params, double Tcoef,
                     int Nvars, const int *index_to_relax, double rel_step, double *grad)
{
    int rc = ORC_OK;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < Nvars; k++) {
        double *p = (double *)malloc(sizeof(double) * (size_t)Nparams);
        double *buf = (double *)malloc(sizeof(double) * (size_t)Nx);
        const int j = index_to_relax[k];
        const double th = params[j];
        const double h = rel_step * (fabs(th) > 1.0 ? fabs(th) : 1.0);
        long double L[4], hh[4];
        if (!p || !buf) { free(p); free(buf); grad[k] = NAN;
#ifdef _OPENMP
#pragma omp atomic write
#endif
            rc = ORC_BAD_LAYOUT;
            continue;
        }
        const double offs[4] = {+1.0, -1.0, +0.5, -0.5};
        for (int e = 0; e < 4; e++) {
            int st;
            memcpy(p, params, sizeof(double) * (size_t)Nparams);
            p[j] = th + offs[e] * h;
            hh[e] = (long double)p[j] - (long double)th;          /* the step actually taken */
            L[e] = orc_chain_logL_wide(model_case, likelihood_case, likelihood_p, plength, Nx, x, y, sigma_y, p, Tcoef, buf, &st);
            if (st != ORC_OK) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                rc = st;
            }
        }
        {
            const long double d1 = (L[0] - L[1]) / (hh[0] - hh[1]);
            const long double d2 = (L[2] - L[3]) / (hh[2] - hh[3]);
            grad[k] = (double)((4.0L * d2 - d1) / 3.0L);
        }
        free(p);
        free(buf);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Analytic gradient of the tempered log-likelihood, d(logL/T)/dparams  (SURVEY.md App. D).
 *
 * NOT in the reference (MALA.cpp:18,317-333: D_MALA() returns zeros), so there is no reference line to
 * follow: App. D is the specification, and this is its second, independent statement -- written from the
 * formulas and from the forward functions ABOVE (same windows [imin,imax), same abs() placement, same
 * lin_interpol bracket, same amplitude form), not from the HIP kernels it checks.  Shape of the code:
 *   w_i = d(logL/T)/dM_i from the oracle's own model;
 *   per multiplet and component, plain loops over the window bins accumulate in long double the adjoints
 *   of the component's (height, centre nu_m) and of the multiplet's (width, asymmetry, mode frequency
 *   inside the asymmetry factor);  then the chain rule of App. D, one statement per dependency.
 * Every accumulator is a pair (value, sum of |terms|): the second member is the exact conditioning of
 * that gradient entry, i.e. what any fp64 evaluation of the same sum loses to cancellation
 * (error <~ eps * sum|terms|); the tests use it as the floor of their per-entry comparison.
 * The truncation window is held fixed (the model is piecewise smooth; App. D caveat).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { long double v, a; } OrcAdj;

static void adj_term(OrcAdj *d, long double t) { d->v += t; d->a += fabsl(t); }
static void adj_axpy(OrcAdj *d, long double coef, OrcAdj s) { d->v += coef * s.v; d->a += fabsl(coef) * s.a; }
static long double orc_sgn(double v) { return (v < 0) ? -1.0L : 1.0L; }

typedef struct {
    OrcAdj hv[7], nu[7], gamma, fA, asym;
    double Q[7], clm[7];
    int fs_sel;          /* which splitting the profile used: 0 f_s1, 1 f_s2, 2 (f_s1+f_s2)/2 */
} OrcMultAdj;

/* Adjoint of one windowed multiplet (forward: orc_optimum_lorentzian_calc + orc_build_l_mode above). */
static int orc_mult_adjoint(int variant, const double *x, long Nx, const long double *wgt, double H_l,
                            double fc_l, double f_s1, double f_s2, double eta, double a3, double asym,
                            double gamma_l, int l, const double *V, double step, double c, OrcMultAdj *o)
{
    long imin, imax;
    const double f_s_win = (variant == 1) ? (f_s1 + f_s2) / 2. : f_s1;
    int st = orc_window(x, Nx, fc_l, f_s_win, gamma_l, l, step, c, &imin, &imax);
    double Qlm = 0, clm = 0, f_s = f_s1;
    const long double g = gamma_l, g2 = (long double)gamma_l * gamma_l, fc = fc_l, al = asym;
    memset(o, 0, sizeof(*o));
    if (st != ORC_OK) return st;
    for (int m = -l; m <= l; m++) {
        double nu;
        if (l != 0) {
            Qlm = (l * (l + 1) - 3 * pow((double)m, 2)) / ((2 * l - 1) * (2 * l + 3));
            if (variant == 1) {
                if (l == 1) { clm = 0; f_s = f_s1; o->fs_sel = 0; }
                if (l == 2) { clm = (5 * pow((double)m, 3) - 17 * m) / 3.; f_s = f_s2; o->fs_sel = 1; }
                if (l == 3) { clm = 0; f_s = (f_s1 + f_s2) / 2.; o->fs_sel = 2; }
            } else {
                if (l == 1) clm = m;
                if (l == 2) clm = (5 * pow((double)m, 3) - 17 * m) / 3.;
                if (l == 3) clm = 0;
            }
            nu = fc_l * (1. + eta * Qlm) + m * f_s + clm * a3;
        } else {
            nu = fc_l;
        }
        o->Q[m + l] = (l != 0) ? Qlm : 0.0;
        o->clm[m + l] = (l != 0) ? clm : 0.0;
        const long double hv = (variant == 2) ? V[m + l] : (long double)H_l * V[m + l];
        const long double cc = 0.5L * g * al / fc;
        for (long i = imin; i < imax; i++) {
            const long double xi = x[i], d = xi - (long double)nu, wi = wgt[i];
            const long double r = 1.0L / (1.0L + 4.0L * d * d / g2);       /* 1/D */
            const long double rel = xi / fc - 1.0L;
            const long double a = 1.0L + al * rel;
            const long double A = a * a + cc * cc;                           /* = 1 when asym == 0 */
            const long double dA_dal = 2.0L * a * rel + al * g2 / (2.0L * fc * fc);
            const long double dA_dg = al * al * g / (2.0L * fc * fc);
            const long double dA_df = -2.0L * a * al * xi / (fc * fc) - al * al * g2 / (2.0L * fc * fc * fc);
            adj_term(&o->hv[m + l], wi * A * r);
            adj_term(&o->nu[m + l], wi * hv * A * (8.0L * d / g2) * r * r);
            adj_term(&o->gamma, wi * hv * (A * (8.0L * d * d / (g2 * g)) * r * r + dA_dg * r));
            adj_term(&o->asym, wi * hv * dA_dal * r);
            adj_term(&o->fA, wi * hv * dA_df * r);
        }
    }
    return ORC_OK;
}

/* d/d(beta) of the squared Wigner column that amplitude_ratio() evaluates numerically; closed forms of
 * SURVEY.md App. D (beta in radians, index m+l).  The forward values of these forms are pinned against
 * orc_amplitude_ratio in tests/test_oracle.py. */
void orc_amplitude_ratio_closed(int l, double beta_rad, double *V, double *dV)
{
    const long double b = beta_rad, c = cosl(b), s = sinl(b);
    long double v[7] = {0}, d[7] = {0};
    if (l == 0) { v[0] = 1; d[0] = 0; }
    if (l == 1) {
        v[1] = c * c;                   d[1] = -2 * c * s;
        v[0] = v[2] = 0.5L * s * s;     d[0] = d[2] = c * s;
    }
    if (l == 2) {
        const long double k = 3 * c * c - 1, s2 = sinl(2 * b), c2 = cosl(2 * b);
        v[2] = 0.25L * k * k;                     d[2] = -3 * k * c * s;
        v[1] = v[3] = 0.375L * s2 * s2;           d[1] = d[3] = 1.5L * s2 * c2;
        v[0] = v[4] = 0.375L * s * s * s * s;     d[0] = d[4] = 1.5L * s * s * s * c;
    }
    if (l == 3) {
        const long double k0 = 5 * cosl(3 * b) + 3 * c, dk0 = -15 * sinl(3 * b) - 3 * s;
        const long double k1 = 5 * cosl(2 * b) + 3, dk1 = -10 * sinl(2 * b);
        v[3] = k0 * k0 / 64;                                d[3] = 2 * k0 * dk0 / 64;
        v[2] = v[4] = 3 * k1 * k1 * s * s / 64;             d[2] = d[4] = 3 * (2 * k1 * dk1 * s * s + k1 * k1 * 2 * s * c) / 64;
        v[1] = v[5] = 15.0L / 8 * c * c * s * s * s * s;    d[1] = d[5] = 15.0L / 8 * (-2 * c * s * s * s * s * s + 4 * c * c * c * s * s * s);
        v[0] = v[6] = 5.0L / 16 * s * s * s * s * s * s;    d[0] = d[6] = 5.0L / 16 * 6 * s * s * s * s * s * c;
    }
    for (int i = 0; i < 2 * l + 1; i++) { if (V) V[i] = (double)v[i]; if (dV) dV[i] = (double)d[i]; }
}

/* Bracket of lin_interpol (interpol.cpp:25-54): the node pair (j, j+1) whose line is evaluated. */
static long orc_lin_interpol_bracket(const double *x, long Nx, double x_int)
{
    long i = 0, j = 0;
    if (x_int >= x[0] && x_int <= x[Nx - 1]) {
        while ((x_int < x[i] || x_int > x[i + 1])) i = i + 1;
        if (i == 0 && (x_int < x[i] || x_int > x[i + 1])) i = i + 1;
        j = i;
    }
    if (x_int < x[0]) j = 0;
    if (x_int > x[Nx - 1]) j = Nx - 2;
    return j;
}

/* Appourchaux width (models.cpp:1408-1411, 1590-1594) and its partial derivatives:
 * W = exp(A1 log(fl/numax) + log(G0) - log(AD)/(1+e^2)),  e = 2 log(fl/nuD) / log(wD/numax);
 * d[0..6] = dW/d(fl, numax, nuD, A1, G0, wD, AD). */
static void orc_app_width_d(long double fl, long double numax, long double nuD, long double A1, long double G0,
                            long double wD, long double AD, long double *d)
{
    const long double Lw = logl(wD / numax), e = 2 * logl(fl / nuD) / Lw, q = 1 + e * e;
    const long double W = expl(A1 * logl(fl / numax) + logl(G0) - logl(AD) / q);
    const long double k = logl(AD) * 2 * e / (q * q);     /* d(-log(AD)/q)/de */
    d[0] = W * (A1 / fl + k * 2 / (fl * Lw));
    d[1] = W * (-A1 / numax + k * e / (Lw * numax));
    d[2] = W * k * (-2 / (nuD * Lw));
    d[3] = W * logl(fl / numax);
    d[4] = W / G0;
    d[5] = W * k * (-e / (Lw * wD));
    d[6] = -W / (AD * q);
}

/* Harvey-like background (orc_harvey_like above) on |params[z..z+Nnoise)|. */
static void orc_noise_adjoint(const double *params, int z, int Nnoise, int Nharvey, const double *x, long Nx,
                              const long double *wgt, OrcAdj *G)
{
    for (int k = 0; k < Nharvey; k++) {
        const double H = fabs(params[z + 3 * k]), tau = fabs(params[z + 3 * k + 1]), p = fabs(params[z + 3 * k + 2]);
        if (tau != 0) {
            const long double s = (long double)((1e-3) * tau);
            OrcAdj gH = {0, 0}, gT = {0, 0}, gP = {0, 0};
            for (long i = 0; i < Nx; i++) {
                const long double sx = s * x[i], t = powl(sx, p), u = 1.0L / (1.0L + t);
                adj_term(&gH, wgt[i] * u);
                adj_term(&gT, wgt[i] * (-(long double)H * p * t * u * u / tau));
                adj_term(&gP, wgt[i] * (-(long double)H * t * logl(sx) * u * u));
            }
            adj_axpy(&G[z + 3 * k], orc_sgn(params[z + 3 * k]), gH);
            adj_axpy(&G[z + 3 * k + 1], orc_sgn(params[z + 3 * k + 1]), gT);
            adj_axpy(&G[z + 3 * k + 2], orc_sgn(params[z + 3 * k + 2]), gP);
        }
    }
    {
        OrcAdj gW = {0, 0};
        for (long i = 0; i < Nx; i++) adj_term(&gW, wgt[i]);
        adj_axpy(&G[z + Nnoise - 1], orc_sgn(params[z + Nnoise - 1]), gW);
    }
}

/* Heights of the amplitude form: H = |p/(pi W)| * V (V >= 0 or 1): dH/dp, dH/dW, dH/dV. */
static void orc_h_amp_d(double p, double W, double V, long double *dp, long double *dW, long double *dV)
{
    const long double u = (long double)p / (ORC_PI_L * (long double)W);
    *dp = orc_sgn((double)u) * V / (ORC_PI_L * (long double)W);
    *dW = -fabsl(u) * V / W;
    *dV = fabsl(u);
}

/* Chain rule of the global models (forward: orc_model_global above, same flavours). */
static int orc_grad_global(int flavour, const double *params, const int *plength, const double *x, long Nx,
                           const long double *wgt, OrcAdj *G)
{
    const double step = x[1] - x[0];
    const int Nmax = plength[0], lmax = plength[1];
    const int Nfl0 = plength[2], Nfl1 = plength[3], Nfl2 = plength[4], Nfl3 = plength[5];
    const int Nsplit = plength[6], Nwidth = plength[7], Nnoise = plength[8], Ninc = plength[9];
    const int Nf = Nfl0 + Nfl1 + Nfl2 + Nfl3;
    const int f0 = Nmax + lmax;
    const int s = Nmax + lmax + Nf, w = s + Nsplit, z = w + Nwidth, q = z + Nnoise;
    const double trunc_c = params[q + Ninc];
    const int do_amp = (params[q + Ninc + 1] != 0.0);
    const int a1l_family = (flavour == 6 || flavour == 7 || flavour == 8);
    const int variant = a1l_family ? 1 : (flavour == 13 ? 2 : 0);
    const int inc_from_split = (flavour == 2 || flavour == 9 || flavour == 10);
    const int off_l[4] = {f0, f0 + Nfl0, f0 + Nfl0 + Nfl1, f0 + Nfl0 + Nfl1 + Nfl2};
    double ratios[4][7] = {{1.0}}, dratios[4][7] = {{0.0}};
    double Vl[4] = {1.0, 0, 0, 0};
    double a1 = 0, inclination = 0, numax = 0, Htot = 0;
    const double eta = params[s + 1], a3 = params[s + 2], asym = params[s + 5];
    OrcAdj g_beta = {0, 0}, g_numax = {0, 0}, g_a1 = {0, 0};
    (void)Nwidth;

    if (lmax < 0 || lmax > 3) return ORC_BAD_LAYOUT;
    if (inc_from_split) {
        a1 = pow(params[s + 3], 2) + pow(params[s + 4], 2);
        inclination = atan(params[s + 4] / params[s + 3]);
        inclination = (double)((long double)(inclination * 180.) / ORC_PI_L);
    } else if (flavour == 3 || a1l_family) {
        inclination = params[q];
    }
    if (flavour == 3 || flavour == 12 || flavour == 13) a1 = fabs(params[s]);
    if (flavour != 13)
        for (int l = 1; l <= lmax; l++) Vl[l] = fabs(params[Nmax + l - 1]);
    if (flavour == 12) {
        const int base[4] = {0, q + 0, q + 2, q + 5};
        for (int l = 1; l <= 3; l++)
            for (int m = -l; m <= l; m++) ratios[l][m + l] = fabs(params[base[l] + (m < 0 ? -m : m)]);
    } else if (flavour != 13) {
        const double PI = 3.141592653589793238462643;
        for (int l = 1; l <= lmax; l++) {
            orc_amplitude_ratio(l, inclination, ratios[l]);
            orc_amplitude_ratio_closed(l, PI * inclination / 180., NULL, dratios[l]);
        }
    }
    if (flavour == 9) {
        for (long n = 0; n < Nmax; n++) {
            numax = numax + params[n] * params[f0 + n];
            Htot = Htot + params[n];
            for (int l = 1; l <= lmax; l++) { numax = numax + params[n] * Vl[l] * params[off_l[l] + n]; Htot = Htot + params[n] * Vl[l]; }
        }
        numax = numax / Htot;
    }

    for (long n = 0; n < Nmax; n++) {
        double f_s1 = a1, f_s2 = a1;
        OrcAdj g_fs1 = {0, 0}, g_fs2 = {0, 0};
        if (flavour == 6) { f_s1 = fabs(params[s]); f_s2 = fabs(params[s + 6]); }
        if (flavour == 7) { f_s1 = fabs(params[s + 6 + n]); f_s2 = f_s1; }
        if (flavour == 8) { f_s1 = fabs(params[s + 6 + n]); f_s2 = fabs(params[s + 6 + Nmax + n]); }

        for (int l = 0; l <= lmax; l++) {
            const int i_f = off_l[l] + (int)n;
            const double fl = params[i_f];
            double Wl, Hl = 0, Hlm[7];
            const double *V;
            long double wd[7];
            long jb = 0;
            double interp = 0;
            OrcMultAdj A;
            OrcAdj gW = {0, 0}, gH = {0, 0}, gVl = {0, 0}, g_fs = {0, 0};
            int st;

            if (flavour == 9) {
                double lnGamma0 = params[w + 1] * log(fl / numax) + log(params[w + 2]);
                double e = 2. * log(fl / params[w + 0]) / log(params[w + 3] / numax);
                Wl = exp(lnGamma0 - log(params[w + 4]) / (1. + pow(e, 2)));
                orc_app_width_d(fl, numax, params[w + 0], params[w + 1], params[w + 2], params[w + 3], params[w + 4], wd);
            } else if (flavour == 10) {
                double lnGamma0 = params[w + 2] * log(fl / params[w + 0]) + log(params[w + 3]);
                double e = 2. * log(fl / params[w + 1]) / log(params[w + 4] / params[w + 0]);
                Wl = exp(lnGamma0 - log(params[w + 5]) / (1. + pow(e, 2)));
                orc_app_width_d(fl, params[w + 0], params[w + 1], params[w + 2], params[w + 3], params[w + 4], params[w + 5], wd);
            } else if (l == 0) {
                Wl = fabs(params[w + n]);
            } else {
                interp = orc_lin_interpol(params + f0, params + w, Nfl0, fl);
                jb = orc_lin_interpol_bracket(params + f0, Nfl0, fl);
                Wl = fabs(interp);
            }

            if (flavour == 13) {
                if (l == 0) {
                    Hlm[0] = do_amp ? orc_h_amp(params[n], Wl) : fabs(params[n]);
                } else {
                    const long pos0 = (long)(l + 1) * n;
                    const double den = do_amp ? (double)(ORC_PI_L * (long double)Wl) : 1.0;
                    for (int k = 0; k < 2 * l + 1; k++) {
                        int am = k - l; if (am < 0) am = -am;
                        Hlm[k] = fabs(params[q + pos0 + am] / den);
                    }
                }
                V = Hlm;
            } else {
                if (l == 0) Hl = do_amp ? orc_h_amp(params[n], Wl) : fabs(params[n]);
                else        Hl = do_amp ? orc_h_amp_v(params[n], Wl, Vl[l]) : fabs(params[n] * Vl[l]);
                V = ratios[l];
            }
            st = orc_mult_adjoint(variant, x, Nx, wgt, Hl, fl, f_s1, f_s2, eta, a3, asym, Wl, l, V, step, trunc_c, &A);
            if (st != ORC_OK) return st;

            /* centres nu_m = f (1 + eta Q_lm) + m f_s + c_lm a3  (l = 0: nu = f) */
            for (int m = -l; m <= l; m++) {
                const OrcAdj gn = A.nu[m + l];
                if (l == 0) { adj_axpy(&G[i_f], 1.0L, gn); continue; }
                adj_axpy(&G[i_f], 1.0L + (long double)eta * A.Q[m + l], gn);
                adj_axpy(&G[s + 1], (long double)fl * A.Q[m + l], gn);
                adj_axpy(&G[s + 2], (long double)A.clm[m + l], gn);
                adj_axpy(&g_fs, (long double)m, gn);
            }
            if (variant == 1) {
                if (A.fs_sel == 0) adj_axpy(&g_fs1, 1.0L, g_fs);
                if (A.fs_sel == 1) adj_axpy(&g_fs2, 1.0L, g_fs);
                if (A.fs_sel == 2) { adj_axpy(&g_fs1, 0.5L, g_fs); adj_axpy(&g_fs2, 0.5L, g_fs); }
            } else {
                adj_axpy(&g_a1, 1.0L, g_fs);
            }
            adj_axpy(&G[i_f], 1.0L, A.fA);          /* the mode frequency inside the asymmetry factor */
            adj_axpy(&G[s + 5], 1.0L, A.asym);
            adj_axpy(&gW, 1.0L, A.gamma);

            /* heights */
            if (flavour == 13) {
                if (l == 0) {
                    if (do_amp) {
                        long double dp, dW, dV;
                        orc_h_amp_d(params[n], Wl, 1.0, &dp, &dW, &dV);
                        adj_axpy(&G[n], dp, A.hv[0]);
                        adj_axpy(&gW, dW, A.hv[0]);
                    } else {
                        adj_axpy(&G[n], orc_sgn(params[n]), A.hv[0]);
                    }
                } else {
                    const long pos0 = (long)(l + 1) * n;
                    for (int k = 0; k < 2 * l + 1; k++) {
                        int am = k - l; if (am < 0) am = -am;
                        const int ip = q + (int)pos0 + am;
                        if (do_amp) {
                            long double dp, dW, dV;
                            orc_h_amp_d(params[ip], Wl, 1.0, &dp, &dW, &dV);
                            adj_axpy(&G[ip], dp, A.hv[k]);
                            adj_axpy(&gW, dW, A.hv[k]);
                        } else {
                            adj_axpy(&G[ip], orc_sgn(params[ip]), A.hv[k]);
                        }
                    }
                }
            } else {
                /* h_m = H_l * V_m */
                for (int k = 0; k < 2 * l + 1; k++) {
                    adj_axpy(&gH, (long double)V[k], A.hv[k]);
                    if (l > 0) {
                        if (flavour == 12) {
                            const int base[4] = {0, q + 0, q + 2, q + 5};
                            int am = k - l; if (am < 0) am = -am;
                            adj_axpy(&G[base[l] + am], (long double)Hl * orc_sgn(params[base[l] + am]), A.hv[k]);
                        } else {
                            adj_axpy(&g_beta, (long double)Hl * dratios[l][k], A.hv[k]);
                        }
                    }
                }
                if (do_amp) {
                    long double dp, dW, dV;
                    orc_h_amp_d(params[n], Wl, Vl[l], &dp, &dW, &dV);
                    adj_axpy(&G[n], dp, gH);
                    adj_axpy(&gW, dW, gH);
                    if (l > 0) adj_axpy(&gVl, dV, gH);
                } else if (l == 0) {
                    adj_axpy(&G[n], orc_sgn(params[n]), gH);
                } else {
                    const long double sg = orc_sgn(params[n] * Vl[l]);
                    adj_axpy(&G[n], sg * Vl[l], gH);
                    adj_axpy(&gVl, sg * (long double)params[n], gH);
                }
                if (l > 0) adj_axpy(&G[Nmax + l - 1], orc_sgn(params[Nmax + l - 1]), gVl);
            }

            /* width */
            if (flavour == 9 || flavour == 10) {
                const int iw[7] = {i_f, -1, w + 0, w + 1, w + 2, w + 3, w + 4};          /* flavour 9 */
                const int jw[7] = {i_f, w + 0, w + 1, w + 2, w + 3, w + 4, w + 5};       /* flavour 10 */
                for (int k = 0; k < 7; k++) {
                    const int ip = (flavour == 9) ? iw[k] : jw[k];
                    if (ip >= 0) adj_axpy(&G[ip], wd[k], gW);
                    else         adj_axpy(&g_numax, wd[k], gW);
                }
            } else if (l == 0) {
                adj_axpy(&G[w + n], orc_sgn(params[w + n]), gW);
            } else {
                /* W = |y_j + a (f - x_j)|, a = (y_{j+1} - y_j)/(x_{j+1} - x_j); x = l=0 frequencies, y = l=0 widths (raw) */
                const long double xj = params[f0 + jb], xj1 = params[f0 + jb + 1], yj = params[w + jb], yj1 = params[w + jb + 1];
                const long double dx = xj1 - xj, a = (yj1 - yj) / dx, t = ((long double)fl - xj) / dx, sg = orc_sgn(interp);
                adj_axpy(&G[i_f], sg * a, gW);
                adj_axpy(&G[w + jb], sg * (1.0L - t), gW);
                adj_axpy(&G[w + jb + 1], sg * t, gW);
                adj_axpy(&G[f0 + jb], sg * (a * t - a), gW);
                adj_axpy(&G[f0 + jb + 1], sg * (-a * t), gW);
            }
        }
        /* per-order splittings of the a1l family */
        if (flavour == 6) { adj_axpy(&G[s], orc_sgn(params[s]), g_fs1); adj_axpy(&G[s + 6], orc_sgn(params[s + 6]), g_fs2); }
        if (flavour == 7) { adj_axpy(&G[s + 6 + n], orc_sgn(params[s + 6 + n]), g_fs1); adj_axpy(&G[s + 6 + n], orc_sgn(params[s + 6 + n]), g_fs2); }
        if (flavour == 8) { adj_axpy(&G[s + 6 + n], orc_sgn(params[s + 6 + n]), g_fs1); adj_axpy(&G[s + 6 + Nmax + n], orc_sgn(params[s + 6 + Nmax + n]), g_fs2); }
    }

    /* a1 */
    if (inc_from_split) {
        adj_axpy(&G[s + 3], 2.0L * params[s + 3], g_a1);
        adj_axpy(&G[s + 4], 2.0L * params[s + 4], g_a1);
    } else if (flavour == 3 || flavour == 12 || flavour == 13) {
        adj_axpy(&G[s], orc_sgn(params[s]), g_a1);
    }
    /* inclination: beta = atan(p4/p3) [rad], or params[q] in degrees */
    if (inc_from_split) {
        const long double p3 = params[s + 3], p4 = params[s + 4], r2 = p3 * p3 + p4 * p4;
        adj_axpy(&G[s + 3], -p4 / r2, g_beta);
        adj_axpy(&G[s + 4], p3 / r2, g_beta);
    } else if (flavour == 3 || a1l_family) {
        adj_axpy(&G[q], ORC_PI_L / 180.0L, g_beta);
    }
    /* nu_max of AppWidth_v1: height-weighted mean frequency of all modes (raw heights, |V_l|) */
    if (flavour == 9) {
        for (long n = 0; n < Nmax; n++) {
            long double fsum = params[f0 + n], vsum = 1.0L;
            for (int l = 1; l <= lmax; l++) { fsum += (long double)Vl[l] * params[off_l[l] + n]; vsum += Vl[l]; }
            adj_axpy(&G[n], (fsum - (long double)numax * vsum) / Htot, g_numax);
            adj_axpy(&G[f0 + n], (long double)params[n] / Htot, g_numax);
            for (int l = 1; l <= lmax; l++) adj_axpy(&G[off_l[l] + n], (long double)params[n] * Vl[l] / Htot, g_numax);
        }
        for (int l = 1; l <= lmax; l++) {
            long double pf = 0, ps = 0;
            for (long n = 0; n < Nmax; n++) { pf += (long double)params[n] * params[off_l[l] + n]; ps += params[n]; }
            adj_axpy(&G[Nmax + l - 1], orc_sgn(params[Nmax + l - 1]) * (pf - (long double)numax * ps) / Htot, g_numax);
        }
    }
    orc_noise_adjoint(params, z, Nnoise, (Nnoise - 1) / 3, x, Nx, wgt, G);
    return ORC_OK;
}

/* Chain rule of the local models (forward: orc_model_local above). */
static int orc_grad_local(int flavour, const double *params, const int *plength, const double *x, long Nx,
                          const long double *wgt, OrcAdj *G)
{
    const double step = x[1] - x[0];
    const int Nmax = plength[0], Nvis = plength[1];
    const int Nfl[4] = {plength[2], plength[3], plength[4], plength[5]};
    const int Nsplit = plength[6], Nwidth = plength[7], Nnoise = plength[8], Ninc = plength[9];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    const int s = Nmax + Nvis + Nf, w = s + Nsplit, z = w + Nwidth;
    const double trunc_c = params[z + Nnoise + Ninc];
    const int do_amp = (params[z + Nnoise + Ninc + 1] != 0.0);
    double ratios[4][7] = {{1.0}}, dratios[4][7] = {{0.0}};
    double a1, inclination = 0;
    const double eta = params[s + 1], a3 = params[s + 2], asym = params[s + 5];
    OrcAdj g_beta = {0, 0}, g_a1 = {0, 0};
    int off = 0;

    if (flavour == 11) {
        const double PI = 3.141592653589793238462643;
        inclination = atan(params[s + 4] / params[s + 3]);
        inclination = (double)((long double)(inclination * 180.) / ORC_PI_L);
        a1 = pow(params[s + 3], 2) + pow(params[s + 4], 2);
        for (int l = 1; l <= 3; l++)
            if (Nfl[l] >= 1) {
                orc_amplitude_ratio(l, inclination, ratios[l]);
                orc_amplitude_ratio_closed(l, PI * inclination / 180., NULL, dratios[l]);
            }
    } else {
        a1 = fabs(params[s]);
    }
    for (int l = 0; l <= 3; l++) {
        for (long n = 0; n < Nfl[l]; n++) {
            const int i_f = Nmax + Nvis + off + (int)n, i_w = w + off + (int)n;
            const double fl = params[i_f], Wl = fabs(params[i_w]);
            double Hl = 0, Hlm[7];
            int ih[7];
            OrcMultAdj A;
            OrcAdj gW = {0, 0}, g_fs = {0, 0};
            int st;
            if (flavour == 11) {
                ih[0] = off + (int)n;
                Hl = do_amp ? orc_h_amp(params[ih[0]], Wl) : fabs(params[ih[0]]);
                st = orc_mult_adjoint(0, x, Nx, wgt, Hl, fl, a1, a1, eta, a3, asym, Wl, l, ratios[l], step, trunc_c, &A);
            } else {
                const double den = do_amp ? (double)(ORC_PI_L * (long double)Wl) : 1.0;
                for (int k = 0; k < 2 * l + 1; k++) {
                    int am = k - l; if (am < 0) am = -am;
                    ih[k] = (l == 0) ? (int)n : off + (l + 1) * (int)n + am;
                    Hlm[k] = fabs(params[ih[k]] / den);
                }
                if (l == 0 && do_amp) Hlm[0] = orc_h_amp(params[n], Wl);
                st = orc_mult_adjoint(2, x, Nx, wgt, 0.0, fl, a1, a1, eta, a3, asym, Wl, l, Hlm, step, trunc_c, &A);
            }
            if (st != ORC_OK) return st;
            for (int m = -l; m <= l; m++) {
                const OrcAdj gn = A.nu[m + l];
                if (l == 0) { adj_axpy(&G[i_f], 1.0L, gn); continue; }
                adj_axpy(&G[i_f], 1.0L + (long double)eta * A.Q[m + l], gn);
                adj_axpy(&G[s + 1], (long double)fl * A.Q[m + l], gn);
                adj_axpy(&G[s + 2], (long double)A.clm[m + l], gn);
                adj_axpy(&g_fs, (long double)m, gn);
            }
            adj_axpy(&g_a1, 1.0L, g_fs);
            adj_axpy(&G[i_f], 1.0L, A.fA);
            adj_axpy(&G[s + 5], 1.0L, A.asym);
            adj_axpy(&gW, 1.0L, A.gamma);
            if (flavour == 11) {
                OrcAdj gH = {0, 0};
                for (int k = 0; k < 2 * l + 1; k++) {
                    adj_axpy(&gH, (long double)ratios[l][k], A.hv[k]);
                    if (l > 0) adj_axpy(&g_beta, (long double)Hl * dratios[l][k], A.hv[k]);
                }
                if (do_amp) {
                    long double dp, dW, dV;
                    orc_h_amp_d(params[ih[0]], Wl, 1.0, &dp, &dW, &dV);
                    adj_axpy(&G[ih[0]], dp, gH);
                    adj_axpy(&gW, dW, gH);
                } else {
                    adj_axpy(&G[ih[0]], orc_sgn(params[ih[0]]), gH);
                }
            } else {
                for (int k = 0; k < 2 * l + 1; k++) {
                    if (do_amp) {
                        long double dp, dW, dV;
                        orc_h_amp_d(params[ih[k]], Wl, 1.0, &dp, &dW, &dV);
                        adj_axpy(&G[ih[k]], dp, A.hv[k]);
                        adj_axpy(&gW, dW, A.hv[k]);
                    } else {
                        adj_axpy(&G[ih[k]], orc_sgn(params[ih[k]]), A.hv[k]);
                    }
                }
            }
            adj_axpy(&G[i_w], orc_sgn(params[i_w]), gW);
        }
        off += Nfl[l];
    }
    if (flavour == 11) {
        const long double p3 = params[s + 3], p4 = params[s + 4], r2 = p3 * p3 + p4 * p4;
        adj_axpy(&G[s + 3], 2.0L * p3, g_a1);
        adj_axpy(&G[s + 4], 2.0L * p4, g_a1);
        adj_axpy(&G[s + 3], -p4 / r2, g_beta);
        adj_axpy(&G[s + 4], p3 / r2, g_beta);
    } else {
        adj_axpy(&G[s], orc_sgn(params[s]), g_a1);
    }
    orc_noise_adjoint(params, z, Nnoise, 0, x, Nx, wgt, G);
    return ORC_OK;
}

/* Gaussian toy models (forward: orc_model_Test_Gaussian / orc_model_Harvey_Gaussian above). */
static void orc_grad_gauss(int model_case, const double *params, const double *x, long Nx, const long double *wgt, OrcAdj *G)
{
    const long double A = (model_case == 0) ? (long double)params[0] : fabsl(params[0]);
    const long double sg = (model_case == 0) ? (long double)params[1] : fabsl(params[1]);
    OrcAdj g0 = {0, 0}, g1 = {0, 0}, g2 = {0, 0};
    for (long i = 0; i < Nx; i++) {
        const long double d = (long double)x[i] - params[2], E = expl(-0.5L * d * d / (sg * sg));
        adj_term(&g0, wgt[i] * E);
        adj_term(&g1, wgt[i] * A * E * d * d / (sg * sg * sg));
        adj_term(&g2, wgt[i] * A * E * d / (sg * sg));
    }
    adj_axpy(&G[0], (model_case == 0) ? 1.0L : orc_sgn(params[0]), g0);
    adj_axpy(&G[1], (model_case == 0) ? 1.0L : orc_sgn(params[1]), g1);
    adj_axpy(&G[2], 1.0L, g2);
    if (model_case == 0) {
        OrcAdj g3 = {0, 0};
        for (long i = 0; i < Nx; i++) adj_term(&g3, wgt[i]);
        adj_axpy(&G[3], 1.0L, g3);
    } else {
        orc_noise_adjoint(params, 3, 4, 1, x, Nx, wgt, G);
    }
}

/* d(logL/T)/dparams[index_to_relax[k]] for one chain; grad_abs (may be NULL) receives sum|terms| per entry,
 * logL (may be NULL) the tempered log-likelihood of the same model evaluation. */
int orc_grad_analytic(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                      long Nx, const double *x, const double *y, const double *sigma_y,
                      int Nparams, const double *params, double Tcoef,
                      int Nvars, const int *index_to_relax, double *grad, double *grad_abs, double *logL)
{
    double *model = (double *)malloc(sizeof(double) * (size_t)(Nx > 0 ? Nx : 1));
    long double *wgt = (long double *)malloc(sizeof(long double) * (size_t)(Nx > 0 ? Nx : 1));
    OrcAdj *G = (OrcAdj *)calloc((size_t)(Nparams > 0 ? Nparams : 1), sizeof(OrcAdj));
    int st = ORC_BAD_LAYOUT;
    double L = NAN;
    if (model && wgt && G) {
        st = orc_chain_logL(model_case, likelihood_case, likelihood_p, plength, Nx, x, y, sigma_y, params, Tcoef, model, &L);
        if (st == ORC_OK) {
            const long double T = Tcoef;
            if (likelihood_case == 0) {
                const long double p = (long double)(long)likelihood_p;
                for (long i = 0; i < Nx; i++) {
                    const long double M = model[i];
                    wgt[i] = (p / T) * ((long double)y[i] / (M * M) - 1.0L / M);
                }
            } else {
                for (long i = 0; i < Nx; i++)
                    wgt[i] = 2.0L * ((long double)y[i] - model[i]) / ((long double)sigma_y[i] * sigma_y[i]) / T;
            }
            switch (model_case) {
            case 0: case 1: orc_grad_gauss(model_case, params, x, Nx, wgt, G); break;
            case 11: case 14: st = orc_grad_local(model_case, params, plength, x, Nx, wgt, G); break;
            default: st = orc_grad_global(model_case, params, plength, x, Nx, wgt, G); break;
            }
        }
    }
    for (int k = 0; k < Nvars; k++) {
        const int j = index_to_relax[k];
        const int ok = (st == ORC_OK && j >= 0 && j < Nparams);
        grad[k] = ok ? (double)G[j].v : NAN;
        if (grad_abs) grad_abs[k] = ok ? (double)G[j].a : NAN;
    }
    if (logL) *logL = L;
    free(model); free(wgt); free(G);
    return st;
}

/* The same for a batch of chains (OpenMP over chains like orc_generate_batch). */
int orc_grad_analytic_batch(int model_case, int likelihood_case, double likelihood_p, const int *plength,
                            long Nx, const double *x, const double *y, const double *sigma_y,
                            int Nchains, int Nparams, const double *params, const double *Tcoefs,
                            int Nvars, const int *index_to_relax, double *grad, double *grad_abs, double *logL,
                            int *status, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int m = 0; m < Nchains; m++) {
        int st = orc_grad_analytic(model_case, likelihood_case, likelihood_p, plength, Nx, x, y, sigma_y, Nparams,
                                   params + (size_t)m * (size_t)Nparams, Tcoefs[m], Nvars, index_to_relax,
                                   grad + (size_t)m * (size_t)Nvars, grad_abs ? grad_abs + (size_t)m * (size_t)Nvars : NULL,
                                   logL ? &logL[m] : NULL);
        if (status) status[m] = st;
    }
    return 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
