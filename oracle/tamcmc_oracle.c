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

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
