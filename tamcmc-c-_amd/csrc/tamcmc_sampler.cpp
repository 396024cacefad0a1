// tamcmc_sampler.cpp -- host-side callers of the hot path: log-priors (N1) and the batched adaptive
// Metropolis + parallel-tempering sampler (N2).  See include/tamcmc_sampler.h for the map to the
// reference.  Plain C++17, no Eigen: matrices are row-major std::vector<double>.
#include "tamcmc_sampler.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

typedef long double ld;
const ld PIl = 3.141592653589793238462643383279502884L;   // stats_dictionary.cpp:23
const ld NEG_INF = -std::numeric_limits<ld>::infinity();

// ------------------------------------------------------------------------------------------------
// N1: primitive priors, stats_dictionary.cpp:31-241 (long double like the reference)
// ------------------------------------------------------------------------------------------------
ld logP_uniform(ld b_min, ld b_max, ld x) { return ((x <= b_max) && (x >= b_min)) ? -std::log(std::fabs(b_max - b_min)) : NEG_INF; }
ld logP_uniform_abs(ld b_min, ld b_max, ld x) { return ((std::fabs(x) <= b_max) && (std::fabs(x) >= b_min)) ? -std::log(std::fabs(b_max - b_min)) : NEG_INF; }
ld logP_uniform_cos(ld b_min, ld b_max, ld x)
{
    const ld x_rad = PIl * x / 180.;
    return ((std::cos(x_rad) < b_max) && (std::cos(x_rad) > b_min)) ? -std::log(std::fabs(b_max - b_min)) : NEG_INF;
}
ld logP_gaussian(ld mean, ld sigma, ld x) { return -std::log(std::sqrt(2 * PIl) * sigma) - 0.5 * std::pow((x - mean) / sigma, (ld)2.); }
ld logP_jeffrey(ld hmin, ld hmax, ld h)
{
    if (h < hmax && h > 0) {
        const ld prior = 1. / (h + hmin), norm = std::log((hmax + hmin) / hmin);
        return std::log(prior / norm);
    }
    return NEG_INF;
}
ld logP_jeffrey_abs(ld hmin, ld hmax, ld h)
{
    if (std::fabs(h) < hmax) {
        const ld prior = 1. / (std::fabs(h) + hmin), norm = std::log((hmax + hmin) / hmin);
        return std::log(prior / norm);
    }
    return NEG_INF;
}
ld logP_uniform_gaussian(ld b_min, ld b_max, ld sigma, ld x)
{
    ld logP = std::numeric_limits<ld>::quiet_NaN();   // uninitialised in the reference when x is NaN
    if (x < b_min) logP = NEG_INF;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * std::pow((x - b_max) / sigma, (ld)2.);
    const ld C = std::log(std::fabs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * sigma);
    return logP - C;
}
ld logP_gaussian_uniform(ld b_min, ld b_max, ld sigma, ld x)
{
    ld logP = std::numeric_limits<ld>::quiet_NaN();
    if (x > b_max) logP = NEG_INF;
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x < b_min) logP = -0.5 * std::pow((x - b_min) / sigma, (ld)2.);
    const ld C = std::log(std::fabs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * sigma);
    return logP - C;
}
ld logP_gaussian_uniform_gaussian(ld b_min, ld b_max, ld sigma1, ld sigma2, ld x)
{
    ld logP = std::numeric_limits<ld>::quiet_NaN();
    if (x < b_min) logP = -0.5 * std::pow((x - b_min) / sigma1, (ld)2.);
    if ((x <= b_max) && (x >= b_min)) logP = 0;
    if (x > b_max) logP = -0.5 * std::pow((x - b_max) / sigma2, (ld)2.);
    const ld C = std::log(std::fabs(b_max - b_min) + 0.5 * std::sqrt(2 * PIl) * (sigma1 + sigma2));
    return logP - C;
}

// One primitive by its primepriors_ctrl.list id; p = column of Input_Data::priors.  *err set for ids the
// reference exits on (3 multivariate Gaussian, unknown ids), priors_calc.cpp:315-417.
ld primitive(int id, const double *p, int nrows, ld x, int *err)
{
    const ld p0 = nrows > 0 ? p[0] : 0, p1 = nrows > 1 ? p[1] : 0, p2 = nrows > 2 ? p[2] : 0, p3 = nrows > 3 ? p[3] : 0;
    switch (id) {
    case 0: return 0;
    case 1: return logP_uniform(p0, p1, x);
    case 2: return logP_gaussian(p0, p1, x);
    case 3: if (err) *err = 1; return 0;
    case 4: return logP_jeffrey(p0, p1, x);
    case 5: return logP_uniform_gaussian(p0, p1, p2, x);
    case 6: return logP_gaussian_uniform(p0, p1, p2, x);
    case 7: return logP_gaussian_uniform_gaussian(p0, p1, p2, p3, x);
    case 8: return logP_uniform_abs(p0, p1, x);
    case 9: return logP_uniform_cos(p0, p1, x);
    case 10: return logP_jeffrey_abs(p0, p1, x);
    case 11: return 0;
    default: if (err) *err = 1; return 0;
    }
}

struct PriorSpec {
    int fct = 0;                       // priors_ctrl.list id
    int Nparams = 0, nrows = 0;
    int plength[11] = {0};
    std::vector<int32_t> sw;           // priors_names_switch
    std::vector<double> pp;            // priors_params, nrows x Nparams
    double extra[4] = {0, 0, 0, 0};
};

// apply_generic_priors, priors_calc.cpp:315-417
ld apply_generic_priors(const PriorSpec &S, const double *params, int *err)
{
    ld pena = 0;
    double col[4];
    for (int i = 0; i < S.Nparams; i++) {
        for (int r = 0; r < 4; r++) col[r] = (r < S.nrows) ? S.pp[(size_t)r * S.Nparams + i] : 0.0;
        pena = pena + primitive(S.sw[i], col, S.nrows, params[i], err);
    }
    return pena;
}

// Frstder_adaptive_reggrid(y).deriv.sum(), derivatives_handler.cpp:342-367 (n >= 3)
ld first_derivative_sum(const double *y, int n)
{
    ld s = 0;
    s += y[1] - y[0];                                   // forward, 1st order, on the first two points
    for (int i = 0; i < n - 2; i++) s += (y[i + 2] - y[i]) / 2.;   // centred
    s += y[n - 1] - y[n - 2];                           // backward on the last two points
    return s;
}

// Scndder_adaptive_reggrid(y).deriv[i], derivatives_handler.cpp:398-423
double second_derivative(const double *y, int n, int i)
{
    if (i == 0) return y[2] - 2. * y[1] + y[0];
    if (i == n - 1) return y[n - 1] - 2. * y[n - 2] + y[n - 3];
    return y[i + 1] - 2. * y[i] + y[i - 1];
}

// priors_MS_Global, priors_calc.cpp:24-172
ld priors_MS_Global(const PriorSpec &S, const double *params, int *err)
{
    ld f = 0;
    const int smooth_switch = (int)S.extra[0];
    const double scoef = S.extra[1], a3ova1_limit = S.extra[2];
    const int impose_normHnlm = (int)S.extra[3];
    const int Nmax = S.plength[0], lmax = S.plength[1];
    const int Nfl[4] = {S.plength[2], S.plength[3], S.plength[4], S.plength[5]};
    const int Nsplit = S.plength[6], Nwidth = S.plength[7], Nnoise = S.plength[8];
    const int Nf = Nfl[0] + Nfl[1] + Nfl[2] + Nfl[3];
    const int s = Nmax + lmax + Nf, q = s + Nsplit + Nwidth + Nnoise, z = s + Nsplit + Nwidth;

    f = f + apply_generic_priors(S, params, err);
    for (int i = Nmax; i <= Nmax + lmax; i++)            // inclusive upper bound as in the reference (:52)
        if (params[i] < 0) f = NEG_INF;
    switch (impose_normHnlm) {
    case 1:
        f = f + logP_uniform(0, 1. + 1e-10, params[q] + 2 * params[q + 1]);
        f = f + logP_uniform(0, 1 + 1e-10, params[q + 2] + 2 * params[q + 3] + 2 * params[q + 4]);
        f = f + logP_uniform(0, 1 + 1e-10, params[q + 5] + 2 * params[q + 6] + 2 * params[q + 7] + 2 * params[q + 8]);
        break;
    case 2: if (err) *err = 1; break;                    // "YET TO BE IMPLEMENTED" + exit in the reference (:83-85)
    }
    if (Nfl[0] < 3) { if (err) *err = 1; return f; }     // the derivative helpers exit on short vectors
    const double *fl0 = params + Nmax + lmax;
    const double Dnu = (double)first_derivative_sum(fl0, Nfl[0]);   // SUM of the derivative array (:90)
    if (Nfl[0] == Nfl[2]) {
        for (int i = 0; i < Nfl[0]; i++) {
            const double d02 = params[Nmax + lmax + i] - params[Nmax + lmax + Nfl[0] + Nfl[1] + i];
            f = f + logP_gaussian_uniform(0, Dnu / 3., 0.015 * Dnu, d02);
        }
    }
    if (smooth_switch == 1) {
        int off = Nmax + lmax;
        for (int l = 0; l < 4; l++) {
            if (Nfl[l] != 0) {
                if (Nfl[l] < 3) { if (err) *err = 1; return f; }
                for (int i = 0; i < Nfl[l]; i++) f = f + logP_gaussian(0, scoef, second_derivative(params + off, Nfl[l], i));
            }
            off += Nfl[l];
        }
    }
    if (std::fabs(params[s + 2] / params[s]) >= a3ova1_limit) f = NEG_INF;
    if (params[s + 4] < 0) f = NEG_INF;
    if (S.sw[z + 3] != 0)
        if ((params[z + 3] < 0) || (params[z + 4] < 0) || (params[z + 5] < 0)) f = NEG_INF;
    if (S.sw[z + 6] != 0)
        if ((params[z + 6] < 0) || (params[z + 7] < 0) || (params[z + 8] < 0)) f = NEG_INF;
    if ((S.sw[s + 9] != 0) && (params[z + 9] < 0)) f = NEG_INF;   // index quirk of the reference (:166) kept
    return f;
}

// priors_local, priors_calc.cpp:175-280
ld priors_local(const PriorSpec &S, const double *params, int *err)
{
    ld f = 0;
    const double a3ova1_limit = S.extra[2];
    const int Nmax = S.plength[0], Nvis = S.plength[1];
    const int Nf = S.plength[2] + S.plength[3] + S.plength[4] + S.plength[5];
    const int s = Nmax + Nvis + Nf, q = s + S.plength[6] + S.plength[7] + S.plength[8];
    f = f + apply_generic_priors(S, params, err);
    if (params[s] != 0) {
        if (std::fabs(params[s + 2] / params[s]) >= a3ova1_limit) f = NEG_INF;
    } else if ((params[s + 3] != 0) && (params[s + 4] != 0)) {
        if (std::fabs(params[s + 2] / (std::pow(params[s + 3], 2) + std::pow(params[s + 4], 2))) >= a3ova1_limit) f = NEG_INF;
    }
    if ((S.sw[q] != 0) && (params[q] < 0)) f = NEG_INF;
    return f;
}

ld log_prior(const PriorSpec &S, const double *params, int *err)
{
    switch (S.fct) {
    case 0: case 1: return apply_generic_priors(S, params, err);   // priors_Test_Gaussian / _Harvey_Gaussian (:288-310)
    case 2: return priors_MS_Global(S, params, err);
    case 3: return priors_local(S, params, err);
    default: if (err) *err = 1; return 0;
    }
}

// ------------------------------------------------------------------------------------------------
// RNG: glibc rand()/srand() (TYPE_3, r[i] = r[i-3] + r[i-31]) + random_JB.cpp's Box-Muller
// ------------------------------------------------------------------------------------------------
struct GlibcRand {
    int32_t r[31];
    int f = 3, b = 0;
    void seed(uint32_t s)
    {
        if (s == 0) s = 1;
        r[0] = (int32_t)s;
        for (int i = 1; i < 31; i++) {
            const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
            long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            r[i] = (int32_t)word;
        }
        f = 3; b = 0;
        for (int i = 0; i < 310; i++) (void)next();
    }
    int32_t next()
    {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        const int32_t out = (int32_t)(v >> 1);
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return out;
    }
    // n calls of next() in one go: the ring is unrolled into a line (no wrap-around tests, no pointer updates per value)
    // and written back with the pointers advanced by n -- the same state the n calls would leave.
    void fill(int32_t *out, int n)
    {
        while (n > 0) {
            const int m = n < 480 ? n : 480;
            uint32_t z[31 + 480];
            for (int i = 0; i < 31; i++) z[i] = (uint32_t)r[(f + i) % 31];          // age order, oldest first
            for (int i = 31; i < 31 + m; i++) { z[i] = z[i - 31] + z[i - 3]; out[i - 31] = (int32_t)(z[i] >> 1); }
            f = (f + m) % 31; b = (b + m) % 31;
            for (int i = 0; i < 31; i++) r[(f + i) % 31] = (int32_t)z[m + i];
            out += m; n -= m;
        }
    }
    // Jump ahead: the state after k more calls of next(), without making them.  The words obey z[n] = z[n-31] + z[n-3]
    // modulo 2^32, a linear recurrence with characteristic polynomial p(x) = x^31 - x^28 - 1, so with
    // G(x) = x^k mod p(x) = sum_j G_j x^j every later word is z[m + k] = sum_j G_j z[m + j].  G costs ~31^2 log2(k)
    // multiply-adds and is kept per k (a sharded sampler skips the same few distances every iteration); applying it
    // costs 30 ordinary steps (the window z[m .. m + 60]) and 31 x 31 multiply-adds.
    void jump(uint64_t k)
    {
        if (k < 64) { for (uint64_t i = 0; i < k; i++) (void)next(); return; }
        static thread_local std::vector<std::pair<uint64_t, std::array<uint32_t, 31>>> cache;
        const std::array<uint32_t, 31> *G = nullptr;
        for (const auto &e : cache) if (e.first == k) { G = &e.second; break; }
        if (!G) {
            auto reduce = [](uint32_t (&c)[61]) {      // x^31 = x^28 + 1
                for (int d = 60; d >= 31; d--) { c[d - 3] += c[d]; c[d - 31] += c[d]; c[d] = 0; }
            };
            uint32_t g[61] = {0};
            g[0] = 1;                                   // x^0
            int top = 63;
            while (top > 0 && !((k >> top) & 1)) top--;
            for (int bit = top; bit >= 0; bit--) {
                uint32_t sq[61] = {0};
                for (int i = 0; i < 31; i++)
                    if (g[i]) for (int j = 0; j < 31; j++) sq[i + j] += g[i] * g[j];
                reduce(sq);
                if ((k >> bit) & 1) {                   // times x
                    for (int d = 31; d >= 1; d--) sq[d] = sq[d - 1];
                    sq[0] = 0;
                    reduce(sq);
                }
                for (int i = 0; i < 61; i++) g[i] = sq[i];
            }
            std::array<uint32_t, 31> a;
            for (int i = 0; i < 31; i++) a[i] = g[i];
            if (cache.size() >= 16) cache.erase(cache.begin());
            cache.emplace_back(k, a);
            G = &cache.back().second;
        }
        // window of the sequence: z[0 .. 30] = the 31 words in age order (oldest first), z[31 .. 60] = the next 30
        uint32_t z[61];
        for (int i = 0; i < 31; i++) z[i] = (uint32_t)r[(f + i) % 31];
        for (int i = 31; i < 61; i++) z[i] = z[i - 31] + z[i - 3];
        uint32_t w[31];
        for (int t = 0; t < 31; t++) {
            uint32_t acc = 0;
            for (int j = 0; j < 31; j++) acc += (*G)[j] * z[t + j];
            w[t] = acc;                                 // z[t + k]
        }
        // the state after k steps, in the same ring layout: the pointers advance by k, the words in age order are w
        f = (int)((f + k) % 31); b = (int)((b + k) % 31);
        for (int i = 0; i < 31; i++) r[(f + i) % 31] = (int32_t)w[i];
    }
};

// Box-Muller pair: r8vec_normal_01 writes cos and sin of one angle side by side (random_JB.cpp:120-190), which an
// optimising compiler turns into ONE sincos() call -- and glibc's sincos differs from its lone sin() / cos() in the last
// bit about once in 3000 values.  The pair is therefore made by sincos() explicitly, everywhere: the numbers do not
// depend on the optimisation level this file is built with (the sanitizer build at -O1 used to differ from the product).
static inline void bm_pair(double r0, double r1, double *c_part, double *s_part)
{
    const double PI = 3.141592653589793;
    double sn, cs;
    ::sincos(2.0 * PI * r1, &sn, &cs);
    const double a = std::sqrt(-2.0 * std::log(r0));
    *c_part = a * cs;
    *s_part = a * sn;
}

struct Rng {
    GlibcRand g;
    int saved = 0;       // random_JB.cpp: static int saved; static double y
    double y = 0.0;
    double uniform() { return (double)g.next() / 2147483647.0; }   // random_JB.cpp:255
    // r8vec_normal_01(n), random_JB.cpp:22-213
    void normals(int n, double *x)
    {
        if (n <= 0) return;
        int x_lo = 1, x_hi = n;
        if (saved == 1) { x[0] = y; saved = 0; x_lo = 2; }
        const int cnt = x_hi - x_lo + 1;
        if (cnt == 0) {
        } else if (cnt == 1) {
            const double r0 = uniform(), r1 = uniform();
            bm_pair(r0, r1, &x[x_hi - 1], &y);
            saved = 1;
        } else if (cnt % 2 == 0) {
            const int m = cnt / 2;
            std::vector<double> r(2 * m);
            for (auto &v : r) v = uniform();
            for (int i = 0; i <= 2 * m - 2; i += 2) bm_pair(r[i], r[i + 1], &x[x_lo + i - 1], &x[x_lo + i]);
        } else {
            x_hi = x_hi - 1;
            const int m = (x_hi - x_lo + 1) / 2 + 1;
            std::vector<double> r(2 * m);
            for (auto &v : r) v = uniform();
            for (int i = 0; i <= 2 * m - 4; i += 2) bm_pair(r[i], r[i + 1], &x[x_lo + i - 1], &x[x_lo + i]);
            const int i = 2 * m - 2;
            bm_pair(r[i], r[i + 1], &x[x_lo + i - 1], &y);
            saved = 1;
        }
    }

    // The same routine split in two so that the transcendental work can run on several threads: `draw` consumes
    // the random stream exactly like normals(n, .) (and settles the carried value y, the one cross-call dependency);
    // `NormalPlan::fill` then produces the n values from the recorded uniforms.  fill(x) == what normals(n, x) returns.
    struct NormalPlan {
        int n = 0, npairs = 0;         // pairs (cos, sin) written to x[first .. first + 2 npairs)
        int first = 0;                 // 0 or 1 (x[0] taken from the carry)
        bool has_carry_in = false, tail = false;   // tail: one more value (cos part only) after the pairs
        double carry_in = 0.0, tail_x = 0.0;
        std::vector<int32_t> raw;      // what rand() returned; the uniforms r = raw / 2147483647 (random_JB.cpp:255) are made in fill
        static double uni(int32_t v) { return (double)v / 2147483647.0; }
        void fill(double *x) const
        {
            if (has_carry_in) x[0] = carry_in;
            for (int k = 0; k < npairs; k++) bm_pair(uni(raw[2 * k]), uni(raw[2 * k + 1]), &x[first + 2 * k], &x[first + 2 * k + 1]);
            if (tail) x[first + 2 * npairs] = tail_x;      // made by draw together with the carried sine (see there)
        }
    };
    void draw(int n, NormalPlan &P)
    {
        P.n = n; P.npairs = 0; P.first = 0; P.has_carry_in = false; P.tail = false;
        if (n <= 0) return;
        int cnt = n;
        if (saved == 1) { P.has_carry_in = true; P.carry_in = y; P.first = 1; saved = 0; cnt = n - 1; }
        if (cnt == 0) return;
        const bool odd = (cnt % 2) != 0;
        const int nr = odd ? cnt + 1 : cnt;          // uniforms consumed (random_JB.cpp: 2 for cnt == 1, 2m otherwise)
        if ((int)P.raw.size() < nr) P.raw.resize(nr);
        g.fill(P.raw.data(), nr);                     // (the serial part of an iteration's draws: only the raw stream)
        P.npairs = cnt / 2;
        P.tail = odd;
        if (odd) {
            // The last pair: its cosine part is the call's last value, its sine part is kept for the next call.  Both are
            // made HERE, side by side as in r8vec_normal_01 (random_JB.cpp:185-190): a compiler turns the adjacent
            // cos / sin of one angle into one sincos call, whose cosine can differ in the last bit from a lone cos().
            bm_pair(NormalPlan::uni(P.raw[nr - 2]), NormalPlan::uni(P.raw[nr - 1]), &P.tail_x, &y);
            saved = 1;
        }
    }
};

// lower Cholesky factor of an n x n SPD matrix (what tmpmat.llt().matrixL() returns, MALA.cpp:344).
// Right-looking form, four columns per pass over the trailing matrix: every element still receives
// a_ij - l_i0 l_j0 - l_i1 l_j1 - ... in that order, each term a separate multiply and subtract (the same operations, in
// the same order, as the textbook left-looking loops -- bitwise the same factor, and the same columns filled in when
// the matrix turns out not to be positive definite), but the inner loops are contiguous elementwise updates the
// compiler vectorises without reassociating anything, and a pass over the trailing matrix applies four columns (64
// factorisations of 44 x 44 per iteration are the longest item of the adapting phase's accept step).
static inline bool chol_col(double *__restrict__ W, double *__restrict__ L, double *__restrict__ c, int n, int col)
{
    const double d = W[(size_t)col * n + col];
    if (!(d > 0.0)) return false;
    const double l = std::sqrt(d);
    L[(size_t)col * n + col] = l;
    for (int i = col + 1; i < n; i++) { c[i] = W[(size_t)i * n + col] / l; L[(size_t)i * n + col] = c[i]; }
    return true;
}

__attribute__((target_clones("avx512f", "avx2", "default")))
bool cholesky(const double *A, int n, double *L, double *W /* n*n + 4*n scratch */)
{
    double *__restrict__ c0 = W + (size_t)n * n, *__restrict__ c1 = c0 + n, *__restrict__ c2 = c1 + n, *__restrict__ c3 = c2 + n;
    std::memset(L, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) W[(size_t)i * n + j] = A[(size_t)i * n + j];
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        if (!chol_col(W, L, c0, n, k)) return false;
        { const double q0 = c0[k + 1];
          for (int i = k + 1; i < n; i++) W[(size_t)i * n + k + 1] -= c0[i] * q0; }
        if (!chol_col(W, L, c1, n, k + 1)) return false;
        { const double q0 = c0[k + 2], q1 = c1[k + 2];
          for (int i = k + 2; i < n; i++) { const double t = W[(size_t)i * n + k + 2] - c0[i] * q0; W[(size_t)i * n + k + 2] = t - c1[i] * q1; } }
        if (!chol_col(W, L, c2, n, k + 2)) return false;
        { const double q0 = c0[k + 3], q1 = c1[k + 3], q2 = c2[k + 3];
          for (int i = k + 3; i < n; i++) { double t = W[(size_t)i * n + k + 3] - c0[i] * q0; t = t - c1[i] * q1; W[(size_t)i * n + k + 3] = t - c2[i] * q2; } }
        if (!chol_col(W, L, c3, n, k + 3)) return false;
        for (int i = k + 4; i < n; i++) {
            const double a0 = c0[i], a1 = c1[i], a2 = c2[i], a3 = c3[i];
            double *__restrict__ w = W + (size_t)i * n;
            for (int j = k + 4; j <= i; j++) {
                double t = w[j] - a0 * c0[j];
                t = t - a1 * c1[j];
                t = t - a2 * c2[j];
                w[j] = t - a3 * c3[j];
            }
        }
    }
    for (; k < n; k++) {
        if (!chol_col(W, L, c0, n, k)) return false;
        for (int i = k + 1; i < n; i++) {
            const double li = c0[i];
            double *__restrict__ w = W + (size_t)i * n;
            for (int j = k + 1; j <= i; j++) w[j] -= li * c0[j];
        }
    }
    return true;
}

// Fork-join helper for the per-chain host work (the reference runs its chain loop under OpenMP, MALA.cpp:632).
// Every chain's arithmetic is sequential and independent of the others, so results do not depend on the thread count.
// Fork-join over chains with a STATIC partition: participant p of P (the caller is participant 0) runs a contiguous
// block of items.  No shared work counter: on a many-CCD host every contended fetch_add costs 100-300 ns,
// and a dynamic queue spent ~20 us per fork on 64 items (measured); here a fork costs one epoch broadcast plus one
// arrival per worker, each on its own cache line.  The fixed item -> thread map also keeps each chain's matrices in
// the same core's cache.
class ChainPool {
public:
    explicit ChainPool(int nthreads) : P_(nthreads < 1 ? 1 : nthreads), slots_((size_t)(nthreads < 1 ? 1 : nthreads))
    {
        for (int t = 1; t < P_; t++) workers_.emplace_back([this, t] { loop(t); });
    }
    ~ChainPool()
    {
        { std::lock_guard<std::mutex> g(mx_); stop_ = true; hdr_.epoch.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    int size() const { return P_; }
    template <class F> void run(int n, F &&fn)
    {
        if (workers_.empty() || n < 2) { for (int i = 0; i < n; i++) fn(i); return; }
        std::function<void(int)> job = std::ref(fn);
        hdr_.job.store(&job, std::memory_order_relaxed); hdr_.n.store(n, std::memory_order_relaxed);
        uint64_t e;
        { std::lock_guard<std::mutex> g(mx_); e = hdr_.epoch.fetch_add(1, std::memory_order_release) + 1; }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        for (int i = 0, hi = block_end(0, n); i < hi; i++) fn(i);
        for (int p = 1; p < P_; p++)
            while (slots_[(size_t)p].done.load(std::memory_order_acquire) != e) __builtin_ia32_pause();
    }

private:
    // what the caller writes and every worker polls / reads: one cache line, never written by a worker
    struct alignas(64) Header {
        std::atomic<uint64_t> epoch{0};
        std::atomic<int> n{0};
        std::atomic<const std::function<void(int)> *> job{nullptr};
    };
    // a worker's arrival flag (the epoch it has finished): its own cache line, so arrivals do not contend
    struct alignas(64) Slot { std::atomic<uint64_t> done{0}; };
    // participant p owns the contiguous items [block_end(p-1), block_end(p)): neighbours in the per-chain arrays
    // belong to the same thread (no false sharing), and sizes differ by at most one item
    int block_end(int p, int n) const { return (int)(((long long)(p + 1) * n) / P_); }
    void loop(int p)
    {
        uint64_t seen = 0;
        for (;;) {
            // spin for a while (the next fork usually comes within one GPU evaluation), then sleep
            uint64_t e = hdr_.epoch.load(std::memory_order_acquire);
            for (int spin = 0; e == seen && spin < 40000; spin++) { __builtin_ia32_pause(); e = hdr_.epoch.load(std::memory_order_acquire); }
            if (e == seen) {
                std::unique_lock<std::mutex> g(mx_);
                sleepers_.fetch_add(1, std::memory_order_release);
                cv_.wait(g, [&] { return hdr_.epoch.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_release);
                e = hdr_.epoch.load(std::memory_order_acquire);
            }
            seen = e;
            if (stop_) return;
            const int n = hdr_.n.load(std::memory_order_relaxed);
            const std::function<void(int)> &job = *hdr_.job.load(std::memory_order_relaxed);
            for (int i = block_end(p - 1, n), hi = block_end(p, n); i < hi; i++) job(i);
            slots_[(size_t)p].done.store(e, std::memory_order_release);
        }
    }
    int P_ = 1;
    Header hdr_;
    std::vector<Slot> slots_;
    std::vector<std::thread> workers_;
    std::mutex mx_;
    std::condition_variable cv_;
    std::atomic<int> sleepers_{0};
    bool stop_ = false;
};

double min1(double e)   // VectorXd(1., e).minCoeff() with the scalar visitor: NaN never replaces 1
{
    double r = 1.0;
    if (e < r) r = e;
    return r;
}

int hip_eval_trampoline(void *user, int32_t n, int32_t np, const double *params, const double *T, double *logL, int32_t *status)
{
    return tamcmc_eval_batch(static_cast<tamcmc_ctx *>(user), n, np, params, T, logL, nullptr, 0, nullptr, nullptr, status);
}

}  // namespace

struct tamcmc_sampler {
    tamcmc_sampler_cfg cfg{};
    tamcmc_eval_fn eval = nullptr;
    void *eval_user = nullptr;
    PriorSpec prior;
    int Nparams = 0, Nvars = 0, nloc = 0;
    std::vector<int32_t> index_to_relax;
    std::vector<double> T;                       // global ladder
    // current state (local chains)
    std::vector<double> params, vars, logL, logPrior, logPost, Pmove;
    std::vector<uint8_t> moved;
    // proposal (local chains)
    std::vector<double> covar, sigma, mu, Lchol;
    std::vector<uint8_t> chol_valid;
    // scratch
    std::vector<double> p_prop, v_prop, L_prop, lpr_prop, u_mh, u_now, z, z_all, chol_scratch;
    std::vector<int> perr_prop;
    bool timing = false;             // developer switch TAMCMC_SAMPLER_TIMING=1: phase times of mh_step, printed at destroy
    double t_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // proposals, launch, priors, draw-ahead, wait, accept, raw draws of chains owned elsewhere, exchange (seconds)
    int64_t t_iters = 0;
    std::vector<int32_t> status;
    std::vector<Rng::NormalPlan> plans;          // one per local chain (+1 for chains owned elsewhere)
    // The random numbers of an iteration do not depend on any outcome, so they are drawn one iteration ahead, while the
    // GPU evaluates the current proposals: MH draws of iteration i+1 (and the parallel-tempering draws of iteration i,
    // which precede them in the stream) are consumed right after the evaluation of iteration i was launched.
    bool drawn_ahead = false;                    // plans / u_mh / z_all already hold the draws of the coming mh_step
    bool pt_cached = false;                      // parallel-tempering draws of the current iteration already consumed
    // With the draws of iteration i+1 in hand, chain m's next proposal needs nothing but chain m's own state after its
    // accept step: it is computed in the SAME pass over the chains (one fork of the thread pool per iteration instead
    // of two, chain m's rows still in its core's cache).  A parallel-tempering swap, which comes in between
    // (MALA.cpp:676), re-proposes the two chains it touched; setters of restored state drop the flag.
    bool proposed_ahead = false;                 // v_prop / p_prop already hold the proposals of the coming mh_step
    // The launches of the NEXT iteration's evaluation are put into the stream while the GPU works on this one
    // (tamcmc_eval_batch_arm: they wait behind a gate kernel), so that between the accept step and the GPU starting on the
    // new proposals there is one store instead of two kernel launches.  Only inside tamcmc_sampler_run / _run_sharded,
    // which know that another iteration follows (arm_next); TAMCMC_SAMPLER_ARM=0 turns it off.
    bool arm_enabled = true, arm_sharded = false, arm_next = false, ctx_armed = false;
    // Results arrive chain by chain (tamcmc_eval_batch_poll): the accept pass is forked BEFORE they are there, every
    // participant watches the chains it owns and runs a chain's accept step the moment its logL has landed -- the fork
    // and most of the pass are then under the evaluation's tail.  TAMCMC_SAMPLER_ARRIVE=0: wait for the whole batch first.
    bool arrive_enabled = true;
    std::vector<uint8_t> arrived;                // [nloc] accept step done on arrival (this iteration)
    // While the proposal matrix of a chain is frozen (Acquire phase, or between two adaptation periods) its step
    // chol(...) z depends on no outcome either: it is computed with the draws, under the GPU evaluation, and the accept
    // pass only adds it to whichever state the chain ends up in (same sum, same order: propose_chain).  A step is used
    // only if it was made from the normals and the factor that are current (generation counters, so that no path that
    // redraws or refactors can leave a stale step behind).
    std::vector<double> step;                    // [nloc][Nvars]
    std::vector<uint64_t> step_zgen, step_cgen, chol_gen;   // per chain: generations the step was made from; factor generation
    uint64_t z_gen = 1;                          // bumped whenever z_all is rewritten
    // Pipelined loop (tamcmc_sampler_run / _run_sharded with the HIP evaluator): the local chains in two halves, each its
    // own sub-batch on its own stream (tamcmc_eval_batch_begin_part).  One half's accept step, next proposals and next
    // launch happen on the host while the GPU evaluates the other half -- chains are independent inside an iteration
    // (MALA.cpp:632-655) and no draw depends on an outcome, so every chain sees exactly the numbers of the plain loop.
    // ... and the random numbers of the iteration after the next come from a thread of their own (DrawAhead below): with
    // the GPU hidden behind the host, the draws (a serial stream, ~30 us per iteration at 64 x 44) would otherwise be the
    // longest item on the host's critical path.  The thread is asked for exactly the packets the plain loop would draw,
    // one iteration early, and never beyond the last iteration of a call: the stream position on return is the plain
    // loop's.
    struct Packet { std::vector<double> u_mh, z_all; double pt_u = 0.0; int32_t pt_A = 0; bool has_pt = false; };
    Packet packet;                               // what the thread fills: PT draws of iteration j, MH draws of j + 1
    std::unique_ptr<ChainPool> draw_pool;        // the thread's helpers for the Box-Muller transforms
    std::thread draw_thread;
    std::atomic<uint64_t> draw_req{0}, draw_done{0};
    std::atomic<bool> draw_stop{false};
    bool draw_req_pt = false;                    // (written before draw_req is bumped) the requested packet has PT draws
    bool draw_pending = false;                   // a request is outstanding (main thread's view)
    int split = 0;                               // > 0: pipelining on (chains per part, rounded up)
    int nparts = 0;                              // sub-batches in flight (2 .. TAMCMC_MAX_PARTS)
    bool inflight = false;                       // both halves of iteration `iter` are launched (only inside a run call)
    bool reserved = false;                       // the context's buffers are sized for nloc chains
    std::atomic<int64_t> bad_chol_ahead{0};
    int32_t pt_A = 0;
    double pt_u = 0.0;
    tamcmc_ctx *hip_ctx = nullptr;               // set by create_hip: the evaluation can be split in begin / end
    std::unique_ptr<ChainPool> pool;
    Rng rng;
    int64_t iter = 0;
    int64_t bad_chol = 0;
};

static int sampler_alloc(tamcmc_sampler **out, const tamcmc_sampler_cfg *cfg, tamcmc_eval_fn eval, void *eval_user,
                         int32_t Nparams, const int32_t plength[11], const double *inputs, const int32_t *relax,
                         const int32_t *sw, const double *pp, int32_t nrows, const double extra[4], const double *err)
{
    if (!out || !cfg || !eval || !plength || !inputs || !relax || !sw || !extra || !err) return TAMCMC_E_INVALID;
    if (cfg->Nchains < 1 || cfg->Nchains_local < 1 || cfg->chain_offset < 0 ||
        cfg->chain_offset + cfg->Nchains_local > cfg->Nchains) return TAMCMC_E_INVALID;
    if (cfg->n_learn < 0 || cfg->n_learn > TAMCMC_MAX_LEARN || nrows < 0 || (nrows > 0 && !pp)) return TAMCMC_E_INVALID;
    int sum = 0;
    for (int i = 0; i < 11; i++) sum += plength[i];
    if (sum != Nparams) return TAMCMC_E_INVALID;
    tamcmc_sampler *s = new (std::nothrow) tamcmc_sampler();
    if (!s) return TAMCMC_E_NOMEM;
    s->cfg = *cfg;
    s->eval = eval; s->eval_user = eval_user;
    s->Nparams = Nparams;
    s->nloc = cfg->Nchains_local;
    for (int i = 0; i < Nparams; i++) if (relax[i] == 1) s->index_to_relax.push_back(i);
    s->Nvars = (int)s->index_to_relax.size();
    if (s->Nvars < 1) { delete s; return TAMCMC_E_INVALID; }
    PriorSpec &P = s->prior;
    P.fct = cfg->prior_fct_switch; P.Nparams = Nparams; P.nrows = nrows;
    std::memcpy(P.plength, plength, sizeof(int) * 11);
    P.sw.assign(sw, sw + Nparams);
    if (nrows > 0) P.pp.assign(pp, pp + (size_t)nrows * Nparams);
    std::memcpy(P.extra, extra, sizeof(double) * 4);
    const int n = s->nloc, nv = s->Nvars;
    s->T.resize(cfg->Nchains);
    for (int m = 0; m < cfg->Nchains; m++) s->T[m] = std::pow(cfg->lambda_temp, m);   // MALA.cpp:103
    s->params.resize((size_t)n * Nparams); s->vars.resize((size_t)n * nv);
    for (int m = 0; m < n; m++) {
        std::memcpy(&s->params[(size_t)m * Nparams], inputs, sizeof(double) * Nparams);
        for (int k = 0; k < nv; k++) s->vars[(size_t)m * nv + k] = inputs[s->index_to_relax[k]];
    }
    s->logL.assign(n, 0); s->logPrior.assign(n, 0); s->logPost.assign(n, 0); s->Pmove.assign(n, 0); s->moved.assign(n, 0);
    // init_proposal, MALA.cpp:246-289
    s->covar.assign((size_t)n * nv * nv, 0.0); s->sigma.resize(n); s->mu.resize((size_t)n * nv);
    s->Lchol.assign((size_t)n * nv * nv, 0.0); s->chol_valid.assign(n, 0);
    for (int m = 0; m < n; m++) {
        for (int k = 0; k < nv; k++) {
            s->covar[((size_t)m * nv + k) * nv + k] = err[k] * err[k];
            s->mu[(size_t)m * nv + k] = s->vars[(size_t)m * nv + k];
        }
        s->sigma[m] = std::pow(2.38, 2) * std::pow(s->T[cfg->chain_offset + m], 0.2) / nv;
    }
    s->p_prop.resize((size_t)n * Nparams); s->v_prop.resize((size_t)n * nv); s->L_prop.resize(n); s->lpr_prop.resize(n); s->perr_prop.resize(n);
    s->u_mh.resize(n); s->z.resize(nv); s->status.resize(n);
    s->plans.resize((size_t)n + 1);
    s->z_all.resize((size_t)n * nv); s->chol_scratch.resize((size_t)n * (2 * (size_t)nv * nv + 4 * (size_t)nv));
    s->step.assign((size_t)n * nv, 0.0); s->step_zgen.assign(n, 0); s->step_cgen.assign(n, 0); s->chol_gen.assign(n, 1);
    {   // host threads for the per-chain work: TAMCMC_SAMPLER_THREADS, default min(cores, 16, chains), and no more than
        // the work of an iteration pays for: a fork costs a few microseconds, a chain's proposal ~nv^2 flops (measured:
        // 10 chains x 9 variables run 25 % faster on one thread than on ten; 64 x 44 want all sixteen)
        int nt = (int)std::thread::hardware_concurrency();
        if (nt > 16) nt = 16;
        const long long work = (long long)n * nv * nv;
        const int by_work = (int)((work + 3999) / 4000);
        if (nt > by_work) nt = by_work;
        if (const char *e = std::getenv("TAMCMC_SAMPLER_THREADS")) nt = std::atoi(e);
        if (nt > n) nt = n;
        if (nt < 1) nt = 1;
        s->pool.reset(new ChainPool(nt));
    { const char *e = getenv("TAMCMC_SAMPLER_TIMING"); s->timing = e && e[0] == '1'; }
    { const char *e = getenv("TAMCMC_SAMPLER_ARM"); s->arm_enabled = !(e && e[0] == '0'); s->arm_sharded = e && e[0] == '2'; }
    { const char *e = getenv("TAMCMC_SAMPLER_ARRIVE"); s->arrive_enabled = !(e && e[0] == '0'); }
    s->arrived.assign((size_t)n, 0);
    {   // Two halves in flight (pipelined_iteration) with the draws on a thread of their own: OFF unless
        // TAMCMC_SAMPLER_PIPELINE=1.  Measured at 64 chains x 1e5 bins, PT every iteration (profiles/README.md, round 3):
        // without the draw thread the loop is host-bound and SLOWER (84 us per iteration against 65 us with one batch: four
        // launches instead of two, two pool forks, and the priors and draws with no GPU evaluation left to hide under);
        // with it 60-62 us (Acquire, +6 %) and 85-90 us (Learning, +-0): the main thread now waits ~26 us per iteration
        // for the GPU, because a half batch is not half the GPU time -- launch + setup kernel + eval floor are ~33 us from
        // launch to result against 42 us for the whole batch -- while the host has only ~18 us of work on the other half.
        // More parts make it worse (3: 74 us, 4: 85 us per Acquire iteration): every part costs two launches (8-9 us of
        // host time) and a pool fork (~10 us whatever the number of chains on this host).
        // Same draws, same decisions either way (tests/test_sampler_gpu.py).
        const char *e = getenv("TAMCMC_SAMPLER_PIPELINE");      // 1: two parts; 2 .. TAMCMC_MAX_PARTS: that many
        int parts = e ? atoi(e) : 0;
        if (parts == 1) parts = 2;
        if (parts > TAMCMC_MAX_PARTS) parts = TAMCMC_MAX_PARTS;
        if (n < 4 * parts) parts = 0;
        s->nparts = parts;
        s->split = parts > 0 ? (n + parts - 1) / parts : 0;
    }
    }
    s->rng.g.seed(cfg->seed);
    *out = s;
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_create(tamcmc_sampler **out, const tamcmc_sampler_cfg *cfg, tamcmc_eval_fn eval, void *eval_user,
                                     int32_t Nparams, const int32_t plength[11], const double *inputs, const int32_t *relax,
                                     const int32_t *sw, const double *pp, int32_t nrows, const double extra[4], const double *err)
{
    return sampler_alloc(out, cfg, eval, eval_user, Nparams, plength, inputs, relax, sw, pp, nrows, extra, err);
}

extern "C" int tamcmc_sampler_create_hip(tamcmc_sampler **out, const tamcmc_sampler_cfg *cfg, tamcmc_ctx *ctx,
                                         int32_t Nparams, const int32_t plength[11], const double *inputs, const int32_t *relax,
                                         const int32_t *sw, const double *pp, int32_t nrows, const double extra[4], const double *err)
{
    if (!ctx) return TAMCMC_E_INVALID;
    const int rc = sampler_alloc(out, cfg, hip_eval_trampoline, ctx, Nparams, plength, inputs, relax, sw, pp, nrows, extra, err);
    if (rc == TAMCMC_OK) (*out)->hip_ctx = ctx;
    return rc;
}

extern "C" int tamcmc_sampler_destroy(tamcmc_sampler *s)
{
    if (s && s->timing && s->t_iters > 0) {
        const double k = 1e6 / (double)s->t_iters;
        fprintf(stderr, "[tamcmc sampler] %lld iterations; us per iteration: proposals %.1f, launch %.1f, arm + priors %.1f, draw-ahead %.1f, wait %.1f, accept %.1f\n",
                (long long)s->t_iters, s->t_phase[0] * k, s->t_phase[1] * k, s->t_phase[2] * k, s->t_phase[3] * k, s->t_phase[4] * k, s->t_phase[5] * k);
    }
    if (s && s->draw_thread.joinable()) {
        s->draw_stop.store(true, std::memory_order_relaxed);
        s->draw_req.fetch_add(1, std::memory_order_release);
        s->draw_thread.join();
    }
    delete s;
    return TAMCMC_OK;
}
extern "C" int64_t tamcmc_sampler_iteration(const tamcmc_sampler *s) { return s ? s->iter : -1; }
extern "C" int32_t tamcmc_sampler_nvars(const tamcmc_sampler *s) { return s ? s->Nvars : -1; }
extern "C" int32_t tamcmc_sampler_nlocal(const tamcmc_sampler *s) { return s ? s->nloc : -1; }
extern "C" int tamcmc_sampler_layout(const tamcmc_sampler *s, int32_t *Nchains, int32_t *chain_offset, int32_t *Nchains_local)
{
    if (!s) return TAMCMC_E_INVALID;
    if (Nchains) *Nchains = s->cfg.Nchains;
    if (chain_offset) *chain_offset = s->cfg.chain_offset;
    if (Nchains_local) *Nchains_local = s->nloc;
    return TAMCMC_OK;
}

// model_def.cpp:139-143 + :358-367 for the starting point of every chain
extern "C" int tamcmc_sampler_init(tamcmc_sampler *s)
{
    if (!s) return TAMCMC_E_INVALID;
    const int n = s->nloc;
    int rc = s->eval(s->eval_user, n, s->Nparams, s->params.data(), &s->T[s->cfg.chain_offset], s->logL.data(), s->status.data());
    if (rc != TAMCMC_OK) return rc;
    int perr = 0;
    for (int m = 0; m < n; m++) {
        s->logPrior[m] = (double)log_prior(s->prior, &s->params[(size_t)m * s->Nparams], &perr);
        s->logPost[m] = s->logL[m] + s->logPrior[m];
    }
    return perr ? TAMCMC_E_INVALID : TAMCMC_OK;
}

// MALA.cpp:292-315 with the projections of :135-176
static void update_proposal(tamcmc_sampler *s, int m, double gamma, double acceptance)
{
    const int nv = s->Nvars;
    double *mu = &s->mu[(size_t)m * nv], *C = &s->covar[(size_t)m * nv * nv];
    const double *v = &s->vars[(size_t)m * nv];
    const double A1 = s->cfg.A1;
    // mu <- p3(mu + gamma (v - mu))
    double nrm = 0.0;
    for (int k = 0; k < nv; k++) { mu[k] = mu[k] + gamma * (v[k] - mu[k]); nrm += mu[k] * mu[k]; }
    nrm = std::sqrt(nrm);
    if (!(nrm <= A1)) for (int k = 0; k < nv; k++) mu[k] = mu[k] * A1 / nrm;
    // Sigma <- p2(Sigma + gamma ((v - mu)(v - mu)^T - Sigma)), with the ALREADY updated mu.  The Frobenius norm of p2 is
    // only ever compared with A1 (1e14 by default) and is summed here in four interleaved parts -- the reference takes
    // Eigen's .norm(), whose order of summation is Eigen's own; one running sum was a chain of nv^2 dependent additions
    // (2 us per chain at 44 variables) in front of every factorisation.
    double *dv = &s->chol_scratch[(size_t)m * (2 * (size_t)nv * nv + 4 * (size_t)nv)];      // (free between factorisations)
    for (int k = 0; k < nv; k++) dv[k] = v[k] - mu[k];
    double f4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < nv; i++) {
        const double di = dv[i];
        double *__restrict__ c = C + (size_t)i * nv;
        int j = 0;
        for (; j + 4 <= nv; j += 4)
            for (int q = 0; q < 4; q++) {
                const double cn = c[j + q] + gamma * (di * dv[j + q] - c[j + q]);
                c[j + q] = cn;
                f4[q] += cn * cn;
            }
        for (; j < nv; j++) {
            const double cn = c[j] + gamma * (di * dv[j] - c[j]);
            c[j] = cn;
            f4[0] += cn * cn;
        }
    }
    double fn = (f4[0] + f4[1]) + (f4[2] + f4[3]);
    fn = std::sqrt(fn);
    if (!(fn <= A1)) for (size_t e = 0; e < (size_t)nv * nv; e++) C[e] = C[e] * A1 / fn;
    // sigma <- p1(sigma + gamma (acceptance - target))
    ld x = (ld)s->sigma[m] + (ld)gamma * ((ld)acceptance - (ld)s->cfg.target_acceptance);
    ld sp = x;
    if (x < s->cfg.epsilon1) sp = s->cfg.epsilon1;
    if (x > A1) sp = A1;
    s->sigma[m] = (double)sp;
    s->chol_valid[m] = 0;
}

static bool learning_now(const tamcmc_sampler *s, int64_t i, int64_t *period)
{
    bool logic = false;
    int which = 0;
    for (int l = 0; l + 1 < s->cfg.n_learn; l++) {                 // MALA.cpp:644-650
        const bool in = (i >= s->cfg.Nt_learn[l]) && (i < s->cfg.Nt_learn[l + 1]);
        logic = logic || in;
        if (in) which = l;
    }
    *period = (s->cfg.n_learn >= 2) ? s->cfg.periods_learn[which] : 1;
    return logic;
}

// MH draws of one iteration in the reference's order: for each chain, u then z (MALA.cpp:451,465,346)
static inline double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Every process consumes the WHOLE stream (that is what makes a sharded run the single-process run, bit for bit), so
// the draws of the chains owned elsewhere are a serial term that grows with the global chain count while the GPU work
// per process shrinks; t_phase[6] keeps its share (tamcmc_sampler_get_timing) so that a sharded bench can report it.
// The raw stream of `count` chains owned elsewhere (for each: one uniform, then what r8vec_normal_01(Nvars) consumes:
// Nvars uniforms, one fewer when a value is carried in, rounded up to an even number, random_JB.cpp:134-209) is passed
// over by a jump of the generator -- except the last chain, which is drawn: whether it leaves a carried value, and
// which, is all the next chain can see of it (a carried value never travels further than one chain).
static void skip_foreign(tamcmc_sampler *s, int count)
{
    const int nv = s->Nvars, n = s->nloc;
    if (count <= 0) return;
    uint64_t K = 0;
    int saved = s->rng.saved;
    for (int c = 0; c + 1 < count; c++) {
        K += 1;
        const int cnt = nv - (saved ? 1 : 0);
        saved = 0;
        if (cnt > 0) { const int odd = cnt & 1; K += (uint64_t)(odd ? cnt + 1 : cnt); saved = odd; }
    }
    if (count > 1) { s->rng.g.jump(K); s->rng.saved = saved; s->rng.y = 0.0; }   // y: only the chain drawn next sees it, and that chain is not ours
    (void)s->rng.uniform();
    s->rng.draw(nv, s->plans[(size_t)n]);
}

static void draw_mh(tamcmc_sampler *s, double *u_out = nullptr)
{
    const int n = s->nloc, nv = s->Nvars, off = s->cfg.chain_offset, N = s->cfg.Nchains;
    if (!u_out) u_out = s->u_mh.data();
    const bool tm = s->timing && n < N;
    double t0 = tm ? wall_now() : 0.0;
    skip_foreign(s, off);
    if (tm) { const double t1 = wall_now(); s->t_phase[6] += t1 - t0; }
    for (int m = 0; m < n; m++) {
        u_out[m] = s->rng.uniform();
        s->rng.draw(nv, s->plans[(size_t)m]);
    }
    if (tm) t0 = wall_now();
    skip_foreign(s, N - off - n);
    if (tm) { const double t1 = wall_now(); s->t_phase[6] += t1 - t0; }
}

static void draw_pt(tamcmc_sampler *s)
{
    s->pt_u = s->rng.uniform();                                   // MALA.cpp:384
    s->pt_A = (int32_t)(s->rng.g.next() % (s->cfg.Nchains - 1));  // random_int_vals(0, Nchains-1), :390 and :178-187
}

extern "C" int tamcmc_sampler_pt_due(const tamcmc_sampler *s);

extern "C" int tamcmc_host_cholesky(const double *A, int32_t n, double *L)
{
    if (!A || !L || n < 1) return TAMCMC_E_INVALID;
    std::vector<double> W((size_t)n * n + 4 * (size_t)n);
    return cholesky(A, n, L, W.data()) ? 0 : 1;
}

// The step chol(...) z of local chain m from the normals in z_all, ahead of its use (only with a factor in hand: the
// matrix may be about to change, and a factorisation here could report a failure that never happens).
static inline void step_chain(tamcmc_sampler *s, int m)
{
    if (!s->chol_valid[m]) return;
    const int nv = s->Nvars;
    const double *z = &s->z_all[(size_t)m * nv];
    const double *Lc = &s->Lchol[(size_t)m * nv * nv];
    double *st = &s->step[(size_t)m * nv];
    for (int a = 0; a < nv; a++) {
        double acc = 0.0;
        for (int b = 0; b <= a; b++) acc += Lc[(size_t)a * nv + b] * z[b];
        st[a] = acc;
    }
    s->step_zgen[m] = s->z_gen; s->step_cgen[m] = s->chol_gen[m];
}

// Proposal of local chain m from its current vars and the normals in z_all: v' = v + chol((Sigma + eps2 I) sigma) z
// (MALA.cpp:335-353), the factor recomputed only when the proposal parameters changed.  Returns false when the matrix
// was not positive definite.
static bool propose_chain(tamcmc_sampler *s, int m, bool with_prior = false)
{
    const int nv = s->Nvars, np = s->Nparams;
    bool ok = true;
    if (!s->chol_valid[m]) {
        double *tmp = &s->chol_scratch[(size_t)m * (2 * (size_t)nv * nv + 4 * (size_t)nv)];
        const double *C = &s->covar[(size_t)m * nv * nv];
        for (int a = 0; a < nv; a++)
            for (int b = 0; b < nv; b++)
                tmp[(size_t)a * nv + b] = (C[(size_t)a * nv + b] + (a == b ? s->cfg.epsilon2 : 0.0)) * s->sigma[m];   // :342
        ok = cholesky(tmp, nv, &s->Lchol[(size_t)m * nv * nv], tmp + (size_t)nv * nv);
        s->chol_valid[m] = 1;
        s->chol_gen[m]++;
    }
    if (s->step_zgen[m] == s->z_gen && s->step_cgen[m] == s->chol_gen[m]) {
        const double *st = &s->step[(size_t)m * nv];                                    // made ahead by step_chain
        for (int a = 0; a < nv; a++) s->v_prop[(size_t)m * nv + a] = s->vars[(size_t)m * nv + a] + st[a];
    } else {
        const double *z = &s->z_all[(size_t)m * nv];
        const double *Lc = &s->Lchol[(size_t)m * nv * nv];
        for (int a = 0; a < nv; a++) {
            double acc = 0.0;
            for (int b = 0; b <= a; b++) acc += Lc[(size_t)a * nv + b] * z[b];
            s->v_prop[(size_t)m * nv + a] = s->vars[(size_t)m * nv + a] + acc;         // :349
        }
    }
    std::memcpy(&s->p_prop[(size_t)m * np], &s->params[(size_t)m * np], sizeof(double) * np);
    for (int k = 0; k < nv; k++) s->p_prop[(size_t)m * np + s->index_to_relax[k]] = s->v_prop[(size_t)m * nv + k];
    if (with_prior) {       // pipelined loop: the proposal's prior in the same pass (no GPU evaluation left to hide it under)
        int perr = 0;
        s->lpr_prop[m] = (double)log_prior(s->prior, &s->p_prop[(size_t)m * np], &perr);
        s->perr_prop[m] = perr;
    }
    return ok;
}

// Accept / reject and adaptation of local chain m (MALA.cpp:475-534, :641-652); with `ahead` the chain's next proposal
// follows at once (the normals of the next iteration are in z_all).
static inline void accept_chain(tamcmc_sampler *s, int m, int64_t i, double gamma, bool learn, int64_t period, bool ahead,
                                std::atomic<int> &perr_any)
{
    const int nv = s->Nvars, np = s->Nparams;
    const double lpr = s->lpr_prop[m];
    if (s->perr_prop[m]) perr_any.store(1, std::memory_order_relaxed);
    const double lpo = s->L_prop[m] + lpr;
    double r;
    if (!std::isnan(s->L_prop[m])) {
        if (lpo == -std::numeric_limits<double>::infinity()) r = 0.;
        else r = min1(std::exp(lpo - s->logPost[m]));
    } else {
        r = 0.;
    }
    if (s->u_now[m] <= r) {
        std::memcpy(&s->params[(size_t)m * np], &s->p_prop[(size_t)m * np], sizeof(double) * np);
        std::memcpy(&s->vars[(size_t)m * nv], &s->v_prop[(size_t)m * nv], sizeof(double) * nv);
        s->logL[m] = s->L_prop[m]; s->logPrior[m] = lpr; s->logPost[m] = lpo;
        s->moved[m] = 1;
    } else {
        s->moved[m] = 0;
    }
    s->Pmove[m] = r;
    if (learn && (i % period) == 0) update_proposal(s, m, gamma, r);
    if (ahead && !propose_chain(s, m, s->split > 0)) s->bad_chol_ahead.fetch_add(1, std::memory_order_relaxed);
}

// ---- the draw thread of the pipelined loop ------------------------------------------------------------------------
static void draw_packet(tamcmc_sampler *s, bool with_pt, ChainPool *pool)
{
    const int n = s->nloc, nv = s->Nvars;
    tamcmc_sampler::Packet &P = s->packet;
    P.has_pt = with_pt;
    if (with_pt) {
        P.pt_u = s->rng.uniform();                                    // MALA.cpp:384
        P.pt_A = (int32_t)(s->rng.g.next() % (s->cfg.Nchains - 1));   // :390
    }
    draw_mh(s, P.u_mh.data());
    pool->run(n, [&](int m) { s->plans[m].fill(&P.z_all[(size_t)m * nv]); });
}

static void draw_thread_main(tamcmc_sampler *s)
{
    uint64_t seen = 0;
    for (;;) {
        uint64_t r = s->draw_req.load(std::memory_order_acquire);
        for (int spin = 0; r == seen && !s->draw_stop.load(std::memory_order_relaxed); spin++) {
            if (spin < 20000) __builtin_ia32_pause(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
            r = s->draw_req.load(std::memory_order_acquire);
        }
        if (s->draw_stop.load(std::memory_order_relaxed)) return;
        seen = r;
        draw_packet(s, s->draw_req_pt, s->draw_pool.get());
        s->draw_done.store(r, std::memory_order_release);
    }
}

static void draw_request(tamcmc_sampler *s, bool with_pt)
{
    if (!s->draw_thread.joinable()) {
        s->packet.u_mh.resize((size_t)s->nloc); s->packet.z_all.resize((size_t)s->nloc * s->Nvars);
        s->draw_pool.reset(new ChainPool(4));
        s->draw_thread = std::thread(draw_thread_main, s);
    }
    s->draw_req_pt = with_pt;
    s->draw_req.fetch_add(1, std::memory_order_release);
    s->draw_pending = true;
}

// the requested packet becomes the sampler's current draws (u_mh / z_all of the coming iteration, PT draws of this one)
static void draw_collect(tamcmc_sampler *s)
{
    const uint64_t want = s->draw_req.load(std::memory_order_relaxed);
    while (s->draw_done.load(std::memory_order_acquire) != want) __builtin_ia32_pause();
    s->draw_pending = false;
    std::swap(s->u_mh, s->packet.u_mh);
    std::swap(s->z_all, s->packet.z_all); s->z_gen++;
    if (s->packet.has_pt) { s->pt_u = s->packet.pt_u; s->pt_A = s->packet.pt_A; s->pt_cached = true; }
}

static void drain_parts(tamcmc_sampler *s)
{
    if (!s->inflight) return;
    std::vector<double> l((size_t)s->nloc);
    for (int q = 0; q < s->nparts; q++) (void)tamcmc_eval_batch_end_part(s->hip_ctx, q, l.data(), nullptr);   // (parts not in flight: refused, harmless)
    s->inflight = false;
}

// One iteration of the pipelined loop.  pt_step() is the caller's parallel-tempering step (draw, local swap or boundary
// exchange, bookkeeping); it is called exactly once, when the accept step of every local chain of the drawn pair is done
// and before those chains are launched again.  moved_row (may be NULL) receives the acceptance flags as they are BEFORE
// the swap (what the plain loop records).  launch_next: leave iteration iter + 1 in flight on return.
template <class PT>
static int pipelined_iteration(tamcmc_sampler *s, bool launch_next, uint8_t *moved_row, PT &&pt_step)
{
    const int n = s->nloc, nv = s->Nvars, np = s->Nparams, off = s->cfg.chain_offset, P = s->nparts;
    const int64_t i = s->iter;
    const double gamma = s->cfg.c0 / (1. + (double)i);             // MALA.cpp:630
    auto now = [&]() { return s->timing ? wall_now() : 0.0; };
    double t0 = now(), t1;
    int m0[TAMCMC_MAX_PARTS + 1];
    for (int q = 0; q <= P; q++) m0[q] = (int)(((long long)q * n) / P);     // part q = chains [m0[q], m0[q + 1])
    auto part_of = [&](int m) { int q = 0; while (q + 1 < P && m >= m0[q + 1]) q++; return q; };
    auto launch = [&](int q) {
        return tamcmc_eval_batch_begin_part(s->hip_ctx, q, m0[q], m0[q + 1] - m0[q], np, &s->p_prop[(size_t)m0[q] * np], &s->T[off + m0[q]]);
    };
    bool flying[TAMCMC_MAX_PARTS] = {};
    auto fail = [&](int code) {      // leave nothing in flight behind an error
        if (s->draw_pending) draw_collect(s);
        std::vector<double> l((size_t)n);
        for (int q = 0; q < P; q++) if (flying[q]) (void)tamcmc_eval_batch_end_part(s->hip_ctx, q, l.data(), nullptr);
        s->inflight = false; s->proposed_ahead = false;
        return code;
    };
    int rc = TAMCMC_OK;
    std::atomic<int64_t> bad{0};
    if (!s->inflight) {
        // pipeline start: this iteration's draws and proposals (unless made ahead), every part launched
        if (!s->reserved) { rc = tamcmc_ctx_reserve(s->hip_ctx, n); if (rc != TAMCMC_OK) return rc; s->reserved = true; }
        if (!s->drawn_ahead) {
            draw_mh(s);
            s->z_gen++; s->pool->run(n, [&](int m) { s->plans[m].fill(&s->z_all[(size_t)m * nv]); });
        }
        s->drawn_ahead = false;
        if (!s->proposed_ahead)
            s->pool->run(n, [&](int m) { if (!propose_chain(s, m, true)) bad.fetch_add(1, std::memory_order_relaxed); });
        else      // proposals made by the plain loop (mh_step) carry no prior yet
            s->pool->run(n, [&](int m) {
                int perr = 0;
                s->lpr_prop[m] = (double)log_prior(s->prior, &s->p_prop[(size_t)m * np], &perr);
                s->perr_prop[m] = perr;
            });
        s->proposed_ahead = false;
        t1 = now(); s->t_phase[0] += t1 - t0; t0 = t1;
        for (int q = 0; q < P; q++) { rc = launch(q); if (rc != TAMCMC_OK) return fail(rc); flying[q] = true; }
        t1 = now(); s->t_phase[1] += t1 - t0; t0 = t1;
    } else {
        for (int q = 0; q < P; q++) flying[q] = true;
    }
    s->inflight = true;
    bad.fetch_add(s->bad_chol_ahead.exchange(0), std::memory_order_relaxed);
    s->bad_chol += bad.load();
    // The stream one iteration ahead: the parallel-tempering draws of THIS iteration (they come first, MALA.cpp:384,390),
    // then the MH draws of the next.  From the draw thread if it was asked for them during the previous iteration, else
    // made here (first iteration of a call); the packet after that is requested at once, unless this call ends first.
    s->u_now = s->u_mh;
    const bool due = tamcmc_sampler_pt_due(s) != 0;
    if (s->draw_pending) {
        draw_collect(s);
    } else {
        if (due && !s->pt_cached) { draw_pt(s); s->pt_cached = true; }
        draw_mh(s);
        s->z_gen++; s->pool->run(n, [&](int m) { s->plans[m].fill(&s->z_all[(size_t)m * nv]); });
    }
    s->drawn_ahead = true;
    if (launch_next) {
        const int64_t inext = i + 1;
        const bool due_next = s->cfg.dN_mixing > 0 && s->cfg.Nchains >= 2 && (inext % s->cfg.dN_mixing == 0) && inext != 0;
        draw_request(s, due_next);
    }
    t1 = now(); s->t_phase[3] += t1 - t0; t0 = t1;

    // Order of the parts: the one(s) holding the local chain(s) of the drawn pair first -- their accept step, the swap,
    // their relaunch -- then the others in turn.  A pair with a chain in each of two parts needs both before the swap.
    int pa = -1, pb = -1;                                 // parts of the pair's local chains (pb: second part, or -1)
    if (due) {
        const int a = s->pt_A - off, b = a + 1;
        const bool inA = a >= 0 && a < n, inB = b >= 0 && b < n;
        if (inA) pa = part_of(a);
        if (inB) { const int q = part_of(b); if (pa < 0) pa = q; else if (q != pa) pb = q; }
    }
    int64_t period = 1;
    const bool learn = learning_now(s, i, &period);
    std::atomic<int> perr_any{0};
    auto finish = [&](int q) {
        int r = tamcmc_eval_batch_end_part(s->hip_ctx, q, &s->L_prop[(size_t)m0[q]], &s->status[(size_t)m0[q]]);
        flying[q] = false;
        t1 = now(); s->t_phase[4] += t1 - t0; t0 = t1;
        if (r != TAMCMC_OK) return r;
        s->pool->run(m0[q + 1] - m0[q], [&](int k) { accept_chain(s, m0[q] + k, i, gamma, learn, period, true, perr_any); });
        if (moved_row) std::memcpy(moved_row + m0[q], &s->moved[(size_t)m0[q]], (size_t)(m0[q + 1] - m0[q]));
        t1 = now(); s->t_phase[5] += t1 - t0; t0 = t1;
        return (int)TAMCMC_OK;
    };
    auto relaunch = [&](int q) {
        const int r = launch(q);
        if (r == TAMCMC_OK) flying[q] = true;
        t1 = now(); s->t_phase[1] += t1 - t0; t0 = t1;
        return r;
    };
    bool pt_done = false;
    if (!launch_next) {
        // last iteration of a call: nothing is launched ahead
        for (int q = 0; q < P; q++) { rc = finish(q); if (rc != TAMCMC_OK) return fail(rc); }
        s->inflight = false;
        s->proposed_ahead = true;
        rc = pt_step();
        if (rc != TAMCMC_OK) return fail(rc);
    } else {
        if (pa >= 0) {
            rc = finish(pa);
            if (rc != TAMCMC_OK) return fail(rc);
            if (pb >= 0) { rc = finish(pb); if (rc != TAMCMC_OK) return fail(rc); }
            s->proposed_ahead = true;                   // (for the pair's chains: a swap re-proposes them, pt_apply)
            rc = pt_step();
            pt_done = true;
            if (rc != TAMCMC_OK) return fail(rc);
            rc = relaunch(pa);
            if (rc != TAMCMC_OK) return fail(rc);
            if (pb >= 0) { rc = relaunch(pb); if (rc != TAMCMC_OK) return fail(rc); }
        }
        for (int q = 0; q < P; q++) {
            if (q == pa || q == pb) continue;
            rc = finish(q);
            if (rc != TAMCMC_OK) return fail(rc);
            rc = relaunch(q);
            if (rc != TAMCMC_OK) return fail(rc);
        }
        if (!pt_done) { rc = pt_step(); if (rc != TAMCMC_OK) return fail(rc); }   // no local chain in the pair (or no attempt): bookkeeping only
        s->inflight = true; s->proposed_ahead = false;
    }
    s->t_iters++;
    if (perr_any.load()) { if (s->draw_pending) draw_collect(s); if (s->inflight) drain_parts(s); s->proposed_ahead = false; return TAMCMC_E_INVALID; }
    return TAMCMC_OK;
}

static bool pipeline_on(const tamcmc_sampler *s) { return s->hip_ctx != nullptr && s->split > 0; }

extern "C" int tamcmc_sampler_mh_step(tamcmc_sampler *s)
{
    if (!s) return TAMCMC_E_INVALID;
    const int n = s->nloc, nv = s->Nvars, np = s->Nparams, off = s->cfg.chain_offset;
    const int64_t i = s->iter;
    const double gamma = s->cfg.c0 / (1. + (double)i);             // MALA.cpp:630
    // 1. proposals.  Random draws first, in the reference's order: for each chain, u then z (MALA.cpp:451,465,346);
    //    then the per-chain linear algebra, chains in parallel.
    auto now = [&]() { return s->timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0; };
    double t0 = now(), t1;
    if (!s->drawn_ahead) {
        draw_mh(s);
        s->z_gen++; s->pool->run(n, [&](int m) { s->plans[m].fill(&s->z_all[(size_t)m * nv]); });
    }
    s->drawn_ahead = false;
    std::atomic<int64_t> bad{0};
    if (!s->proposed_ahead)
        s->pool->run(n, [&](int m) { if (!propose_chain(s, m)) bad.fetch_add(1, std::memory_order_relaxed); });
    s->proposed_ahead = false;
    bad.fetch_add(s->bad_chol_ahead.exchange(0), std::memory_order_relaxed);
    s->bad_chol += bad.load();
    t1 = now(); s->t_phase[0] += t1 - t0; t0 = t1;
    // 2. the hot path: every local chain in one call.  While the GPU works, consume the stream one iteration ahead:
    //    the parallel-tempering draws of THIS iteration (MALA.cpp:384,390; they come first in the stream), then the
    //    MH draws of the next one and their Box-Muller transforms.  u_mh of this iteration is kept aside first.
    int rc;
    int perr_arrive = 0;
    bool arrive_used = false;
    if (s->hip_ctx) {
        if (s->ctx_armed) {
            s->ctx_armed = false;
            rc = tamcmc_eval_batch_fire(s->hip_ctx, n, np, s->p_prop.data(), &s->T[off]);
        } else {
            rc = tamcmc_eval_batch_begin(s->hip_ctx, n, np, s->p_prop.data(), &s->T[off]);
        }
        if (rc != TAMCMC_OK) return rc;
        t1 = now(); s->t_phase[1] += t1 - t0; t0 = t1;
        if (s->arm_next && s->arm_enabled) {       // the next iteration's launches, under this one's evaluation
            s->ctx_armed = tamcmc_eval_batch_arm(s->hip_ctx, n) == TAMCMC_OK;      // (refused: the next step launches as usual)
        }
        s->u_now = s->u_mh;
        // the priors of the proposals need nothing from the GPU: off the critical path
        s->pool->run(n, [&](int m) {
            int perr = 0;
            s->lpr_prop[m] = (double)log_prior(s->prior, &s->p_prop[(size_t)m * np], &perr);
            s->perr_prop[m] = perr;
        });
        t1 = now(); s->t_phase[2] += t1 - t0; t0 = t1;
        if (tamcmc_sampler_pt_due(s) && !s->pt_cached) { draw_pt(s); s->pt_cached = true; }
        draw_mh(s);
        {
            // (the accept step of THIS iteration adapts the proposal: its factor is about to change, no step ahead)
            int64_t per_ = 1;
            const bool steps_ahead = !(learning_now(s, i, &per_) && (i % per_) == 0);
            s->z_gen++;
            s->pool->run(n, [&](int m) { s->plans[m].fill(&s->z_all[(size_t)m * nv]); if (steps_ahead) step_chain(s, m); });
        }
        s->drawn_ahead = true;
        t1 = now(); s->t_phase[3] += t1 - t0; t0 = t1;
        int64_t period_a = 1;
        const bool learn_a = learning_now(s, i, &period_a);
        // Only where the accept step is long -- the iterations that adapt the proposal (covariance update and a
        // factorisation per chain): with a frozen proposal the pass is a few microseconds, and sixteen threads watching
        // the result lines cost the evaluation's last stores more than the earlier start gains (measured).
        arrive_used = s->arrive_enabled && learn_a && (i % period_a) == 0;
        if (arrive_used) {
            // accept on arrival (the pass below then only picks up chains whose result was slow to come, if any)
            std::atomic<int> perr_a{0};
            std::fill(s->arrived.begin(), s->arrived.end(), (uint8_t)0);
            s->pool->run(n, [&](int m) {
                double Lm = 0.0;
                int32_t stm = 0;
                int rp = TAMCMC_PENDING;
                for (int spin = 0; spin < (1 << 20); spin++) {       // bounded: a failed launch is reported by _end below
                    rp = tamcmc_eval_batch_poll(s->hip_ctx, m, &Lm, &stm);
                    if (rp != TAMCMC_PENDING) break;
                    for (int b = 0; b < 8; b++) __builtin_ia32_pause();      // (easy on the lines the GPU is about to write)
                }
                if (rp != TAMCMC_OK) return;
                s->L_prop[m] = Lm; s->status[m] = stm;
                accept_chain(s, m, i, gamma, learn_a, period_a, true, perr_a);
                s->arrived[m] = 1;
            });
            if (perr_a.load()) perr_arrive = 1;
        }
        rc = tamcmc_eval_batch_end(s->hip_ctx, n, s->L_prop.data(), s->status.data());
        t1 = now(); s->t_phase[4] += t1 - t0; t0 = t1;
    } else {
        s->u_now = s->u_mh;
        s->pool->run(n, [&](int m) {
            int perr = 0;
            s->lpr_prop[m] = (double)log_prior(s->prior, &s->p_prop[(size_t)m * np], &perr);
            s->perr_prop[m] = perr;
        });
        rc = s->eval(s->eval_user, n, np, s->p_prop.data(), &s->T[off], s->L_prop.data(), s->status.data());
    }
    if (rc != TAMCMC_OK) return rc;
    // 3. accept / reject and adaptation, MALA.cpp:475-534, :641-652
    int64_t period = 1;
    const bool learn = learning_now(s, i, &period);
    std::atomic<int> perr_any{0};
    const bool ahead = s->drawn_ahead;      // the normals of the next iteration are in z_all: propose in the same pass
    auto accept = [&](int m) { accept_chain(s, m, i, gamma, learn, period, ahead, perr_any); };
    // (the fixed chain -> thread map of the pool keeps a chain's rows in the cache of the core that proposed them:
    // even the short accept step without adaptation is cheaper forked than pulled over to the calling thread)
    if (arrive_used) {
        int left = 0;
        for (int m = 0; m < n; m++) left += s->arrived[m] ? 0 : 1;
        if (left) s->pool->run(n, [&](int m) { if (!s->arrived[m]) accept(m); });
    } else {
        s->pool->run(n, accept);
    }
    s->proposed_ahead = ahead;
    t1 = now(); s->t_phase[5] += t1 - t0; s->t_iters++;
    const int perr = perr_any.load() | perr_arrive;
    return perr ? TAMCMC_E_INVALID : TAMCMC_OK;
}

extern "C" int tamcmc_sampler_pt_due(const tamcmc_sampler *s)
{
    if (!s || s->cfg.dN_mixing <= 0 || s->cfg.Nchains < 2) return 0;
    return (s->iter % s->cfg.dN_mixing == 0 && s->iter != 0) ? 1 : 0;   // MALA.cpp:676
}

extern "C" int tamcmc_sampler_pt_draw(tamcmc_sampler *s, int32_t *A, double *u)
{
    if (!s || !A || !u) return TAMCMC_E_INVALID;
    if (!s->pt_cached) draw_pt(s);          // (already consumed during the evaluation when drawing ahead)
    s->pt_cached = false;
    *u = s->pt_u;
    *A = s->pt_A;
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_pt_record_size(const tamcmc_sampler *s) { return s ? 4 + s->Nparams + s->Nvars : -1; }

extern "C" int tamcmc_sampler_pt_export(const tamcmc_sampler *s, int32_t chain, double *rec)
{
    if (!s || !rec) return TAMCMC_E_INVALID;
    const int m = chain - s->cfg.chain_offset;
    if (m < 0 || m >= s->nloc) return TAMCMC_E_INVALID;
    rec[0] = s->logL[m]; rec[1] = s->logPrior[m]; rec[2] = s->moved[m]; rec[3] = s->Pmove[m];
    std::memcpy(rec + 4, &s->params[(size_t)m * s->Nparams], sizeof(double) * s->Nparams);
    std::memcpy(rec + 4 + s->Nparams, &s->vars[(size_t)m * s->Nvars], sizeof(double) * s->Nvars);
    return TAMCMC_OK;
}

// MALA.cpp:393-434 for the end(s) of the pair (A, A+1) owned here.  recA / recB: records of chain A and
// A+1 BEFORE the swap (own export or the peer's).
static void pt_apply(tamcmc_sampler *s, int A, double u, const double *recA, const double *recB, int32_t *swapped, double *r_out)
{
    const int B = A + 1, np = s->Nparams, nv = s->Nvars, off = s->cfg.chain_offset;
    const double TA = s->T[A], TB = s->T[B];
    const double L_A = recA[0], L_B = recB[0];
    const ld logL_A_TB = (ld)(L_A * TA / TB), logL_B_TA = (ld)(L_B * TB / TA);
    const double r_T = min1(std::exp((double)(logL_A_TB + logL_B_TA - (ld)L_A - (ld)L_B)));
    const bool sw = (u <= r_T);
    if (sw) {
        const int a = A - off, b = B - off;
        if (a >= 0 && a < s->nloc) {   // A <- B
            std::memcpy(&s->params[(size_t)a * np], recB + 4, sizeof(double) * np);
            std::memcpy(&s->vars[(size_t)a * nv], recB + 4 + np, sizeof(double) * nv);
            s->logL[a] = (double)logL_B_TA;
            s->logPrior[a] = recB[1];
            s->logPost[a] = (double)(logL_B_TA + (ld)recB[1]);
            s->moved[a] = (uint8_t)recB[2]; s->Pmove[a] = recB[3];
        }
        if (b >= 0 && b < s->nloc) {   // B <- A ; logPosterior[B] uses logPrior[ind_A] AFTER it was overwritten (quirk A.6-8)
            std::memcpy(&s->params[(size_t)b * np], recA + 4, sizeof(double) * np);
            std::memcpy(&s->vars[(size_t)b * nv], recA + 4 + np, sizeof(double) * nv);
            s->logL[b] = (double)logL_A_TB;
            s->logPrior[b] = recA[1];
            s->logPost[b] = (double)(logL_A_TB + (ld)recB[1]);
            s->moved[b] = (uint8_t)recA[2]; s->Pmove[b] = recA[3];
        }
    }
    if (sw && s->proposed_ahead) {      // the proposals computed ahead started from the rows just replaced
        const int a = A - off, b = B - off;
        if (a >= 0 && a < s->nloc && !propose_chain(s, a, s->split > 0)) s->bad_chol_ahead.fetch_add(1, std::memory_order_relaxed);
        if (b >= 0 && b < s->nloc && !propose_chain(s, b, s->split > 0)) s->bad_chol_ahead.fetch_add(1, std::memory_order_relaxed);
    }
    if (swapped) *swapped = sw ? 1 : 0;
    if (r_out) *r_out = r_T;
}

extern "C" int tamcmc_sampler_pt_local(tamcmc_sampler *s, int32_t A, double u, int32_t *swapped, double *r_T)
{
    if (!s || A < 0 || A + 1 >= s->cfg.Nchains) return TAMCMC_E_INVALID;
    std::vector<double> ra(tamcmc_sampler_pt_record_size(s)), rb(ra.size());
    if (tamcmc_sampler_pt_export(s, A, ra.data()) != TAMCMC_OK || tamcmc_sampler_pt_export(s, A + 1, rb.data()) != TAMCMC_OK)
        return TAMCMC_E_INVALID;
    pt_apply(s, A, u, ra.data(), rb.data(), swapped, r_T);
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_pt_import(tamcmc_sampler *s, int32_t A, double u, const double *peer, int32_t *swapped, double *r_T)
{
    if (!s || !peer || A < 0 || A + 1 >= s->cfg.Nchains) return TAMCMC_E_INVALID;
    const int off = s->cfg.chain_offset;
    const bool ownA = (A >= off && A < off + s->nloc), ownB = (A + 1 >= off && A + 1 < off + s->nloc);
    if (ownA == ownB) return TAMCMC_E_INVALID;   // must own exactly one end
    std::vector<double> mine(tamcmc_sampler_pt_record_size(s));
    tamcmc_sampler_pt_export(s, ownA ? A : A + 1, mine.data());
    if (ownA) pt_apply(s, A, u, mine.data(), peer, swapped, r_T);
    else      pt_apply(s, A, u, peer, mine.data(), swapped, r_T);
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_end_iteration(tamcmc_sampler *s) { if (!s) return TAMCMC_E_INVALID; s->iter++; return TAMCMC_OK; }

// An armed evaluation must not outlive the loop that armed it (an error return, or a caller that goes on to use the
// context directly): on the way out of a run call its gate is opened and the batch waited for.
struct ArmGuard {
    tamcmc_sampler *s;
    ~ArmGuard()
    {
        s->arm_next = false;
        if (s->ctx_armed && s->hip_ctx) (void)tamcmc_eval_batch_disarm(s->hip_ctx);
        s->ctx_armed = false;
    }
};

extern "C" int tamcmc_sampler_run(tamcmc_sampler *s, int64_t n_iter, uint8_t *moved_hist, int32_t *swap_hist)
{
    if (!s || s->nloc != s->cfg.Nchains) return TAMCMC_E_INVALID;   // single process only
    ArmGuard guard{s};
    for (int64_t k = 0; k < n_iter; k++) {
        int32_t sh = -1;
        auto pt_step = [&]() {
            if (!tamcmc_sampler_pt_due(s)) return (int)TAMCMC_OK;
            int32_t A, swapped; double u, r;
            tamcmc_sampler_pt_draw(s, &A, &u);
            const int rc2 = tamcmc_sampler_pt_local(s, A, u, &swapped, &r);
            if (rc2 == TAMCMC_OK) sh = 2 * A + swapped;
            return rc2;
        };
        int rc;
        if (pipeline_on(s)) {
            rc = pipelined_iteration(s, k + 1 < n_iter, moved_hist ? moved_hist + (size_t)k * s->nloc : nullptr, pt_step);
            if (rc != TAMCMC_OK) return rc;
        } else {
            s->arm_next = (k + 1 < n_iter);
            rc = tamcmc_sampler_mh_step(s);
            if (rc != TAMCMC_OK) return rc;
            if (moved_hist) std::memcpy(moved_hist + (size_t)k * s->nloc, s->moved.data(), s->nloc);
            rc = pt_step();
            if (rc != TAMCMC_OK) return rc;
        }
        if (swap_hist) swap_hist[k] = sh;
        s->iter++;
    }
    return TAMCMC_OK;
}

// ------------------------------------------------------------------------------------------------
// Sharded runs: the iteration loop of MALA::execute (MALA.cpp:608-737) for the block of chains this process owns.
// ------------------------------------------------------------------------------------------------
struct tamcmc_shard_block {
    int64_t cap = 0, n = 0;
    int nloc = 0, nv = 0;
    std::vector<double> vars, stat, moved, pt;         // [cap][nloc][nv], [cap][3][nloc], [cap][nloc], [cap][4]
    std::vector<double> sum_sigma, sum_mu, sum_covar, sum_vars;
};

extern "C" int tamcmc_shard_block_create(tamcmc_shard_block **out, const tamcmc_sampler *s, int64_t capacity)
{
    if (!out || !s || capacity < 1) return TAMCMC_E_INVALID;
    tamcmc_shard_block *b = new (std::nothrow) tamcmc_shard_block();
    if (!b) return TAMCMC_E_NOMEM;
    b->cap = capacity; b->nloc = s->nloc; b->nv = s->Nvars;
    const size_t n = (size_t)s->nloc, nv = (size_t)s->Nvars, c = (size_t)capacity;
    try {
        b->vars.resize(c * n * nv); b->stat.resize(c * 3 * n); b->moved.resize(c * n); b->pt.resize(c * 4);
        b->sum_sigma.assign(n, 0.0); b->sum_mu.assign(n * nv, 0.0); b->sum_covar.assign(n * nv * nv, 0.0); b->sum_vars.assign(n * nv, 0.0);
    } catch (const std::bad_alloc &) {
        delete b;
        return TAMCMC_E_NOMEM;
    }
    *out = b;
    return TAMCMC_OK;
}
extern "C" int tamcmc_shard_block_destroy(tamcmc_shard_block *b) { delete b; return TAMCMC_OK; }
extern "C" int64_t tamcmc_shard_block_count(const tamcmc_shard_block *b) { return b ? b->n : -1; }
extern "C" int tamcmc_shard_block_reset(tamcmc_shard_block *b)
{
    if (!b) return TAMCMC_E_INVALID;
    b->n = 0;
    std::fill(b->sum_sigma.begin(), b->sum_sigma.end(), 0.0); std::fill(b->sum_mu.begin(), b->sum_mu.end(), 0.0);
    std::fill(b->sum_covar.begin(), b->sum_covar.end(), 0.0); std::fill(b->sum_vars.begin(), b->sum_vars.end(), 0.0);
    return TAMCMC_OK;
}
extern "C" int tamcmc_shard_block_data(tamcmc_shard_block *b, int32_t which, double **ptr, int64_t *count)
{
    if (!b || !ptr || !count) return TAMCMC_E_INVALID;
    std::vector<double> *v = nullptr;
    size_t used = 0;
    const size_t n = (size_t)b->nloc, nv = (size_t)b->nv, k = (size_t)b->n;
    switch (which) {
    case 0: v = &b->vars; used = k * n * nv; break;
    case 1: v = &b->stat; used = k * 3 * n; break;
    case 2: v = &b->moved; used = k * n; break;
    case 3: v = &b->pt; used = k * 4; break;
    case 4: v = &b->sum_sigma; used = v->size(); break;
    case 5: v = &b->sum_mu; used = v->size(); break;
    case 6: v = &b->sum_covar; used = v->size(); break;
    case 7: v = &b->sum_vars; used = v->size(); break;
    default: return TAMCMC_E_INVALID;
    }
    *ptr = v->data(); *count = (int64_t)used;
    return TAMCMC_OK;
}

// one sample of the local chains (what Outputs::update_buffer_* keep, outputs.cpp:863-1027) and the running sums of
// the proposal parameters (the *_mean entries of the restore files)
static void shard_block_record(tamcmc_shard_block *b, const tamcmc_sampler *s, double att, double A, double r, double sw)
{
    const size_t n = (size_t)b->nloc, nv = (size_t)b->nv, k = (size_t)b->n;
    std::memcpy(&b->vars[k * n * nv], s->vars.data(), sizeof(double) * n * nv);
    std::memcpy(&b->stat[(k * 3 + 0) * n], s->logL.data(), sizeof(double) * n);
    std::memcpy(&b->stat[(k * 3 + 1) * n], s->logPrior.data(), sizeof(double) * n);
    std::memcpy(&b->stat[(k * 3 + 2) * n], s->logPost.data(), sizeof(double) * n);
    for (size_t m = 0; m < n; m++) b->moved[k * n + m] = (double)s->moved[m];
    b->pt[k * 4 + 0] = att; b->pt[k * 4 + 1] = A; b->pt[k * 4 + 2] = r; b->pt[k * 4 + 3] = sw;
    for (size_t e = 0; e < n; e++) b->sum_sigma[e] += s->sigma[e];
    for (size_t e = 0; e < n * nv; e++) { b->sum_mu[e] += s->mu[e]; b->sum_vars[e] += s->vars[e]; }
    for (size_t e = 0; e < n * nv * nv; e++) b->sum_covar[e] += s->covar[e];
    b->n++;
}

extern "C" int tamcmc_sampler_run_sharded(tamcmc_sampler *s, int64_t n_iter, tamcmc_exchange_fn exchange, void *user,
                                          tamcmc_shard_block *block, uint8_t *moved_hist, int32_t *swap_hist, int64_t *done)
{
    if (!s || n_iter < 0) return TAMCMC_E_INVALID;
    if (block && (block->nloc != s->nloc || block->nv != s->Nvars)) return TAMCMC_E_INVALID;
    const int off = s->cfg.chain_offset, nloc = s->nloc, nrec = tamcmc_sampler_pt_record_size(s);
    std::vector<double> send((size_t)nrec), recv((size_t)nrec);
    int64_t k = 0;
    struct Done { int64_t *p; const int64_t &k; ~Done() { if (p) *p = k; } } report{done, k};   // also on the error returns
    ArmGuard guard{s};
    for (; k < n_iter; k++) {
        if (block && block->n >= block->cap) break;                    // the caller gathers the block, resets it, calls again
        double att = 0.0, Ad = -1.0, r = std::numeric_limits<double>::quiet_NaN(), swd = -1.0;
        int32_t sh = -1;
        auto pt_step = [&]() {
            if (!tamcmc_sampler_pt_due(s)) return (int)TAMCMC_OK;
            int32_t A, swapped = 0; double u, rT = 0.0;
            tamcmc_sampler_pt_draw(s, &A, &u);                        // same values in every process (replicated stream)
            att = 1.0; Ad = (double)A; sh = -2;                       // -2: attempted, this process owns neither end
            const bool ownA = (A >= off && A < off + nloc), ownB = (A + 1 >= off && A + 1 < off + nloc);
            int rc2 = TAMCMC_OK;
            if (ownA && ownB) {
                rc2 = tamcmc_sampler_pt_local(s, A, u, &swapped, &rT);
                if (rc2 != TAMCMC_OK) return rc2;
                r = rT; swd = (double)swapped; sh = 2 * A + swapped;
            } else if (ownA || ownB) {
                if (!exchange) return (int)TAMCMC_E_INVALID;
                const int mine = ownA ? A : A + 1, peer = ownA ? A + 1 : A;
                tamcmc_sampler_pt_export(s, mine, send.data());
                const double t0 = s->timing ? wall_now() : 0.0;
                rc2 = exchange(user, mine, peer, send.data(), recv.data(), nrec);
                if (s->timing) s->t_phase[7] += wall_now() - t0;
                if (rc2 != 0) return (int)TAMCMC_E_INVALID;
                rc2 = tamcmc_sampler_pt_import(s, A, u, recv.data(), &swapped, &rT);
                if (rc2 != TAMCMC_OK) return rc2;
                r = rT; swd = (double)swapped; sh = 2 * A + swapped;
            }
            return (int)TAMCMC_OK;
        };
        int rc;
        if (pipeline_on(s)) {
            const bool more = (k + 1 < n_iter) && !(block && block->n + 1 >= block->cap);
            rc = pipelined_iteration(s, more, moved_hist ? moved_hist + (size_t)k * nloc : nullptr, pt_step);
            if (rc != TAMCMC_OK) return rc;
        } else {
            // (chains spread over several processes: between this step and the next lies the boundary exchange, which
            // runs the caller's communication library on the same GPU while the armed launches would wait behind their
            // gate.  Nothing in that is known to synchronise the device, but it cannot be tried on the one-GPU boxes this
            // was developed on, and a device-wide wait under an armed batch costs the gate's whole patience: armed only
            // when this process owns the whole ladder, or on request -- TAMCMC_SAMPLER_ARM=2.)
            s->arm_next = (k + 1 < n_iter) && !(block && block->n + 1 >= block->cap) && (nloc == s->cfg.Nchains || s->arm_sharded);
            rc = tamcmc_sampler_mh_step(s);
            if (rc != TAMCMC_OK) return rc;
            if (moved_hist) std::memcpy(moved_hist + (size_t)k * nloc, s->moved.data(), (size_t)nloc);
            rc = pt_step();
            if (rc != TAMCMC_OK) return rc;
        }
        if (swap_hist) swap_hist[k] = sh;
        if (block) shard_block_record(block, s, att, Ad, r, swd);
        s->iter++;
    }
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_set_timing(tamcmc_sampler *s, int32_t enable)
{
    if (!s) return TAMCMC_E_INVALID;
    s->timing = enable != 0;
    for (double &t : s->t_phase) t = 0.0;
    s->t_iters = 0;
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_get_timing(const tamcmc_sampler *s, double seconds[8], int64_t *iterations)
{
    if (!s || !seconds) return TAMCMC_E_INVALID;
    for (int i = 0; i < 8; i++) seconds[i] = s->t_phase[i];
    if (iterations) *iterations = s->t_iters;
    return TAMCMC_OK;
}

extern "C" int tamcmc_sampler_get(const tamcmc_sampler *s, int32_t which, double *out, int64_t cap)
{
    if (!s || !out) return TAMCMC_E_INVALID;
    const std::vector<double> *v = nullptr;
    std::vector<double> tmp;
    switch (which) {
    case 0: v = &s->vars; break;
    case 1: v = &s->params; break;
    case 2: v = &s->logL; break;
    case 3: v = &s->logPrior; break;
    case 4: v = &s->logPost; break;
    case 5: v = &s->Pmove; break;
    case 6: v = &s->sigma; break;
    case 7: v = &s->mu; break;
    case 8: v = &s->covar; break;
    case 9: tmp.assign(s->T.begin() + s->cfg.chain_offset, s->T.begin() + s->cfg.chain_offset + s->nloc); v = &tmp; break;
    case 10: tmp.assign(s->moved.begin(), s->moved.end()); v = &tmp; break;
    default: return TAMCMC_E_INVALID;
    }
    if ((int64_t)v->size() > cap) return TAMCMC_E_INVALID;
    std::memcpy(out, v->data(), sizeof(double) * v->size());
    return TAMCMC_OK;
}

// Restored state (Config::read_restore_files -> Model_def ctor model_def.cpp:100-137, MALA::restore_proposal
// MALA.cpp:190-238): vars of every local chain (params rows follow), sigma, mu, covarmat.
extern "C" int tamcmc_sampler_set(tamcmc_sampler *s, int32_t which, const double *in, int64_t count)
{
    if (!s || !in) return TAMCMC_E_INVALID;
    const size_t n = (size_t)s->nloc, nv = (size_t)s->Nvars;
    s->proposed_ahead = false;      // proposals computed ahead used the state being replaced
    switch (which) {
    case 0:
        if ((size_t)count != n * nv) return TAMCMC_E_INVALID;
        std::memcpy(s->vars.data(), in, sizeof(double) * n * nv);
        for (size_t m = 0; m < n; m++)
            for (size_t k = 0; k < nv; k++) s->params[m * s->Nparams + s->index_to_relax[k]] = in[m * nv + k];
        return TAMCMC_OK;
    case 6:
        if ((size_t)count != n) return TAMCMC_E_INVALID;
        std::memcpy(s->sigma.data(), in, sizeof(double) * n);
        std::fill(s->chol_valid.begin(), s->chol_valid.end(), 0);   // Lchol factors (covar + eps2 I) * sigma
        return TAMCMC_OK;
    case 7:
        if ((size_t)count != n * nv) return TAMCMC_E_INVALID;
        std::memcpy(s->mu.data(), in, sizeof(double) * n * nv);
        return TAMCMC_OK;
    case 8:
        if ((size_t)count != n * nv * nv) return TAMCMC_E_INVALID;
        std::memcpy(s->covar.data(), in, sizeof(double) * n * nv * nv);
        std::fill(s->chol_valid.begin(), s->chol_valid.end(), 0);
        return TAMCMC_OK;
    default:
        return TAMCMC_E_INVALID;
    }
}

extern "C" int tamcmc_sampler_set_iteration(tamcmc_sampler *s, int64_t iteration)
{
    if (!s || iteration < 0) return TAMCMC_E_INVALID;
    s->iter = iteration;                       // initial_i of MALA.cpp:97-101 (do_restore_last_index)
    return TAMCMC_OK;
}

extern "C" double tamcmc_logP_primitive(int32_t id, const double p[4], double x)
{
    int err = 0;
    return (double)primitive(id, p, 4, x, &err);
}

extern "C" double tamcmc_log_prior(int32_t fct, int32_t Nparams, const double *params, const int32_t plength[11],
                                   const int32_t *sw, const double *pp, int32_t nrows, const double extra[4], int32_t *error)
{
    PriorSpec S;
    S.fct = fct; S.Nparams = Nparams; S.nrows = nrows;
    std::memcpy(S.plength, plength, sizeof(int) * 11);
    S.sw.assign(sw, sw + Nparams);
    if (nrows > 0) S.pp.assign(pp, pp + (size_t)nrows * Nparams);
    std::memcpy(S.extra, extra, sizeof(double) * 4);
    int err = 0;
    const double v = (double)log_prior(S, params, &err);
    if (error) *error = err;
    return v;
}

// test hook: consecutive r8vec_normal_01 calls of the given sizes after srand(seed); split = 0 uses the one-piece
// routine, 1 the draw / fill pair the sampler runs on several threads
extern "C" void tamcmc_normals(uint32_t seed, int32_t ncalls, const int32_t *sizes, double *out, int32_t split)
{
    Rng rng;
    rng.g.seed(seed);
    Rng::NormalPlan plan;
    for (int c = 0; c < ncalls; c++) {
        if (split) { rng.draw(sizes[c], plan); plan.fill(out); }
        else rng.normals(sizes[c], out);
        out += sizes[c];
    }
}

extern "C" void tamcmc_glibc_rand_jump(uint32_t seed, uint64_t skip, int32_t n, int32_t *out)
{
    GlibcRand g;
    g.seed(seed);
    g.jump(skip);
    for (int i = 0; i < n; i++) out[i] = g.next();
}

extern "C" void tamcmc_glibc_rand(uint32_t seed, int32_t n, int32_t *out)
{
    GlibcRand g;
    g.seed(seed);
    for (int i = 0; i < n; i++) out[i] = g.next();
}
