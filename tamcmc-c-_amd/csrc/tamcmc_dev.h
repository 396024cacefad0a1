// tamcmc_dev.h -- structures shared by the host side of the C ABI and the gfx950 kernels.
//
// Data layout in HBM (all fp64 unless noted), one set per ctx (= one star on one GPU):
//   x2[Nx]=2x, y[Nx], lx[Nx]=log(x), isig2[Nx]=1/sigma_y^2 (chi_square only)   resident, written at create
//   params[Nchains][Nparams], Tcoefs[Nchains]                                 per call (or caller-resident)
//   mult[Nchains][n_mult]   : TmMult  -- per-chain multiplet table written by the setup kernel
//   noise[Nchains]          : TmNoise -- Harvey / white-noise / Gaussian terms per chain
//   cell[Nchains][cells]    : TmCellRec -- background polynomial per fixed 4096-bin cell
//   thdr[Nchains][tiles]    : TmTileHdr -- per-chain tile boundaries (equal-cost tiles), active-multiplet counts
//   part[Nchains][tiles][4] : per-tile partial sums of the likelihood (fixed-order reduction)
//   gmult[Nchains][tiles][n_mult][TM_GSLOTS], gnoise[Nchains][tiles][2][TM_NSLOTS] : gradient partials
//   hser[Nchains][cells][TM_MAXH][TM_HSER] : per-profile series of u = 1/(1+t) in (log x - lxc), gradient path only
//   logL[Nchains], status[Nchains] (int32), grad[Nchains][Nvars]
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define TM_HD __host__ __device__
#else
#define TM_HD
#endif

#define TM_MAXM 7      // components of a multiplet: 2l+1, l <= 3 (build_lorentzian.cpp:74)
#define TM_MAXH 3      // Harvey profiles per chain (Nnoise = 3*Nharvey+1 = 10 in every .model file, io_ms_global.cpp)
#define TM_THREADS 256 // threads per workgroup of the eval kernel (4 waves of 64)
#define TM_MAXMULT 256 // multiplets per chain (Nmax*(lmax+1) or sum Nfl); staged in LDS, 160 B each
#define TM_GSLOTS 24   // gradient partials per (tile, multiplet): 3 per component + 3 asymmetry sums
#define TM_NSLOTS 16   // gradient partials per tile for the noise terms: 3 per Harvey + N0 (+pad)
#define TM_PDEG 8      // degree of the in-tile Taylor polynomials of the Harvey profiles
#define TM_ORDER_MAX 1024   // tile counts up to this are launched costliest-first (rank table built in LDS)
#define TM_HSER 10     // per-profile series coefficients kept for the gradient path (degree TM_PDEG + 1: one more for the derivative)

// model families (how the params row is unpacked)
#define TM_FAM_GAUSS  0  // ids 0, 1          models.cpp:1968-2034
#define TM_FAM_GLOBAL 1  // ids 2,3,6-10,12,13 models.cpp:15-1674
#define TM_FAM_LOCAL  2  // ids 11, 14        models.cpp:1683-1963

struct TmLayout {
    int32_t model_case;   // models_ctrl.list id
    int32_t family;
    int32_t variant;      // 0 a1etaa3, 1 a1l_etaa3, 2 a1etaa3_v2 (heights per m)   build_lorentzian.cpp
    int32_t Nmax, lmax;   // plength[0], plength[1] (local models: Nvis)
    int32_t Nfl[4];       // plength[2..5]
    int32_t Nsplit, Nwidth, Nnoise, Ninc;
    int32_t off_f[4];     // first index of the degree-l frequencies
    int32_t s, w, z, q;   // splitting, widths, noise, inclination blocks
    int32_t Nparams;      // sum(plength)
    int32_t n_mult;       // multiplets per chain
    int32_t nharvey;      // Harvey profiles evaluated (0 for local models, models.cpp:1818)
    int32_t likelihood_case;
    int32_t Nx;
    int32_t bg_exact;     // developer switch (env TAMCMC_BG_EXACT=1): no in-cell polynomials, exp() per bin and profile
    int32_t asym_var;     // the asymmetry parameter (params[s+5]) is one of the gradient variables (tamcmc_ctx_set_vars):
                          // the gradient launch then takes the asymmetric code path even where asym == 0 -- the factor
                          // A(x) is 1 there, but dA/d(asym) = 2 (x/f - 1) is not
    int32_t pad_l;
    double  x0, xlast, step;  // x[0], x[Nx-1], x[1]-x[0] (models.cpp:489)
    double  like_p;           // (double)(long)likelihood_params
};

struct TmMult {
    double g2;            // Gamma^2
    double aA, aB, c2;    // asymmetry: a = x*aA + aB, A(x) = a*a + c2   (build_lorentzian.cpp:96)
    double nu2[TM_MAXM];  // 2*nu_lm
    double hq[TM_MAXM];   // h_lm * Gamma^2
    int32_t imin, imax;   // truncation window [imin, imax)  (build_lorentzian.cpp:423-427)
    int32_t ncomp;        // 2l+1
    int32_t has_asym;
};
static_assert(sizeof(TmMult) == 160, "TmMult layout");
#define TM_MULT_DOUBLES 20

struct TmNoise {
    double H[TM_MAXH];    // |H_k|                 (harvey1985: |H_k|*|tau_k|)
    double lt[TM_MAXH];   // log(1e-3*|tau_k|)     (harvey1985: log(1e-3*2pi*|tau_k|))
    double p[TM_MAXH];    // |p_k|
    double N0;            // white noise (last noise parameter)
    double gA, gnu0, gs2; // Gaussian term amp*exp(-0.5 (x-nu0)^2 / s2) of ids 0, 1
    int32_t nh;           // active Harvey profiles (tau != 0)
    int32_t has_gauss;
    int32_t status;       // TAMCMC_CHAIN_* from the setup kernel
    int32_t pad;
};

static_assert(sizeof(TmNoise) == 120, "TmNoise layout");
#define TM_NOISE_DOUBLES 15

// ---- geometry --------------------------------------------------------------------------------------------
// The grid is cut into UNITS of TM_UNIT_BINS = 512 bins (two rows of 256: thread t of a workgroup owns bins
// unit*512 + k*256 + t, k = 0, 1, so every load is a coalesced 8-byte-per-lane stream).
// CELLS of TM_CELL_UNITS = 8 units (4096 bins) are a fixed function of the grid: each carries the background as a
// polynomial in (log x - log x_centre) (and, for the gradient, the Harvey series the backward kernel needs).
// TILES -- the work of one workgroup -- are runs of at most TM_TILE_MAXU consecutive units whose boundaries the setup
// kernel chooses PER CHAIN so that every tile of a chain costs about the same (prefix sum of the per-unit cost the
// chain's truncation windows imply): tile count and cell geometry depend on the grid only, a chain's boundaries on that
// chain's own parameters only -- never on the batch -- so a chain's result does not change with the chains beside it.
#define TM_UNIT_BINS 512
#define TM_UNIT_SHIFT 9
#define TM_CELL_UNITS 8
#define TM_CELL_SHIFT 3
#ifndef TM_TILE_MAXU
#define TM_TILE_MAXU 8      // units per gradient tile at most: 32 KB of weights in LDS; a tile meets at most 2 cells
#endif
#define TM_TILE_MAXU_L 16   // units per likelihood-only tile at most (no weights to keep)
#define TM_EQ_MAXU 4096     // grids of up to this many units (2M bins) get equal-cost tiles (cost prefix lives in LDS)

// Per (chain, cell) record written by the setup kernel and read by the eval kernel through scalar loads.
struct TmCellRec {
    double bg[TM_PDEG + 1];   // background N0 + sum_h H_h/(1 + t_h) as a polynomial in (log x - lxc)
    double t0[TM_MAXH];       // t_h = (1e-3 tau_h x_c)^p_h at the cell centre
    double lxc;               // log x at the cell centre
    int32_t npoly;            // 1: polynomials valid on this cell (|p_h (log x - lxc)| <= 0.04, t0 finite), 0: use exp
    int32_t pad;
};
static_assert(sizeof(TmCellRec) == 8 * (TM_PDEG + 1 + TM_MAXH + 2), "TmCellRec layout");

// Per (chain, tile) header: the tile's units [u0, u1), the number of multiplets whose window meets it (their indices
// are tidx[0..nact)) and its cost (launch-rank key).
struct TmTileHdr {
    int32_t u0, u1, nact, cost;
};
static_assert(sizeof(TmTileHdr) == 16, "TmTileHdr layout");

// One entry of a tile's active list: the multiplet's index in the chain's table and -- so that the eval kernel can test
// the window and pick the code path without waiting for the record itself -- its window and shape.  One 16-byte scalar load.
struct TmActive {
    int32_t idx;          // index into mult[chain][..]
    int32_t imin, imax;   // truncation window [imin, imax)
    int32_t shape;        // ncomp | (has_asym << 8)
};
static_assert(sizeof(TmActive) == 16, "TmActive layout");

// Cost model of the tile balancer, in VALU instructions per bin: a unit costs c0 + sum over the multiplets whose window
// meets it of (a * ncomp + b).
struct TmCostModel {
    int32_t c0, a, b, pad;
    // equal-length tiles of TWO sizes (tail shaping): tiles [0, t1) have su1 units, tiles [t1, tiles) su2 <= su1 units;
    // t1 = 0: all tiles alike, ceil(units / tiles) units each.  A function of the grid only, like the tile count.
    int32_t t1, su1, su2, pad2;
};
// first unit of tile t / tile holding unit u under that geometry (su = ceil(units / tiles) for the one-size case)
static inline TM_HD int tm_tile_first_unit(const TmCostModel &g, int su, int t) { return (g.t1 <= 0) ? t * su : (t < g.t1 ? t * g.su1 : g.t1 * g.su1 + (t - g.t1) * g.su2); }
static inline TM_HD int tm_tile_units(const TmCostModel &g, int su, int t) { return (g.t1 <= 0) ? su : (t < g.t1 ? g.su1 : g.su2); }
static inline TM_HD int tm_tile_of_unit(const TmCostModel &g, int su, int u) { return (g.t1 <= 0) ? u / su : (u < g.t1 * g.su1 ? u / g.su1 : g.t1 + (u - g.t1 * g.su1) / g.su2); }

struct TmEvalArgs {
    const double *x2, *y, *lx, *isig2;   // x2 = 2 x (the kernels only ever need d = 2x - 2nu), log x, 1 / sigma^2
    const TmMult *mult;
    const TmNoise *noise;
    const TmCellRec *cell;      // [Nchains][cells]
    const TmTileHdr *thdr;      // [Nchains][tiles]
    const TmActive *tidx;       // [Nchains][tiles][n_mult] active multiplets, table order
    const int32_t *spec;        // NULL, or [Nchains]: which of the context's spectra (y, 1/sigma^2 blocks of Nx) a chain is fitted to
    const double *wt;           // [Nchains][2] {Tcoefs[chain], p/T or 2/T} copied by the setup kernel into device memory
    double *part;               // [Nchains][tiles][4]: sum y/M, sum log M as (mantissa product, exponent sum), pad
    double *gmult;              // [Nchains][tiles][n_mult][TM_GSLOTS] or NULL
    double *gnoise;             // [Nchains][tiles][2][TM_NSLOTS] or NULL: one set per cell the tile meets (at most 2)
    int32_t *ticket;            // [Nchains] arrival counters (zero between launches) for the in-launch finalize, or NULL
    double *logL;               // [Nchains] outputs of the in-launch finalize (likelihood-only path)
    int32_t *status;
    const int32_t *row_of_chain;// NULL or [Nchains]: row of model_out to fill, -1 none
    double *model_out;
    int32_t Nx, n_mult, tiles, cells, likelihood_case;
    const int32_t *order;       // [Nchains][tiles] launch rank -> tile, costliest first (setup kernel), or NULL
    int32_t order_mode;
    int32_t prio;               // 1: issue priority by launch rank (s_setprio), 0: all workgroups alike
    int32_t generic;            // 0: chi(2,2p), no Gaussian term, no model rows -> the specialised eval kernel (tamcmc_eval_body.h, GEN)
    double like_p;
    unsigned long long tile_magic;   // ceil(2^40 / tiles): n / tiles == (n * tile_magic) >> 40 for n < 2^20 (slot -> tile rotation)
};

static inline int tm_units(long long Nx) { return (int)((Nx + TM_UNIT_BINS - 1) / TM_UNIT_BINS); }
static inline int tm_cells(int units) { return (units + TM_CELL_UNITS - 1) / TM_CELL_UNITS; }
// Tile counts (a function of the grid only).  Grids of <= 4 units (2048 bins) are one tile (then the prologue and the
// evaluation share a launch, tamcmc_fused.hip); short grids get ~10 tiles per chain (a workgroup's run time is the
// floor of a launch); long grids tiles of TM_TILE_MAXU = 8 units -- 25 tiles at 1e5 bins (196 units) for both
// launches: the per-tile costs (prologue, and in the gradient kernel one wave reduction per multiplet: 14 % of that
// kernel at 6 units per tile) are spread over as many bins as the 32 KB of gradient weights allow; measured flat or
// slower for every other count (profiles/README.md).  tiles * TM_TILE_MAXU >= units always holds.
static inline int tm_tiles(int units, int grad)
{
    (void)grad;
    if (units <= 4) return 1;
    if (units < 70) { const int s = units / 10 > 0 ? units / 10 : 1; return (units + s - 1) / s; }
    return (units + TM_TILE_MAXU - 1) / TM_TILE_MAXU;
}

#if defined(__HIPCC__)
// sum of log M over a tile from its (mantissa product, exponent sum): log(mant) + e ln 2 with ONE rounding of the
// product-sum (explicit fma), so that the likelihood launch's in-launch finalize (tamcmc_eval_body.h, contracted build)
// and the gradient path's finalize in the backward kernel (built with -ffp-contract=off) give the same bits.
#define TM_LN2 0.693147180559945309417232
static __device__ __forceinline__ double tm_tile_logsum(double mant, double e) { return __builtin_fma(e, TM_LN2, log(mant)); }
#endif
// Whether the setup kernel balances a chain's tiles (per-chain boundaries) or cuts tiles of equal length: the balancer
// needs slack (tiles * TM_TILE_MAXU > units), its bound is TM_TILE_MAXU units per tile, and its tables live in LDS.
// One predicate for the setup launcher, the setup body and the backward kernel (which finds the tiles a window meets
// by division when they have equal length, by a search over the tile starts otherwise).
static inline TM_HD int tm_setup_balances(int units, int tiles, int equal_cost, int max_units_per_tile)
{
    return (equal_cost != 0 && max_units_per_tile == TM_TILE_MAXU && tiles > 1 && tiles <= TM_ORDER_MAX && units <= TM_EQ_MAXU &&
            (long long)tiles * TM_TILE_MAXU > units) ? 1 : 0;
}

// Equal-cost tile boundary t (0 < t < tiles) from the INCLUSIVE prefix `pre` of the unit costs (C = pre[units-1] the
// total, cmin the cheapest unit): the smallest u in [0, units] with Q'(u) >= t/T of the total, where Q' is the prefix
// of cost(u) + lambda and lambda >= 0 is the uniform surcharge that keeps every tile within S = TM_TILE_MAXU units:
// with D = T S - U (> 0 by the tile count) and N = max(0, C - T S cmin), lambda = N / D gives
//     (tile size - 1) * (cmin + lambda) < (C + lambda U) / T = S (cmin + lambda),   i.e. size <= S.
// Integer arithmetic throughout (cross-multiplied in 64 bit: T <= 2^10, D <= 2^13, C < 2^31 -- the int32 prefix; the
// accepted cost models keep it there, tamcmc_api.cpp: env_cost -- so every product stays below 2^55), so the boundaries are
// a pure function of the costs.  Shared by the setup kernel and tests/cpp/geometry_check.cpp.
static inline TM_HD int tm_tile_bound(int t, int tiles, int units, const int *pre, long long C, long long cmin)
{
    const long long T = tiles, U = units, S = TM_TILE_MAXU;
    const long long D = T * S - U;
    const long long Nl = (C - T * S * cmin > 0) ? C - T * S * cmin : 0;
    const long long rhs = (long long)t * (D * C + Nl * U);
    int lo = 0, hi = units;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const long long q = (mid == 0) ? 0 : (long long)pre[mid - 1];
        if (T * (D * q + Nl * mid) >= rhs) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// What the fused small-grid launch (tamcmc_fused.hip) needs on top of TmEvalArgs: the inputs and the gradient-path
// outputs of the prologue.
struct TmFusedArgs {
    const double *params, *Tcoefs;
    void *chain_rec, *aux;      // NULL on the likelihood-only path
    double *hser;               // NULL on the likelihood-only path
    int32_t p_doubles, pad;     // LDS doubles reserved for the params row (Nparams rounded up to even)
};

#ifdef __cplusplus
extern "C++" {
// launchers implemented in the .hip files
struct ihipStream_t;
// units / cells / tiles: geometry of the eval launch that follows (tile headers and active lists are built for it);
// equal_cost = 0 cuts every chain into tiles of equal length instead
int tm_launch_setup(const TmLayout &L, int Nchains, const double *d_params, const double *d_Tcoefs, double *d_wt, const double *d_lx,
                    int units, int cells, int tiles, int equal_cost, TmCostModel cm, TmMult *d_mult, TmNoise *d_noise, TmCellRec *d_cell,
                    TmTileHdr *d_thdr, TmActive *d_tidx, void *d_chain_rec /* may be NULL */, void *d_aux /* may be NULL */,
                    double *d_hser /* may be NULL */, int32_t *d_order /* may be NULL */, void *stream);
size_t tm_sizeof_chain_rec();
size_t tm_sizeof_aux();
int tm_launch_eval(const TmEvalArgs &a, int Nchains, bool grad, void *stream);
// setup + eval in one launch; requires a.tiles == 1
int tm_launch_fused(const TmLayout &L, const TmFusedArgs &f, const TmEvalArgs &a, int Nchains, bool grad, void *stream);
// backward also performs the finalize step (logL, status) of the gradient path
int tm_launch_backward(const TmLayout &L, int Nchains, int units, int cells, int tiles, int equal_cost /* as given to tm_launch_setup */,
                       TmCostModel geom /* the cost model / tile geometry given to tm_launch_setup */,
                       const double *d_params,
                       const double *d_Tcoefs, const void *d_chain_rec, const void *d_aux, const TmNoise *d_noise,
                       const double *d_part, const double *d_gmult, const double *d_gnoise, const TmCellRec *d_cell,
                       const TmTileHdr *d_thdr, const double *d_hser, int Nvars, const int32_t *d_index_to_relax, double *d_grad,
                       double *d_logL, int32_t *d_status, void *stream);
}
#endif
