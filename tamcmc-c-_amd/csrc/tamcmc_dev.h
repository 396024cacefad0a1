// tamcmc_dev.h -- structures shared by the host side of the C ABI and the gfx950 kernels.
//
// Data layout in HBM (all fp64 unless noted), one set per ctx (= one star on one GPU):
//   x[Nx], y[Nx], lx[Nx]=log(x), isig2[Nx]=1/sigma_y^2 (chi_square only)      resident, written at create
//   params[Nchains][Nparams], Tcoefs[Nchains]                                 per call (or caller-resident)
//   mult[Nchains][n_mult]   : TmMult  -- per-chain multiplet table written by the setup kernel
//   noise[Nchains]          : TmNoise -- Harvey / white-noise / Gaussian terms per chain
//   part[Nchains][tiles][2] : per-tile partial sums of the likelihood (fixed-order reduction)
//   gmult[Nchains][tiles][n_mult][TM_GSLOTS], gnoise[Nchains][tiles][TM_NSLOTS] : gradient partials
//   hser[Nchains][tiles][TM_MAXH][TM_HSER] : per-profile series of u = 1/(1+t) in (log x - lxc), gradient path only
//   logL[Nchains], status[Nchains] (int32), grad[Nchains][Nvars]
#pragma once
#include <stddef.h>
#include <stdint.h>

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
    int32_t bg_exact;     // developer switch (env TAMCMC_BG_EXACT=1): no in-tile polynomials, exp() per bin and profile
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

// Per (chain, tile) descriptor written by the setup kernel and read by the eval kernel through scalar loads:
// everything a tile needs besides the multiplet records themselves.
struct TmTileRec {
    double bg[TM_PDEG + 1];   // background N0 + sum_h H_h/(1 + t_h) as a polynomial in (log x - lxc)
    double t0[TM_MAXH];       // t_h = (1e-3 tau_h x_c)^p_h at the tile centre
    double lxc;               // log x at the tile centre
    int32_t npoly;            // 1: polynomials valid on this tile (|p_h (log x - lxc)| <= 0.04, t0 finite), 0: use exp
    int32_t nact;             // multiplets whose window meets the tile; their indices are tidx[0..nact)
};
static_assert(sizeof(TmTileRec) == 8 * (TM_PDEG + 1 + TM_MAXH + 2), "TmTileRec layout");

struct TmEvalArgs {
    const double *x, *y, *lx, *isig2;
    const TmMult *mult;
    const TmNoise *noise;
    const TmTileRec *trec;      // [Nchains][tiles]
    const int32_t *tidx;        // [Nchains][tiles][n_mult] active multiplet indices, table order
    const int32_t *spec;        // NULL, or [Nchains]: which of the context's spectra (y, 1/sigma^2 blocks of Nx) a chain is fitted to
    const double *wt;            // [Nchains][2] {Tcoefs[chain], p/T or 2/T} copied by the setup kernel into device memory
    double *part;               // [Nchains][tiles][2]
    double *gmult;              // [Nchains][tiles][n_mult][TM_GSLOTS] or NULL
    double *gnoise;             // [Nchains][tiles][TM_NSLOTS] or NULL
    int32_t *ticket;            // [Nchains] arrival counters (zero between launches) for the in-launch finalize, or NULL
    double *logL;               // [Nchains] outputs of the in-launch finalize (likelihood-only path)
    int32_t *status;
    const int32_t *row_of_chain;// NULL or [Nchains]: row of model_out to fill, -1 none
    double *model_out;
    int32_t Nx, n_mult, tiles, likelihood_case;
    const int32_t *order;       // [Nchains][tiles] launch rank -> tile, costliest first (setup kernel), or NULL
    int32_t units, order_mode;  // the grid is cut into `units` sub-blocks of 256*KU bins (see TM_TILE_U0 below)
    int32_t tile_big, tile_small;   // sub-blocks of the even / odd tiles
    double like_p;
    unsigned long long tile_magic;   // ceil(2^40 / tiles): n / tiles == (n * tile_magic) >> 40 for n < 2^20 (slot -> tile rotation)
};

// Tile geometry.  The grid is cut into `units` sub-blocks of 256*KU bins; tiles alternate between `big` and `small`
// sub-blocks: tile 2k starts at sub-block k*(big+small) and owns `big` of them, tile 2k+1 owns the `small` ones that
// follow; the last tile is cut at `units`.  big == small gives uniform tiles.  Two sizes exist for the sake of the
// launch's tail: tiles are launched costliest-first, so the small ones run last and the launch ends on short
// workgroups (profiles/README.md).  The geometry depends on the grid only, never on the batch or the parameters.
#define TM_TILE_U0(t, big, small) ((((int)(t)) >> 1) * ((int)(big) + (int)(small)) + ((((int)(t)) & 1) ? (int)(big) : 0))
#define TM_TILE_S(t, big, small, units)                                                                              \
    ((((((int)(t)) & 1) ? (int)(small) : (int)(big)) < (int)(units) - TM_TILE_U0(t, big, small))                      \
         ? ((((int)(t)) & 1) ? (int)(small) : (int)(big))                                                             \
         : (int)(units) - TM_TILE_U0(t, big, small))
static inline int tm_tile_count(int units, int big, int small)
{
    const int P = big + small, n = units / P, rem = units - n * P;
    return 2 * n + (rem > 0 ? 1 : 0) + (rem > big ? 1 : 0);
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
// unit_bins / units / big / small: geometry of the eval launch that follows (the tile descriptors are built for it)
int tm_launch_setup(const TmLayout &L, int Nchains, const double *d_params, const double *d_Tcoefs, double *d_wt, const double *d_lx, int unit_bins,
                    int units, int big, int small, TmMult *d_mult, TmNoise *d_noise, TmTileRec *d_trec, int32_t *d_tidx,
                    void *d_chain_rec /* may be NULL */, void *d_aux /* may be NULL */, double *d_hser /* may be NULL */,
                    int32_t *d_order /* may be NULL */, void *stream);
size_t tm_sizeof_chain_rec();
size_t tm_sizeof_aux();
int tm_launch_eval(const TmEvalArgs &a, int Nchains, int KU, bool grad, void *stream);
// setup + eval in one launch; requires a.tiles == 1
int tm_launch_fused(const TmLayout &L, const TmFusedArgs &f, const TmEvalArgs &a, int Nchains, int KU, bool grad, void *stream);
// backward also performs the finalize step (logL, status) of the gradient path
int tm_launch_backward(const TmLayout &L, int Nchains, int unit_bins, int units, int big, int small, const double *d_params,
                       const double *d_Tcoefs, const void *d_chain_rec, const void *d_aux, const TmNoise *d_noise,
                       const double *d_part, const double *d_gmult, const double *d_gnoise,
                       const TmTileRec *d_trec, const double *d_hser, int Nvars, const int32_t *d_index_to_relax, double *d_grad, double *d_logL, int32_t *d_status,
                       void *stream);
}
#endif
