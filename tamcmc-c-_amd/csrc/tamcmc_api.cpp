// tamcmc_api.cpp -- host side of the C ABI declared in include/tamcmc_accel.h.
// Owns the device buffers of one context, validates arguments the way Model_def's callers rely on
// (plength sums to Nparams, model / likelihood ids from the *.list tables), and strings the launches of
// one evaluation on the context stream:
//     setup (params -> multiplet table, tile descriptors)
//     -> eval (model + likelihood; the last workgroup of a chain sums its tiles in fixed order: -p(..)/T)
//     or eval<grad> (+ gradient partials) -> backward (tile sums, chain rule to d/dvars, logL)
// No CPU fallback exists in this library.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "tamcmc_accel.h"
#include "tamcmc_dev.h"

static thread_local char g_hip_err[256] = "";

#define TM_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            snprintf(g_hip_err, sizeof(g_hip_err), "%s -> %s", #call, hipGetErrorString(e_)); \
            return TAMCMC_E_HIP;                                                             \
        }                                                                                    \
    } while (0)

struct tamcmc_ctx {
    int device = 0;
    TmLayout L{};
    // Geometry (tamcmc_dev.h): units of 512 bins, cells of 8 units, tiles_l / tiles_g tiles per chain for the likelihood-only
    // and the gradient launch -- functions of the grid alone; a chain's tile BOUNDARIES are chosen by the setup kernel.
    int units = 0, cells = 0;
    int tiles_l = 1, tiles_g = 1;
    int tiles_max = 1;
    int equal_cost = 0;            // TAMCMC_EQUAL_COST=1: per-chain tile boundaries of equal cost instead of equal length
    int prio = 0;                  // TAMCMC_PRIO=1: issue priority by launch rank (s_setprio)
    TmCostModel cost_l{60, 5, 9, TM_TILE_MAXU_L}, cost_g{110, 13, 24, TM_TILE_MAXU};   // (.pad = units per tile at most)   // VALU instructions per bin: c0 + sum(a * ncomp + b) (TAMCMC_COST / TAMCMC_COST_GRAD)
    int last_tiles = 0;            // T of the most recent likelihood-only call (tamcmc_ctx_geometry)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // resident data
    double *d_x2 = nullptr, *d_y = nullptr, *d_lx = nullptr, *d_isig2 = nullptr;   // 2 x, y, log x, 1 / sigma^2
    int nspec = 1;                 // spectra resident in d_y / d_isig2 (blocks of Nx); tamcmc_ctx_set_spectra
    int32_t *d_spec = nullptr;     // [spec_n] chain -> spectrum map (tamcmc_ctx_set_chain_spectrum), or NULL: all chains use spectrum 0
    int spec_n = 0;
    // per-batch buffers (capacity in chains)
    int cap = 0;
    bool cap_grad = false;
    void *d_slab = nullptr;        // one allocation behind every per-batch buffer below (ensure_capacity)
    double *d_params = nullptr, *d_T = nullptr, *d_logL = nullptr, *d_part = nullptr;
    double *d_gmult = nullptr, *d_gnoise = nullptr, *d_hser = nullptr;
    int32_t *d_order = nullptr; int order_mode = 2;
    int fuse = 1;                  // one tile per chain -> prologue and evaluation in one launch (TAMCMC_FUSED=0 disables)
    int32_t *d_status = nullptr, *d_rows = nullptr;
    TmMult *d_mult = nullptr;
    TmNoise *d_noise = nullptr;
    void *d_chain_rec = nullptr, *d_aux = nullptr;   // TmChain / TmMultFull records kept for the backward kernel
    double *d_wt = nullptr;        // [cap][2] {T, wscale} device copies written by the setup kernel
    int32_t *d_ticket = nullptr;   // [cap] arrival counters of the in-launch finalize (kept at zero between launches)
    TmCellRec *d_cell = nullptr;   // [cap][cells] background polynomials
    TmTileHdr *d_thdr = nullptr;   // [cap][tiles_max] tile headers (per-chain boundaries)
    TmActive *d_tidx = nullptr;    // [cap][tiles_max][n_mult] active multiplet lists
    double *d_model = nullptr;
    size_t model_cap = 0;
    // host-pointer entry point: pinned, device-mapped staging the kernels read / write directly over PCIe
    // (no copy-engine round trips): h_in = [params | Tcoefs], h_out = [logL | grad], h_status
    double *h_in = nullptr, *h_out = nullptr;
    int32_t *h_status = nullptr;
    int h_cap = 0, h_nvars = -1;
    double *dv_in = nullptr, *dv_out = nullptr;   // device views of h_in / h_out / h_status (looked up once per allocation)
    int32_t *dv_status = nullptr;
    hipEvent_t ev_done = nullptr;  // completion of a host-pointer call, polled (see wait_done)
    bool ev_recorded = false;      // wait_data: the event of the current call has been recorded (lazily)
    int in_flight = 0;             // chains of a tamcmc_eval_batch_begin not yet collected by _end
    int armed = 0;                 // chains of a tamcmc_eval_batch_arm whose launches wait behind the gate for _fire
    uint32_t *h_gate = nullptr, *dv_gate = nullptr;   // pinned word the gate kernel watches
    uint32_t gate_seq = 0;         // value that opens the gate of the armed batch
    int gate_patience = 1 << 21;   // polls (~2 us each) before the gate gives up; TAMCMC_GATE_PATIENCE (tests)
    // tamcmc_eval_batch_begin_part / _end_part: two sub-batches of the context's chains in flight at once, part 1 on a
    // stream of its own; a part's rows of every per-chain buffer start at its first chain
    hipStream_t part_streams[TAMCMC_MAX_PARTS] = {};      // [0] unused (part 0 runs on the context stream), created on demand
    int part_first[TAMCMC_MAX_PARTS] = {}, part_n[TAMCMC_MAX_PARTS] = {};
    bool part_ev_recorded[TAMCMC_MAX_PARTS] = {};
    hipEvent_t part_ev[TAMCMC_MAX_PARTS] = {};
    // (an armed batch counts: nothing but _fire / _disarm / _end may touch the context while launches wait behind a gate)
    bool parts_busy() const { if (armed) return true; for (int n : part_n) if (n) return true; return false; }
    // variables
    int Nvars = 0;
    int32_t *d_relax = nullptr;
    // shader-clock probe (tamcmc_ctx_clock_probe_begin / _end): one wave on its own stream beside the evaluation
    hipStream_t probe_stream = nullptr;
    unsigned long long *h_probe = nullptr, *dv_probe = nullptr;   // pinned: {core cycles, 100 MHz ticks}
    // profiling
    bool profile = false;
    int profile_stride = 1;       // events around every n-th eval launch (tamcmc_ctx_profile(ctx, n))
    long long profile_count = 0;
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
    size_t ev_used = 0;
};

static int pick_tiles(const tamcmc_ctx *c, int Nchains, bool grad);

static int model_supported(int id)
{
    if (id == 4 || id == 5) return TAMCMC_E_MODEL_DISABLED;
    if (id < 0 || id > 14) return TAMCMC_E_UNKNOWN_MODEL;
    return TAMCMC_OK;
}

// Build the layout of the params row (models.cpp:492-506, :1696-1710).
static int build_layout(TmLayout &L, int model_case, int likelihood_case, double like_p, const int32_t plength[11],
                        int64_t Nx, const double *x)
{
    std::memset(&L, 0, sizeof(L));
    L.model_case = model_case;
    L.likelihood_case = likelihood_case;
    L.like_p = (double)(long)like_p;           // `long p`, likelihoods.cpp:17
    L.Nx = (int32_t)Nx;
    L.x0 = x[0];
    L.xlast = x[Nx - 1];
    L.step = x[1] - x[0];
    int sum = 0;
    for (int i = 0; i < 11; i++) { if (plength[i] < 0) return TAMCMC_E_INVALID; sum += plength[i]; }
    L.Nparams = sum;
    if (model_case == 0 || model_case == 1) {
        L.family = TM_FAM_GAUSS;
        L.n_mult = 0;
        if (sum < (model_case == 0 ? 4 : 7)) return TAMCMC_E_INVALID;
        L.nharvey = (model_case == 1) ? 1 : 0;
        return TAMCMC_OK;
    }
    L.family = (model_case == 11 || model_case == 14) ? TM_FAM_LOCAL : TM_FAM_GLOBAL;
    L.variant = (model_case == 6 || model_case == 7 || model_case == 8) ? 1 : ((model_case == 13 || model_case == 14) ? 2 : 0);
    L.Nmax = plength[0];
    L.lmax = plength[1];
    for (int l = 0; l < 4; l++) L.Nfl[l] = plength[2 + l];
    L.Nsplit = plength[6]; L.Nwidth = plength[7]; L.Nnoise = plength[8]; L.Ninc = plength[9];
    const int Nf = L.Nfl[0] + L.Nfl[1] + L.Nfl[2] + L.Nfl[3];
    L.off_f[0] = L.Nmax + L.lmax;
    for (int l = 1; l < 4; l++) L.off_f[l] = L.off_f[l - 1] + L.Nfl[l - 1];
    L.s = L.Nmax + L.lmax + Nf;
    L.w = L.s + L.Nsplit;
    L.z = L.w + L.Nwidth;
    L.q = L.z + L.Nnoise;
    if (L.q + L.Ninc + 2 > sum) return TAMCMC_E_INVALID;   // trunc_c and do_amp must exist
    if (L.Nsplit < 6) return TAMCMC_E_INVALID;
    if (L.Nnoise < 1) return TAMCMC_E_INVALID;
    if (L.family == TM_FAM_GLOBAL) {
        if (L.lmax < 0 || L.lmax > 3 || L.Nmax < 1) return TAMCMC_E_INVALID;
        for (int l = 0; l <= L.lmax; l++) if (L.Nfl[l] != L.Nmax) return TAMCMC_E_INVALID; // models.cpp:485-486
        L.n_mult = L.Nmax * (L.lmax + 1);
        if (L.n_mult > TM_MAXMULT) return TAMCMC_E_INVALID;
        L.nharvey = (L.Nnoise - 1) / 3;
        if (L.nharvey > TM_MAXH) return TAMCMC_E_INVALID;
        const bool interp = !(model_case == 9 || model_case == 10);
        if (interp && L.lmax >= 1 && L.Nfl[0] < 2) return TAMCMC_E_INVALID;  // lin_interpol needs two nodes
        if (interp && L.Nwidth < L.Nmax) return TAMCMC_E_INVALID;
        if (model_case == 9 && L.Nwidth < 5) return TAMCMC_E_INVALID;
        if (model_case == 10 && L.Nwidth < 6) return TAMCMC_E_INVALID;
        if (model_case == 6 && L.Nsplit < 7) return TAMCMC_E_INVALID;
        if (model_case == 7 && L.Nsplit < 6 + L.Nmax) return TAMCMC_E_INVALID;
        if (model_case == 8 && L.Nsplit < 6 + 2 * L.Nmax) return TAMCMC_E_INVALID;
        if (model_case == 12 && L.Ninc < 2 + (L.lmax >= 2 ? 3 : 0) + (L.lmax >= 3 ? 4 : 0)) return TAMCMC_E_INVALID;
        if (model_case == 13 && L.lmax >= 1 && L.Ninc < (L.lmax + 1) * (L.Nmax - 1) + L.lmax + 1) return TAMCMC_E_INVALID;
        if ((model_case == 3 || model_case == 6 || model_case == 7 || model_case == 8) && L.Ninc < 1) return TAMCMC_E_INVALID;
    } else {
        L.n_mult = Nf;
        if (L.n_mult > TM_MAXMULT) return TAMCMC_E_INVALID;
        L.nharvey = 0;                         // models.cpp:1818
        if (Nf < 1) return TAMCMC_E_INVALID;
        if (L.Nwidth < Nf) return TAMCMC_E_INVALID;
        if (model_case == 11 && L.Nmax < Nf) return TAMCMC_E_INVALID;
        if (model_case == 14) {
            int off = 0;
            for (int l = 0; l < 4; l++) {
                if (L.Nfl[l] > 0 && off + (l + 1) * (L.Nfl[l] - 1) + l >= sum) return TAMCMC_E_INVALID;
                off += L.Nfl[l];
            }
        }
    }
    return TAMCMC_OK;
}

// The per-batch buffers live in ONE allocation (a slab, carved at 256-byte boundaries): the kernels of a step touch a
// dozen of them within their first microseconds, and every separate hipMalloc sits in pages of its own.
static void free_batch(tamcmc_ctx *c)
{
    (void)hipFree(c->d_slab); c->d_slab = nullptr;
    c->d_params = c->d_T = c->d_logL = c->d_part = c->d_gmult = c->d_gnoise = c->d_hser = c->d_wt = nullptr;
    c->d_status = c->d_rows = c->d_order = c->d_ticket = nullptr;
    c->d_mult = nullptr; c->d_noise = nullptr; c->d_cell = nullptr; c->d_thdr = nullptr; c->d_tidx = nullptr;
    c->d_chain_rec = c->d_aux = nullptr;
    c->cap = 0; c->cap_grad = false;
}

static int ensure_capacity(tamcmc_ctx *c, int Nchains, bool grad)
{
    if (Nchains <= c->cap && (!grad || c->cap_grad)) return TAMCMC_OK;
    const int cap = Nchains > c->cap ? Nchains : c->cap;
    const bool g = grad || c->cap_grad;
    TM_HIP(hipStreamSynchronize(c->stream));
    for (hipStream_t ps : c->part_streams) if (ps) TM_HIP(hipStreamSynchronize(ps));
    free_batch(c);
    const size_t n = (size_t)cap;
    const size_t nm = (size_t)(c->L.n_mult > 0 ? c->L.n_mult : 1);
    // two passes over the same list: sizes first, then the pointers
    char *base = nullptr;
    size_t off = 0;
    auto carve = [&](auto **ptr, size_t bytes) {
        if (base) *ptr = reinterpret_cast<std::remove_reference_t<decltype(**ptr)> *>(base + off);
        off += (bytes + 255) & ~(size_t)255;
    };
    auto layout = [&]() {
        off = 0;
        carve(&c->d_params, n * c->L.Nparams * sizeof(double));
        carve(&c->d_T, n * sizeof(double));
        carve(&c->d_logL, n * sizeof(double));
        carve(&c->d_wt, n * 2 * sizeof(double));
        carve(&c->d_status, n * sizeof(int32_t));
        carve(&c->d_rows, n * sizeof(int32_t));
        carve(&c->d_ticket, n * sizeof(int32_t));
        carve(&c->d_noise, n * sizeof(TmNoise));
        carve(&c->d_order, n * c->tiles_max * sizeof(int32_t));
        carve(&c->d_thdr, n * c->tiles_max * sizeof(TmTileHdr));
        carve(&c->d_part, n * c->tiles_max * 4 * sizeof(double));
        carve(&c->d_cell, n * c->cells * sizeof(TmCellRec));
        carve(&c->d_mult, n * nm * sizeof(TmMult));
        carve(&c->d_tidx, n * c->tiles_max * nm * sizeof(TmActive));
        if (g) {
            char *cr = nullptr, *ax = nullptr;
            carve(&cr, n * tm_sizeof_chain_rec());
            carve(&ax, n * nm * tm_sizeof_aux());
            if (base) { c->d_chain_rec = cr; c->d_aux = ax; }
            carve(&c->d_gnoise, n * c->tiles_g * 2 * TM_NSLOTS * sizeof(double));
            carve(&c->d_hser, n * c->cells * TM_MAXH * TM_HSER * sizeof(double));
            carve(&c->d_gmult, n * c->tiles_g * nm * TM_GSLOTS * sizeof(double));
        }
    };
    layout();
    const size_t total = off;
    if (hipMalloc(&c->d_slab, total) != hipSuccess) { c->d_slab = nullptr; free_batch(c); return TAMCMC_E_NOMEM; }
    base = static_cast<char *>(c->d_slab);
    layout();
    TM_HIP(hipMemset(c->d_ticket, 0, n * sizeof(int32_t)));
    c->cap = cap;
    c->cap_grad = g;
    return TAMCMC_OK;
}

extern "C" int tamcmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int tamcmc_ctx_create(tamcmc_ctx **out, int device_id, int model_case, int likelihood_case,
                                 double likelihood_p, const int32_t plength[11], int64_t Nx,
                                 const double *x, const double *y, const double *sigma_y)
{
    if (!out) return TAMCMC_E_INVALID;
    *out = nullptr;
    if (!plength || !x || !y || Nx < 2 || Nx > (1LL << 28)) return TAMCMC_E_INVALID;   // 32-bit byte offsets in the eval kernel (the reference reads at most 1e6 rows, config.cpp:531)
    int rc = model_supported(model_case);
    if (rc != TAMCMC_OK) return rc;
    if (likelihood_case != 0 && likelihood_case != 1) return TAMCMC_E_UNKNOWN_MODEL;
    if (likelihood_case == 1 && !sigma_y) return TAMCMC_E_INVALID;
    const int ndev = tamcmc_device_count();
    if (ndev <= 0 || device_id < 0 || device_id >= ndev) return TAMCMC_E_NODEVICE;

    tamcmc_ctx *c = new (std::nothrow) tamcmc_ctx();
    if (!c) return TAMCMC_E_NOMEM;
    c->device = device_id;
    rc = build_layout(c->L, model_case, likelihood_case, likelihood_p, plength, Nx, x);
    if (rc != TAMCMC_OK) { delete c; return rc; }

    auto env_int = [](const char *name, int lo, int hi, int *dst) {
        const char *e = getenv(name);
        if (e) { int v = atoi(e); if (v >= lo && v <= hi) *dst = v; }
    };
    // Developer switches (documented in include/tamcmc_accel.h); none of them changes a result beyond rounding.
    { int v = 0; env_int("TAMCMC_BG_EXACT", 0, 1, &v); c->L.bg_exact = v; }
    env_int("TAMCMC_ORDER", 0, 2, &c->order_mode);
    env_int("TAMCMC_FUSED", 0, 1, &c->fuse);
    env_int("TAMCMC_EQUAL_COST", 0, 1, &c->equal_cost);
    env_int("TAMCMC_GATE_PATIENCE", 1, 1 << 30, &c->gate_patience);
    env_int("TAMCMC_PRIO", 0, 1, &c->prio);
    auto env_cost = [](const char *name, TmCostModel *m) {
        const char *e = getenv(name);
        int c0, a, b;
        // accepted only while the balancer's int32 cost prefix cannot overflow: the worst unit costs
        // c0 + TM_MAXMULT * (7 a + b) and at most TM_EQ_MAXU units are balanced (tm_tile_bound itself works in 64 bit);
        // a model outside that range is ignored (the defaults stay) rather than clamped
        if (e && sscanf(e, "%d,%d,%d", &c0, &a, &b) == 3 && c0 >= 1 && c0 <= 10000 && a >= 0 && a <= 1000 && b >= 0 && b <= 1000 &&
            ((long long)c0 + (long long)TM_MAXMULT * (7LL * a + b)) * (long long)TM_EQ_MAXU < (1LL << 31)) { m->c0 = c0; m->a = a; m->b = b; }
    };
    env_cost("TAMCMC_COST", &c->cost_l);
    env_cost("TAMCMC_COST_GRAD", &c->cost_g);
    {
        c->units = tm_units(Nx);
        c->cells = tm_cells(c->units);
        c->tiles_l = tm_tiles(c->units, 0);
        c->tiles_g = tm_tiles(c->units, 1);
        // a tile count given by hand must still keep every tile within TM_TILE_MAXU units
        const int tmin_g = (c->units + TM_TILE_MAXU - 1) / TM_TILE_MAXU + (c->units > TM_TILE_MAXU ? 1 : 0);
        const int tmin_l = (c->units + TM_TILE_MAXU_L - 1) / TM_TILE_MAXU_L + (c->units > TM_TILE_MAXU_L ? 1 : 0);
        if (c->units > 4) {
            env_int("TAMCMC_TILES", tmin_l, 1 << 20, &c->tiles_l);
            env_int("TAMCMC_TILES_GRAD", tmin_g, 1 << 20, &c->tiles_g);
        }
        // Tail shaping of the gradient launch (long grids, equal-length tiles): the first tiles keep TM_TILE_MAXU units, the
        // rest -- cheaper, hence launched last under the costliest-first order -- get su2 units, so that the workgroups
        // that end the launch are short (the launch is 1.56 rounds of resident workgroups at C2, and a round's last
        // workgroups run on a nearly empty chip).  The long tiles cover `frac` percent of the units: 85 %, 4 units by
        // default -- at 1e5 bins 21 tiles of 8 units + 7 of 4 (measured, profiles/README.md: C2 80.2 -> 78.1 us, C4 258 ->
        // 251 us, 16 / 32 chains -1.1 us, 256 chains +1 %).  TAMCMC_TAIL="frac,su2" sets it, TAMCMC_TAIL=0 switches it off.
        // A function of the grid only, like the tile count: a chain's result does not depend on the batch.
        c->cost_g.t1 = 0;
        {
            const char *e = getenv("TAMCMC_TAIL");
            int frac = 85, su2 = 4;
            if (e && sscanf(e, "%d,%d", &frac, &su2) != 2) frac = 0;
            if (frac >= 1 && frac <= 99 && su2 >= 1 && su2 <= TM_TILE_MAXU &&
                c->units >= 70 && !c->equal_cost && !getenv("TAMCMC_TILES_GRAD")) {
                const int t1 = (int)(((long long)c->units * frac / 100 + TM_TILE_MAXU / 2) / TM_TILE_MAXU);
                const int rest = c->units - t1 * TM_TILE_MAXU;
                if (t1 >= 1 && rest > 0) {
                    c->cost_g.t1 = t1; c->cost_g.su1 = TM_TILE_MAXU; c->cost_g.su2 = su2;
                    c->tiles_g = t1 + (rest + su2 - 1) / su2;
                }
            }
        }
        // the same for the likelihood launch (experiment: TAMCMC_TAIL_L="frac,su2"; off by default)
        c->cost_l.t1 = 0;
        {
            const char *e = getenv("TAMCMC_TAIL_L");
            int frac = 0, su2 = 0;
            if (e && sscanf(e, "%d,%d", &frac, &su2) == 2 && frac >= 1 && frac <= 99 && su2 >= 1 && su2 <= TM_TILE_MAXU &&
                c->units >= 70 && !c->equal_cost && !getenv("TAMCMC_TILES")) {
                const int t1 = (int)(((long long)c->units * frac / 100 + TM_TILE_MAXU / 2) / TM_TILE_MAXU);
                const int rest = c->units - t1 * TM_TILE_MAXU;
                if (t1 >= 1 && rest > 0) {
                    c->cost_l.t1 = t1; c->cost_l.su1 = TM_TILE_MAXU; c->cost_l.su2 = su2;
                    c->tiles_l = t1 + (rest + su2 - 1) / su2;
                }
            }
        }
        // the balancer's guarantee is TM_TILE_MAXU units per tile; equal-length likelihood tiles may be longer
        c->cost_l.pad = (c->equal_cost && (long long)c->tiles_l * TM_TILE_MAXU > c->units) ? TM_TILE_MAXU : TM_TILE_MAXU_L;
    }
    c->tiles_max = c->tiles_l > c->tiles_g ? c->tiles_l : c->tiles_g;

    auto fail = [&](int code) { tamcmc_ctx_destroy(c); return code; };
    if (hipSetDevice(device_id) != hipSuccess) return fail(TAMCMC_E_NODEVICE);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(TAMCMC_E_HIP);
    c->stream = c->own_stream;
    const size_t bytes = (size_t)Nx * sizeof(double);
    if (hipMalloc(&c->d_x2, bytes) != hipSuccess || hipMalloc(&c->d_y, bytes) != hipSuccess ||
        hipMalloc(&c->d_lx, bytes) != hipSuccess)
        return fail(TAMCMC_E_NOMEM);
    std::vector<double> tmp((size_t)Nx);
    for (int64_t i = 0; i < Nx; i++) tmp[(size_t)i] = 2.0 * x[i];        // exact; the Lorentzians are written in d = 2x - 2nu
    if (hipMemcpy(c->d_x2, tmp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(TAMCMC_E_HIP);
    for (int64_t i = 0; i < Nx; i++) tmp[(size_t)i] = std::log(x[i]);   // log x table for the Harvey powers
    if (hipMemcpy(c->d_y, y, bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_lx, tmp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
        return fail(TAMCMC_E_HIP);
    if (likelihood_case == 1) {
        for (int64_t i = 0; i < Nx; i++) tmp[(size_t)i] = 1.0 / (sigma_y[i] * sigma_y[i]);  // likelihoods.cpp:36
        if (hipMalloc(&c->d_isig2, bytes) != hipSuccess) return fail(TAMCMC_E_NOMEM);
        if (hipMemcpy(c->d_isig2, tmp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(TAMCMC_E_HIP);
    }
    *out = c;
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_destroy(tamcmc_ctx *c)
{
    if (!c) return TAMCMC_OK;
    (void)hipSetDevice(c->device);
    if (c->armed && c->h_gate) { __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE); c->armed = 0; }   // let the gate go
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_batch(c);
    (void)hipHostFree(c->h_gate);
    (void)hipFree(c->d_x2); (void)hipFree(c->d_y); (void)hipFree(c->d_lx); (void)hipFree(c->d_isig2); (void)hipFree(c->d_spec);
    (void)hipFree(c->d_model); (void)hipFree(c->d_relax);
    (void)hipHostFree(c->h_in); (void)hipHostFree(c->h_out); (void)hipHostFree(c->h_status);
    if (c->probe_stream) { (void)hipStreamSynchronize(c->probe_stream); (void)hipStreamDestroy(c->probe_stream); }
    (void)hipHostFree(c->h_probe);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    for (hipStream_t ps : c->part_streams) if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
    for (hipEvent_t e : c->part_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_set_vars(tamcmc_ctx *c, int32_t Nvars, const int32_t *index_to_relax)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c || Nvars < 0 || (Nvars > 0 && !index_to_relax)) return TAMCMC_E_INVALID;
    for (int i = 0; i < Nvars; i++)
        if (index_to_relax[i] < 0 || index_to_relax[i] >= c->L.Nparams) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_relax); c->d_relax = nullptr;
    c->Nvars = Nvars;
    // the asymmetry as a variable: its derivative does not vanish at asym == 0 although the factor is 1 there
    c->L.asym_var = 0;
    if (c->L.family != TM_FAM_GAUSS)
        for (int i = 0; i < Nvars; i++)
            if (index_to_relax[i] == c->L.s + 5) c->L.asym_var = 1;
    if (Nvars > 0) {
        TM_HIP(hipMalloc(&c->d_relax, (size_t)Nvars * sizeof(int32_t)));
        TM_HIP(hipMemcpy(c->d_relax, index_to_relax, (size_t)Nvars * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_set_spectra(tamcmc_ctx *c, int32_t Nspectra, const double *y, const double *sigma_y)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c || Nspectra < 1 || !y || (c->L.likelihood_case == 1 && !sigma_y)) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    const size_t nx = (size_t)c->L.Nx, bytes = nx * (size_t)Nspectra * sizeof(double);
    // the new blocks first, the swap last: a failure leaves the context as it was
    double *ny = nullptr, *nis = nullptr;
    if (hipMalloc(&ny, bytes) != hipSuccess) return TAMCMC_E_NOMEM;
    if (hipMemcpy(ny, y, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(ny); return TAMCMC_E_HIP; }
    if (c->L.likelihood_case == 1) {
        std::vector<double> tmp(nx * (size_t)Nspectra);
        for (size_t i = 0; i < tmp.size(); i++) tmp[i] = 1.0 / (sigma_y[i] * sigma_y[i]);  // likelihoods.cpp:36
        if (hipMalloc(&nis, bytes) != hipSuccess) { (void)hipFree(ny); return TAMCMC_E_NOMEM; }
        if (hipMemcpy(nis, tmp.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(ny); (void)hipFree(nis); return TAMCMC_E_HIP; }
    }
    (void)hipFree(c->d_y); c->d_y = ny;
    if (c->L.likelihood_case == 1) { (void)hipFree(c->d_isig2); c->d_isig2 = nis; }
    (void)hipFree(c->d_spec); c->d_spec = nullptr; c->spec_n = 0;
    c->nspec = Nspectra;
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_set_chain_spectrum(tamcmc_ctx *c, int32_t Nchains, const int32_t *spectrum_of_chain)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c || Nchains < 0 || (Nchains > 0 && !spectrum_of_chain)) return TAMCMC_E_INVALID;
    for (int m = 0; m < Nchains; m++)
        if (spectrum_of_chain[m] < 0 || spectrum_of_chain[m] >= c->nspec) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_spec); c->d_spec = nullptr; c->spec_n = 0;
    if (Nchains > 0) {
        TM_HIP(hipMalloc(&c->d_spec, (size_t)Nchains * sizeof(int32_t)));
        TM_HIP(hipMemcpy(c->d_spec, spectrum_of_chain, (size_t)Nchains * sizeof(int32_t), hipMemcpyHostToDevice));
        c->spec_n = Nchains;
    }
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_set_stream(tamcmc_ctx *c, void *hip_stream)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_synchronize(tamcmc_ctx *c)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_profile(tamcmc_ctx *c, int enable)
{
    if (!c) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    c->profile = enable != 0;
    c->profile_stride = enable > 1 ? enable : 1;
    c->profile_count = 0;
    c->ev_used = 0;
    // a pool of events up front: creating one costs ~10 us, which would land inside the caller's timed region
    while (c->profile && c->ev.size() < 256) {
        hipEvent_t e;
        TM_HIP(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_kernel_time(tamcmc_ctx *c, double *total_ms, int64_t *launches)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c || !total_ms || !launches) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->stream));
    double t = 0.0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.f;
        TM_HIP(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        t += (double)ms;
    }
    *total_ms = t;
    *launches = (int64_t)(c->ev_used / 2);
    return TAMCMC_OK;
}

// One wave that does nothing but watch two counters for `ticks` ticks of the constant 100 MHz clock: s_memtime counts
// shader-core cycles, s_memrealtime the constant clock, so their ratio is the core clock the GPU ran at meanwhile --
// launched on a stream of its own beside the evaluation, it reads the clock UNDER THAT LOAD (bench.py: roofline.valu).
__global__ void tamcmc_clock_probe_kernel(unsigned long long ticks, unsigned long long *out)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0, c1 = c0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = wall_clock64(); c1 = clock64(); }
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

extern "C" int tamcmc_ctx_clock_probe_begin(tamcmc_ctx *c, double milliseconds)
{
    if (!c || !(milliseconds > 0.0) || milliseconds > 2000.0) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    if (!c->probe_stream) TM_HIP(hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    if (!c->h_probe) {
        TM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_probe), 2 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        TM_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dv_probe), c->h_probe, 0));
    }
    c->h_probe[0] = c->h_probe[1] = 0;
    hipLaunchKernelGGL(tamcmc_clock_probe_kernel, dim3(1), dim3(64), 0, c->probe_stream,
                       (unsigned long long)(milliseconds * 1e5), c->dv_probe);
    TM_HIP(hipGetLastError());
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_clock_probe_end(tamcmc_ctx *c, double *core_GHz, double *seconds)
{
    if (!c || !c->probe_stream || !core_GHz) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    TM_HIP(hipStreamSynchronize(c->probe_stream));
    const double cyc = (double)c->h_probe[0], t = (double)c->h_probe[1] / 1e8;
    *core_GHz = (t > 0.0) ? cyc / t / 1e9 : 0.0;
    if (seconds) *seconds = t;
    return TAMCMC_OK;
}

extern "C" int tamcmc_ctx_geometry(tamcmc_ctx *c, int32_t *bins_per_tile, int32_t *tiles, int32_t *threads_per_block,
                                   int32_t *n_multiplets)
{
    if (!c) return TAMCMC_E_INVALID;
    const int T = c->last_tiles > 0 ? c->last_tiles : pick_tiles(c, 64, false);
    if (bins_per_tile) *bins_per_tile = TM_UNIT_BINS * c->cost_l.pad;   // largest tile of the likelihood launch (TM_TILE_MAXU_L units; TM_TILE_MAXU when balanced, and always for the gradient launch)
    if (tiles) *tiles = T;
    if (threads_per_block) *threads_per_block = TM_THREADS;
    if (n_multiplets) *n_multiplets = c->L.n_mult;
    return TAMCMC_OK;
}

// Number of tiles.  It depends on the grid only, never on the batch: a chain's result must not change with the number
// of chains evaluated beside it (a sharded run and a single-process run have to produce bit-identical chains).
static int pick_tiles(const tamcmc_ctx *c, int /*Nchains*/, bool grad)
{
    return grad ? c->tiles_g : c->tiles_l;
}

// Enqueue setup -> eval (-> backward) for device-resident inputs.
// base / stream: the sub-batch starts at chain `base` of the context's per-chain buffers (0 for a whole batch) and runs
// on `stream` (the context stream for a whole batch).
static int enqueue(tamcmc_ctx *c, int Nchains, const double *d_params, const double *d_T, double *d_logL,
                   double *d_grad, int32_t *d_status, const int32_t *d_rows, double *d_model, int base = 0, hipStream_t stream = nullptr)
{
    const bool grad = d_grad != nullptr;
    if (stream == nullptr) stream = c->stream;
    // several spectra resident: every chain of the batch must have been told which one it is fitted to (a batch longer
    // than the map used to fall back to spectrum 0 for all chains -- silently the wrong data)
    if (c->nspec > 1 && (c->d_spec == nullptr || base + Nchains > c->spec_n)) return TAMCMC_E_INVALID;
    const int units = c->units, cells = c->cells;
    const int tiles = pick_tiles(c, Nchains, grad);
    if (!grad) c->last_tiles = tiles;
    const size_t b = (size_t)base, nmx = (size_t)(c->L.n_mult > 0 ? c->L.n_mult : 1);
    TmMult *const p_mult = c->d_mult + b * nmx;
    TmNoise *const p_noise = c->d_noise + b;
    TmCellRec *const p_cell = c->d_cell + b * cells;
    TmTileHdr *const p_thdr = c->d_thdr + b * tiles;
    TmActive *const p_tidx = c->d_tidx + b * tiles * nmx;
    double *const p_wt = c->d_wt + b * 2, *const p_part = c->d_part + b * tiles * 4;
    int32_t *const p_order = c->d_order + b * tiles, *const p_ticket = c->d_ticket + b;
    double *const p_gmult = grad ? c->d_gmult + b * tiles * nmx * TM_GSLOTS : nullptr;
    double *const p_gnoise = grad ? c->d_gnoise + b * tiles * 2 * TM_NSLOTS : nullptr;
    double *const p_hser = grad ? c->d_hser + b * cells * TM_MAXH * TM_HSER : nullptr;
    void *const p_chain_rec = grad ? (void *)((char *)c->d_chain_rec + b * tm_sizeof_chain_rec()) : nullptr;
    void *const p_aux = grad ? (void *)((char *)c->d_aux + b * nmx * tm_sizeof_aux()) : nullptr;
    TmEvalArgs a{};
    a.x2 = c->d_x2; a.y = c->d_y; a.lx = c->d_lx; a.isig2 = c->d_isig2;
    a.spec = (c->nspec > 1) ? c->d_spec + b : nullptr;
    a.mult = p_mult; a.noise = p_noise; a.cell = p_cell; a.thdr = p_thdr; a.tidx = p_tidx; a.wt = p_wt;
    a.part = p_part; a.gmult = p_gmult; a.gnoise = p_gnoise;
    a.row_of_chain = d_rows; a.model_out = d_model;
    a.ticket = grad ? nullptr : p_ticket; a.logL = d_logL; a.status = d_status;
    a.Nx = c->L.Nx; a.n_mult = c->L.n_mult; a.tiles = tiles; a.cells = cells; a.likelihood_case = c->L.likelihood_case;
    a.like_p = c->L.like_p;
    a.order = p_order; a.order_mode = (tiles <= 65535) ? c->order_mode : 0; a.prio = c->prio;
    a.generic = (c->L.likelihood_case != 0 || c->L.family == TM_FAM_GAUSS || d_rows != nullptr) ? 1 : 0;
    if (tiles == 1 && a.order_mode == 2) a.order_mode = 1;     // nothing to rank
    a.tile_magic = ((1ULL << 40) + (unsigned long long)tiles - 1) / (unsigned long long)tiles;
    // one tile per chain (short grids): prologue and evaluation share a launch
    const bool fused = (tiles == 1) && c->fuse != 0 && units <= TM_TILE_MAXU;   // (TAMCMC_TILES=1 on a 9..16-unit grid: two launches)
    int rc = 0;
    if (!fused) {
        rc = tm_launch_setup(c->L, Nchains, d_params, d_T, p_wt, c->d_lx, units, cells, tiles, c->equal_cost, grad ? c->cost_g : c->cost_l,
                             p_mult, p_noise, p_cell, p_thdr, p_tidx, p_chain_rec, p_aux,
                             p_hser, a.order_mode == 2 ? p_order : nullptr, stream);
        if (rc != 0) { snprintf(g_hip_err, sizeof(g_hip_err), "setup launch -> %s", hipGetErrorString((hipError_t)rc)); return TAMCMC_E_HIP; }
    }
    const bool timed = c->profile && (c->profile_count++ % c->profile_stride == 0);
    if (timed) {
        while (c->ev.size() < c->ev_used + 2) {
            hipEvent_t e;
            TM_HIP(hipEventCreate(&e));
            c->ev.push_back(e);
        }
        TM_HIP(hipEventRecord(c->ev[c->ev_used], stream));
    }
    if (fused) {
        TmFusedArgs f{};
        f.params = d_params; f.Tcoefs = d_T;
        f.chain_rec = p_chain_rec; f.aux = p_aux; f.hser = p_hser;
        f.p_doubles = (c->L.Nparams + 1) & ~1;
        rc = tm_launch_fused(c->L, f, a, Nchains, grad, stream);
    } else {
        rc = tm_launch_eval(a, Nchains, grad, stream);
    }
    if (rc != 0) {
        snprintf(g_hip_err, sizeof(g_hip_err), "eval launch -> %s", hipGetErrorString((hipError_t)rc));
        (void)hipMemsetAsync(p_ticket, 0, (size_t)Nchains * sizeof(int32_t), stream);   // arrival counters back to zero
        return TAMCMC_E_HIP;
    }
    if (timed) {
        TM_HIP(hipEventRecord(c->ev[c->ev_used + 1], stream));
        c->ev_used += 2;
    }
    if (!grad) {
        // finalize happens inside the eval launch (last-arriving workgroup per chain)
    } else {
        rc = tm_launch_backward(c->L, Nchains, units, cells, tiles, tm_setup_balances(units, tiles, c->equal_cost, (grad ? c->cost_g : c->cost_l).pad), c->cost_g, d_params, p_wt, p_chain_rec, p_aux, p_noise, p_part,
                                p_gmult, p_gnoise, p_cell, p_thdr, p_hser, c->Nvars, c->d_relax, d_grad, d_logL, d_status,
                                stream);
        if (rc != 0) { snprintf(g_hip_err, sizeof(g_hip_err), "backward launch -> %s", hipGetErrorString((hipError_t)rc)); return TAMCMC_E_HIP; }
    }
    return TAMCMC_OK;
}

static int grad_supported(const tamcmc_ctx *c)
{
    if (c->Nvars <= 0 || !c->d_relax) return TAMCMC_E_NOVARS;
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_device(tamcmc_ctx *c, int32_t Nchains, int32_t Nparams,
                                        const double *d_params, const double *d_Tcoefs,
                                        double *d_logL, double *d_grad, int32_t *d_status)
{
    if (c && c->armed) return TAMCMC_E_INVALID;     // launches wait behind a gate: only _fire / _end / _disarm / destroy (tamcmc_accel.h)
    if (!c || Nchains < 1 || !d_params || !d_Tcoefs || !d_logL) return TAMCMC_E_INVALID;
    if (Nparams != c->L.Nparams || c->parts_busy()) return TAMCMC_E_INVALID;
    if (d_grad) { int rc = grad_supported(c); if (rc != TAMCMC_OK) return rc; }
    TM_HIP(hipSetDevice(c->device));
    int rc = ensure_capacity(c, Nchains, d_grad != nullptr);
    if (rc != TAMCMC_OK) return rc;
    return enqueue(c, Nchains, d_params, d_Tcoefs, d_logL, d_grad, d_status, nullptr, nullptr);
}

// Wait for everything enqueued so far by polling an event.  hipStreamSynchronize may park the calling thread on an
// interrupt; on this platform that path showed rare stalls of 1-40 ms after a ~160 us batch (profiles/README.md),
// and a sampler calls this thousands of times per second.
static int wait_done(tamcmc_ctx *c)
{
    if (!c->ev_done) TM_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    TM_HIP(hipEventRecord(c->ev_done, c->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(c->ev_done);
        if (e == hipSuccess) return TAMCMC_OK;
        if (e != hipErrorNotReady) { snprintf(g_hip_err, sizeof(g_hip_err), "hipEventQuery -> %s", hipGetErrorString(e)); return TAMCMC_E_HIP; }
        __builtin_ia32_pause();
    }
}

// Host path without model rows: instead of waiting for the launch to retire, watch the results arrive.  Every logL and
// gradient entry is one aligned 8-byte store and every status one 4-byte store into coherent pinned memory, written
// exactly once per launch, so a slot that no longer holds the marker put there before the launch holds its final
// value -- no ordering between slots is assumed.  The completion event is still recorded and consulted now and then:
// a failed launch ends the wait with an error instead of a hang, and should a result ever equal the marker (a kernel
// NaN does not have this payload) the wait ends when the launch retires.
static const uint64_t TM_PENDING_BITS = 0x7FF8DEADBEEF5A5AULL;
// nw = doubles to watch in h_out: n (logL) or n * (1 + Nvars) (logL, then the gradient rows)
static void mark_pending(tamcmc_ctx *c, int n, size_t nw)
{
    uint64_t *o = reinterpret_cast<uint64_t *>(c->h_out);
    for (size_t m = 0; m < nw; m++) o[m] = TM_PENDING_BITS;
    for (int m = 0; m < n; m++) c->h_status[m] = -1;
}
// o / st: first logL (then gradient) slot and first status slot of the (sub-)batch; ev / recorded / stream: its completion
// event, recorded lazily; ticket / nticket: its arrival counters (re-armed when a launch retired without finalizing).
static int wait_slots(tamcmc_ctx *c, volatile const uint64_t *o, volatile const int32_t *st, int n, size_t nw,
                      hipEvent_t *ev, bool *recorded, hipStream_t stream, int32_t *ticket, int nticket)
{
    unsigned spins = 0;
    for (size_t m = 0; m < nw;) {
        if (o[m] != TM_PENDING_BITS && (m >= (size_t)n || st[m] != -1)) { m++; continue; }
        __builtin_ia32_pause();
        if ((++spins & 2047u) == 0) {
            // The completion event is recorded only now, behind the kernels already in the stream (it completes once they
            // have): a call that gets its results within the first ~2000 polls -- every healthy call -- never pays for an
            // event on the launch path (~1.5 us of host time per call in a sampler loop).
            if (!*recorded) {
                if (!*ev && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) return TAMCMC_E_HIP;
                if (hipEventRecord(*ev, stream) != hipSuccess) return TAMCMC_E_HIP;
                *recorded = true;
            }
            const hipError_t e = hipEventQuery(*ev);
            if (e == hipSuccess) {
                // The launch has retired: whatever the slots hold is final.  A logL / status slot that still holds its
                // marker was never written -- a chain whose finalize did not run (e.g. an arrival counter left non-zero
                // by an earlier failed launch).  Report it instead of handing the marker out as a result, and re-arm
                // the counters so that the context is usable again.
                for (size_t k = 0; k < (size_t)n; k++)
                    if (o[k] == TM_PENDING_BITS || st[k] == -1) {
                        snprintf(g_hip_err, sizeof(g_hip_err), "chain %zu was not finalized by a retired launch", k);
                        (void)hipMemsetAsync(ticket, 0, (size_t)nticket * sizeof(int32_t), stream);
                        return TAMCMC_E_HIP;
                    }
                return TAMCMC_OK;
            }
            if (e != hipErrorNotReady) {
                snprintf(g_hip_err, sizeof(g_hip_err), "hipEventQuery -> %s", hipGetErrorString(e));
                (void)hipMemsetAsync(ticket, 0, (size_t)nticket * sizeof(int32_t), stream);
                return TAMCMC_E_HIP;
            }
        }
    }
    return TAMCMC_OK;
}
static int wait_data(tamcmc_ctx *c, int n, size_t nw)
{
    return wait_slots(c, reinterpret_cast<volatile const uint64_t *>(c->h_out), c->h_status, n, nw, &c->ev_done, &c->ev_recorded,
                      c->stream, c->d_ticket, c->cap);
}

// pinned, device-mapped staging of the host-pointer entry points
static int ensure_staging(tamcmc_ctx *c, int Nchains)
{
    const int Nparams = c->L.Nparams;
    if (Nchains <= c->h_cap && c->h_nvars == c->Nvars) return TAMCMC_OK;
    TM_HIP(hipStreamSynchronize(c->stream));
    (void)hipHostFree(c->h_in); (void)hipHostFree(c->h_out); (void)hipHostFree(c->h_status);
    c->h_in = c->h_out = nullptr; c->h_status = nullptr; c->h_cap = 0;
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
    const size_t cap = (size_t)(Nchains > c->cap ? Nchains : c->cap);
    TM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_in), cap * ((size_t)Nparams + 1) * sizeof(double), flags));
    TM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_out), cap * ((size_t)(c->Nvars > 0 ? c->Nvars : 0) + 1) * sizeof(double), flags));
    TM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_status), cap * sizeof(int32_t), flags));
    TM_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dv_in), c->h_in, 0));
    TM_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dv_out), c->h_out, 0));
    TM_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dv_status), c->h_status, 0));
    c->h_cap = (int)cap; c->h_nvars = c->Nvars;
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_begin(tamcmc_ctx *c, int32_t Nchains, int32_t Nparams, const double *params, const double *Tcoefs)
{
    if (!c || Nchains < 1 || !params || !Tcoefs || Nparams != c->L.Nparams || c->in_flight || c->armed || c->parts_busy()) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    int rc = ensure_capacity(c, Nchains, false);
    if (rc != TAMCMC_OK) return rc;
    rc = ensure_staging(c, Nchains);
    if (rc != TAMCMC_OK) return rc;
    const size_t n = (size_t)Nchains;
    std::memcpy(c->h_in, params, n * Nparams * sizeof(double));
    std::memcpy(c->h_in + n * Nparams, Tcoefs, n * sizeof(double));
    double *dv_in = nullptr, *dv_out = nullptr;
    int32_t *dv_status = nullptr;
    dv_in = c->dv_in; dv_out = c->dv_out; dv_status = c->dv_status;
    mark_pending(c, Nchains, (size_t)Nchains);
    rc = enqueue(c, Nchains, dv_in, dv_in + n * Nparams, dv_out, nullptr, dv_status, nullptr, nullptr);
    if (rc != TAMCMC_OK) return rc;
    c->ev_recorded = false;          // wait_data records the completion event only if the results are slow to arrive
    c->in_flight = Nchains;
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_end(tamcmc_ctx *c, int32_t Nchains, double *logL, int32_t *status)
{
    if (!c || !logL || c->in_flight != Nchains) return TAMCMC_E_INVALID;
    c->in_flight = 0;
    { const int rc = wait_data(c, Nchains, (size_t)Nchains); if (rc != TAMCMC_OK) return rc; }
    std::memcpy(logL, c->h_out, (size_t)Nchains * sizeof(double));
    if (status) std::memcpy(status, c->h_status, (size_t)Nchains * sizeof(int32_t));
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_poll(const tamcmc_ctx *c, int32_t chain, double *logL, int32_t *status)
{
    if (!c || !logL || !status || chain < 0 || chain >= c->in_flight) return TAMCMC_E_INVALID;
    const uint64_t v = reinterpret_cast<volatile const uint64_t *>(c->h_out)[chain];
    const int32_t st = reinterpret_cast<volatile const int32_t *>(c->h_status)[chain];
    if (v == TM_PENDING_BITS || st == -1) return TAMCMC_PENDING;
    std::memcpy(logL, &v, sizeof(double));
    *status = st;
    return TAMCMC_OK;
}

int tm_launch_gate(uint32_t *dv_gate, uint32_t target, int patience, void *stream);      // tamcmc_setup.hip
#define TM_GATE_EXPIRED 16
#define TM_GATE_FIRING 32

// The host's side of the gate's expiry protocol (tamcmc_setup.hip): announce, then look.  true: the gate has given up and
// the armed launches ran (or are running) on stale input -- the caller opens the word anyway, lets them drain and starts over.
static bool gate_claim(tamcmc_ctx *c)
{
    __atomic_store_n(c->h_gate + TM_GATE_FIRING, c->gate_seq, __ATOMIC_SEQ_CST);
    return __atomic_load_n(c->h_gate + TM_GATE_EXPIRED, __ATOMIC_SEQ_CST) == c->gate_seq;
}

// An armed batch: its launches are put into the stream AHEAD of its parameters, behind a one-wave gate kernel that
// watches a pinned word.  A host loop arms batch i+1 while the GPU evaluates batch i (the launch calls, ~6 us, are then
// hidden under that evaluation) and fires it with one store once the parameters are known.
extern "C" int tamcmc_eval_batch_arm(tamcmc_ctx *c, int32_t Nchains)
{
    if (!c || Nchains < 1 || c->armed || c->parts_busy()) return TAMCMC_E_INVALID;
    if (c->in_flight && c->in_flight != Nchains) return TAMCMC_E_INVALID;
    // never (re)allocate under a batch in flight: tamcmc_ctx_reserve (or an earlier batch of this size) sized the buffers
    if (Nchains > c->cap || Nchains > c->h_cap || c->h_nvars != c->Nvars) {
        if (c->in_flight) return TAMCMC_E_INVALID;
        const int rc = tamcmc_ctx_reserve(c, Nchains);
        if (rc != TAMCMC_OK) return rc;
    }
    TM_HIP(hipSetDevice(c->device));
    if (!c->h_gate) {
        TM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_gate), 256, hipHostMallocMapped | hipHostMallocCoherent));
        TM_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dv_gate), c->h_gate, 0));
        std::memset(c->h_gate, 0, 256);
        c->gate_seq = 1;               // (0 is what the expiry / firing words hold before the first batch)
        *c->h_gate = c->gate_seq;
    }
    const uint32_t target = c->gate_seq + 1;
    int rc = tm_launch_gate(c->dv_gate, target, c->gate_patience, c->stream);
    if (rc != 0) { snprintf(g_hip_err, sizeof(g_hip_err), "gate launch -> %s", hipGetErrorString((hipError_t)rc)); return TAMCMC_E_HIP; }
    const size_t n = (size_t)Nchains;
    rc = enqueue(c, Nchains, c->dv_in, c->dv_in + n * c->L.Nparams, c->dv_out, nullptr, c->dv_status, nullptr, nullptr);
    c->gate_seq = target;
    if (rc != TAMCMC_OK) {            // the gate is in the stream: open it, nothing sits behind it
        __atomic_store_n(c->h_gate, target, __ATOMIC_RELEASE);
        return rc;
    }
    c->armed = Nchains;
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_fire(tamcmc_ctx *c, int32_t Nchains, int32_t Nparams, const double *params, const double *Tcoefs)
{
    if (!c || !params || !Tcoefs || Nparams != c->L.Nparams || c->armed != Nchains || Nchains < 1 || c->in_flight) return TAMCMC_E_INVALID;
    if (gate_claim(c)) {
        // the gate gave up waiting (the host was held up for seconds): the armed launches used stale input.  Let them
        // drain and evaluate this batch the plain way.
        __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE);
        c->armed = 0;
        TM_HIP(hipStreamSynchronize(c->stream));
        return tamcmc_eval_batch_begin(c, Nchains, Nparams, params, Tcoefs);
    }
    const size_t n = (size_t)Nchains;
    std::memcpy(c->h_in, params, n * Nparams * sizeof(double));
    std::memcpy(c->h_in + n * Nparams, Tcoefs, n * sizeof(double));
    mark_pending(c, Nchains, (size_t)Nchains);
    __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE);      // (after the parameters and the markers)
    c->ev_recorded = false;
    c->armed = 0;
    c->in_flight = Nchains;
    return TAMCMC_OK;
}

// Opens the gate of an armed batch that will not be fired (the loop ended, or failed): it runs on whatever the input
// buffer holds -- the previous batch's parameters -- and is waited for here; nothing is handed out.
extern "C" int tamcmc_eval_batch_disarm(tamcmc_ctx *c)
{
    if (!c) return TAMCMC_E_INVALID;
    if (!c->armed) return TAMCMC_OK;
    if (c->in_flight) return TAMCMC_E_INVALID;        // collect the batch in flight first (_end)
    const int n = c->armed;
    if (gate_claim(c)) {              // it has run (or is running) already: just let it retire
        __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE);
        c->armed = 0;
        TM_HIP(hipStreamSynchronize(c->stream));
        return TAMCMC_OK;
    }
    mark_pending(c, n, (size_t)n);
    __atomic_store_n(c->h_gate, c->gate_seq, __ATOMIC_RELEASE);
    c->ev_recorded = false;
    c->armed = 0;
    return wait_data(c, n, (size_t)n);
}

extern "C" int tamcmc_ctx_reserve(tamcmc_ctx *c, int32_t Nchains)
{
    if (!c || Nchains < 1 || c->in_flight || c->armed || c->parts_busy()) return TAMCMC_E_INVALID;
    TM_HIP(hipSetDevice(c->device));
    int rc = ensure_capacity(c, Nchains, false);
    if (rc != TAMCMC_OK) return rc;
    return ensure_staging(c, Nchains);
}

extern "C" int tamcmc_eval_batch_begin_part(tamcmc_ctx *c, int32_t part, int32_t first, int32_t Nchains, int32_t Nparams,
                                            const double *params, const double *Tcoefs)
{
    if (!c || part < 0 || part >= TAMCMC_MAX_PARTS || first < 0 || Nchains < 1 || !params || !Tcoefs || Nparams != c->L.Nparams) return TAMCMC_E_INVALID;
    if (c->in_flight || c->part_n[part]) return TAMCMC_E_INVALID;
    for (int o = 0; o < TAMCMC_MAX_PARTS; o++)      // ranges of parts in flight must not overlap
        if (o != part && c->part_n[o] && first < c->part_first[o] + c->part_n[o] && c->part_first[o] < first + Nchains) return TAMCMC_E_INVALID;
    // buffers are never (re)allocated under a part in flight: tamcmc_ctx_reserve sizes them beforehand
    if (first + Nchains > c->cap || first + Nchains > c->h_cap || c->h_nvars != c->Nvars) {
        if (c->parts_busy()) return TAMCMC_E_INVALID;
        const int rc = tamcmc_ctx_reserve(c, first + Nchains);
        if (rc != TAMCMC_OK) return rc;
    }
    TM_HIP(hipSetDevice(c->device));
    if (part > 0 && !c->part_streams[part]) TM_HIP(hipStreamCreateWithFlags(&c->part_streams[part], hipStreamNonBlocking));
    hipStream_t stream = (part == 0) ? c->stream : c->part_streams[part];
    const size_t f = (size_t)first, n = (size_t)Nchains, np = (size_t)Nparams, hc = (size_t)c->h_cap;
    std::memcpy(c->h_in + f * np, params, n * np * sizeof(double));
    std::memcpy(c->h_in + hc * np + f, Tcoefs, n * sizeof(double));
    uint64_t *o = reinterpret_cast<uint64_t *>(c->h_out) + f;
    for (size_t m = 0; m < n; m++) { o[m] = TM_PENDING_BITS; c->h_status[f + m] = -1; }
    const int rc = enqueue(c, Nchains, c->dv_in + f * np, c->dv_in + hc * np + f, c->dv_out + f, nullptr, c->dv_status + f, nullptr, nullptr,
                           first, stream);
    if (rc != TAMCMC_OK) return rc;
    c->part_ev_recorded[part] = false;
    c->part_first[part] = first; c->part_n[part] = Nchains;
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch_end_part(tamcmc_ctx *c, int32_t part, double *logL, int32_t *status)
{
    if (!c || part < 0 || part >= TAMCMC_MAX_PARTS || !logL || !c->part_n[part]) return TAMCMC_E_INVALID;
    const int n = c->part_n[part], first = c->part_first[part];
    c->part_n[part] = 0;
    hipStream_t stream = (part == 0) ? c->stream : c->part_streams[part];
    const int rc = wait_slots(c, reinterpret_cast<volatile const uint64_t *>(c->h_out) + first, c->h_status + first, n, (size_t)n,
                              &c->part_ev[part], &c->part_ev_recorded[part], stream, c->d_ticket + first, n);
    if (rc != TAMCMC_OK) return rc;
    std::memcpy(logL, c->h_out + first, (size_t)n * sizeof(double));
    if (status) std::memcpy(status, c->h_status + first, (size_t)n * sizeof(int32_t));
    return TAMCMC_OK;
}

extern "C" int tamcmc_eval_batch(tamcmc_ctx *c, int32_t Nchains, int32_t Nparams,
                                 const double *params, const double *Tcoefs,
                                 double *logL, double *grad,
                                 int32_t n_rows, const int32_t *model_rows, double *model_out,
                                 int32_t *status)
{
    if (!c || Nchains < 1 || !params || !Tcoefs || !logL) return TAMCMC_E_INVALID;
    if (Nparams != c->L.Nparams) return TAMCMC_E_INVALID;
    if (n_rows < 0 || (n_rows > 0 && (!model_rows || !model_out))) return TAMCMC_E_INVALID;
    for (int r = 0; r < n_rows; r++)
        if (model_rows[r] < 0 || model_rows[r] >= Nchains) return TAMCMC_E_INVALID;
    if (grad) { int rc = grad_supported(c); if (rc != TAMCMC_OK) return rc; }
    if (c->in_flight || c->parts_busy()) return TAMCMC_E_INVALID;   // (buffers may move below)
    TM_HIP(hipSetDevice(c->device));
    int rc = ensure_capacity(c, Nchains, grad != nullptr);
    if (rc != TAMCMC_OK) return rc;

    const size_t n = (size_t)Nchains;
    rc = ensure_staging(c, Nchains);
    if (rc != TAMCMC_OK) return rc;
    std::memcpy(c->h_in, params, n * Nparams * sizeof(double));
    std::memcpy(c->h_in + n * Nparams, Tcoefs, n * sizeof(double));
    double *dv_in = nullptr, *dv_out = nullptr;
    int32_t *dv_status = nullptr;
    dv_in = c->dv_in; dv_out = c->dv_out; dv_status = c->dv_status;
    const int32_t *d_rows = nullptr;
    if (n_rows > 0) {
        std::vector<int32_t> rows(n, -1);
        for (int r = 0; r < n_rows; r++) rows[(size_t)model_rows[r]] = r;   // a chain listed twice keeps the last row
        const size_t need = (size_t)n_rows * (size_t)c->L.Nx;
        if (need > c->model_cap) {
            TM_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_model); c->d_model = nullptr; c->model_cap = 0;
            TM_HIP(hipMalloc(&c->d_model, need * sizeof(double)));
            c->model_cap = need;
        }
        TM_HIP(hipMemcpyAsync(c->d_rows, rows.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        TM_HIP(hipStreamSynchronize(c->stream));   // rows is a local
        d_rows = c->d_rows;
    }
    // logL / status (a few hundred bytes) are written straight into the mapped host buffer by the last kernel; the
    // the backward kernel writes each chain's gradient row as one run of consecutive stores, straight into the mapped
    // host buffer (a copy-engine transfer of these ~20 KB would add ~20 us of latency)
    const bool watch = (n_rows == 0);                         // no model rows to copy back: watch the results arrive (wait_data)
    const size_t nwatch = n * (grad ? (size_t)c->Nvars + 1 : 1);
    if (watch) mark_pending(c, Nchains, nwatch);
    rc = enqueue(c, Nchains, dv_in, dv_in + n * Nparams, dv_out, grad ? dv_out + n : nullptr, dv_status, d_rows, c->d_model);
    if (rc != TAMCMC_OK) return rc;
    if (n_rows > 0) {
        // rows whose chain was listed more than once share one device row
        for (int r = 0; r < n_rows; r++) {
            int src = r;
            for (int r2 = n_rows - 1; r2 > r; r2--) if (model_rows[r2] == model_rows[r]) { src = r2; break; }
            TM_HIP(hipMemcpyAsync(model_out + (size_t)r * c->L.Nx, c->d_model + (size_t)src * c->L.Nx,
                                  (size_t)c->L.Nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (watch) {
        c->ev_recorded = false;
        rc = wait_data(c, Nchains, nwatch);
    } else {
        rc = wait_done(c);
    }
    if (rc != TAMCMC_OK) return rc;
    TM_HIP(hipGetLastError());
    std::memcpy(logL, c->h_out, n * sizeof(double));
    if (status) std::memcpy(status, c->h_status, n * sizeof(int32_t));
    if (grad) std::memcpy(grad, c->h_out + n, n * (size_t)c->Nvars * sizeof(double));
    return TAMCMC_OK;
}

extern "C" int tamcmc_model_explicit(tamcmc_ctx *c, int32_t Nparams, const double *params, double *model_out, int32_t *status)
{
    if (!c || !params || !model_out) return TAMCMC_E_INVALID;
    const double T = 1.0;
    double logL = 0.0;
    const int32_t row = 0;
    int32_t st = 0;
    int rc = tamcmc_eval_batch(c, 1, Nparams, params, &T, &logL, nullptr, 1, &row, model_out, &st);
    if (status) *status = st;
    return rc;
}

extern "C" const char *tamcmc_strerror(int code)
{
    switch (code) {
    case TAMCMC_OK: return "ok";
    case TAMCMC_E_INVALID: return "invalid argument";
    case TAMCMC_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case TAMCMC_E_HIP: return "HIP runtime error (see tamcmc_last_hip_error)";
    case TAMCMC_E_MODEL_DISABLED: return "model id disabled in the reference (ids 4, 5)";
    case TAMCMC_E_UNKNOWN_MODEL: return "unknown model or likelihood id";
    case TAMCMC_E_NOMEM: return "out of memory";
    case TAMCMC_E_NOVARS: return "gradient requested before tamcmc_ctx_set_vars";
    case TAMCMC_E_NOGRAD: return "gradient not available for this model";
    default: return "unknown error code";
    }
}

extern "C" const char *tamcmc_last_hip_error(void) { return g_hip_err; }
#ifndef TM_KERNEL_HASH
#define TM_KERNEL_HASH "unknown"
#endif
extern "C" const char *tamcmc_version(void) { return "tamcmc_accel 0.3 (gfx950) kernels:" TM_KERNEL_HASH; }
