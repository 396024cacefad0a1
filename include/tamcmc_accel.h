/*
 * tamcmc_accel.h -- C ABI of the MI355X (gfx950) accelerator for TAMCMC's hot path:
 * for every parallel-tempered chain at once, params row -> model spectrum M(x) -> tempered
 * log-likelihood (-> gradient with respect to the relaxed variables).
 *
 * This is the drop-in boundary.  It replaces, for the whole batch of chains in one call, what the
 * reference does per chain inside its OpenMP loop (MALA.cpp:632-639):
 *
 *     Model_def::generate_model(Data*, long m, VectorXd Tcoefs)        model_def.cpp:358-367
 *       -> Model_def::call_model(Data*, int m)                          model_def.cpp:210-289
 *            -> model_MS_Global_* / model_MS_local_* / model_*_Gaussian models.cpp
 *       -> Model_def::call_likelihood(Data*, int m, VectorXd Tcoefs)    model_def.cpp:291-320
 *            -> likelihood_chi22p / likelihood_chi_square               likelihoods.cpp:17-39
 *
 * Conventions kept from the reference: all arithmetic fp64; logL is returned ALREADY DIVIDED by the
 * chain temperature Tcoefs[m] (model_def.cpp:302); the parameter layout is the flat `params` row
 * described by plength[0..10] (models.cpp:492-506); model / likelihood ids are the
 * integers of Config/default/models_ctrl.list and likelihoods_ctrl.list.  Where the reference would
 * print and exit() from inside the path the library reports a per-chain status instead.
 *
 * All entry points return TAMCMC_OK (0) or a TAMCMC_E_* code; none of them throws, prints or exits.
 * There is NO CPU fallback: without a usable HIP device tamcmc_ctx_create fails with
 * TAMCMC_E_NODEVICE.
 *
 * Environment switches read by tamcmc_ctx_create (developer knobs; none is needed in normal use, none changes a result
 * beyond rounding, and the tests exercise every one of them):
 *   TAMCMC_TILES, TAMCMC_TILES_GRAD   tiles per chain of the likelihood-only / gradient launch (default 8 units of 512 bins)
 *   TAMCMC_EQUAL_COST=1               per-chain tile boundaries of equal cost instead of equal length
 *   TAMCMC_COST, TAMCMC_COST_GRAD     "c0,a,b": the balancer's cost model
 *   TAMCMC_PRIO=1                     issue priority by launch rank
 *   TAMCMC_ORDER=0|1|2                launch order (default 2: tile-major, each chain's tiles costliest-first)
 *   TAMCMC_FUSED=0                    one-tile grids: prologue and evaluation as two launches instead of one
 *   TAMCMC_BG_EXACT=1                 Harvey background by exp() per bin instead of the per-cell polynomial
 *   TAMCMC_TAIL="frac,su2" | 0        gradient launch on long grids: 8-unit tiles for frac % of the units, su2-unit tiles for the
 *                                     rest (default "85,4"; 0: all tiles alike); TAMCMC_TAIL_L the same for the likelihood launch (default off)
 *   TAMCMC_GATE_PATIENCE=n            polls (~2 us each) before the gate of an armed batch gives up (default 2^21: ~4 s; tests)
 * (tamcmc_sampler.h: TAMCMC_SAMPLER_THREADS, TAMCMC_SAMPLER_TIMING, TAMCMC_SAMPLER_PIPELINE, TAMCMC_SAMPLER_ARM, TAMCMC_SAMPLER_ARRIVE.)
 *
 * Threading: one ctx = one device + one stream; calls on one ctx must be serialised by the caller;
 * different ctx objects (other GPUs, other stars) may be driven concurrently from different threads.
 * Ownership: the library copies x, y, sigma_y to the device at create time and never keeps caller
 * pointers after a call returns.
 */
#ifndef TAMCMC_ACCEL_H
#define TAMCMC_ACCEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamcmc_ctx tamcmc_ctx;

/* error codes */
#define TAMCMC_OK                  0
#define TAMCMC_E_INVALID           1  /* bad argument (NULL pointer, negative size, plength/Nparams mismatch) */
#define TAMCMC_E_NODEVICE          2  /* no usable HIP device / device_id out of range                        */
#define TAMCMC_E_HIP               3  /* a HIP runtime call failed (see tamcmc_last_hip_error)                */
#define TAMCMC_E_MODEL_DISABLED    4  /* model id 4 or 5: the reference itself exits for these                */
#define TAMCMC_E_UNKNOWN_MODEL     5  /* id not in models_ctrl.list / likelihoods_ctrl.list                   */
#define TAMCMC_E_NOMEM             6
#define TAMCMC_E_NOVARS            7  /* gradient requested before tamcmc_ctx_set_vars                        */
#define TAMCMC_E_NOGRAD            8  /* gradient not available for this model/likelihood id                  */
#define TAMCMC_PENDING            -1  /* tamcmc_eval_batch_poll only: not an error, the chain's result has not arrived yet */

/* per-chain status written by the eval calls */
#define TAMCMC_CHAIN_OK            0
#define TAMCMC_CHAIN_NAN           1  /* logL is NaN: legal, means "reject" (MALA.cpp:475,507-509).  Also where a
                                         width over- or underflows (AppWidth with an absurd exponent): the reference's
                                         formula stays finite there and loses the move by ~1e4 in logL -- same decision */
#define TAMCMC_CHAIN_EMPTY_WINDOW  2  /* a truncation window is empty: the reference would exit(EXIT_FAILURE),
                                         build_lorentzian.cpp:428-443; logL is set to NaN                      */
#define TAMCMC_CHAIN_INTERNAL      3  /* internal consistency check of the tile balancer failed (never expected);
                                         logL is set to NaN                                                    */

/* Replaces: Config::setup() handing `Data` (data.h:24-36) + the integer switches
 * (config.cpp:95-98) to the Model_def constructor (model_def.cpp:27-55).
 *   device_id        HIP device ordinal (>= 0).
 *   model_case       models_ctrl.list id (0..14; 4 and 5 -> TAMCMC_E_MODEL_DISABLED).
 *   likelihood_case  0 = chi(2,2p), 1 = chi_square.
 *   likelihood_p     Model_def::likelihood_params; truncated to an integer like the reference's
 *                    `long p` argument (likelihoods.cpp:17).
 *   plength          the 11 block lengths of the params row.
 *   x, y, sigma_y    Nx doubles each on the host; sigma_y may be NULL unless likelihood_case == 1.
 *                    x must be the regular grid the reference assumes (build_lorentzian.cpp:423).
 *   Nx               2 <= Nx <= 2^28 (TAMCMC_E_INVALID outside): the kernels address the grid with 32-bit byte
 *                    offsets; the reference itself reads at most 1e6 rows (config.cpp:531). */
int tamcmc_ctx_create(tamcmc_ctx **out, int device_id, int model_case, int likelihood_case,
                      double likelihood_p, const int32_t plength[11], int64_t Nx,
                      const double *x, const double *y, const double *sigma_y);

/* Replaces: Model_def::index_to_relax (model_def.cpp:81-88).  Declares which params columns are
 * the free variables; needed only for gradients.  grad column k = d(logL/T)/d params[index_to_relax[k]]. */
int tamcmc_ctx_set_vars(tamcmc_ctx *ctx, int32_t Nvars, const int32_t *index_to_relax);

/* Extension (no counterpart in the reference, which fits one spectrum per process): several spectra ON THE SAME GRID
 * and model layout in one context -- an ensemble of synthetic stars, noise realisations of one star -- so that their
 * chains share a batch (one launch at the full-batch rate; separate contexts on separate streams overlap only ~1.5x,
 * DESIGN.md section 6).  set_spectra replaces the resident spectrum by Nspectra blocks of Nx (sigma_y likewise, NULL
 * unless the likelihood is chi_square) and clears the map; set_chain_spectrum says which spectrum chain m of the
 * following batches is fitted to.  While more than one spectrum is resident every batch must be covered by the map:
 * an evaluation of more chains than the map holds (or without a map) returns TAMCMC_E_INVALID instead of fitting the
 * uncovered chains to spectrum 0.  A chain's result is bit for bit what a context holding only its spectrum returns. */
int tamcmc_ctx_set_spectra(tamcmc_ctx *ctx, int32_t Nspectra, const double *y, const double *sigma_y);
int tamcmc_ctx_set_chain_spectrum(tamcmc_ctx *ctx, int32_t Nchains, const int32_t *spectrum_of_chain);

/* Replaces: the `for chain` loop of generate_model() calls (MALA.cpp:632-639, model_def.cpp:139-143).
 * Host pointers, row-major.  Synchronous: results are valid on return.
 *   params      Nchains x Nparams
 *   Tcoefs      Nchains temperatures (MALA.cpp:103)
 *   logL        Nchains, tempered: logL/T
 *   grad        NULL, or Nchains x Nvars
 *   model_rows  n_rows chain indices whose model spectrum is wanted (Model_def::model rows), or NULL
 *   model_out   n_rows x Nx, or NULL
 *   status      NULL or Nchains TAMCMC_CHAIN_* codes */
int tamcmc_eval_batch(tamcmc_ctx *ctx, int32_t Nchains, int32_t Nparams,
                      const double *params, const double *Tcoefs,
                      double *logL, double *grad,
                      int32_t n_rows, const int32_t *model_rows, double *model_out,
                      int32_t *status);

/* Same computation with DEVICE pointers (hipMalloc'ed on the ctx device), enqueued on the ctx stream
 * without synchronising: for a sampler that keeps chain state resident in HBM.
 * d_grad / d_status may be NULL. */
int tamcmc_eval_batch_device(tamcmc_ctx *ctx, int32_t Nchains, int32_t Nparams,
                             const double *d_params, const double *d_Tcoefs,
                             double *d_logL, double *d_grad, int32_t *d_status);

/* tamcmc_eval_batch split in two for callers that have host work to overlap with the GPU (the sampler draws the next
 * iteration's random numbers meanwhile): _begin copies params / Tcoefs and enqueues the likelihood evaluation, _end
 * waits and delivers logL / status.  Likelihood only (no gradient, no model rows); one batch in flight per ctx.
 * Both host-pointer entry points return as soon as every result has arrived in the library's pinned staging buffer,
 * which can be a few microseconds before the launch itself retires on the ctx stream; everything else in this API
 * that touches the ctx is ordered after it on that stream (or synchronises it). */
int tamcmc_eval_batch_begin(tamcmc_ctx *ctx, int32_t Nchains, int32_t Nparams, const double *params, const double *Tcoefs);
int tamcmc_eval_batch_end(tamcmc_ctx *ctx, int32_t Nchains, double *logL, int32_t *status);

#define TAMCMC_MAX_PARTS 4
/* Armed batch -- for a host loop whose next parameters depend on the results of the batch in flight (a sampler):
 * _arm puts the launches of the NEXT likelihood-only batch into the stream behind a one-wave gate kernel, while the
 * current batch is still being evaluated (allowed between _begin / _fire and _end of a batch of the same size; the
 * buffers must already be sized: tamcmc_ctx_reserve or an earlier batch), so that the launch calls cost nothing on the
 * critical path; _fire copies the parameters into the pinned input buffer and opens the gate with one store -- it takes
 * the place of _begin, and _end collects the results as usual.  While a batch is armed every other entry point of the
 * context returns TAMCMC_E_INVALID except _fire, _end (of the batch in flight), _disarm and tamcmc_ctx_destroy.  _disarm
 * opens the gate of a batch that will not be fired (it runs on the previous parameters, is waited for, and nothing is
 * handed out).  The gate itself gives up after ~4 s, so that a wave never outlives a host that died before firing; a
 * host that was merely held up that long loses time, not a result: _fire notices that the gate has expired (the armed
 * launches ran on stale input), lets them drain and evaluates the batch the plain way. */
int tamcmc_eval_batch_arm(tamcmc_ctx *ctx, int32_t Nchains);
/* One chain of the batch in flight (_begin / _fire, not yet _end): TAMCMC_OK with its logL and status once they have
 * arrived, TAMCMC_PENDING before.  Results arrive chain by chain (each is finalized by the last of its tiles), so a host
 * loop can start on a chain's accept step while the others are still being evaluated.  Read-only: any number of threads
 * may poll (different or the same chains) while no other entry point of the context is called; _end is still due, and
 * is the call that reports a failed launch -- a poller must bound its patience and then go there. */
int tamcmc_eval_batch_poll(const tamcmc_ctx *ctx, int32_t chain, double *logL, int32_t *status);
int tamcmc_eval_batch_fire(tamcmc_ctx *ctx, int32_t Nchains, int32_t Nparams, const double *params, const double *Tcoefs);
int tamcmc_eval_batch_disarm(tamcmc_ctx *ctx);

/* The same in up to TAMCMC_MAX_PARTS PARTS that may be in flight together (part = 0 .. 3; every part but 0 runs on a stream
 * of its own): chains
 * [first, first + Nchains) of the context's numbering -- the ranges of parts in flight must not overlap; with several
 * spectra resident `first` also indexes the chain -> spectrum map.  For a sampler that splits its chains in parts
 * and handles one part's results on the host while the GPU evaluates the others (chains are independent inside an
 * iteration, MALA.cpp:632-655).  params / Tcoefs point at the part's first row / entry.  A chain's result is bit for
 * bit that of tamcmc_eval_batch (tests/test_parity_gpu.py).  tamcmc_ctx_reserve sizes the context's buffers for Nchains
 * chains in total beforehand: they are never reallocated under a part in flight (a begin that would need to returns
 * TAMCMC_E_INVALID).  No whole-batch call may be made while a part is in flight. */
int tamcmc_ctx_reserve(tamcmc_ctx *ctx, int32_t Nchains);
int tamcmc_eval_batch_begin_part(tamcmc_ctx *ctx, int32_t part, int32_t first, int32_t Nchains, int32_t Nparams,
                                 const double *params, const double *Tcoefs);
int tamcmc_eval_batch_end_part(tamcmc_ctx *ctx, int32_t part, double *logL, int32_t *status);

/* Replaces: Model_def::call_model_explicit (model_def.cpp:199-208) as used by tools/getmodel.cpp:111.
 * One params row -> model spectrum (Nx doubles, host).  *status gets the TAMCMC_CHAIN_* code. */
int tamcmc_model_explicit(tamcmc_ctx *ctx, int32_t Nparams, const double *params,
                          double *model_out, int32_t *status);

/* Stream plumbing.  hip_stream is a hipStream_t created on the ctx device (NULL = the ctx's own stream). */
int tamcmc_ctx_set_stream(tamcmc_ctx *ctx, void *hip_stream);
int tamcmc_ctx_synchronize(tamcmc_ctx *ctx);

/* Kernel timing with HIP events recorded on the ctx stream around the dominant kernel of every eval call while
 * enabled (enable = 1), or of every n-th call (enable = n > 1: an event pair costs ~3 us of stream time, which a
 * throughput measurement over the same calls would otherwise carry in full).  tamcmc_ctx_kernel_time synchronises the stream, then returns the summed
 * duration and the number of launches since profiling was enabled. */
int tamcmc_ctx_profile(tamcmc_ctx *ctx, int enable);
int tamcmc_ctx_kernel_time(tamcmc_ctx *ctx, double *total_ms, int64_t *launches);

/* Shader-clock probe: _begin launches ONE wave on a stream of its own that watches the core-cycle counter against the
 * constant 100 MHz counter for `milliseconds`; _end waits for it and returns the mean core clock over that window.
 * Evaluations enqueued between the two calls run beside it, so this is the clock under that load (used by bench.py to
 * turn instruction counts into a fraction of the fp64 issue rate with a clock measured in the same run). */
int tamcmc_ctx_clock_probe_begin(tamcmc_ctx *ctx, double milliseconds);
int tamcmc_ctx_clock_probe_end(tamcmc_ctx *ctx, double *core_GHz, double *seconds);

/* Launch geometry actually used (for DESIGN.md / bench bookkeeping): bins_per_tile = the largest tile of the
 * likelihood-only launch (16 units of 512 bins; 8 when TAMCMC_EQUAL_COST balances the tiles, and always for the gradient
 * launch), tiles = tiles per chain of the most recent likelihood-only call. */
int tamcmc_ctx_geometry(tamcmc_ctx *ctx, int32_t *bins_per_tile, int32_t *tiles, int32_t *threads_per_block,
                        int32_t *n_multiplets);

int tamcmc_ctx_destroy(tamcmc_ctx *ctx);

int tamcmc_device_count(void);
const char *tamcmc_strerror(int code);
const char *tamcmc_last_hip_error(void);
const char *tamcmc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TAMCMC_ACCEL_H */
