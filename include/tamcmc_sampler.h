/*
 * tamcmc_sampler.h -- host-side callers of the hot path, in C++
 * behind a C ABI like the rest of the library:
 *
 *   N1  log-priors            priors_calc.cpp:24-417, stats_dictionary.cpp:31-241,
 *                             derivatives_handler.cpp:342-455
 *   N2  batched sampler       MALA.cpp:53-112 (setup), :246-289 (init_proposal), :292-315 (update_proposal),
 *                             :335-353 (new_prop_values), :381-445 (parallel_tempering), :447-538
 *                             (update_position_MH), :608-692 (execute loop); random_JB.cpp:22-259 (RNG)
 *
 * The reference's "MALA" is an adaptive random-walk Metropolis with parallel tempering (no gradient,
 * MALA.cpp:18); this restates exactly that, with three deliberate differences, none of which changes a
 * result: (1) every chain's proposal is drawn first (same RNG order: u then z per chain, chain 0..N-1),
 * then ALL chains are evaluated in ONE evaluator call, then accepted -- legal because draws do not
 * depend on outcomes; (2) the seed is an argument instead of time(NULL), and the
 * generator is a private copy of glibc's rand() algorithm (TYPE_3 additive feedback), so runs are
 * reproducible and independent of other users of rand(); (3) the Cholesky factor is cached while the
 * proposal is not being adapted.  MaxChain = 24 (MALA.cpp:565-572) is not enforced.
 *
 * Sharded runs (one process per GPU): every rank creates the sampler with the GLOBAL chain count and its
 * own [chain_offset, chain_offset + n_local) block; every rank consumes the whole random stream (the draws of
 * chains owned elsewhere are passed over by a jump-ahead of the generator, not made), so the
 * chains are bit-identical to a single-process run.  A parallel-tempering pair that straddles two ranks
 * is exchanged by the caller (torch.distributed send/recv) through tamcmc_sampler_pt_* below.
 *
 * Environment switches read by tamcmc_sampler_create*: TAMCMC_SAMPLER_THREADS (host threads of the per-chain fork-join
 * pool), TAMCMC_SAMPLER_TIMING=1 (phase times printed at destroy), TAMCMC_SAMPLER_PIPELINE=2 (HIP evaluator, loops in the
 * library: the local chains as two sub-batches in flight, one handled on the host while the GPU evaluates the other;
 * same draws and decisions; off by default, measured slower at 64 chains x 1e5 bins), TAMCMC_SAMPLER_ARM=0 (HIP evaluator:
 * do NOT put the next iteration's launches into the stream ahead of its parameters -- tamcmc_eval_batch_arm / _fire, on by
 * default inside tamcmc_sampler_run, and inside _run_sharded when the process owns every chain; =2: also when the chains
 * are spread over several processes, where the boundary exchange runs between arming and firing), TAMCMC_SAMPLER_ARRIVE=0 (HIP evaluator: wait for the whole batch
 * before the accept pass instead of running a chain's accept step when its own result has arrived --
 * tamcmc_eval_batch_poll, on by default).  None of them changes a draw or a decision.
 */
#ifndef TAMCMC_SAMPLER_H
#define TAMCMC_SAMPLER_H

#include <stdint.h>
#include "tamcmc_accel.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamcmc_sampler tamcmc_sampler;

/* Evaluator: tempered log-likelihood of n chains (the hot path).  Returns 0 or an error code.
 * tamcmc_eval_batch has this shape once its ctx is bound (tamcmc_sampler_create_hip does that). */
typedef int (*tamcmc_eval_fn)(void *user, int32_t nchains, int32_t nparams, const double *params,
                              const double *Tcoefs, double *logL, int32_t *status);

#define TAMCMC_MAX_LEARN 8

typedef struct {
    int32_t Nchains;            /* global number of tempered chains                         MALA_cfg::Nchains     */
    int32_t chain_offset;       /* first chain owned by this process (0 for a single process)                      */
    int32_t Nchains_local;      /* chains owned by this process (= Nchains for a single process)                   */
    double  lambda_temp;        /* Tcoefs[m] = lambda_temp^m                                 MALA.cpp:103          */
    double  target_acceptance;  /* 0.234                                                     config_default.cfg:9  */
    double  c0, epsilon1, epsilon2, A1;  /* Robbins-Monro constants                          config_default.cfg:10-13 */
    int64_t dN_mixing;          /* PT attempt every dN_mixing iterations (i != 0)            MALA.cpp:676          */
    int32_t n_learn;            /* entries of Nt_learn (periods_learn has n_learn-1)         MALA.cpp:641-652      */
    int64_t Nt_learn[TAMCMC_MAX_LEARN];
    int64_t periods_learn[TAMCMC_MAX_LEARN];
    uint32_t seed;              /* srand(seed)                                               MALA.cpp:62-63        */
    int32_t prior_fct_switch;   /* priors_ctrl.list: 0 Test_Gaussian, 1 Harvey_Gaussian, 2 io_MS_Global, 3 io_local */
} tamcmc_sampler_cfg;

/* priors_params: n_prior_rows x Nparams, row-major (Input_Data::priors, one column per parameter);
 * priors_names_switch: Nparams primitive-prior ids (primepriors_ctrl.list); extra_priors: 4 values
 * (smooth switch, smoothness coefficient, a3/a1 limit, impose_normHnlm; priors_calc.cpp:28-31).
 * err: Nvars initial proposal errors (init_proposal's error[i] = var*frac + offset, MALA.cpp:249-257). */
int tamcmc_sampler_create(tamcmc_sampler **out, const tamcmc_sampler_cfg *cfg,
                          tamcmc_eval_fn eval, void *eval_user,
                          int32_t Nparams, const int32_t plength[11], const double *inputs, const int32_t *relax,
                          const int32_t *priors_names_switch, const double *priors_params, int32_t n_prior_rows,
                          const double extra_priors[4], const double *err);

/* Convenience: the evaluator is a HIP context (tamcmc_eval_batch on it). */
int tamcmc_sampler_create_hip(tamcmc_sampler **out, const tamcmc_sampler_cfg *cfg, tamcmc_ctx *ctx,
                              int32_t Nparams, const int32_t plength[11], const double *inputs, const int32_t *relax,
                              const int32_t *priors_names_switch, const double *priors_params, int32_t n_prior_rows,
                              const double extra_priors[4], const double *err);

/* Model_def constructor's initial generate_model() of every chain (model_def.cpp:139-143). */
int tamcmc_sampler_init(tamcmc_sampler *s);

/* One iteration of MALA::execute's loop body WITHOUT the parallel-tempering step (MALA.cpp:630-655). */
int tamcmc_sampler_mh_step(tamcmc_sampler *s);

/* Parallel tempering, split so that a boundary pair can be exchanged by the caller:
 *   pt_due    1 if this iteration attempts a swap (i % dN_mixing == 0 && i != 0)
 *   pt_draw   consumes u and A from the random stream (every rank calls it: same values everywhere)
 *   pt_local  performs the attempt when both chains are owned here; returns swapped 0/1 in *swapped
 *   pt_export copies {logL, logPrior, moved, Pmove, params[Nparams], vars[Nvars]} of an owned chain
 *   pt_import finishes a boundary attempt on this rank given the peer's exported record
 *   end_iteration advances the iteration counter (call once per iteration, after the PT step if any) */
int tamcmc_sampler_pt_due(const tamcmc_sampler *s);
int tamcmc_sampler_pt_draw(tamcmc_sampler *s, int32_t *A, double *u);
int tamcmc_sampler_pt_local(tamcmc_sampler *s, int32_t A, double u, int32_t *swapped, double *r_T);
int tamcmc_sampler_pt_record_size(const tamcmc_sampler *s);
int tamcmc_sampler_pt_export(const tamcmc_sampler *s, int32_t chain, double *record);
int tamcmc_sampler_pt_import(tamcmc_sampler *s, int32_t A, double u, const double *peer_record, int32_t *swapped, double *r_T);
int tamcmc_sampler_end_iteration(tamcmc_sampler *s);

/* Single process: n full iterations (mh_step + PT when due + end_iteration).
 * moved_hist (may be NULL): n x Nchains_local acceptance flags; swap_hist (may be NULL): n entries,
 * -1 no attempt, else 2*A + swapped. */
int tamcmc_sampler_run(tamcmc_sampler *s, int64_t n_iter, uint8_t *moved_hist, int32_t *swap_hist);

/* Sharded runs, the whole loop in the library (MALA.cpp:608-737 for this process's block of chains): n_iter iterations
 * of mh_step + parallel tempering + end_iteration.  A pair (A, A+1) owned here is swapped locally; when this process
 * owns exactly one end, `exchange` is called with this end's exported record (n_doubles = tamcmc_sampler_pt_record_size)
 * and must return the peer's record in recv -- the caller implements it as a neighbour send/recv (torch.distributed:
 * RCCL on GPUs, gloo in the CPU tests; tamcmc-c-_amd/sharded.py); it returns 0 or non-zero on failure.  Processes that
 * own neither end do not communicate.  moved_hist / swap_hist as in tamcmc_sampler_run, except that an attempt this
 * process took no part in is recorded as -2.  With a block (below) the loop stops early when the block is full; *done
 * (may be NULL) receives the number of iterations completed, also when the call returns an error.
 * Errors: the call returns at once when the evaluator, a swap step or the exchange fails on THIS process; it cannot go on
 * answering its neighbours (the random stream of the failed iteration is spent, its chains have no state to send), so a
 * neighbour that reaches a boundary pair with it would wait in its send/recv.  The caller must therefore take the whole
 * process group down on a non-zero return (tamcmc-c-_amd/sharded.py does: it aborts the group, the peers' pending
 * send/recv fail and they raise in turn); a rank must not simply leave the loop and idle. */
typedef int (*tamcmc_exchange_fn)(void *user, int32_t my_chain, int32_t peer_chain, const double *send, double *recv,
                                  int32_t n_doubles);
typedef struct tamcmc_shard_block tamcmc_shard_block;
int tamcmc_sampler_run_sharded(tamcmc_sampler *s, int64_t n_iter, tamcmc_exchange_fn exchange, void *user,
                               tamcmc_shard_block *block, uint8_t *moved_hist, int32_t *swap_hist, int64_t *done);
/* A process's share of an output block (Outputs::update_buffer_*, outputs.cpp:863-1027): the samples of its chains and
 * the running sums of the proposal parameters, filled by tamcmc_sampler_run_sharded, gathered by the caller once per
 * block (rank 0 then writes the files through tamcmc_outputs_push_block).  data(which): 0 vars [n][nloc][Nvars],
 * 1 stat [n][3][nloc] = logL | logPrior | logPost, 2 moved [n][nloc], 3 pt [n][4] = attempted, chain A, Pswitch (NaN
 * unless an owner), switched (-1 unless an owner); sums over the block: 4 sigma, 5 mu, 6 covarmat, 7 vars. */
int tamcmc_shard_block_create(tamcmc_shard_block **out, const tamcmc_sampler *s, int64_t capacity);
int tamcmc_shard_block_data(tamcmc_shard_block *b, int32_t which, double **ptr, int64_t *count);
int64_t tamcmc_shard_block_count(const tamcmc_shard_block *b);
int tamcmc_shard_block_reset(tamcmc_shard_block *b);
int tamcmc_shard_block_destroy(tamcmc_shard_block *b);

/* Wall time spent per phase of the loop since set_timing(s, 1): seconds[0..7] = proposals, launch, priors, draws of the
 * next iteration (overlapped with the GPU), wait, accept, raw draws of the chains owned by OTHER processes (the
 * replicated-stream term of a sharded run, included in [3] or [0]), boundary exchange. */
int tamcmc_sampler_set_timing(tamcmc_sampler *s, int32_t enable);
int tamcmc_sampler_get_timing(const tamcmc_sampler *s, double seconds[8], int64_t *iterations);

/* State access (local chains, row-major). which: 0 vars, 1 params, 2 logLikelihood (tempered), 3 logPrior,
 * 4 logPosterior, 5 Pmove, 6 sigma, 7 mu, 8 covarmat (n_local x Nvars x Nvars), 9 Tcoefs (local), 10 moved (0/1) */
int tamcmc_sampler_get(const tamcmc_sampler *s, int32_t which, double *out, int64_t capacity);
/* Restore a saved state before tamcmc_sampler_init (Config::read_restore_files, config.cpp:1322-1577; used by
 * Model_def's ctor, model_def.cpp:100-137, and MALA::restore_proposal, MALA.cpp:190-238).  which: 0 vars (the params
 * rows follow), 6 sigma, 7 mu, 8 covarmat; count = number of doubles.  set_iteration = do_restore_last_index. */
int tamcmc_sampler_set(tamcmc_sampler *s, int32_t which, const double *in, int64_t count);
int tamcmc_sampler_set_iteration(tamcmc_sampler *s, int64_t iteration);
int64_t tamcmc_sampler_iteration(const tamcmc_sampler *s);
int32_t tamcmc_sampler_nvars(const tamcmc_sampler *s);
int32_t tamcmc_sampler_nlocal(const tamcmc_sampler *s);   /* chains owned by this process */
int tamcmc_sampler_layout(const tamcmc_sampler *s, int32_t *Nchains, int32_t *chain_offset, int32_t *Nchains_local);
int tamcmc_sampler_destroy(tamcmc_sampler *s);

/* N1 entry points, exported for tests (known answers of stats_dictionary.cpp:252-326). */
double tamcmc_logP_primitive(int32_t prior_id, const double p[4], double x);
double tamcmc_log_prior(int32_t prior_fct_switch, int32_t Nparams, const double *params, const int32_t plength[11],
                        const int32_t *priors_names_switch, const double *priors_params, int32_t n_prior_rows,
                        const double extra_priors[4], int32_t *error);
/* consecutive r8vec_normal_01 calls (random_JB.cpp:22-213) of the given sizes after srand(seed); split selects the
 * one-piece routine (0) or the draw/fill pair the sampler uses (1): both must give the same numbers (test hook) */
void tamcmc_normals(uint32_t seed, int32_t ncalls, const int32_t *sizes, double *out, int32_t split);
/* the private copy of glibc's rand(): fills out[n] after srand(seed) (test hook) */
void tamcmc_glibc_rand(uint32_t seed, int32_t n, int32_t *out);
/* the same after `skip` values were passed over by the generator's jump-ahead (a sharded sampler jumps over the draws of
 * the chains other processes own): must equal tamcmc_glibc_rand's values skip .. skip + n - 1 (test hook) */
void tamcmc_glibc_rand_jump(uint32_t seed, uint64_t skip, int32_t n, int32_t *out);
/* the sampler's lower Cholesky factor of a row-major n x n matrix (MALA.cpp:344, tmpmat.llt().matrixL()): L row-major,
 * zeros above the diagonal; returns 0, or 1 when a pivot is not positive (the columns before it are filled in) -- test
 * hook: the blocked form must be the textbook factor bit for bit */
int tamcmc_host_cholesky(const double *A, int32_t n, double *L);

#ifdef __cplusplus
}
#endif
#endif
