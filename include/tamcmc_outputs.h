/*
 * tamcmc_outputs.h -- result files of a run in the reference's formats, restore files, and the phase driver
 * (SURVEY.md section 8f, row N4).  Host C++ behind a C ABI; nothing here touches the GPU except through the
 * sampler's evaluator.
 *
 *   <root>params_chain-<m>.bin + <root>params.hdr      Outputs::write_bin_params            outputs.cpp:1231-1334
 *   <root>stat_criteria.bin + .hdr                     Outputs::write_bin_stat_criteria     outputs.cpp:1472-1549
 *   <root>parallel_tempering.bin + .hdr                Outputs::write_bin_parallel_temp_params  outputs.cpp:1336-1404
 *   the same three as text (file_format=text|debug)    Outputs::write_txt_*                 outputs.cpp:510-676,791-861
 *   <root>acceptance.txt                               Outputs::write_txt_acceptance        outputs.cpp:747-789
 *   <restore_dir><restore_file_out>{1,2,3}.dat         Outputs::write_buffer_restore        outputs.cpp:863-1027
 *   reading them back                                  Config::read_restore_files           config.cpp:1322-1577
 *   buffering cadence (Nbuffer)                        Outputs::update_buffer_*             outputs.cpp:1552-1818
 *   the loop                                           MALA::execute                        MALA.cpp:608-720
 *
 * Binary layouts (little-endian, no padding): params_chain-<m>.bin = Nvars doubles per sample;
 * stat_criteria.bin = 3*Nchains doubles per sample [logL | logPrior | logPost]; parallel_tempering.bin =
 * {uint8 attempted, int32 chain0, double Pswitch, uint8 switched} = 14 bytes per sample.
 *
 * Deliberate differences from the reference, none of which changes a format:
 *  - the reference's buffer logic writes, as the LAST row of every file, a row of its buffer that has not been
 *    filled yet (stale or uninitialised memory) and drops the last sample (outputs.cpp:1555-1585: the flush branch
 *    runs before the new sample is stored).  Here the last row is the last sample.
 *  - acceptance.txt: the rate of a block is counted over the samples of that block (the reference counts over
 *    Nbuffer buffer rows, some of them stale, after overwriting row 0; outputs.cpp:1730-1747,1824-1857).
 *  - restore files: "*_mean" entries are plain means over the samples of the last block (the reference divides a
 *    sum of count+1 terms by count, outputs.cpp:1772-1781).  Numbers are written with 17 significant digits when
 *    restore_precision=17 is requested (default 6, the reference's stream default, which does not round-trip).
 *  - proposal-parameter and model dumps (get_proposal_params, get_models; debug aids, off by default, the second one
 *    disabled by the reference itself for Nx <= 5000 through an inverted test, outputs.cpp:119-127) are not written.
 */
#ifndef TAMCMC_OUTPUTS_H
#define TAMCMC_OUTPUTS_H

#include <stdint.h>
#include "tamcmc_io.h"
#include "tamcmc_sampler.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamcmc_outputs tamcmc_outputs;

/* Outputs::Outputs.  File names, Nsamples, Nbuffer, erase_old_files, get_* and file_format come from the setup's
 * Outputs group; names / relax / plength / constants from the loaded .model.  iteration0 = samples already done by a
 * previous run that is being appended to (restored_vals.iteration), else 0. */
int tamcmc_outputs_create(tamcmc_outputs **out, const tamcmc_setup *setup, int32_t Nchains, const double *Tcoefs,
                          int64_t iteration0, int32_t restore_precision);
/* One call per iteration after the parallel-tempering step (MALA.cpp:690-702).  attempted/chain_A/Pswap/swapped as
 * MALA::execute passes them (chain_A = -1 and Pswap = last value when no attempt was made). */
int tamcmc_outputs_record(tamcmc_outputs *o, const tamcmc_sampler *s, int32_t attempted, int32_t chain_A, double Pswap,
                          int32_t swapped);
/* A whole block of n <= Nbuffer samples at once, from arrays covering ALL chains (sharded runs: rank 0 gathers the
 * ranks' local blocks, tamcmc-c-_amd/sharded.py).  vars[n][Nchains][Nvars], stat[n][3 Nchains] = [logL | logPrior |
 * logPost], moved[n][Nchains]; attempted / chain0 / Pswitch / switched [n]; last_* = state after the block's last sample,
 * sum_* = sums over the block (for the *_mean entries of the restore files).  Writes the block and the restore files. */
int tamcmc_outputs_push_block(tamcmc_outputs *o, int64_t n, const double *vars, const double *stat, const uint8_t *moved,
                              const uint8_t *attempted, const int32_t *chain0, const double *Pswitch, const uint8_t *switched,
                              const double *last_vars, const double *last_sigma, const double *last_mu, const double *last_covar,
                              const double *sum_vars, const double *sum_sigma, const double *sum_mu, const double *sum_covar);
/* Flush what is buffered and write the restore files from the sampler's current state. */
int tamcmc_outputs_finish(tamcmc_outputs *o, const tamcmc_sampler *s);
int tamcmc_outputs_destroy(tamcmc_outputs *o);
const char *tamcmc_outputs_error(const tamcmc_outputs *o);

/* Config::read_restore_files + the use MALA / Model_def make of it: according to the setup's Outputs keys
 * (do_restore_variables, do_restore_proposal, do_restore_proposal_mean, do_restore_last_index, restore_dir,
 * restore_file_in) load <restore_dir><restore_file_in>{1,2,3}.dat into the sampler.  Call before tamcmc_sampler_init.
 * iteration (may be NULL) receives the restored iteration (0 unless do_restore_last_index). */
int tamcmc_restore_apply(const tamcmc_setup *setup, tamcmc_sampler *s, int64_t *iteration, char *errbuf, int32_t errcap);

/* One phase of MALA::execute in a single process: restore (if configured) -> init -> iterate up to Outputs.Nsamples
 * -> result + restore files.  The sampler must have been created from the same setup.  progress (may be NULL) is
 * called every Nbuffer samples with (iteration, Nsamples, user). */
typedef void (*tamcmc_progress_fn)(int64_t iteration, int64_t Nsamples, void *user);
int tamcmc_run_phase(const tamcmc_setup *setup, tamcmc_sampler *s, tamcmc_progress_fn progress, void *user,
                     int32_t restore_precision, char *errbuf, int32_t errcap);

#ifdef __cplusplus
}
#endif
#endif
