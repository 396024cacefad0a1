/*
 * tamcmc_io.h -- file-format front door of the hot path, host C++ behind a C ABI.
 *
 * Reads the reference's own configuration and input files and produces exactly what its `Config::setup` hands to
 * `MALA` and `Model_def`: the flat params row with `plength[11]`, the relax mask, the prior table, the cropped
 * spectrum and the integer ids of model / likelihood / prior.  Nothing here touches the GPU.
 *
 *   .data                      Config::read_data_ascii_Ncols        config.cpp:519-647
 *   crop to the .model range   Config::setup                        config.cpp:80-189
 *   .model (raw sections)      read_MCMC_file_MS_Global / _local    io_ms_global.cpp:25-306, io_local.cpp:26-302
 *   .model -> Input_Data       build_init_MS_Global / _local        io_ms_global.cpp:308-1088, io_local.cpp:304-1158
 *                              set_noise_params[_local]             io_ms_global.cpp:1091-1184, io_local.cpp:1160-1220
 *                              IO_models::fill_param / add_param    io_models.cpp:24-127
 *   config_default.cfg         Config::read_cfg_file                config.cpp:810-1320
 *   errors_default.cfg         Config::read_defautlerrors           config.cpp:1608-1667
 *   *_ctrl.list                Config::read_listfiles               config.cpp:1763-1813
 *   initial proposal errors    MALA::init_proposal (err = A var + B) MALA.cpp:246-257
 *   slice ranges               get_slices_range                     main.cpp:379-444
 *   phase presets              Config_presets::apply_presets        config_presets.cpp:39-200  (via tamcmc_setup_set)
 *
 * Where the reference prints a message and exit()s, these functions return TAMCMC_IO_E_* and keep the message
 * (tamcmc_setup_error).  What the reference prints as warnings while it interprets a .model file is kept in
 * tamcmc_setup_log.  The "simple matrix" .model reader of the two Gaussian toy models
 * (Config::read_inputs_prior_Simple_Matrix) is not provided: in the reference those models cannot be selected
 * (their names in Config/templates do not match models_ctrl.list / priors_ctrl.list).
 */
#ifndef TAMCMC_IO_H
#define TAMCMC_IO_H

#include <stdint.h>
#include "tamcmc_sampler.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tamcmc_setup tamcmc_setup;

enum {
    TAMCMC_IO_OK = 0,
    TAMCMC_IO_E_INVALID = 1,     /* bad argument / call order                        */
    TAMCMC_IO_E_OPEN = 2,        /* a file could not be opened                       */
    TAMCMC_IO_E_SYNTAX = 3,      /* the reference would print a syntax error + exit  */
    TAMCMC_IO_E_NAME = 4,        /* unknown model / prior / likelihood / keyword     */
    TAMCMC_IO_E_RANGE = 5,       /* frequency range incompatible with data / modes   */
    TAMCMC_IO_E_CAPACITY = 6     /* caller's buffer too small                        */
};

/* Config::Config: reads <config_dir>/config_default.cfg, errors_default.cfg and the four *_ctrl.list files
 * (config_dir = the reference's Config/default). */
int tamcmc_setup_create(tamcmc_setup **out, const char *config_dir);
int tamcmc_setup_create_files(tamcmc_setup **out, const char *cfg_file, const char *errors_file,
                              const char *models_list, const char *priors_list, const char *likelihoods_list,
                              const char *primepriors_list);

/* Override / query one keyword of config_default.cfg (what Config_presets::apply_presets does field by field).
 * group: "MALA", "Modeling", "Data", "Outputs", "Diagnostics"; values are the strings the .cfg would hold. */
int tamcmc_setup_set(tamcmc_setup *s, const char *group, const char *key, const char *value);
int tamcmc_setup_get(const tamcmc_setup *s, const char *group, const char *key, char *buf, int32_t cap);
/* Config_presets' phase rules: phase = "Burn-in" | "Learning" | "Acquire"  (config_presets.cpp:69-93,133-134) */
int tamcmc_setup_apply_phase(tamcmc_setup *s, const char *phase, int64_t Nsamples, double c0);

/* Config::setup(slice_ind): read the .data file, interpret the .model file with the reader that
 * Modeling.prior_fct_name selects (io_MS_Global | io_local), crop the data to the model's range. */
int tamcmc_setup_load(tamcmc_setup *s, const char *model_file, const char *data_file, int32_t slice_ind);

/* Stand-alone pieces for tools (tools/getmodel.cpp:76-84): the whole .data file as a malloc'ed row-major matrix
 * (free it with tamcmc_buffer_free), and the id of a name in a *_ctrl.list file. */
int tamcmc_data_file_read(const char *data_file, double **data, int64_t *nrows, int32_t *ncols);
void tamcmc_buffer_free(void *p);
int tamcmc_list_file_lookup(const char *list_file, const char *name, int32_t *id);

/* get_slices_range: the `* fmin fmax` lines of a .model file; ranges = n x 2 row-major. */
int tamcmc_model_file_slices(const char *model_file, double *ranges, int32_t cap_rows, int32_t *n);

/* ---- results of tamcmc_setup_load ---- */
int tamcmc_setup_sizes(const tamcmc_setup *s, int32_t *Nparams, int32_t *Nvars, int64_t *Nx, int32_t plength[11],
                       int32_t *model_case, int32_t *likelihood_case, int32_t *prior_case, double *likelihood_p);
/* inputs[Nparams], relax[Nparams], priors_names_switch[Nparams], priors[4 x Nparams] (row-major, like
 * Input_Data::priors), extra_priors[4], err[Nvars] (initial proposal errors).  Any pointer may be NULL. */
int tamcmc_setup_inputs(const tamcmc_setup *s, double *inputs, int32_t *relax, int32_t *priors_names_switch,
                        double *priors, double extra_priors[4], double *err);
int tamcmc_setup_data(const tamcmc_setup *s, double *x, double *y, double *sigma_y);
/* which: 0 inputs_names[i], 1 priors_names[i], 2 model_fullname, 3 star ID, 4 x label, 5 y label, 6 x unit, 7 y unit */
int tamcmc_setup_name(const tamcmc_setup *s, int32_t which, int32_t i, char *buf, int32_t cap);
/* raw sections of the .model file (MCMC_files): which: 0 Dnu, 1 numax, 2 C_l, 3 fmin, 4 fmax, 5 resolution */
double tamcmc_setup_scalar(const tamcmc_setup *s, int32_t which);
/* MALA group -> sampler configuration (Nchains, lambda_temp, c0, epsilon1/2, A1, dN_mixing, Nt_learn, periods_learn,
 * target_acceptance, prior id); seed and the chain block are left to the caller. */
int tamcmc_setup_sampler_cfg(const tamcmc_setup *s, tamcmc_sampler_cfg *cfg);

const char *tamcmc_setup_error(const tamcmc_setup *s);
const char *tamcmc_setup_log(const tamcmc_setup *s);
int tamcmc_setup_destroy(tamcmc_setup *s);

#ifdef __cplusplus
}
#endif
#endif
