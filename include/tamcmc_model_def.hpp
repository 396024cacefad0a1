// tamcmc_model_def.hpp -- C++ host adapter with the shape of the reference's Model_def class
// (tamcmc/headers/model_def.h:23-83, tamcmc/sources/model_def.cpp) on top of the C ABI of
// tamcmc_accel.h.  Header-only, no Eigen: matrices are row-major std::vector<double>.
//
// Same public members and methods as the reference class, so a sampler written against Model_def
// (MALA.cpp:447-538) reads the same:
//   params (Nmodels x Nparams), vars (Nmodels x Nvars), model (Nmodels x Nx, rows filled on demand),
//   logLikelihood (ALREADY divided by the chain temperature, model_def.cpp:302), logPrior,
//   logPosterior, Pmove, moved, swaped, Pswap, comparator_MH, comparator_PT;
//   call_model / call_model_explicit / update_params_with_vars / call_likelihood / call_prior /
//   generate_model.
// New: generate_models() evaluates every chain in ONE device call (what the OpenMP loop of
// MALA.cpp:632-639 becomes), optionally with d(logL/T)/dvars.
// Errors: the reference prints and exit()s; this throws std::runtime_error with the library's message.
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "tamcmc_accel.h"
#include "tamcmc_sampler.h"

namespace tamcmc {

struct Data {                      // data.h:24-36
    std::vector<double> x, y, sigma_y;
    long Nx() const { return (long)x.size(); }
};

class Model_def {
public:
    std::vector<double> params;          // Nmodels x Nparams
    std::vector<double> vars;            // Nmodels x Nvars
    std::vector<double> model;           // Nmodels x Nx
    std::vector<double> logLikelihood, logPrior, logPosterior, Pmove;
    std::vector<double> gradLogLikelihood;   // Nmodels x Nvars (new)
    std::vector<int32_t> status;             // per-chain TAMCMC_CHAIN_* (new; replaces exit())
    std::vector<bool> moved;
    bool swaped = false;
    long double Pswap = 0;
    std::vector<double> comparator_MH;
    long double comparator_PT = 0;

    long Nmodels = 0, Nparams = 0, Nvars = 0, Ncons = 0;
    std::vector<int32_t> plength, relax, index_to_relax;
    int model_fct_name_switch = 0, likelihood_fct_name_switch = 0;
    double likelihood_params = 1.0;
    std::function<long double(const double *params_row)> prior_fct;   // call_prior hook (host side)

    // call_prior as the reference defines it (model_def.cpp:322-356 -> priors_calc.cpp): prior_fct_name_switch from
    // priors_ctrl.list, the primitive-prior ids and the 4 x Nparams prior table of Input_Data, extra_priors[4]
    void use_reference_priors(int prior_fct_name_switch, const std::vector<int32_t> &priors_names_switch,
                              const std::vector<double> &priors /* 4 x Nparams, row-major */, const std::vector<double> &extra_priors)
    {
        if ((long)priors_names_switch.size() != Nparams || (long)priors.size() != 4 * Nparams || extra_priors.size() != 4)
            throw std::runtime_error("use_reference_priors: table sizes do not match Nparams");
        auto sw = priors_names_switch; auto pp = priors; auto ex = extra_priors; auto pl = plength;
        const int np = (int)Nparams;
        prior_fct = [sw, pp, ex, pl, np, prior_fct_name_switch](const double *row) -> long double {
            int32_t err = 0;
            const double v = tamcmc_log_prior(prior_fct_name_switch, np, row, pl.data(), sw.data(), pp.data(), 4, ex.data(), &err);
            if (err) throw std::runtime_error("tamcmc_log_prior: invalid prior configuration");
            return v;
        };
    }

    // Model_def(Config*, VectorXd Tcoefs, bool) -- model_def.cpp:27-181
    Model_def(const Data &data, int model_case, const std::vector<int32_t> &plength_, const std::vector<double> &inputs,
              const std::vector<int32_t> &relax_, const std::vector<double> &Tcoefs, int likelihood_case = 0,
              double likelihood_p = 1.0, int device_id = 0)
        : plength(plength_), relax(relax_), model_fct_name_switch(model_case),
          likelihood_fct_name_switch(likelihood_case), likelihood_params(likelihood_p), data_(data)
    {
        if (plength.size() != 11) throw std::runtime_error("plength must have 11 entries");
        Nmodels = (long)Tcoefs.size();
        for (int v : plength) Nparams += v;
        if ((long)inputs.size() != Nparams || (long)relax.size() != Nparams) throw std::runtime_error("inputs/relax size");
        for (long i = 0; i < Nparams; i++) if (relax[i] == 1) index_to_relax.push_back((int32_t)i);   // model_def.cpp:81-94
        Nvars = (long)index_to_relax.size();
        Ncons = Nparams - Nvars;
        params.resize(Nmodels * Nparams);
        vars.resize(Nmodels * Nvars);
        for (long m = 0; m < Nmodels; m++) {
            for (long i = 0; i < Nparams; i++) params[m * Nparams + i] = inputs[i];
            for (long k = 0; k < Nvars; k++) vars[m * Nvars + k] = inputs[index_to_relax[k]];
        }
        model.assign(Nmodels * data.Nx(), 0.0);
        logLikelihood.assign(Nmodels, 0.0); logPrior.assign(Nmodels, 0.0); logPosterior.assign(Nmodels, 0.0);
        Pmove.assign(Nmodels, 0.0); comparator_MH.assign(Nmodels, 0.0); moved.assign(Nmodels, false);
        gradLogLikelihood.assign(Nmodels * Nvars, 0.0); status.assign(Nmodels, 0);
        check(tamcmc_ctx_create(&ctx_, device_id, model_case, likelihood_case, likelihood_p, plength.data(),
                                (int64_t)data.Nx(), data.x.data(), data.y.data(),
                                data.sigma_y.empty() ? nullptr : data.sigma_y.data()), "tamcmc_ctx_create");
        if (Nvars > 0) check(tamcmc_ctx_set_vars(ctx_, (int32_t)Nvars, index_to_relax.data()), "tamcmc_ctx_set_vars");
    }
    ~Model_def() { tamcmc_ctx_destroy(ctx_); }
    Model_def(const Model_def &) = delete;
    Model_def &operator=(const Model_def &) = delete;

    // model_def.cpp:370-378
    void update_params_with_vars(long m)
    {
        for (long k = 0; k < Nvars; k++) params[m * Nparams + index_to_relax[k]] = vars[m * Nvars + k];
    }

    // model_def.cpp:210-289
    std::vector<double> call_model(const Data *, int m)
    {
        std::vector<double> out(data_.Nx());
        check(tamcmc_model_explicit(ctx_, (int32_t)Nparams, &params[m * Nparams], out.data(), &status[m]), "tamcmc_model_explicit");
        return out;
    }

    // model_def.cpp:199-208 (tools/getmodel.cpp:111)
    static std::vector<double> call_model_explicit(const Data &data, const std::vector<int32_t> &plength0,
                                                   const std::vector<double> &params0, int model_case, int device_id = 0)
    {
        tamcmc_ctx *c = nullptr;
        int rc = tamcmc_ctx_create(&c, device_id, model_case, 0, 1.0, plength0.data(), (int64_t)data.Nx(), data.x.data(),
                                   data.y.data(), nullptr);
        if (rc != TAMCMC_OK) throw std::runtime_error(std::string("tamcmc_ctx_create: ") + tamcmc_strerror(rc));
        std::vector<double> out(data.Nx());
        int32_t st = 0;
        rc = tamcmc_model_explicit(c, (int32_t)params0.size(), params0.data(), out.data(), &st);
        tamcmc_ctx_destroy(c);
        if (rc != TAMCMC_OK) throw std::runtime_error(std::string("tamcmc_model_explicit: ") + tamcmc_strerror(rc));
        return out;
    }

    // model_def.cpp:291-320
    long double call_likelihood(const Data *, int m, const std::vector<double> &Tcoefs)
    {
        double L = 0;
        check(tamcmc_eval_batch(ctx_, 1, (int32_t)Nparams, &params[m * Nparams], &Tcoefs[m], &L, nullptr, 0, nullptr,
                                nullptr, &status[m]), "tamcmc_eval_batch");
        return L;
    }

    // model_def.cpp:322-356
    long double call_prior(const Data *, int m) { return prior_fct ? prior_fct(&params[m * Nparams]) : 0.0L; }

    // model_def.cpp:358-367
    long double generate_model(const Data *d, long m, const std::vector<double> &Tcoefs)
    {
        const int32_t row = 0;
        double L = 0;
        check(tamcmc_eval_batch(ctx_, 1, (int32_t)Nparams, &params[m * Nparams], &Tcoefs[m], &L, nullptr, 1, &row,
                                &model[m * data_.Nx()], &status[m]), "tamcmc_eval_batch");
        logLikelihood[m] = L;
        logPrior[m] = (double)call_prior(d, (int)m);
        logPosterior[m] = logLikelihood[m] + logPrior[m];
        return logPosterior[m];
    }

    // every chain in one device call; with_grad fills gradLogLikelihood
    void generate_models(const std::vector<double> &Tcoefs, bool with_grad = false)
    {
        check(tamcmc_eval_batch(ctx_, (int32_t)Nmodels, (int32_t)Nparams, params.data(), Tcoefs.data(), logLikelihood.data(),
                                with_grad ? gradLogLikelihood.data() : nullptr, 0, nullptr, nullptr, status.data()),
              "tamcmc_eval_batch");
        for (long m = 0; m < Nmodels; m++) {
            logPrior[m] = (double)call_prior(nullptr, (int)m);
            logPosterior[m] = logLikelihood[m] + logPrior[m];
        }
    }

    tamcmc_ctx *ctx() { return ctx_; }

private:
    static void check(int rc, const char *where)
    {
        if (rc != TAMCMC_OK)
            throw std::runtime_error(std::string(where) + ": " + tamcmc_strerror(rc) + " | " + tamcmc_last_hip_error());
    }
    const Data &data_;
    tamcmc_ctx *ctx_ = nullptr;
};

}  // namespace tamcmc
