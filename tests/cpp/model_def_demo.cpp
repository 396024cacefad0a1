// Drives the C++ Model_def-shaped adapter (include/tamcmc_model_def.hpp) the way the reference's
// sampler drives Model_def: per-chain generate_model() and the batched generate_models().
// Prints "logL <chain> <value>" lines; tests/test_cpp_adapter.py rebuilds the same inputs in numpy
// and checks them against the oracle.  Without a GPU the constructor throws (no CPU fallback) and
// the program prints "NODEVICE" and exits with code 3.
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "tamcmc_model_def.hpp"

int main()
{
    const long Nx = 3000;
    tamcmc::Data d;
    d.x.resize(Nx); d.y.resize(Nx);
    for (long i = 0; i < Nx; i++) {
        d.x[i] = 1000.0 + 2000.0 / Nx * i;
        d.y[i] = 1.0 + 0.5 * std::sin(0.01 * i) * std::sin(0.01 * i);
    }
    // model_Harvey_Gaussian (id 1): |p0| exp(-0.5 (x-p2)^2/p1^2) + harvey_like(|p3..p5|) + |p6|
    const std::vector<int32_t> plength = {3, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0};
    const std::vector<double> inputs = {5.0, 150.0, 2100.0, 3.0, 2.5, 2.2, 0.7};
    const std::vector<int32_t> relax = {1, 1, 1, 1, 0, 0, 1};
    const std::vector<double> T = {1.0, 2.5, 6.25};
    try {
        tamcmc::Model_def md(d, 1, plength, inputs, relax, T);
        for (long m = 0; m < md.Nmodels; m++) {
            for (long k = 0; k < md.Nvars; k++) md.vars[m * md.Nvars + k] *= 1.0 + 0.01 * (m + 1) * (k + 1);
            md.update_params_with_vars(m);
        }
        md.prior_fct = [](const double *p) { return (long double)(-0.001 * p[0]); };
        md.generate_models(T, true);
        for (long m = 0; m < md.Nmodels; m++) std::printf("logL %ld %.17g\n", m, md.logLikelihood[m]);
        for (long m = 0; m < md.Nmodels; m++) std::printf("post %ld %.17g\n", m, md.logPosterior[m]);
        for (long m = 0; m < md.Nmodels; m++) {
            const double one = (double)md.generate_model(&d, m, T);   // likelihood-only kernel: other tile size
            std::printf("single %ld %.17g\n", m, one);
        }
        std::printf("grad0");
        for (long k = 0; k < md.Nvars; k++) std::printf(" %.17g", md.gradLogLikelihood[k]);
        std::printf("\nmodel0 %.17g %.17g\n", md.model[0], md.model[Nx - 1]);
        // the reference's own prior function (priors_ctrl.list id 1 = priors_Harvey_Gaussian): Uniform, Jeffreys,
        // Gaussian on the first three parameters, the rest Fix (primepriors_ctrl.list ids 1, 4, 2, 0)
        const std::vector<int32_t> sw = {1, 4, 2, 0, 0, 0, 0};
        std::vector<double> pp(4 * 7, -9999.0);
        pp[0 * 7 + 0] = 0.0;   pp[1 * 7 + 0] = 20.0;      // Uniform(0, 20)
        pp[0 * 7 + 1] = 10.0;  pp[1 * 7 + 1] = 1000.0;    // Jeffreys(10, 1000)
        pp[0 * 7 + 2] = 2100.; pp[1 * 7 + 2] = 50.0;      // Gaussian(2100, 50)
        md.use_reference_priors(1, sw, pp, {0.0, 0.0, 0.0, 0.0});
        for (long m = 0; m < md.Nmodels; m++) std::printf("refprior %ld %.17g\n", m, (double)md.call_prior(&d, (int)m));
    } catch (const std::runtime_error &e) {
        std::printf("NODEVICE %s\n", e.what());
        return 3;
    }
    return 0;
}
