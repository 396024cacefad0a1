"""The hot path driven from the reference's own input files (tests/golden/ref_inputs: the .model / .data files of
the reference's test directory): reader (N3) -> HIP evaluation (C ABI) vs the oracle on the same arrays, and the
sampler (N1 + N2) started from a .model file.  The measured spectrum TF_3443483_local-v3.data is the only real data
set the reference ships."""
import os

import numpy as np
import pytest

import gradcheck
from tamcmc_amd import sampler as S
from tamcmc_amd.setup_io import Setup, model_file_slices

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
CFG = os.path.join(G, "Config_default")
TF_MODEL = os.path.join(G, "TF_3443483_local-v3.model")
TF_DATA = os.path.join(G, "TF_3443483_local-v3.data")
RTOL_LOGL = 1e-10          # north-star tolerance
RTOL_MODEL = 1e-12


def chains_around(s, n, seed=3, scale=0.3):
    """n parameter rows: the file's initial guesses, then Gaussian steps of `scale` x the initial proposal errors."""
    rng = np.random.default_rng(seed)
    P = np.tile(s.inputs, (n, 1))
    P[1:, s.index_to_relax] += scale * s.err * rng.standard_normal((n - 1, s.Nvars))
    return P


def check_against_oracle(accel_mod, orc, s, P, T, grad=True):
    with accel_mod.Accel(s.model_case, s.plength, s.x, s.y, likelihood_case=s.likelihood_case,
                         likelihood_p=s.likelihood_p) as acc:
        acc.set_vars(s.index_to_relax)
        logL, st, models = acc.eval_batch(P, T, model_rows=[0, len(P) - 1])
        ref_logL, ref_st = orc.generate_batch(s.model_case, s.plength, s.x, s.y, P, T, likelihood_p=s.likelihood_p)[:2]
        assert np.array_equal(st, ref_st) and np.all(st == 0)
        assert np.allclose(logL, ref_logL, rtol=RTOL_LOGL, atol=0)
        for row, m in zip((0, len(P) - 1), models):
            ref_m = orc.model(s.model_case, P[row], s.plength, s.x)[0]
            assert np.allclose(m, ref_m, rtol=RTOL_MODEL, atol=0)
        if grad:
            logL2, st2, g = acc.eval_batch(P, T, grad=True)
            assert np.allclose(logL2, logL, rtol=1e-13) and np.all(np.isfinite(g)) and g.shape == (len(P), s.Nvars)
            # every gradient entry against the oracle's analytic gradient (tolerance: tests/gradcheck.py), whatever the
            # file's truncation constant
            ref, ref_abs, _, gst = orc.grad_analytic(s.model_case, s.plength, s.x, s.y, P, T, s.index_to_relax,
                                                     likelihood_p=s.likelihood_p)
            assert np.all(gst == 0)
            gradcheck.assert_grad_entrywise(g, ref, ref_abs, tag=f"model {s.model_case}, {s.Nx} bins")
    return logL


def test_every_slice_of_the_reference_spectrum(accel_mod, orc):
    for k in range(len(model_file_slices(TF_MODEL))):
        s = Setup(CFG).load(TF_MODEL, TF_DATA, k)
        P = chains_around(s, 6, seed=10 + k)
        T = 1.7 ** np.arange(6)                                 # lambda_temp of config_default.cfg
        logL = check_against_oracle(accel_mod, orc, s, P, T)
        assert np.all(np.isfinite(logL))


def test_gradient_on_the_reference_spectrum(accel_mod, orc):
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 2)
    P = chains_around(s, 3, seed=5, scale=0.1)
    T = np.array([1.0, 1.7, 2.89])
    with accel_mod.Accel(s.model_case, s.plength, s.x, s.y) as acc:
        acc.set_vars(s.index_to_relax)
        _, _, g = acc.eval_batch(P, T, grad=True)
    # trunc_c = 10000 in this file (no `trunc_c` keyword): logL is smooth, finite differences of the oracle apply
    for m in range(3):
        fd, fst = orc.grad_fd(s.model_case, s.plength, s.x, s.y, P[m], T[m], s.index_to_relax)
        assert fst == 0
        scale = np.max(np.abs(fd))
        assert np.allclose(g[m], fd, rtol=0, atol=2e-5 * scale), (m, np.max(np.abs(g[m] - fd)) / scale)


@pytest.mark.parametrize("name,reader", [("kplr008379927_kasoc-psd_slc_v2_1000.model", "io_MS_Global"),
                                         ("00088.0.model", "io_MS_Global"), ("02194.0.model", "io_MS_Global"),
                                         ("kplr008379927_kasoc-psd_slc_v2_1000_local-v2.model", "io_local")])
def test_reference_model_files_on_synthetic_spectra(accel_mod, orc, tmp_path, name, reader):
    """The other .model files of the reference come without their spectra: a chi^2_2 realisation of the file's own
    initial model stands in for the data."""
    path = os.path.join(G, name)
    lo, hi = model_file_slices(path)[0]
    step = 0.02
    x = np.arange(lo - 5 * step, hi + 5 * step, step)
    d = str(tmp_path / "flat.data")
    with open(d, "w") as f:
        f.write("# flat\n! frequency power\n* (microHz) (ppm^2/microHz)\n" + "".join("%.8f 1.0\n" % v for v in x))
    s = Setup(CFG)
    s.set("Modeling", "prior_fct_name", reader)
    s.load(path, d, 0)
    with accel_mod.Accel(s.model_case, s.plength, s.x, np.ones(s.Nx)) as a0:
        m_true, st = a0.model_explicit(s.inputs)
    assert st == 0 and np.all(m_true > 0)
    rng = np.random.default_rng(11)
    s.y = m_true * (-np.log(rng.uniform(size=s.Nx)))
    P = chains_around(s, 5, seed=21, scale=0.2)
    T = 1.7 ** np.arange(5)
    check_against_oracle(accel_mod, orc, s, P, T)


def test_sampler_from_reference_files(accel_mod, orc):
    """Adaptive Metropolis + parallel tempering started from TF_3443483 slice 1 with the priors of the .model file:
    HIP evaluator and oracle evaluator must give the same accept/reject and swap history."""
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 0)
    s.set("MALA", "Nchains", 6)
    s.set("MALA", "Nt_learn", "20, 60, 100000")
    cfg = s.sampler_cfg(seed=1234)

    def oracle_eval(P, T):
        logL, st = orc.generate_batch(s.model_case, s.plength, s.x, s.y, P, T, likelihood_p=s.likelihood_p)[:2]
        return logL, st

    hist = []
    with accel_mod.Accel(s.model_case, s.plength, s.x, s.y) as acc:
        for ev in (acc, oracle_eval):
            smp = S.Sampler(cfg, ev, s.plength, s.inputs, s.relax, s.err, s.priors_names_switch, s.priors, s.extra_priors)
            smp.init()
            lp0 = smp.get("logPrior")
            assert np.all(np.isfinite(lp0)), "the file's initial guesses must lie inside its own priors"
            moved, swaps = smp.run(150)
            hist.append((moved.copy(), swaps.copy(), smp.get("vars"), smp.get("logL")))
            smp.close()
    assert np.array_equal(hist[0][0], hist[1][0]) and np.array_equal(hist[0][1], hist[1][1])
    assert np.allclose(hist[0][2], hist[1][2], rtol=1e-9) and np.allclose(hist[0][3], hist[1][3], rtol=1e-9)
    acc_rate = hist[0][0][:, 0].mean()
    assert 0.02 < acc_rate < 0.95
