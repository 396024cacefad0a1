"""Gradient of the tempered log-likelihood (new functionality: the reference has none, MALA.cpp:18).
Checked (a) ENTRY BY ENTRY against the oracle's analytic gradient (orc_grad_analytic: an independent statement of
SURVEY.md App. D, itself pinned per entry against finite differences on the CPU, tests/test_oracle_grad.py) -- with the
truncation window on, every model id, and at the benchmarked shape; tolerance in tests/gradcheck.py;
(b) as in round 1/2 against Richardson-extrapolated central differences of the ORACLE log-likelihood with trunc_c = 10000
(no truncation window, so logL is smooth; SURVEY.md App. D caveat), 2e-5 of the row's largest entry."""
import numpy as np
import pytest

import gradcheck
import workloads as W
from tamcmc_amd import synth

pytestmark = pytest.mark.gpu

TOL = 2e-5


def fd_check(accel_mod, orc, mid, w, y, sigma=None, like=0, nchains=2, rel_step=1e-6):
    P = W.perturbed(w, nchains, scale=0.002, seed=5)
    T = np.array([1.0, 3.7])[:nchains]
    idx = w["index_to_relax"]
    with accel_mod.Accel(mid, w["plength"], w["x"], y, sigma_y=sigma, likelihood_case=like) as acc:
        acc.set_vars(idx)
        logL, st, g = acc.eval_batch(P, T, grad=True)
        logL0, _ = acc.eval_batch(P, T)
    assert np.all(st == 0)
    assert np.allclose(logL, logL0, rtol=1e-13)      # K differs between the two kernels: same value
    for k in range(nchains):
        gfd, st2 = orc.grad_fd(mid, w["plength"], w["x"], y, P[k], T[k], idx, rel_step=rel_step, sigma_y=sigma,
                               likelihood_case=like)
        assert st2 == 0
        scale = np.max(np.abs(gfd))
        err = np.abs(g[k] - gfd) / scale
        assert np.max(err) <= TOL, (mid, k, np.argmax(err), g[k][np.argmax(err)], gfd[np.argmax(err)])


@pytest.mark.parametrize("mid", [2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("kw", [dict(), dict(asym=25.0, do_amp=True)], ids=["plain", "asym-amp"])
def test_gradient_vs_finite_differences(accel_mod, orc, mid, kw):
    w = W.any_model(mid, Nx=3000, trunc_c=10000.0, **kw)
    if kw:   # let the asymmetry vary too
        b = W.split(w) if mid not in (11, 14) else None
        relax = w["relax"].copy()
        s = (b["s"] if b else int(w["plength"][0] + w["plength"][1] + w["plength"][2:6].sum()))
        relax[s + 5] = 1
        relax[s + 1] = 1     # eta
        relax[s + 2] = 1     # a3
        w["relax"] = relax
        w["index_to_relax"] = np.flatnonzero(relax).astype(np.int32)
    m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=23)
    fd_check(accel_mod, orc, mid, w, y)


def _all_shape_variables(w, mid):
    """eta, a3 and the asymmetry join the variables (fixed in the synthetic star's .model)."""
    relax = w["relax"].copy()
    pl = w["plength"]
    s = int(pl[0] + pl[1] + pl[2:6].sum())
    relax[[s + 1, s + 2, s + 5]] = 1
    w["relax"] = relax
    w["index_to_relax"] = np.flatnonzero(relax).astype(np.int32)
    return w


@pytest.mark.parametrize("mid", [2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("kw", [dict(trunc_c=20.0), dict(trunc_c=7.0, asym=-40.0, do_amp=True), dict(trunc_c=20.0, asym=25.0),
                                dict(trunc_c=10000.0, do_amp=True)],
                         ids=["c20", "c7-asym-amp", "c20-asym", "notrunc-amp"])
def test_gradient_entrywise_vs_oracle_analytic(accel_mod, orc, mid, kw):
    """Every live Lorentzian model id, truncation window ON (trunc_c 20 and 7) and off, all variables incl. eta, a3 and
    asymmetry, 5000 bins (not a multiple of any tile size), 6 tempered chains: every entry against orc_grad_analytic."""
    w = _all_shape_variables(W.any_model(mid, Nx=5000, **kw), mid)
    m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
    assert st == 0
    y = synth.make_spectrum(m, seed=17)
    P = W.perturbed(w, 6, scale=0.004)
    T = synth.temperatures(6)
    with accel_mod.Accel(mid, w["plength"], w["x"], y) as acc:
        acc.set_vars(w["index_to_relax"])
        gradcheck.check_against_oracle(acc, orc, mid, w, y, P, T, tag=f"id {mid} {kw}")


@pytest.mark.parametrize("mid", [0, 1, 2, 11])
def test_gradient_entrywise_chi_square_and_p(accel_mod, orc, mid):
    """chi_square likelihood (likelihoods.cpp:31-39) and a truncated likelihood exponent p = 2.9 -> 2 (model_def.cpp:300)."""
    w = W.any_model(mid, Nx=3000)
    if mid in (2, 11):
        w = _all_shape_variables(w, mid)
    m, _ = orc.model(mid, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=19)
    sig = 0.05 + 0.2 * np.abs(np.sin(np.arange(y.size)))
    P = W.perturbed(w, 4, scale=0.003)
    T = synth.temperatures(4)
    with accel_mod.Accel(mid, w["plength"], w["x"], y, sigma_y=sig, likelihood_case=1) as acc:
        acc.set_vars(w["index_to_relax"])
        gradcheck.check_against_oracle(acc, orc, mid, w, y, P, T, sigma=sig, like=1, tag=f"id {mid} chi_square")
    with accel_mod.Accel(mid, w["plength"], w["x"], y, likelihood_p=2.9) as acc:
        acc.set_vars(w["index_to_relax"])
        gradcheck.check_against_oracle(acc, orc, mid, w, y, P, T, likelihood_p=2.9, tag=f"id {mid} p=2.9")


@pytest.mark.parametrize("mid", [2, 3])
def test_gradient_entrywise_c2_full_size(accel_mod, orc, mid):
    """BASELINE config C2 -- the benchmarked shape: 64 tempered chains x 1e5 bins, trunc_c = 20, all 44 (id 2) variables,
    every one of the 64 x 44 entries against the oracle's analytic gradient; logL of the gradient path vs the oracle."""
    w = synth.workload_c2(model_case=mid)
    m, _ = orc.model(mid, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m)
    P = synth.chain_params(w, 64)
    T = synth.temperatures(64)
    with accel_mod.Accel(mid, w["plength"], w["x"], y) as acc:
        acc.set_vars(w["index_to_relax"])
        L, st, g = acc.eval_batch(P, T, grad=True)
    assert np.all(st == 0)
    rL, rst = orc.generate_batch(mid, w["plength"], w["x"], y, P, T)
    assert np.max(np.abs(L - rL) / np.abs(rL)) <= 1e-10
    gradcheck.check_against_oracle(None, orc, mid, w, y, P, T, tag=f"C2 id {mid}", g=g)


@pytest.mark.parametrize("mid", [0, 1])
def test_gradient_gaussian_models(accel_mod, orc, mid):
    w = W.make_gauss(mid, Nx=2500)
    m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=29)
    fd_check(accel_mod, orc, mid, w, y)
    sig = 0.3 + 0.1 * np.cos(np.arange(y.size)) ** 2
    fd_check(accel_mod, orc, mid, w, y, sigma=sig, like=1)


def test_gradient_full_size_is_reproducible(accel_mod, orc):
    w = synth.workload_c2()
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m)
    P = synth.chain_params(w, 8)
    T = synth.temperatures(8)
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        acc.set_vars(w["index_to_relax"])
        L1, _, g1 = acc.eval_batch(P, T, grad=True)
        L2, _, g2 = acc.eval_batch(P, T, grad=True)
        L0, _, g0 = acc.eval_batch(P, np.ones(8), grad=True)
        L3, _, g3 = acc.eval_batch(P, T, grad=True)
    assert np.array_equal(g1, g2) and np.array_equal(L1, L2)
    assert np.array_equal(g1, g3) and np.array_equal(L1, L3)        # nothing of the launch in between is left behind
    assert np.all(np.isfinite(g1))
    # tempering scales the gradient like the likelihood: g(T) * T == g(T = 1)  (model_def.cpp:302); entries are sums
    # with cancellation, so the comparison is relative to the largest entry of the row
    assert np.max(np.abs(g1 * T[:, None] - g0) / np.max(np.abs(g0), axis=1, keepdims=True)) <= 1e-12
    assert np.allclose(L1 * T, L0, rtol=1e-13, atol=0)


def test_gradient_on_polynomial_background_tiles(accel_mod, orc, monkeypatch):
    """On the C2 grid (2300-3140 uHz) every tile evaluates the Harvey background as a polynomial in log x and the
    gradient kernel keeps moments of the weights instead of per-profile sums (tamcmc_backward.hip converts them).
    (a) finite differences of the oracle on a short piece of that grid, all variables incl. the 10 noise parameters;
    (b) full size: same logL and gradient as the exp()-per-bin path (developer switch TAMCMC_BG_EXACT=1)."""
    w = synth.workload_c2(Nx=6000, trunc_c=10000.0)
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    fd_check(accel_mod, orc, 2, w, synth.make_spectrum(m, seed=31))

    w = synth.workload_c2()
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m)
    P = synth.chain_params(w, 6)
    T = synth.temperatures(6)
    out = []
    for exact in ("0", "1"):
        monkeypatch.setenv("TAMCMC_BG_EXACT", exact)
        with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
            acc.set_vars(w["index_to_relax"])
            out.append(acc.eval_batch(P, T, grad=True))
    (L0, s0, g0), (L1, s1, g1) = out
    assert np.all(s0 == 0) and np.all(s1 == 0)
    assert np.allclose(L0, L1, rtol=1e-13, atol=0)
    scale = np.max(np.abs(g1), axis=1, keepdims=True)
    assert np.max(np.abs(g0 - g1) / scale) < 1e-11
    assert not np.array_equal(g0, g1)      # the two paths really are different code
    for g_path, name in ((g0, "moments"), (g1, "exact")):      # both against the oracle, entry by entry, at full size
        gradcheck.check_against_oracle(None, orc, 2, w, y, P, T, tag=f"background {name}", g=g_path)


def test_gradient_requires_vars(accel_mod, orc):
    w = W.make(2, Nx=1000)
    y = np.ones(1000)
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        with pytest.raises(accel_mod.AccelError) as e:
            acc.eval_batch(w["params_true"][None, :], np.ones(1), grad=True)
        assert e.value.code == accel_mod.capi.E_NOVARS
