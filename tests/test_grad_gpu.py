"""Gradient of the tempered log-likelihood (new functionality: the reference has none, MALA.cpp:18).
Checked against Richardson-extrapolated central differences of the ORACLE log-likelihood with
trunc_c = 10000 (no truncation window, so logL is smooth; SURVEY.md App. D caveat)."""
import numpy as np
import pytest

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


def test_gradient_requires_vars(accel_mod, orc):
    w = W.make(2, Nx=1000)
    y = np.ones(1000)
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        with pytest.raises(accel_mod.AccelError) as e:
            acc.eval_batch(w["params_true"][None, :], np.ones(1), grad=True)
        assert e.value.code == accel_mod.capi.E_NOVARS
