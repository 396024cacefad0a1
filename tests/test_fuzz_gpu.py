"""Randomized HIP-vs-oracle parity sweep: model id, grid length (2 ... 60000 bins: fused one-tile launches, short-grid
tiles, full-size tiles), truncation constant, asymmetry, amplitude mode, likelihood, chain count and parameter scatter
are drawn at random.  60 cases by default; TAMCMC_FUZZ_CASES / TAMCMC_FUZZ_SEED widen the sweep."""
import os

import numpy as np
import pytest

import gradcheck
import workloads as W
from tamcmc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("balanced", [0, 1], ids=["equal-length-tiles", "equal-cost-tiles"])
def test_random_configurations_match_the_oracle(accel_mod, orc, monkeypatch, balanced):
    monkeypatch.setenv("TAMCMC_EQUAL_COST", str(balanced))     # 1: per-chain tile boundaries from the setup kernel's balancer
    ncases = int(os.environ.get("TAMCMC_FUZZ_CASES", "60"))
    rng = np.random.default_rng(int(os.environ.get("TAMCMC_FUZZ_SEED", "1")) + 1000 * balanced)
    worst_L = worst_M = worst_g = worst_c = worst_r = 0.0
    an_done = 0
    done, failures = 0, []
    fd_done, fd_max = 0, max(8, ncases // 10)
    for case in range(ncases):
        mid = int(rng.choice(W.ALL_IDS))
        Nx = int(rng.choice([rng.integers(2, 600), rng.integers(600, 2100), rng.integers(2100, 12000), rng.integers(12000, 60000)]))
        kw = dict(Nx=Nx)
        if mid not in (0, 1):
            kw.update(trunc_c=float(rng.choice([3.0, 7.0, 20.0, 50.0, 10000.0])), asym=float(rng.choice([0.0, 0.0, 25.0, -40.0, 5.0])),
                      do_amp=bool(rng.integers(0, 2)))
        w = W.any_model(mid, **kw)
        m, st0 = orc.model(mid, w["params_true"], w["plength"], w["x"])
        if st0 != 0 or not np.all(np.isfinite(m)) or np.any(m <= 0):
            continue                                  # not a usable truth spectrum (e.g. 2 bins and an empty window)
        y = synth.make_spectrum(m, seed=int(rng.integers(1, 1 << 30)))
        n = int(rng.integers(1, 9))
        P = W.perturbed(w, n, scale=float(rng.choice([0.0005, 0.003, 0.01])), seed=int(rng.integers(1, 1 << 30)))
        T = synth.temperatures(n) if n > 1 else np.ones(1)
        like = int(rng.integers(0, 2)) if mid in (0, 1, 2, 11) else 0
        sig = 0.05 + 0.2 * np.abs(np.sin(np.arange(y.size))) if like == 1 else None
        row = int(rng.integers(0, n))
        with accel_mod.Accel(mid, w["plength"], w["x"], y, sigma_y=sig, likelihood_case=like) as acc:
            acc.set_vars(w["index_to_relax"])
            L, st, models = acc.eval_batch(P, T, model_rows=[row])
            Lg, stg, g = acc.eval_batch(P, T, grad=True)
        g_ref = None
        if balanced:
            # the same batch with tiles of equal length: other tile boundaries, so other partial sums -- logL and every
            # gradient entry must agree to rounding (the finite-difference check below only sees the few smooth cases)
            monkeypatch.setenv("TAMCMC_EQUAL_COST", "0")
            with accel_mod.Accel(mid, w["plength"], w["x"], y, sigma_y=sig, likelihood_case=like) as acc:
                acc.set_vars(w["index_to_relax"])
                _, _, g_ref = acc.eval_batch(P, T, grad=True)
            monkeypatch.setenv("TAMCMC_EQUAL_COST", "1")
        rL, rst, rm = orc.generate_batch(mid, w["plength"], w["x"], y, P, T, sigma_y=sig, likelihood_case=like, want_models=True)
        ok = np.array_equal(st, rst) and np.array_equal(stg, rst)
        good = (rst == 0) & np.isfinite(rL)
        if np.any(good):
            eL = float(np.max(np.abs(L[good] - rL[good]) / np.abs(rL[good])))
            eG = float(np.max(np.abs(Lg[good] - rL[good]) / np.abs(rL[good])))
            worst_L = max(worst_L, eL, eG)
            ok = ok and eL <= 1e-10 and eG <= 1e-10 and bool(np.all(np.isfinite(g[good])))
            if g_ref is not None:
                scale = np.max(np.abs(g_ref[good]), axis=1, keepdims=True)
                ok = ok and bool(np.all(np.abs(g[good] - g_ref[good]) <= 1e-10 * np.maximum(scale, 1e-300)))
            if good[row]:
                eM = float(np.max(np.abs(models[0] - rm[row]) / np.abs(rm[row])))
                worst_M = max(worst_M, eM)
                ok = ok and eM <= 1e-12
        ok = ok and bool(np.all(np.isnan(L[~good]) == np.isnan(rL[~good])))
        # every gradient entry of every healthy chain against the oracle's analytic gradient (window on or off)
        if ok and np.any(good):
            ga_ref, ga_abs, _, ga_st = orc.grad_analytic(mid, w["plength"], w["x"], y, P[good], T[good], w["index_to_relax"],
                                                         sigma_y=sig, likelihood_case=like)
            try:
                assert np.all(ga_st == 0)
                ec, er = gradcheck.assert_grad_entrywise(g[good], ga_ref, ga_abs, tag=f"case {case}")
                worst_c, worst_r = max(worst_c, ec), max(worst_r, er)
                an_done += int(good.sum())
            except AssertionError as e:
                ok = False
                failures.append(str(e))
        # gradient against Richardson-extrapolated central differences of the oracle, where logL is smooth (no
        # truncation window) and the grid is short enough for ~100 oracle evaluations
        if ok and good[0] and kw.get("trunc_c") == 10000.0 and Nx <= 3000 and fd_done < fd_max:
            gfd, st2 = orc.grad_fd(mid, w["plength"], w["x"], y, P[0], T[0], w["index_to_relax"], rel_step=1e-6, sigma_y=sig,
                                   likelihood_case=like)
            if st2 == 0 and np.all(np.isfinite(gfd)):
                eg = float(np.max(np.abs(g[0] - gfd)) / np.max(np.abs(gfd)))
                worst_g = max(worst_g, eg)
                ok = ok and eg <= 2e-5
                fd_done += 1
        done += 1
        if not ok:
            failures.append((case, mid, kw, n, like, st.tolist(), rst.tolist()))
    print(f"fuzz: {done} cases, worst relative logL error {worst_L:.2e}, model {worst_M:.2e}; {fd_done} gradients against "
          f"finite differences, worst {worst_g:.2e} of the largest entry; {an_done} gradient rows against the analytic oracle, "
          f"worst entry {worst_c:.2e} of its sum|terms|, {worst_r:.2e} relative")
    assert not failures, failures
    assert done >= ncases // 2
