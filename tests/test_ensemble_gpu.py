"""BASELINE config C5 (ensemble: independent stars, a few chains each, one context per star on its own stream):
contexts driven concurrently from several host threads must give exactly what they give one after the other
(include/tamcmc_accel.h: "distinct ctx objects may be driven from different host threads concurrently")."""
import threading

import numpy as np
import pytest

from tamcmc_amd import synth

pytestmark = pytest.mark.gpu


def test_concurrent_contexts_equal_sequential(accel_mod, orc):
    nstars, nch, nrep = 6, 16, 8
    stars = []
    for k in range(nstars):
        w = synth.workload_c2(model_case=2 if k % 2 == 0 else 3, Nx=20000 + 1000 * k)
        m, st = orc.model(w["model_case"], w["params_true"], w["plength"], w["x"])
        assert st == 0
        y = synth.make_spectrum(m, seed=100 + k)
        P = synth.chain_params(w, nch, seed=0x9E3779B97F4A7C15 + k)
        stars.append((w, y, P, synth.temperatures(nch)))
    ctxs = [accel_mod.Accel(w["model_case"], w["plength"], w["x"], y) for (w, y, P, T) in stars]
    try:
        for c, (w, y, P, T) in zip(ctxs, stars):
            c.set_vars(w["index_to_relax"])
        seq = [c.eval_batch(P, T, grad=True) for c, (w, y, P, T) in zip(ctxs, stars)]
        seq_l = [c.eval_batch(P, T) for c, (w, y, P, T) in zip(ctxs, stars)]      # the likelihood-only kernel tiles differently
        for (logL, st, g), (w, y, P, T) in zip(seq, stars):
            ref, rst = orc.generate_batch(w["model_case"], w["plength"], w["x"], y, P, T)
            assert np.array_equal(st, rst) and np.allclose(logL, ref, rtol=1e-10, atol=0)
        out = [[None] * nrep for _ in range(nstars)]
        errs = []

        def work(k):
            try:
                w, y, P, T = stars[k]
                for r in range(nrep):
                    out[k][r] = ctxs[k].eval_batch(P, T, grad=(r % 2 == 0))
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(k,)) for k in range(nstars)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for k in range(nstars):
            for r in range(nrep):
                res = out[k][r]
                want = seq[k] if r % 2 == 0 else seq_l[k]
                assert np.array_equal(res[0], want[0]) and np.array_equal(res[1], want[1])
                if r % 2 == 0:
                    assert np.array_equal(res[2], want[2])
    finally:
        for c in ctxs:
            c.close()


def test_several_spectra_in_one_batch(accel_mod, orc):
    """Ensemble on ONE context: several spectra on the same grid, each chain fitted to its own (tamcmc_ctx_set_spectra /
    tamcmc_ctx_set_chain_spectrum).  Bit for bit what one context per spectrum returns, and within tolerance of the oracle."""
    nspec, nch = 4, 5
    w = synth.workload_c2(Nx=30000)
    m, st = orc.model(2, w["params_true"], w["plength"], w["x"])
    assert st == 0
    Y = np.stack([synth.make_spectrum(m, seed=500 + k) for k in range(nspec)])
    P = synth.chain_params(w, nspec * nch, seed=77)
    T = np.tile(synth.temperatures(nch), nspec)
    spec = np.repeat(np.arange(nspec, dtype=np.int32), nch)
    perm = np.random.default_rng(5).permutation(nspec * nch)         # chains of one spectrum need not be adjacent
    P, T, spec = P[perm], T[perm], spec[perm]
    with accel_mod.Accel(2, w["plength"], w["x"], Y[0]) as acc:
        acc.set_vars(w["index_to_relax"])
        acc.set_spectra(Y)
        acc.set_chain_spectrum(spec)
        L, st, models = acc.eval_batch(P, T, model_rows=[0])
        Lg, stg, g = acc.eval_batch(P, T, grad=True)
        with pytest.raises(accel_mod.AccelError):
            acc.set_chain_spectrum(np.full(3, nspec, dtype=np.int32))     # out of range
    for k in range(nspec):
        sel = np.flatnonzero(spec == k)
        with accel_mod.Accel(2, w["plength"], w["x"], Y[k]) as one:
            one.set_vars(w["index_to_relax"])
            L1, st1 = one.eval_batch(P[sel], T[sel])
            Lg1, _, g1 = one.eval_batch(P[sel], T[sel], grad=True)
        assert np.array_equal(L[sel], L1) and np.array_equal(st[sel], st1)
        assert np.array_equal(Lg[sel], Lg1) and np.array_equal(g[sel], g1)
        rL, rst = orc.generate_batch(2, w["plength"], w["x"], Y[k], P[sel], T[sel])
        assert np.array_equal(st[sel], rst)
        assert np.max(np.abs(L[sel] - rL) / np.abs(rL)) <= 1e-10
    # chi-square likelihood with per-spectrum sigma, on the fused one-tile path as well (Nx = 900)
    w = synth.workload_c1(Nx=900)
    m, st = orc.model(w["model_case"], w["params_true"], w["plength"], w["x"])
    Y = np.stack([synth.make_spectrum(m, seed=900 + k) for k in range(3)])
    S = np.stack([0.05 + 0.1 * (k + 1) * np.abs(np.sin(np.arange(900))) for k in range(3)])
    P = synth.chain_params(w, 6, seed=3)
    T = synth.temperatures(6)
    spec = np.array([2, 0, 1, 1, 0, 2], dtype=np.int32)
    with accel_mod.Accel(w["model_case"], w["plength"], w["x"], Y[0], sigma_y=S[0], likelihood_case=1) as acc:
        acc.set_spectra(Y, S)
        acc.set_chain_spectrum(spec)
        L, st = acc.eval_batch(P, T)
    for k in range(3):
        sel = np.flatnonzero(spec == k)
        rL, rst = orc.generate_batch(w["model_case"], w["plength"], w["x"], Y[k], P[sel], T[sel], sigma_y=S[k], likelihood_case=1)
        assert np.array_equal(st[sel], rst)
        assert np.max(np.abs(L[sel] - rL) / np.abs(rL)) <= 1e-10
