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
