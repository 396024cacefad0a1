"""BASELINE.json configs C3, C4 and C5 at their stated shape on ONE MI355X (the multi-GPU configs as one-GPU
stand-ins: the chains / stars a node would spread over 8 GPUs are evaluated here in one or several batches, and the
property that makes the spreading legal -- a chain's result does not depend on the batch it is evaluated in -- is
checked bit for bit).  HIP path against the CPU oracle; tolerances as in test_parity_gpu.py."""
import numpy as np
import pytest

import gradcheck
from tamcmc_amd import synth

pytestmark = pytest.mark.gpu

RTOL_LOGL = 1e-10     # north-star: logL within 1e-10 relative
GRAD_TOL = 2e-5       # of the largest entry of the finite-difference gradient (tests/test_grad_gpu.py)


def _spectrum(orc, w, seed=None):
    m, st = orc.model(w["model_case"], w["params_true"], w["plength"], w["x"])
    assert st == 0
    return synth.make_spectrum(m) if seed is None else synth.make_spectrum(m, seed=seed)


def test_c4_gradient_at_full_size(accel_mod, orc):
    """C4: model_MS_Global with 106 parameters (Nmax = 14, l = 0..3, 56 multiplets, 94 variables) on 1e5 bins -- the
    gradient-width stress.  (a) trunc_c = 20 as configured: logL of the likelihood and of the gradient path vs the oracle
    for 16 chains, gradient finite and reproducible; (b) trunc_c = 10000 (no window, smooth logL): ten gradient columns
    spread over every parameter block against Richardson finite differences of the oracle."""
    w = synth.workload_c4()
    assert w["x"].size == 100000 and int(w["plength"].sum()) == 106
    y = _spectrum(orc, w)
    P = synth.chain_params(w, 16)
    T = synth.temperatures(16)
    idx = w["index_to_relax"]
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        acc.set_vars(idx)
        L0, st0 = acc.eval_batch(P, T)
        L, st, g = acc.eval_batch(P, T, grad=True)
        L2, _, g2 = acc.eval_batch(P, T, grad=True)
    rL, rst = orc.generate_batch(2, w["plength"], w["x"], y, P, T)
    assert np.array_equal(st, rst) and np.array_equal(st0, rst) and np.all(st == 0)
    assert np.max(np.abs(L - rL) / np.abs(rL)) <= RTOL_LOGL
    assert np.max(np.abs(L0 - rL) / np.abs(rL)) <= RTOL_LOGL
    assert np.all(np.isfinite(g)) and np.array_equal(g, g2) and np.array_equal(L, L2)
    # (a') every one of the 16 x 94 entries, window on, against the oracle's analytic gradient (tests/gradcheck.py)
    assert idx.size == 94
    gradcheck.check_against_oracle(None, orc, 2, w, y, P, T, tag="C4 16 x 1e5, trunc_c 20", g=g)

    w = synth.workload_c4(trunc_c=10000.0)
    y = _spectrum(orc, w)
    idx = w["index_to_relax"]
    P = synth.chain_params(w, 16)
    T = synth.temperatures(16)
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        acc.set_vars(idx)
        L, st, g = acc.eval_batch(P, T, grad=True)
    assert np.all(st == 0)
    # ten columns spread over the variables: heights, visibilities, l = 0..3 frequencies, splitting, widths, noise, inclination
    cols = np.unique(np.linspace(0, idx.size - 1, 10).round().astype(int))
    assert cols.size >= 10
    for k in (0, 9):                      # coldest chain of the batch and a warm one
        gfd, st2 = orc.grad_fd(2, w["plength"], w["x"], y, P[k], T[k], idx[cols])
        assert st2 == 0
        scale = np.max(np.abs(g[k]))
        err = np.abs(g[k][cols] - gfd) / scale
        assert np.max(err) <= GRAD_TOL, (k, cols[np.argmax(err)], g[k][cols], gfd)


def test_c3_256_chains_one_batch_and_as_eight_blocks(accel_mod, orc):
    """C3: 256 tempered chains x 1e5 bins.  On the node they are 8 contiguous temperature blocks of 32, one per GPU; here
    the whole ladder is evaluated in one batch against the oracle, and again as the 8 blocks a rank each would see:
    bitwise equal (so the sharded run is the single-GPU run), likelihood and gradient."""
    w = synth.workload_c2()
    y = _spectrum(orc, w)
    P = synth.chain_params(w, 256)
    T = synth.temperatures(256)
    with accel_mod.Accel(2, w["plength"], w["x"], y) as acc:
        acc.set_vars(w["index_to_relax"])
        L, st = acc.eval_batch(P, T)
        Lg, stg, g = acc.eval_batch(P, T, grad=True)
        parts = [acc.eval_batch(P[b * 32:(b + 1) * 32], T[b * 32:(b + 1) * 32]) for b in range(8)]
        gparts = [acc.eval_batch(P[b * 32:(b + 1) * 32], T[b * 32:(b + 1) * 32], grad=True) for b in range(8)]
    rL, rst = orc.generate_batch(2, w["plength"], w["x"], y, P, T)
    assert np.array_equal(st, rst) and np.array_equal(stg, rst)
    assert np.max(np.abs(L - rL) / np.abs(rL)) <= RTOL_LOGL
    assert np.max(np.abs(Lg - rL) / np.abs(rL)) <= RTOL_LOGL
    assert np.array_equal(np.concatenate([p[0] for p in parts]), L)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), st)
    assert np.array_equal(np.concatenate([p[0] for p in gparts]), Lg)
    assert np.array_equal(np.concatenate([p[2] for p in gparts]), g)
    gradcheck.check_against_oracle(None, orc, 2, w, y, P, T, tag="C3 256 x 1e5", g=g)      # all 256 x 44 entries


def test_c5_ensemble_32_stars_16_chains(accel_mod, orc):
    """C5: 32 independent synthetic stars x 16 chains on 1e5 bins.  (a) as one context holding the 32 spectra
    (tamcmc_ctx_set_spectra: 512 chains in one batch) vs the oracle star by star; (b) four of the stars as contexts of
    their own (what a GPU of the node holds: 4 stars on 4 streams) -- bit for bit what the shared batch returned."""
    nstars, nch = 32, 16
    w = synth.workload_c2()
    m, st = orc.model(2, w["params_true"], w["plength"], w["x"])
    assert st == 0
    Y = np.stack([synth.make_spectrum(m, seed=1000 + k) for k in range(nstars)])
    P = synth.chain_params(w, nstars * nch, seed=11)
    T = np.tile(synth.temperatures(nch), nstars)
    spec = np.repeat(np.arange(nstars, dtype=np.int32), nch)
    with accel_mod.Accel(2, w["plength"], w["x"], Y[0]) as acc:
        acc.set_vars(w["index_to_relax"])
        acc.set_spectra(Y)
        acc.set_chain_spectrum(spec)
        L, stt = acc.eval_batch(P, T)
        Lg, stg, g = acc.eval_batch(P, T, grad=True)
        with pytest.raises(accel_mod.AccelError):         # a batch the map does not cover is refused, not fitted to star 0
            acc.set_chain_spectrum(spec[:100])
            acc.eval_batch(P, T)
    for k in range(nstars):
        sel = slice(k * nch, (k + 1) * nch)
        rL, rst = orc.generate_batch(2, w["plength"], w["x"], Y[k], P[sel], T[sel])
        assert np.array_equal(stt[sel], rst) and np.array_equal(stg[sel], rst)
        assert np.max(np.abs(L[sel] - rL) / np.abs(rL)) <= RTOL_LOGL
        assert np.max(np.abs(Lg[sel] - rL) / np.abs(rL)) <= RTOL_LOGL
    assert np.all(np.isfinite(g))
    for k in (0, 13, 31):               # gradient of three of the stars, entry by entry, against the oracle
        sel = slice(k * nch, (k + 1) * nch)
        gradcheck.check_against_oracle(None, orc, 2, w, Y[k], P[sel], T[sel], tag=f"C5 star {k}", g=g[sel])
    for k in (0, 7, 19, 31):
        sel = slice(k * nch, (k + 1) * nch)
        with accel_mod.Accel(2, w["plength"], w["x"], Y[k]) as one:
            one.set_vars(w["index_to_relax"])
            L1, st1 = one.eval_batch(P[sel], T[sel])
            Lg1, _, g1 = one.eval_batch(P[sel], T[sel], grad=True)
        assert np.array_equal(L[sel], L1) and np.array_equal(stt[sel], st1)
        assert np.array_equal(Lg[sel], Lg1) and np.array_equal(g[sel], g1)
