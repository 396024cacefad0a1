"""Per-entry comparison of a gradient with the oracle's analytic gradient (oracle/tamcmc_oracle.c: orc_grad_analytic).
Test helper only.

Tolerance for entry k:      |g_k - ref_k|  <=  1e-10 * |ref_k|  +  3e-14 * S_k,      S_k = sum over bins of |terms of ref_k|.

The first term is the bar SURVEY.md section 8(c) sets ("CPU-analytic <-> HIP-analytic <= 1e-10 rel").  The second is the
cancellation floor, and it is per entry, not per row: every gradient entry is a sum over ~1e5 bins of terms of both
signs (w_i = y_i/M_i^2 - 1/M_i changes sign with the noise), S_k is the sum of their magnitudes as the oracle
accumulates it beside the value, and NO fp64 evaluation of such a sum can be trusted below a few eps * S_k (one rounding
of M_i alone moves term i by eps * |term_i| * O(1)).  3e-14 = 135 eps covers the ~sqrt(N) growth of the rounding of the
three parties (oracle model in fp64, HIP model in fp64, HIP sums in fp64) and the few-ulp differences of the device's
exp/log/atan.  Observed: <= 6e-15 * S_k in the suite's fixed cases; entries whose value is > 1e-4 of S_k agree to <= 1e-12
relative; the worst relative figure there is 5.7e-10 on a C2 entry (an l=0 height at a hot chain) whose value is 3e-7 of
its S_k.  A widened random sweep (tests/test_fuzz_gpu.py, 1500 cases, 6700 gradient rows: asymmetric, amplitude-form and
trunc_c = 3 models among them, whose bin terms cancel internally as well) reached 1.7e-13 * S_k on an entry that agreed to
7e-11 relative, i.e. 60 % of its tolerance.
A wrong factor, sign, index or window edge in any term moves an entry by >= 1e-3 of S_k."""
import numpy as np

GRAD_RTOL = 1e-10
GRAD_COND = 3e-14


def assert_grad_entrywise(g, ref, ref_abs, tag=""):
    g, ref, ref_abs = np.atleast_2d(g), np.atleast_2d(ref), np.atleast_2d(ref_abs)
    assert g.shape == ref.shape == ref_abs.shape, (g.shape, ref.shape)
    assert np.all(np.isfinite(ref)) and np.all(np.isfinite(g)), tag
    err = np.abs(g - ref)
    tol = GRAD_RTOL * np.abs(ref) + GRAD_COND * ref_abs
    bad = err > tol
    if np.any(bad):
        i = np.unravel_index(np.argmax(err / np.maximum(tol, 1e-300)), err.shape)
        raise AssertionError(f"{tag}: {int(bad.sum())} gradient entries off; worst chain {i[0]} var {i[1]}: "
                             f"hip {g[i]!r} oracle {ref[i]!r} sum|terms| {ref_abs[i]!r}")
    nz = ref_abs > 0
    return (float(np.max(err[nz] / ref_abs[nz])) if np.any(nz) else 0.0,
            float(np.max(err[nz] / np.maximum(np.abs(ref[nz]), 1e-300))) if np.any(nz) else 0.0)


def check_against_oracle(acc, orc, mid, w, y, P, T, sigma=None, like=0, likelihood_p=1.0, tag="", g=None):
    """acc: an open Accel with set_vars(w['index_to_relax']) done (or g given).  Returns the HIP gradient."""
    idx = w["index_to_relax"]
    if g is None:
        _, st, g = acc.eval_batch(P, T, grad=True)
    ref, ref_abs, rL, rst = orc.grad_analytic(mid, w["plength"], w["x"], y, P, T, idx, sigma_y=sigma,
                                              likelihood_case=like, likelihood_p=likelihood_p)
    ok = rst == 0
    assert np.any(ok), tag
    assert_grad_entrywise(np.asarray(g)[ok], ref[ok], ref_abs[ok], tag)
    return g
