"""HIP vs oracle at parameter values where a formula changes branch or degenerates: zero / negative splittings and
asymmetry, inclination 0 / 45 / 90 / beyond, a vanishing visibility or height, a mode exactly on a bin, two modes on one
frequency, equal / negative / tiny widths, a switched-off Harvey profile.  Status, logL (both launch paths) and every
gradient entry are compared as everywhere else (tests/gradcheck.py), with two documented exceptions:

* d/d(inclination) AT 90 degrees is analytically zero (the ratios are even about 90) and both sides return rounding noise
  of the order 1e-16 of the row; within 1e-4 degrees of 90 the factor cos(i) carries the rounding of i * pi / 180, so
  the entry agrees to ~2e-9 relative only.  Asserted there: the difference is below 1e-12 of the row's largest entry.
  (The same holds where the inclination is atan(p4 / p3) with p3 = 0.)
* a width that overflows or underflows (AppWidth with an absurd exponent: exp(+-3000)) gives a NaN model here (status 1,
  the move is rejected) where the reference's formula happens to stay finite -- and loses the move by 1e4 in logL.
  Asserted there: the oracle's logL is far below the base chain's.
"""
import numpy as np
import pytest

import gradcheck
import workloads as W
from tamcmc_amd import synth

pytestmark = pytest.mark.gpu


def _cases(mid, w):
    pl = np.asarray(w["plength"])
    off = np.concatenate([[0], np.cumsum(pl)])
    base = np.asarray(w["params_true"], dtype=float)
    out = [("base", base, None)]

    def setp(name, idx, val, kind=None):
        p = base.copy()
        p[idx] = val
        out.append((name, p, kind))
    s6 = off[6]
    atan_pair = mid in (2, 9, 10, 11, 12, 13, 14)          # params[s+3], [s+4] = sqrt(a1) cos i, sqrt(a1) sin i
    for j in range(min(int(pl[6]), 6)):
        setp(f"split[{j}]=0", s6 + j, 0.0, "inc90" if (atan_pair and j == 3) else None)
        setp(f"split[{j}]<0", s6 + j, -abs(base[s6 + j]) - 1e-3)
    if pl[9] >= 1:
        for v in (0.0, 90.0, 45.0, 1e-9, 89.999999, 150.0, -20.0):
            setp(f"inc={v}", off[9], v, "inc90" if abs(v - 90.0) < 1e-4 else None)
    if pl[1] >= 1:
        setp("V1=0", off[1], 0.0)
    if pl[0] >= 1:
        setp("H0=0", off[0], 0.0)
        setp("H0<0", off[0], -base[off[0]])
    if pl[2] >= 2:
        setp("f on a bin", off[2], w["x"][np.searchsorted(w["x"], base[off[2]])])
        setp("f0==f1", off[2] + 1, base[off[2]])
    wd = off[7]
    if pl[7] >= 2:
        setp("W[1]=W[0]", wd + 1, base[wd], "overflow" if mid == 9 else None)      # id 9: exponent := nu_dip (~2600)
        setp("W[0]<0", wd, -base[wd])
        setp("W[0] tiny", wd, 1e-6)
    if pl[8] >= 3:
        setp("Harvey H=0", off[8], 0.0)
        setp("Harvey tau=0", off[8] + 1, 0.0)
    return out


@pytest.mark.parametrize("mid", [2, 3, 6, 7, 9, 11, 12])
def test_special_parameter_values(accel_mod, orc, mid):
    n_cases = 0
    for kw in (dict(asym=0.0), dict(asym=25.0), dict(asym=0.0, do_amp=True)):
        w = W.any_model(mid, Nx=9000, **kw)
        m, st0 = orc.model(mid, w["params_true"], w["plength"], w["x"])
        assert st0 == 0
        y = synth.make_spectrum(m, seed=11)
        cases = _cases(mid, w)
        P = np.stack([c[1] for c in cases])
        T = np.linspace(1.0, 3.0, len(cases))
        with accel_mod.Accel(mid, w["plength"], w["x"], y) as acc:
            acc.set_vars(w["index_to_relax"])
            L, st, g = acc.eval_batch(P, T, grad=True)
            L2, st2 = acc.eval_batch(P, T)
        rL, rst = orc.generate_batch(mid, w["plength"], w["x"], y, P, T)
        ref, ref_abs, _, gst = orc.grad_analytic(mid, w["plength"], w["x"], y, P, T, w["index_to_relax"])
        for k, (name, _, kind) in enumerate(cases):
            tag = f"id {mid} {kw} '{name}'"
            n_cases += 1
            if kind == "overflow":
                assert st[k] == 1 and st2[k] == 1 and np.isnan(L[k]), tag
                assert rst[k] == 0 and rL[k] * T[k] < rL[0] * T[0] - 1e3, tag        # the reference loses this move anyway
                continue
            assert st[k] == rst[k] and st2[k] == rst[k], (tag, st[k], st2[k], rst[k])
            if rst[k] != 0 or not np.isfinite(rL[k]):
                continue
            assert abs(L[k] - rL[k]) <= 1e-10 * abs(rL[k]), (tag, L[k], rL[k])
            assert abs(L[k] - L2[k]) <= 1e-12 * abs(L2[k]), tag       # (the two launches may cut the grid into different tiles)
            if gst[k] != 0:
                continue
            if kind == "inc90":
                err = np.abs(g[k] - ref[k])
                tol = gradcheck.GRAD_RTOL * np.abs(ref[k]) + gradcheck.GRAD_COND * ref_abs[k]
                floor = 1e-12 * np.max(np.abs(ref[k]))
                assert np.all((err <= tol) | (err <= floor)), (tag, float(np.max(err)), floor)
            else:
                gradcheck.assert_grad_entrywise(g[k], ref[k], ref_abs[k], tag)
    assert n_cases > 50
