"""CPU tests of the oracle (oracle/tamcmc_oracle.c) itself.

The reference holds no golden vectors for its model/likelihood functions and
cannot be built here, so the oracle is checked against (a) the only reference-produced numbers
available -- amplitude_ratio(2, 55 deg) recorded in SURVEY.md App. D -- (b) closed forms,
(c) an independent numpy restatement of the Lorentzian sum written from the formulas of
SURVEY.md App. A.3/A.4, and (d) properties of the truncation window.
"""
import math

import numpy as np
import pytest

import workloads as W
from tamcmc_amd import synth


# ---------------------------------------------------------------- function_rot.cpp
def closed_form_ratios(l, beta_deg):
    b = math.radians(beta_deg)
    c, s = math.cos(b), math.sin(b)
    if l == 1:
        return np.array([0.5 * s * s, c * c, 0.5 * s * s])
    if l == 2:
        v0 = 0.25 * (3 * c * c - 1) ** 2
        v1 = 0.375 * math.sin(2 * b) ** 2
        v2 = 0.375 * s ** 4
        return np.array([v2, v1, v0, v1, v2])
    v0 = (5 * math.cos(3 * b) + 3 * c) ** 2 / 64.0
    v1 = 3 * (5 * math.cos(2 * b) + 3) ** 2 * s * s / 64.0
    v2 = 15.0 / 8.0 * c * c * s ** 4
    v3 = 5.0 / 16.0 * s ** 6
    return np.array([v3, v2, v1, v0, v1, v2, v3])


def test_amplitude_ratio_reference_values(orc):
    # values printed by the reference's own amplitude_ratio(2, 55 deg), index m+l
    r = orc.amplitude_ratio(2, 55.0)
    assert r[0] == pytest.approx(0.168845443602361, abs=5e-15)
    assert r[1] == pytest.approx(0.331133333084808, abs=5e-15)
    assert r[2] == pytest.approx(0.000042446625662, abs=5e-15)


@pytest.mark.parametrize("l", [1, 2, 3])
@pytest.mark.parametrize("beta", [0.0, 1e-3, 17.0, 45.0, 55.0, 89.999, 90.0, 120.0])
def test_amplitude_ratio_closed_form(orc, l, beta):
    r = orc.amplitude_ratio(l, beta)
    assert np.allclose(r, closed_form_ratios(l, beta), rtol=0, atol=2e-15)
    assert r.sum() == pytest.approx(1.0, abs=1e-14)
    assert np.array_equal(r, r[::-1])


# ---------------------------------------------------------------- interpol.cpp
def test_lin_interpol(orc):
    x = np.array([1.0, 2.0, 4.0, 7.0])
    y = np.array([10.0, 20.0, 0.0, 3.0])
    assert orc.lin_interpol(x, y, 1.5) == pytest.approx(15.0)
    assert orc.lin_interpol(x, y, 3.0) == pytest.approx(10.0)
    assert orc.lin_interpol(x, y, 2.0) == pytest.approx(20.0)       # node: first bracket that contains it
    assert orc.lin_interpol(x, y, 1.0) == pytest.approx(10.0)
    assert orc.lin_interpol(x, y, 7.0) == pytest.approx(3.0)
    assert orc.lin_interpol(x, y, 0.0) == pytest.approx(0.0)        # extrapolation, slope of first segment
    assert orc.lin_interpol(x, y, 10.0) == pytest.approx(6.0)       # extrapolation, slope of last segment


# ---------------------------------------------------------------- truncation window
def test_window_branches(orc):
    x = synth.grid(100000, 2300.0, 0.0084)
    c = 20.0
    # Gamma >= 1, f_s >= 1, l = 2  ->  half width c (l f_s + Gamma)
    st, a, b = orc.truncation_window(x, 2700.0, 1.4, 1.5, 2, c)
    hw = c * (2 * 1.4 + 1.5)
    assert st == 0 and a == math.floor((2700.0 - hw - 2300.0) / 0.0084) and b == math.ceil((2700.0 + hw - 2300.0) / 0.0084)
    # Gamma <= 1, f_s <= 1, l = 0  ->  c * 2.2
    st, a, b = orc.truncation_window(x, 2700.0, 0.4, 0.5, 0, c)
    assert st == 0 and b - a == pytest.approx(2 * 44.0 / 0.0084, abs=2)
    # Gamma >= 1, f_s <= 1, l = 1  ->  c (l + Gamma)
    st, a, b = orc.truncation_window(x, 2700.0, 0.4, 2.0, 1, c)
    assert st == 0 and b - a == pytest.approx(2 * c * 3.0 / 0.0084, abs=2)
    # clamps to the grid
    st, a, b = orc.truncation_window(x, 2301.0, 1.4, 1.5, 2, c)
    assert st == 0 and a == 0
    st, a, b = orc.truncation_window(x, 3139.0, 1.4, 1.5, 2, c)
    assert st == 0 and b == x.size
    # mode far below the grid: pmax is reset to x0 + c (build_lorentzian.cpp:415-417)
    st, a, b = orc.truncation_window(x, 1000.0, 1.4, 1.5, 0, c)
    assert st == 0 and a == 0 and b == math.ceil(c / 0.0084)
    # mode far above: pmin reset to x_last - c
    st, a, b = orc.truncation_window(x, 9000.0, 1.4, 1.5, 0, c)
    assert st == 0 and b == x.size and a == math.floor((x[-1] - c - 2300.0) / 0.0084)
    # no truncation
    st, a, b = orc.truncation_window(x, 2700.0, 1.4, 1.5, 2, 10000.0)
    assert (st, a, b) == (0, 0, x.size)
    # NaN width: no branch fires -> the reference would exit
    st, a, b = orc.truncation_window(x, 2700.0, 1.4, float("nan"), 2, c)
    assert st == 2


# ---------------------------------------------------------------- independent numpy restatement
def numpy_global_a1etaa3(w, x):
    """Model id 2 straight from SURVEY.md App. A (formulas, not the oracle's code)."""
    p, pl = w["params_true"], w["plength"]
    b = W.split(w)
    Nmax, lmax, s, wq, z, q = b["Nmax"], b["lmax"], b["s"], b["w"], b["z"], b["q"]
    trunc_c, do_amp = p[q + 1], p[q + 2] != 0
    a1 = p[s + 3] ** 2 + p[s + 4] ** 2
    inc = math.degrees(math.atan(p[s + 4] / p[s + 3]))
    eta, a3, asym = p[s + 1], p[s + 2], p[s + 5]
    fl0 = p[Nmax + lmax:Nmax + lmax + Nmax]
    Wl0 = p[wq:wq + Nmax]
    M = np.zeros_like(x)
    step = x[1] - x[0]
    for n in range(Nmax):
        for l in range(lmax + 1):
            f = p[Nmax + lmax + l * Nmax + n]
            if l == 0:
                G = abs(Wl0[n])
            else:
                G = abs(np.interp(f, fl0, Wl0)) if fl0[0] <= f <= fl0[-1] else None
                if G is None:   # linear extrapolation with the edge segment
                    i = 0 if f < fl0[0] else Nmax - 2
                    sl = (Wl0[i + 1] - Wl0[i]) / (fl0[i + 1] - fl0[i])
                    G = abs(Wl0[i] + sl * (f - fl0[i]))
            V = 1.0 if l == 0 else abs(p[Nmax + l - 1])
            H = abs(p[n] / (math.pi * G)) * V if do_amp else abs(p[n] * V)
            ratios = np.ones(1) if l == 0 else closed_form_ratios(l, inc)
            if G >= 1 and a1 >= 1:
                hw = trunc_c * (l * a1 + G) if l else trunc_c * G * 2.2
            elif G <= 1 and a1 >= 1:
                hw = trunc_c * (l * a1 + 1) if l else trunc_c * 2.2
            elif G >= 1:
                hw = trunc_c * (l + G) if l else trunc_c * 2.2 * G
            else:
                hw = trunc_c * (l + 1) if l else trunc_c * 2.2
            imin = max(0, math.floor((f - hw - x[0]) / step))
            imax = min(x.size, math.ceil((f + hw - x[0]) / step))
            xs = x[imin:imax]
            A = 1.0 if asym == 0 else (1 + asym * (xs / f - 1)) ** 2 + (0.5 * G * asym / f) ** 2
            for m in range(-l, l + 1):
                if l == 0:
                    nu = f
                else:
                    Q = (l * (l + 1) - 3 * m * m) / ((2 * l - 1) * (2 * l + 3))
                    clm = m if l == 1 else ((5 * m ** 3 - 17 * m) / 3.0 if l == 2 else 0.0)
                    nu = f * (1 + eta * Q) + m * a1 + clm * a3
                M[imin:imax] += H * ratios[m + l] * A / (1 + 4 * (xs - nu) ** 2 / G ** 2)
    noise = np.abs(p[z:z + 10])
    for k in range(3):
        if noise[3 * k + 1] != 0:
            M += noise[3 * k] / (1 + (1e-3 * noise[3 * k + 1] * x) ** noise[3 * k + 2])
    return M + noise[9]


@pytest.mark.parametrize("kw", [dict(), dict(trunc_c=10000.0), dict(asym=30.0), dict(do_amp=True), dict(asym=-12.0, trunc_c=5.0)])
def test_model_id2_against_numpy_restatement(orc, kw):
    w = W.make(2, Nx=6000, **kw)
    m, st = orc.model(2, w["params_true"], w["plength"], w["x"])
    assert st == 0
    ref = numpy_global_a1etaa3(w, w["x"])
    assert np.max(np.abs(m - ref) / ref) < 1e-13


def test_ids_2_and_3_agree_on_equivalent_parameters(orc):
    # id 3 takes a1 and the inclination directly; id 2 derives them from sqrt(a1) cos i / sin i
    w2 = W.make(2, Nx=5000)
    w3 = W.make(3, Nx=5000)
    m2, _ = orc.model(2, w2["params_true"], w2["plength"], w2["x"])
    m3, _ = orc.model(3, w3["params_true"], w3["plength"], w3["x"])
    assert np.max(np.abs(m2 - m3) / m2) < 1e-12


def test_linearity_in_heights(orc):
    w = W.make(2, Nx=5000)
    b = W.split(w)
    p = w["params_true"]
    p0 = p.copy(); p0[:b["Nmax"]] = 0.0
    m, _ = orc.model(2, p, w["plength"], w["x"])
    mb, _ = orc.model(2, p0, w["plength"], w["x"])        # background only
    p2 = p.copy(); p2[:b["Nmax"]] *= 3.0
    m3, _ = orc.model(2, p2, w["plength"], w["x"])
    assert np.allclose(m3 - mb, 3.0 * (m - mb), rtol=1e-12, atol=1e-14)


def test_truncation_is_a_hard_window(orc):
    # one l=0 mode, zero background: the model must be exactly 0 outside [imin, imax)
    w = W.make(2, Nx=8000)
    b = W.split(w)
    p = w["params_true"].copy()
    p[:b["Nmax"]] = 0.0
    p[3] = 2.0
    p[b["Nmax"]:b["Nmax"] + b["lmax"]] = 0.0     # visibilities: the height is shared by the l>0 modes of the order
    p[b["z"]:b["z"] + 10] = 0.0
    m, st = orc.model(2, p, w["plength"], w["x"])
    f, G = p[b["Nmax"] + b["lmax"] + 3], abs(p[b["w"] + 3])
    st2, imin, imax = orc.truncation_window(w["x"], f, p[b["s"] + 3] ** 2 + p[b["s"] + 4] ** 2, G, 0, 20.0)
    assert st == 0 and st2 == 0
    assert np.all(m[:imin] == 0.0) and np.all(m[imax:] == 0.0) and np.all(m[imin:imax] > 0.0)


def test_disabled_and_unknown_ids(orc):
    w = W.make(2, Nx=1000)
    for mid, code in ((4, orc.MODEL_DISABLED), (5, orc.MODEL_DISABLED), (15, orc.UNKNOWN_MODEL), (-1, orc.UNKNOWN_MODEL)):
        _, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
        assert st == code


def test_empty_window_status(orc):
    w = W.make(2, Nx=1000)
    b = W.split(w)
    p = w["params_true"].copy()
    p[b["q"] + 1] = -1.0     # negative trunc_c: pmax < pmin -> empty (the reference exits)
    _, st = orc.model(2, p, w["plength"], w["x"])
    assert st == orc.EMPTY_WINDOW


# ---------------------------------------------------------------- noise / likelihood
def test_harvey_like_and_likelihood(orc):
    w = W.make(2, Nx=3000)
    b = W.split(w)
    p = w["params_true"].copy()
    p[:b["Nmax"]] = 0.0
    m, _ = orc.model(2, p, w["plength"], w["x"])
    n = np.abs(p[b["z"]:b["z"] + 10])
    ref = n[9] + n[3] / (1 + (1e-3 * n[4] * w["x"]) ** n[5]) + n[6] / (1 + (1e-3 * n[7] * w["x"]) ** n[8])
    assert np.allclose(m, ref, rtol=1e-14)
    y = synth.make_spectrum(m, seed=5)
    L = orc.likelihood_chi22p(y, m, 1.0)
    assert L == pytest.approx(-(np.sum(y / m) + np.sum(np.log(m))), rel=1e-13)
    assert orc.likelihood_chi22p(y, m, 2.7) == pytest.approx(2.0 * L, rel=1e-15)   # p is truncated to a long
    sig = 0.1 + 0.01 * np.arange(m.size)
    assert orc.likelihood_chi_square(y, m, sig) == pytest.approx(-np.sum((y - m) ** 2 / sig ** 2), rel=1e-13)


def test_gaussian_models(orc):
    for mid in (0, 1):
        w = W.make_gauss(mid, Nx=2000)
        m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
        p, x = w["params_true"], w["x"]
        if mid == 0:
            ref = p[0] * np.exp(-0.5 * (x - p[2]) ** 2 / p[1] ** 2) + p[3]
        else:
            ref = abs(p[0]) * np.exp(-0.5 * (x - p[2]) ** 2 / p[1] ** 2) + p[3] / (1 + (1e-3 * p[4] * x) ** p[5]) + p[6]
        assert st == 0 and np.allclose(m, ref, rtol=1e-14)


def test_batch_matches_single_and_tempering(orc):
    w = W.make(2, Nx=4000)
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=9)
    P = W.perturbed(w, 5)
    T = synth.temperatures(5)
    logL, st = orc.generate_batch(2, w["plength"], w["x"], y, P, T)
    assert np.all(st == 0)
    for k in range(5):
        mk, _ = orc.model(2, P[k], w["plength"], w["x"])
        assert logL[k] == pytest.approx(orc.likelihood_chi22p(y, mk, 1.0) / T[k], rel=1e-15)
    logL1, _ = orc.generate_batch(2, w["plength"], w["x"], y, P, np.ones(5), nthreads=1)
    assert np.allclose(logL1 / T, logL, rtol=1e-15)


@pytest.mark.parametrize("mid", W.ALL_IDS)
def test_every_model_id_runs_and_is_positive(orc, mid):
    for kw in (dict(), dict(asym=25.0, do_amp=True, trunc_c=10000.0)):
        w = W.any_model(mid, Nx=2048, **kw)
        m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
        assert st == 0 and np.all(np.isfinite(m)) and np.all(m > 0)


def test_a1l_window_uses_mean_splitting(orc):
    # switch without break in optimum_lorentzian_calc_a1l_etaa3 (build_lorentzian.cpp:269-278):
    # id 6 with a1(l=1) = 1.4, a1(l=2) = 0.9 -> every window uses f_s = 1.15
    w = W.make(6, Nx=8000)
    b = W.split(w)
    p = w["params_true"].copy()
    p[:b["Nmax"]] = 0.0
    p[2] = 1.0
    p[b["z"]:b["z"] + 10] = 0.0
    p[b["Nmax"]] = 0.0           # V1 = 0: only l=0 and l=2 of order 2 survive
    m, st = orc.model(6, p, w["plength"], w["x"])
    assert st == 0
    nz = np.flatnonzero(m)
    f2 = p[b["Nmax"] + b["lmax"] + 2 * b["Nmax"] + 2]
    G2 = abs(orc.lin_interpol(p[b["Nmax"] + b["lmax"]:b["Nmax"] + b["lmax"] + b["Nmax"]], p[b["w"]:b["w"] + b["Nmax"]], f2))
    st2, imin, imax = orc.truncation_window(w["x"], f2, 0.5 * (1.4 + 0.9), G2, 2, 20.0)
    assert st2 == 0 and nz[0] == imin
