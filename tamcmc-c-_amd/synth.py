"""Synthetic workloads of SURVEY.md section 8d (self-contained: same generator on every box).

A workload is a dict with
  model_case, plength (11 ints), x (Nx), params_true (Nparams), relax (Nparams 0/1),
  index_to_relax, err (Nvars initial step sizes, errors_default.cfg rule err = A*theta + B, MALA.cpp:256),
  names (one per parameter).
`make_spectrum(model_true)` multiplies a model spectrum by chi^2 2-dof noise; the model itself comes
from whoever calls (the HIP library in bench.py, the oracle in CPU-only tests): this module does
no model arithmetic of its own.

Mode table, noise levels, visibilities and splitting of the C2/C3/C5 star are the numbers of the
reference's example input test/inputs/kplr008379927_kasoc-psd_slc_v2_1000.model:5,36-61,76-82.
"""
import math

import numpy as np

MASK64 = (1 << 64) - 1


class XorShift64:
    """s ^= s<<13; s ^= s>>7; s ^= s<<17."""

    def __init__(self, seed):
        self.s = seed & MASK64
        if self.s == 0:
            self.s = 88172645463325252

    def next_u64(self):
        s = self.s
        s ^= (s << 13) & MASK64
        s ^= s >> 7
        s ^= (s << 17) & MASK64
        self.s = s
        return s

    def uniform(self):
        return ((self.next_u64() >> 11) + 0.5) / 9007199254740992.0  # 2^53

    def uniforms(self, n):
        out = np.empty(n)
        for i in range(n):
            out[i] = self.uniform()
        return out

    def normals(self, n):
        out = np.empty(n + (n & 1))
        for i in range(0, out.size, 2):
            r0, r1 = self.uniform(), self.uniform()
            rad = math.sqrt(-2.0 * math.log(r0))
            out[i] = rad * math.cos(2.0 * math.pi * r1)
            out[i + 1] = rad * math.sin(2.0 * math.pi * r1)
        return out[:n]


# test/inputs/kplr008379927_kasoc-psd_slc_v2_1000.model:36-56  (l, nu, Gamma, H)
_KPLR_MODES = [
    (0, 2324.48999, 1.06971, 0.80080), (0, 2442.87988, 1.11876, 1.08911), (0, 2563.39990, 1.17982, 1.29912),
    (0, 2683.52002, 1.31250, 1.35009), (0, 2804.55005, 1.59687, 1.22185), (0, 2924.45996, 2.11491, 0.96493),
    (0, 3044.98999, 2.96253, 0.66326),
    (1, 2380.04004, 1.09640, 1.20120), (1, 2499.64990, 1.14045, 1.63366), (1, 2620.07007, 1.21997, 1.94867),
    (1, 2741.30005, 1.40819, 2.02514), (1, 2860.90991, 1.78100, 1.83278), (1, 2981.33008, 2.43618, 1.44740),
    (1, 3102.36011, 3.46559, 0.99489),
    (2, 2313.12988, 1.06409, 0.42442), (2, 2433.05005, 1.11526, 0.57723), (2, 2552.25000, 1.17293, 0.68853),
    (2, 2673.59009, 1.29802, 0.71555), (2, 2794.01001, 1.56543, 0.64758), (2, 2913.91992, 2.05953, 0.51141),
    (2, 3035.25000, 2.88200, 0.35153),
]
_KPLR_NOISE = [0.0, 0.0, 1.0, 11.049588, 49.669854, 4.0, 0.93569041, 1.3516447, 2.0, 0.13392108]

# errors_default.cfg:5-27  name -> (A, B)
_ERR = {
    "Frequency_l": (0.0, 0.07), "Height_l": (0.02, 0.01), "Width_l": (0.015, 0.005),
    "Visibility_l1": (0.0, 0.1), "Visibility_l2": (0.0, 0.05), "Visibility_l3": (0.0, 0.05),
    "Splitting_a1": (0.02, 0.0), "sqrt(splitting_a1).cosi": (0.1, 0.05), "sqrt(splitting_a1).sini": (0.1, 0.05),
    "Asphericity_eta": (0.2, 0.000001), "Splitting_a3": (0.15, 0.005), "Lorentzian_asymetry": (0.02, 1.0),
    "Inclination": (0.0, 5.0), "Harvey-Noise_H": (0.03, 0.0), "Harvey-Noise_tc": (0.015, 0.0),
    "Harvey-Noise_p": (0.015, 0.0), "White_Noise_N0": (0.015, 0.0002), "fixed": (0.0, 0.0),
}


def _finish(w):
    w["plength"] = np.asarray(w["plength"], dtype=np.int32)
    w["params_true"] = np.asarray(w["params_true"], dtype=np.float64)
    w["relax"] = np.asarray(w["relax"], dtype=np.int32)
    assert w["params_true"].size == int(w["plength"].sum()) == w["relax"].size == len(w["names"])
    w["index_to_relax"] = np.flatnonzero(w["relax"]).astype(np.int32)
    err = []
    for i in w["index_to_relax"]:
        a, b = _ERR[w["names"][i]]
        err.append(a * w["params_true"][i] + b)
    w["err"] = np.asarray(err)
    return w


def _global_workload(model_case, modes_by_l, V, a1, inc_deg, eta, a3, asym, noise, trunc_c, do_amp, x):
    """Params row of a model_MS_Global_a1etaa3_HarveyLike-style model (ids 2, 3)."""
    lmax = len(modes_by_l) - 1
    Nmax = len(modes_by_l[0])
    p, names, relax = [], [], []

    def add(v, name, r):
        p.append(float(v)); names.append(name); relax.append(int(r))

    for (_, _, H) in modes_by_l[0]:
        add(H, "Height_l", 1)
    for l in range(1, lmax + 1):
        add(V[l - 1], f"Visibility_l{l}", 1)
    for l in range(lmax + 1):
        assert len(modes_by_l[l]) == Nmax
        for (nu, _, _) in modes_by_l[l]:
            add(nu, "Frequency_l", 1)
    inc = math.radians(inc_deg)
    free_inc = model_case == 2
    add(a1, "Splitting_a1", 0 if free_inc else 1)
    add(eta, "Asphericity_eta", 0)
    add(a3, "Splitting_a3", 0)
    add(math.sqrt(a1) * math.cos(inc), "sqrt(splitting_a1).cosi", 1 if free_inc else 0)
    add(math.sqrt(a1) * math.sin(inc), "sqrt(splitting_a1).sini", 1 if free_inc else 0)
    add(asym, "Lorentzian_asymetry", 0)
    for (_, G, _) in modes_by_l[0]:
        add(G, "Width_l", 1)
    kinds = ["Harvey-Noise_H", "Harvey-Noise_tc", "Harvey-Noise_p"]
    for k in range(9):
        free = (noise[3 * (k // 3) + 1] != 0.0) and (k % 3 != 2)   # H and tau of the active Harvey profiles
        add(noise[k], kinds[k % 3], 1 if free else 0)
    add(noise[9], "White_Noise_N0", 1)
    add(inc_deg, "Inclination", 0 if free_inc else 1)
    add(trunc_c, "fixed", 0)
    add(1.0 if do_amp else 0.0, "fixed", 0)
    Nf = [len(m) for m in modes_by_l] + [0] * (3 - lmax)
    plength = [Nmax, lmax] + Nf + [6, Nmax, 10, 1, 2]
    return _finish(dict(model_case=model_case, plength=plength, x=x, params_true=p, relax=relax, names=names))


def grid(Nx, x0, step):
    return x0 + step * np.arange(Nx, dtype=np.float64)


def workload_c2(model_case=2, Nx=100000, asym=0.0, trunc_c=20.0, do_amp=False):
    """C2/C3/C5 star: 21 modes l=0..2, 56 params, x = 2300 + 0.0084 i."""
    x = grid(Nx, 2300.0, 0.0084)
    modes = [[(nu, G, H) for (l, nu, G, H) in _KPLR_MODES if l == ll] for ll in range(3)]
    return _global_workload(model_case, modes, [1.5, 0.53], 1.4, 55.0, 1e-5, 0.01, asym, _KPLR_NOISE, trunc_c, do_amp, x)


def workload_c4(model_case=2, Nx=100000, asym=0.0, trunc_c=20.0, Nmax=14):
    """C4: Nmax=14, l=0..3 -> 106 params; nu = Dnu (n + l/2 + eps) - l(l+1) D0."""
    Dnu, eps, D0, n0 = 60.0, 1.4, 0.9, 38
    numax = Dnu * (n0 + Nmax / 2.0 + eps)
    modes = []
    for l in range(4):
        row = []
        for k in range(Nmax):
            nu = Dnu * (n0 + k + l / 2.0 + eps) - l * (l + 1) * D0
            G = 1.0 + 2.0 * k / (Nmax - 1)
            H = 2.0 * math.exp(-0.5 * ((nu - numax) / (3.0 * Dnu)) ** 2)
            row.append((nu, G, H))
        modes.append(row)
    x = grid(Nx, modes[2][0][0] - 20.0, (modes[1][-1][0] + 40.0 - (modes[2][0][0] - 20.0)) / Nx)
    return _global_workload(model_case, modes, [1.5, 0.53, 0.08], 1.4, 55.0, 1e-5, 0.01, asym, _KPLR_NOISE, trunc_c, False, x)


def workload_c1(Nx=10000, trunc_c=20.0):
    """C1: model_MS_local_basic (id 11), one slice: x0 = 94.30, step = 0.00812."""
    x = grid(Nx, 94.30, 0.00812)
    # (l, nu, Gamma, H) inside the slice
    modes = [(0, 110.2, 0.15, 12.0), (0, 152.8, 0.18, 9.0), (1, 131.5, 0.16, 14.0), (1, 168.9, 0.2, 7.0),
             (2, 105.6, 0.17, 6.0), (2, 148.1, 0.19, 5.0)]
    p, names, relax = [], [], []

    def add(v, name, r):
        p.append(float(v)); names.append(name); relax.append(int(r))

    by_l = [[m for m in modes if m[0] == l] for l in range(4)]
    for l in range(4):
        for m in by_l[l]:
            add(m[3], "Height_l", 1)
    for l in range(4):
        for m in by_l[l]:
            add(m[1], "Frequency_l", 1)
    a1, inc = 0.4, math.radians(60.0)
    add(a1, "Splitting_a1", 0); add(0.0, "Asphericity_eta", 0); add(0.0, "Splitting_a3", 0)
    add(math.sqrt(a1) * math.cos(inc), "sqrt(splitting_a1).cosi", 1)
    add(math.sqrt(a1) * math.sin(inc), "sqrt(splitting_a1).sini", 1)
    add(0.0, "Lorentzian_asymetry", 0)
    for l in range(4):
        for m in by_l[l]:
            add(m[2], "Width_l", 1)
    add(0.8, "White_Noise_N0", 1)
    add(60.0, "Inclination", 0)
    add(trunc_c, "fixed", 0); add(0.0, "fixed", 0)
    Nf = [len(b) for b in by_l]
    Nmax = sum(Nf)
    plength = [Nmax, 0] + Nf + [6, Nmax, 1, 1, 2]
    return _finish(dict(model_case=11, plength=plength, x=x, params_true=p, relax=relax, names=names))


def make_spectrum(model_true, seed=88172645463325252):
    """y_i = M_i * (-ln u_i): chi^2 with 2 d.o.f. around the model."""
    rng = XorShift64(seed)
    u = rng.uniforms(model_true.size)
    return model_true * (-np.log(u))


def chain_params(w, Nchains, scale=0.5, seed=0x9E3779B97F4A7C15):
    """Chain m = theta_true + scale * err_k * g_{m,k} on the relaxed entries."""
    rng = XorShift64(seed)
    idx = w["index_to_relax"]
    g = rng.normals(Nchains * idx.size).reshape(Nchains, idx.size)
    P = np.tile(w["params_true"], (Nchains, 1))
    P[:, idx] += scale * w["err"][None, :] * g
    return P


def temperatures(Nchains, Tmax=150.0):
    """Tcoefs[m] = lambda^m (MALA.cpp:103) with lambda = Tmax^(1/(Nchains-1))."""
    if Nchains == 1:
        return np.ones(1)
    lam = Tmax ** (1.0 / (Nchains - 1))
    return lam ** np.arange(Nchains, dtype=np.float64)
