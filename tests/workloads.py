"""Parameter rows for every model id of models_ctrl.list, derived from the synthetic C2 star
(tamcmc_amd.synth).  Test helper only."""
import math

import numpy as np

import tamcmc_amd
from tamcmc_amd import synth


def _base(Nx, lmax=2, asym=0.0, trunc_c=20.0, do_amp=False):
    w = synth.workload_c2(model_case=2, Nx=Nx, asym=asym, trunc_c=trunc_c, do_amp=do_amp)
    return w


def split(w):
    """Blocks of a global params row."""
    pl = w["plength"]
    Nmax, lmax = int(pl[0]), int(pl[1])
    Nf = int(pl[2] + pl[3] + pl[4] + pl[5])
    s = Nmax + lmax + Nf
    wq = s + int(pl[6])
    z = wq + int(pl[7])
    q = z + int(pl[8])
    return dict(Nmax=Nmax, lmax=lmax, Nf=Nf, s=s, w=wq, z=z, q=q, cfg=q + int(pl[9]))


def make(model_case, Nx=4096, asym=0.0, trunc_c=20.0, do_amp=False, seed=7):
    """Returns (plength, params, index_to_relax) for `model_case` on the C2 grid (first Nx bins... the grid
    is rescaled so that all 21 modes stay inside it)."""
    rng = np.random.default_rng(seed)
    w = _base(100000, asym=asym, trunc_c=trunc_c, do_amp=do_amp)
    x = synth.grid(Nx, 2300.0, 840.0 / Nx)
    p = w["params_true"].copy()
    pl = w["plength"].copy()
    b = split(w)
    Nmax, lmax, s, wq, z, q, cfg = b["Nmax"], b["lmax"], b["s"], b["w"], b["z"], b["q"], b["cfg"]
    relax = w["relax"].copy()
    if do_amp:
        p[:Nmax] = p[:Nmax] * math.pi * p[wq:wq + Nmax]   # amplitudes^2 giving similar heights
    if model_case in (2,):
        pass
    elif model_case == 3:
        w3 = synth.workload_c2(model_case=3, Nx=100000, asym=asym, trunc_c=trunc_c, do_amp=do_amp)
        relax = w3["relax"].copy()
    elif model_case in (6, 7, 8):
        extra = {6: 1, 7: Nmax, 8: 2 * Nmax}[model_case]
        a1 = 1.4
        head, tail = p[:s + 6], p[s + 6:]
        ext = a1 * (1.0 + 0.2 * rng.standard_normal(extra))
        if model_case == 6:
            ext = np.array([0.9])         # a1(l=2) < 1 while a1(l=1) > 1: both window branches
        p = np.concatenate([head, ext, tail])
        p[s] = a1
        pl[6] = 6 + extra
        relax = np.concatenate([relax[:s + 6], np.ones(extra, dtype=np.int32), relax[s + 6:]])
        relax[s] = 1
        relax[s + 3] = relax[s + 4] = 0
        relax[-3] = 1     # inclination
    elif model_case in (9, 10):
        if model_case == 9:
            wp = [2600.0, 4.0, 2.0, 3500.0, 2.0]
        else:
            wp = [2700.0, 2600.0, 4.0, 2.0, 3500.0, 2.0]
        p = np.concatenate([p[:wq], wp, p[wq + Nmax:]])
        relax = np.concatenate([relax[:wq], np.ones(len(wp), dtype=np.int32), relax[wq + Nmax:]])
        pl[7] = len(wp)
    elif model_case == 12:
        r = [0.33, 0.335, 0.0001, 0.33, 0.17, 0.05, 0.15, 0.28, 0.09]
        p = np.concatenate([p[:q], r, p[q + 1:]])
        relax = np.concatenate([relax[:q], np.ones(9, dtype=np.int32), relax[q + 1:]])
        pl[9] = 9
        p[s] = 1.4; relax[s] = 1; relax[s + 3] = relax[s + 4] = 0
    elif model_case == 13:
        nh = (lmax + 1) * (Nmax - 1) + lmax + 1
        h = 0.2 + rng.random(nh)
        if do_amp:
            h = h * 3.0
        p = np.concatenate([p[:q], h, p[q + 1:]])
        relax = np.concatenate([relax[:q], np.ones(nh, dtype=np.int32), relax[q + 1:]])
        pl[9] = nh
        p[s] = 1.4; relax[s] = 1; relax[s + 3] = relax[s + 4] = 0
        relax[Nmax:Nmax + lmax] = 0       # visibilities are not used by this model
    else:
        raise ValueError(model_case)
    return dict(model_case=model_case, plength=pl.astype(np.int32), params_true=p, x=x,
                index_to_relax=np.flatnonzero(relax).astype(np.int32), relax=relax)


def make_local(model_case, Nx=4096, asym=0.0, trunc_c=20.0, do_amp=False, seed=11):
    rng = np.random.default_rng(seed)
    w = synth.workload_c1(Nx=10000, trunc_c=trunc_c)
    x = synth.grid(Nx, 94.30, 81.2 / Nx)
    p = w["params_true"].copy()
    pl = w["plength"].copy()
    relax = w["relax"].copy()
    Nf = [int(v) for v in pl[2:6]]
    Nmax = int(pl[0])
    s = Nmax + int(pl[1]) + sum(Nf)
    p[s + 5] = asym
    cfg = len(p) - 2
    p[cfg + 1] = 1.0 if do_amp else 0.0
    if do_amp:
        wq = s + int(pl[6])
        p[:Nmax] = p[:Nmax] * math.pi * p[wq:wq + Nmax]
    if model_case == 11:
        pass
    elif model_case == 14:
        # heights block: l=0 -> Nfl0 entries, then the reference's literal (overlapping) indexing
        # off_l + (l+1) n + |m|; make the block long enough for every read
        need = 0
        off = 0
        for l in range(4):
            if Nf[l] > 0:
                need = max(need, off + (l + 1) * (Nf[l] - 1) + l + 1)
            off += Nf[l]
        NmaxH = max(need, Nmax)
        h = 0.5 + 10.0 * rng.random(NmaxH)
        p = np.concatenate([h, p[Nmax:]])
        relax = np.concatenate([np.ones(NmaxH, dtype=np.int32), relax[Nmax:]])
        pl[0] = NmaxH
        s = NmaxH + int(pl[1]) + sum(Nf)
        p[s] = 0.4; relax[s] = 1; relax[s + 3] = relax[s + 4] = 0
    else:
        raise ValueError(model_case)
    return dict(model_case=model_case, plength=pl.astype(np.int32), params_true=p, x=x,
                index_to_relax=np.flatnonzero(relax).astype(np.int32), relax=relax)


def make_gauss(model_case, Nx=4096):
    x = synth.grid(Nx, 1000.0, 2000.0 / Nx)
    if model_case == 0:
        p = np.array([5.0, 150.0, 2100.0, 0.7])
        pl = [4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    else:
        p = np.array([5.0, -150.0, 2100.0, 3.0, 2.5, 2.2, 0.7])
        pl = [3, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0]
    return dict(model_case=model_case, plength=np.array(pl, dtype=np.int32), params_true=p, x=x,
                index_to_relax=np.arange(p.size, dtype=np.int32), relax=np.ones(p.size, dtype=np.int32))


def any_model(model_case, **kw):
    if model_case in (0, 1):
        kw.pop("asym", None); kw.pop("trunc_c", None); kw.pop("do_amp", None)
        return make_gauss(model_case, **kw)
    if model_case in (11, 14):
        return make_local(model_case, **kw)
    return make(model_case, **kw)


def perturbed(w, Nchains, scale=0.01, seed=3):
    """Chains = truth * (1 + scale * N(0,1)) on the relaxed entries."""
    rng = np.random.default_rng(seed)
    P = np.tile(w["params_true"], (Nchains, 1))
    idx = w["index_to_relax"]
    P[:, idx] *= 1.0 + scale * rng.standard_normal((Nchains, idx.size))
    return P


ALL_IDS = [0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14]
