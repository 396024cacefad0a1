"""Every model name the global / local readers know, reached through the reference's kplr008379927 .model file with its
`model_fullname` (and the keywords that model needs) edited: parameter layout on the CPU, and -- on the GPU -- the HIP
path against the oracle on the arrays the reader produced (reader -> C ABI -> kernels, for all live model ids)."""
import os

import numpy as np
import pytest

from tamcmc_amd.setup_io import Setup, SetupError, model_file_slices

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
CFG = os.path.join(G, "Config_default")
KPLR = os.path.join(G, "kplr008379927_kasoc-psd_slc_v2_1000.model")
KPLR_LOCAL = os.path.join(G, "kplr008379927_kasoc-psd_slc_v2_1000_local-v2.model")

# name -> (id in models_ctrl.list, Nsplit, Nwidth, Ninc) for Nmax = 7, lmax = 2
GLOBAL = {
    "model_MS_Global_a1etaa3_HarveyLike": (2, 6, 7, 1),
    "model_MS_Global_a1etaa3_HarveyLike_Classic": (3, 6, 7, 1),
    "model_MS_Global_a1l_etaa3_HarveyLike": (6, 7, 7, 1),
    "model_MS_Global_a1n_etaa3_HarveyLike": (7, 13, 7, 1),
    "model_MS_Global_a1nl_etaa3_HarveyLike": (8, 20, 7, 1),
    "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v1": (9, 6, 7, 1),
    "model_MS_Global_a1etaa3_AppWidth_HarveyLike_v2": (10, 6, 7, 1),
    "model_MS_Global_a1etaa3_HarveyLike_Classic_v2": (12, 6, 7, 9),
    "model_MS_Global_a1etaa3_HarveyLike_Classic_v3": (13, 6, 7, 35),
}


def write_files(tmp_path, text, lo, hi, step=0.04):
    m = str(tmp_path / "star.model")
    open(m, "w").write(text)
    x = np.arange(lo - 5 * step, hi + 5 * step, step)
    d = str(tmp_path / "star.data")
    with open(d, "w") as f:
        f.write("# synthetic\n! frequency power\n* (microHz) (ppm^2/microHz)\n" + "".join("%.8f 1.0\n" % v for v in x))
    return m, d


def load_variant(tmp_path, name, extra="", trunc=True, reader="io_MS_Global", base=KPLR):
    txt = open(base).read()
    old = [ln for ln in txt.splitlines() if ln.split()[:1] == ["model_fullname"]][0]
    txt = txt.replace(old, "           model_fullname             " + name + ("\n                  trunc_c                 Fix         30.0" if trunc else ""))
    if not txt.endswith("\n"):
        txt += "\n"
    txt += extra
    lo, hi = model_file_slices(base)[0]
    m, d = write_files(tmp_path, txt, lo, hi)
    s = Setup(CFG)
    s.set("Modeling", "prior_fct_name", reader)
    return s.load(m, d, 0)


@pytest.mark.parametrize("name", sorted(GLOBAL))
def test_global_model_layouts(tmp_path, name):
    mid, nsplit, nwidth, ninc = GLOBAL[name]
    s = load_variant(tmp_path, name)
    assert s.model_case == mid and s.model_fullname == name
    assert list(s.plength) == [7, 2, 7, 7, 7, 0, nsplit, nwidth, 10, ninc, 2]
    assert s.inputs[-2] == 30.0 and s.Nparams == s.plength.sum() and np.all(np.isfinite(s.inputs))
    q = 9 + 21
    if mid in (2, 9, 10):            # a1 / inclination projected (io_ms_global.cpp:872-909)
        assert s.inputs_names[q + 3:q + 5] == ["sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini"] and s.inputs_names[q] == "Empty"
    if mid == 6:                     # a1(l): slot 0 and slot 6 carry the splitting (:710-714)
        assert s.inputs_names[q] == "Splitting_a1" and s.inputs_names[q + 6] == "Splitting_a1" and s.inputs[q + 6] == 1.4
    if mid == 7:                     # a1(n): slots 6..6+Nmax, slot 0 emptied (:715-722)
        assert s.inputs_names[q] == "Empty" and all(n == "Splitting_a1" for n in s.inputs_names[q + 6:q + 13])
    if mid == 8:
        assert s.inputs_names[q] == "Empty" and all(n == "Splitting_a1" for n in s.inputs_names[q + 6:q + 20])
    if mid in (9, 10):               # Appourchaux+2016 width relation from numax (height-weighted mean frequency)
        w = q + 6
        n = 5 if mid == 9 else 6
        assert all(nm.startswith("width:Appourchaux_v%d:" % (1 if mid == 9 else 2)) for nm in s.inputs_names[w:w + n])
        assert s.inputs_names[w + n:w + 7] == ["Empty"] * (7 - n) and s.priors_names[w] == "Gaussian"
        h, f = s.inputs[:7], s.inputs[9:30]
        numax = (f[:7] @ h + 1.5 * (f[7:14] @ h) + 0.53 * (f[14:21] @ h)) / (h.sum() * (1 + 1.5 + 0.53))
        assert s.inputs[w + (0 if mid == 9 else 1)] == pytest.approx(numax, rel=1e-12)
        assert s.inputs[w + (1 if mid == 9 else 2)] == pytest.approx(4. / 2150. * numax + (1. - 1000. * 4. / 2150.), rel=1e-12)
    if mid == 13:                    # Classic_v3: one height per (n, l, m >= 0); visibilities but the last emptied (:941-973)
        k = q + 6 + 7 + 10
        assert s.inputs_names[k] == "Inc: H0,1,0" and s.inputs_names[k + 34] == "Inc: H6,2,2" and s.extra_priors[3] == 2
        assert s.inputs_names[7] == "Empty" and s.inputs_names[8] == "Visibility_l2"     # the reference's `el < lmax` loop


def test_keyword_variants(tmp_path):
    # squared amplitudes instead of heights (io_ms_global.cpp:512-520)
    s = load_variant(tmp_path, "model_MS_Global_a1etaa3_HarveyLike", extra="fit_squareAmplitude_instead_Height   bool   1\n")
    raw = [ln.split() for ln in open(KPLR) if len(ln.split()) == 6 and ln.split()[0] == "0"]
    assert s.inputs_names[0] == "Amplitude_l0" and s.inputs[-1] == 1.0
    assert s.inputs[0] == pytest.approx(np.pi * float(raw[0][4]) * float(raw[0][5]), rel=1e-15)
    # a frequency keyword replaces the default GUG wings / switches to Uniform (:634-651)
    s = load_variant(tmp_path, "model_MS_Global_a1etaa3_HarveyLike", extra="Frequency   GUG   -1  -1  -1   0.5   0.25\n")
    assert s.priors_names[9] == "GUG" and list(s.priors[2:, 9]) == [0.5, 0.25]
    s = load_variant(tmp_path, "model_MS_Global_a1etaa3_HarveyLike", extra="Frequency   Uniform   -1\n")
    assert s.priors_names[9] == "Uniform" and s.priors_names_switch[9] == 1 and list(s.priors[2:, 9]) == [-9999, -9999]
    # the projected splitting given directly
    s = load_variant(tmp_path, "model_MS_Global_a1etaa3_HarveyLike",
                     extra="sqrt(splitting_a1).cosi   Uniform   0.7   0.0   2.0\nsqrt(splitting_a1).sini   Uniform   0.9   0.0   2.0\n")
    q = 30
    assert list(s.inputs[q + 3:q + 5]) == [0.7, 0.9] and list(s.priors[:2, q + 3]) == [0.0, 2.0] and s.inputs_names[q] == "Empty"
    with pytest.raises(SetupError):          # ... but not with the Classic model (:978-985)
        load_variant(tmp_path, "model_MS_Global_a1etaa3_HarveyLike_Classic",
                     extra="sqrt(splitting_a1).cosi   Uniform   0.7   0.0   2.0\nsqrt(splitting_a1).sini   Uniform   0.9   0.0   2.0\n")
    with pytest.raises(SetupError):          # a1(n) needs as many l=1 as l=2 modes -- and no projected splitting
        load_variant(tmp_path, "model_MS_Global_a1n_etaa3_HarveyLike",
                     extra="sqrt(splitting_a1).cosi   Uniform   0.7   0.0   2.0\nsqrt(splitting_a1).sini   Uniform   0.9   0.0   2.0\n")
    # local reader: Fix_Auto heights scale the Jeffreys bounds with each mode's own height (io_local.cpp:728-781)
    txt_extra = ""
    base = open(KPLR_LOCAL).read().replace("                   Height            Jeffreys          1.000000          1000.000",
                                           "                   Height            Fix_Auto          10.0          3.0")
    p = tmp_path / "loc.model"
    open(p, "w").write(base.replace("model_MS_local_Hnlm", "model_MS_local_basic"))
    s = load_variant(tmp_path, "model_MS_local_basic", extra=txt_extra, trunc=False, reader="io_local", base=str(p))
    assert s.model_case == 11 and s.priors_names[0] == "Jeffreys"
    assert s.priors[0, 0] == pytest.approx(s.inputs[0] / 10.0) and s.priors[1, 0] == pytest.approx(s.inputs[0] * 3.0)
    assert s.priors[0, 1] == pytest.approx(s.inputs[1] / 10.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GLOBAL) + ["model_MS_local_basic", "model_MS_local_Hnlm"])
def test_reader_to_kernels_parity(accel_mod, orc, tmp_path, name):
    if name.startswith("model_MS_local"):
        base = open(KPLR_LOCAL).read()
        p = tmp_path / "loc.model"
        open(p, "w").write(base)
        s = load_variant(tmp_path, name, trunc=True, reader="io_local", base=str(p))
    else:
        s = load_variant(tmp_path, name)
    with accel_mod.Accel(s.model_case, s.plength, s.x, np.ones(s.Nx)) as a0:
        m_true, st = a0.model_explicit(s.inputs)
    assert st == 0
    rm, rst = orc.model(s.model_case, s.inputs, s.plength, s.x)
    assert rst == 0 and np.allclose(m_true, rm, rtol=1e-12, atol=0)
    rng = np.random.default_rng(5)
    y = rm * (-np.log(rng.uniform(size=s.Nx)))
    P = np.tile(s.inputs, (4, 1))
    P[1:, s.index_to_relax] += 0.2 * s.err * rng.standard_normal((3, s.Nvars))
    T = 1.7 ** np.arange(4)
    with accel_mod.Accel(s.model_case, s.plength, s.x, y) as acc:
        acc.set_vars(s.index_to_relax)
        logL, st = acc.eval_batch(P, T)
        logLg, stg, g = acc.eval_batch(P, T, grad=True)
    ref, rst = orc.generate_batch(s.model_case, s.plength, s.x, y, P, T)
    assert np.array_equal(st, rst) and np.array_equal(stg, rst)
    ok = rst == 0
    assert ok.sum() >= 1
    assert np.allclose(logL[ok], ref[ok], rtol=1e-10, atol=0) and np.allclose(logLg[ok], ref[ok], rtol=1e-10, atol=0)
    assert np.all(np.isfinite(g[ok]))
