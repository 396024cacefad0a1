"""N3: readers of the reference's own input files (include/tamcmc_io.h).

Fixtures: tests/golden/ref_inputs/ holds the input DATA files of the reference's test directory
(test/inputs/*.model, *.data) and its Config/default/*.cfg / *.list files, unchanged.  The reference keeps no
record of what its readers produce from them, so the expected values below are worked out by hand from the file
contents and the rules in io_ms_global.cpp / io_local.cpp / config.cpp (cited per check) -- "parity unpinned"
against the reference's executable, pinned against its documented rules."""
import math
import os

import numpy as np
import pytest

from tamcmc_amd.setup_io import (IO_E_NAME, IO_E_OPEN, IO_E_RANGE, IO_E_SYNTAX, Setup, SetupError, model_file_slices)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs")
CFG = os.path.join(G, "Config_default")
TF_MODEL = os.path.join(G, "TF_3443483_local-v3.model")
TF_DATA = os.path.join(G, "TF_3443483_local-v3.data")
KPLR = os.path.join(G, "kplr008379927_kasoc-psd_slc_v2_1000.model")
KPLR_LOCAL = os.path.join(G, "kplr008379927_kasoc-psd_slc_v2_1000_local-v2.model")
M88 = os.path.join(G, "00088.0.model")


def synth_data(path, lo, hi, step=0.05, trailing_newline=True, labels=True, units=True):
    x = np.arange(lo - 5 * step, hi + 5 * step, step)
    lines = ["# synthetic"]
    if labels:
        lines.append("! frequency power")
    if units:
        lines.append("* (microHz) (ppm^2/microHz)")
    lines += ["%18.8f %18.8f" % (v, 1.0 + 0.001 * i) for i, v in enumerate(x)]
    with open(path, "w") as f:
        f.write("\n".join(lines) + ("\n" if trailing_newline else ""))
    return x


def load_global(path, tmp_path, reader="io_MS_Global", step=0.05):
    sl = model_file_slices(path)
    d = str(tmp_path / "synth.data")
    synth_data(d, sl[0, 0], sl[0, 1], step)
    s = Setup(CFG)
    s.set("Modeling", "prior_fct_name", reader)
    return s.load(path, d, 0)


# ---------------------------------------------------------------- config files
def test_config_default_and_lists():
    s = Setup(CFG)
    # Config/default/config_default.cfg
    assert s.get("MALA", "target_acceptance") == "0.234" and s.get("MALA", "Nchains") == "10"
    assert s.get("Modeling", "prior_fct_name") == "io_local" and s.get("Modeling", "likelihood_fct_name") == "chi(2,2p)"
    assert s.get("Outputs", "output_root_name") == "" and s.get("Outputs", "file_format") == "binary"
    cfg = s.sampler_cfg(seed=7)
    assert (cfg.Nchains, cfg.Nchains_local, cfg.chain_offset) == (10, 10, 0)
    assert cfg.lambda_temp == 1.70 and cfg.c0 == 10 and cfg.epsilon1 == 1e-12 and cfg.A1 == 1e14
    assert cfg.dN_mixing == 1                                  # "dN_mixing=1.;" read with str_to_int
    assert cfg.n_learn == 3 and list(cfg.Nt_learn)[:3] == [500, 1500, 100000] and list(cfg.periods_learn)[:2] == [1, 1]
    assert cfg.seed == 7 and cfg.prior_fct_switch == 3          # priors_ctrl.list: 3 io_local
    with pytest.raises(SetupError) as e:
        s.set("MALA", "no_such_key", "1")
    assert e.value.code == IO_E_NAME
    with pytest.raises(SetupError) as e:
        Setup(os.path.join(G, "nowhere"))
    assert e.value.code == IO_E_OPEN


def test_phase_presets():
    # config_presets.cpp:69-93,133-134
    s = Setup(CFG)
    s.apply_phase("Burn-in", 5000, 1.8)
    c = s.sampler_cfg()
    assert list(c.Nt_learn)[:3] == [500, 1500, 5001] and c.dN_mixing == 1 and c.c0 == 1.8
    s.apply_phase("Learning", 50000, 1.7)
    c = s.sampler_cfg()
    assert list(c.Nt_learn)[:3] == [500, 1500, 50001] and c.dN_mixing == 50001
    s = Setup(CFG)
    s.apply_phase("Acquire", 1000, 0.0)
    c = s.sampler_cfg()
    assert list(c.Nt_learn)[:3] == [1001, 1002, 1003] and c.dN_mixing == 1 and s.get("Outputs", "Nsamples") == "1000"
    with pytest.raises(SetupError):
        s.apply_phase("Cooling", 10, 1.0)


def test_cfg_syntax_errors(tmp_path):
    import shutil
    d = tmp_path / "cfg"
    shutil.copytree(CFG, d)
    txt = open(d / "config_default.cfg").read()
    open(d / "config_default.cfg", "w").write(txt.replace("target_acceptance=0.234;", "target_acceptance=0.234"))
    with pytest.raises(SetupError) as e:
        Setup(d)
    assert e.value.code == IO_E_SYNTAX                         # "Terminator not found!", config.cpp:649-698
    open(d / "config_default.cfg", "w").write(txt.replace("c0=10;", "c00=10;"))
    with pytest.raises(SetupError) as e:
        Setup(d)
    assert e.value.code == IO_E_NAME                           # "is not a known keyword", config.cpp:911-917


# ---------------------------------------------------------------- .data
def test_reference_data_file_and_crop():
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 0)
    raw = np.loadtxt(TF_DATA, comments=["#", "!", "*"])
    assert s.resol == raw[2, 0] - raw[1, 0]                    # config.cpp:354
    lo, hi = 94.30, 102.20                                     # slice 1 of the .model file
    keep = raw[(np.arange(len(raw)) >= np.argmax(raw[:, 0] >= lo)) & (raw[:, 0] < hi)]
    assert s.Nx == len(keep) == 973
    assert np.array_equal(s.x, keep[:, 0]) and np.array_equal(s.y, keep[:, 1])
    assert np.all(s.sigma_y == 1.0)                            # ysig_col=-1 -> ones, config.cpp:165-173
    assert (s.xlabel, s.ylabel, s.xunit, s.yunit) == ("frequency", "power", "(microHz)", "(ppm^2/microHz)")
    for k, (a, b) in enumerate(model_file_slices(TF_MODEL)):
        sk = Setup(CFG).load(TF_MODEL, TF_DATA, k)
        assert sk.x[0] >= a and sk.x[-1] < b and sk.fmin == a and sk.fmax == b
        assert sk.Nx == np.count_nonzero((raw[:, 0] >= a) & (raw[:, 0] < b))


def test_data_reader_quirks(tmp_path):
    s = Setup(CFG)
    s.set("Modeling", "prior_fct_name", "io_MS_Global")
    # a last line without a trailing newline is never parsed (while(!eof) loop, config.cpp:608-620)
    d1, d2 = str(tmp_path / "a.data"), str(tmp_path / "b.data")
    x = synth_data(d1, 1890.0, 2280.0, 0.5, trailing_newline=True)
    synth_data(d2, 1890.0, 2280.0, 0.5, trailing_newline=False)
    # widen the model range beyond the data so that the crop keeps everything up to the end
    txt = open(M88).read().replace("* 1850.0 2300.0", "* 1000  3000")
    assert "* 1000  3000" in txt
    m = str(tmp_path / "wide.model")
    open(m, "w").write(txt)
    n1 = s.load(m, d1, 0).Nx
    n2 = s.load(m, d2, 0).Nx
    assert n1 == len(x) and n2 == len(x) - 1
    # labels but no units: the reader skips one data row (config.cpp:585-606)
    d3 = str(tmp_path / "c.data")
    synth_data(d3, 1890.0, 2280.0, 0.5, units=False)
    assert s.load(m, d3, 0).Nx == len(x) - 1
    # neither labels nor units: nothing is skipped
    d4 = str(tmp_path / "d.data")
    synth_data(d4, 1890.0, 2280.0, 0.5, labels=False, units=False)
    assert s.load(m, d4, 0).Nx == len(x)
    # a range entirely above the data
    open(m, "w").write(txt.replace("* 1000  3000", "* 5000  6000"))
    with pytest.raises(SetupError) as e:
        s.load(m, d1, 0)
    assert e.value.code == IO_E_RANGE


# ---------------------------------------------------------------- .model, local reader
def test_local_basic_slice0():
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 0)
    assert (s.model_fullname, s.model_case, s.likelihood_case, s.prior_case) == ("model_MS_local_basic", 11, 0, 3)
    assert s.ID == "003443483" and s.Dnu == 10.6795 and s.C_l == 2.59256 and s.numax == -9999
    assert list(s.plength) == [2, 0, 1, 0, 1, 0, 6, 2, 1, 1, 2] and s.Nparams == 16 and s.Nvars == 9
    # modes strictly inside (94.30, 102.20): l=0 99.1056, l=2 97.7716 (io_local.cpp:483-540)
    assert s.inputs_names[:4] == ["Height_l", "Height_l", "Frequency_l", "Frequency_l"]
    assert list(s.inputs[:4]) == [1199.06221, 635.50294, 99.10560, 97.77160]
    # "Height Jeffreys 1.0 1 10000": values after the first one are the prior parameters (i0_prior=1, :797-804)
    assert list(s.priors[:, 0]) == [1, 10000, -9999, -9999] and s.priors_names[0] == "Jeffreys" and s.priors_names_switch[0] == 4
    # frequencies: GUG on the eigen-solution window with 1% wings (:613-620)
    assert s.priors_names[2] == "GUG" and s.priors_names_switch[2] == 7
    assert list(s.priors[:2, 2]) == [98.76572, 99.64391]
    assert s.priors[2, 2] == s.priors[3, 2] == 0.01 * abs(99.64391 - 98.76572)
    # Splitting_a1 Uniform(0.4; 0, 1.5) + Inclination Uniform(45) -> sqrt(a1) cos i, sqrt(a1) sin i  (:956-986)
    q = 4
    assert s.inputs_names[q:q + 6] == ["Empty", "Asphericity_eta", "Splitting_a3", "sqrt(splitting_a1).cosi",
                                       "sqrt(splitting_a1).sini", "Lorentzian_asymetry"]
    assert s.inputs[q + 3] == pytest.approx(math.sqrt(0.4) * math.cos(math.radians(45)), rel=1e-15)
    assert s.inputs[q + 4] == pytest.approx(math.sqrt(0.4) * math.sin(math.radians(45)), rel=1e-15)
    assert list(s.relax[q:q + 6]) == [0, 0, 0, 1, 1, 0]
    assert s.priors[0, q + 3] == 0 and s.priors[1, q + 3] == pytest.approx(math.sqrt(1.5), rel=1e-15)
    # Asphericity_eta Fix_Auto 1: eta = 4/3 pi Dnl (a1 1e-6)^2 / (rho G), rho from Dnu (:855-860, :310-323)
    rho_sun = 1.98855e30 * 1e3 / (4 * math.pi * (6.96342e5 * 1e5) ** 3 / 3)
    rho = (10.6795 / 135.1) ** 2 * rho_sun
    assert s.inputs[q + 1] == pytest.approx(4. / 3. * math.pi * 0.75 * (0.4e-6) ** 2 / (rho * 6.667e-8), rel=1e-13)
    # Width Fix_Auto: Jeffreys(resolution, Dnu/3) (:809-822)
    w = 10
    assert s.inputs_names[w:w + 2] == ["Width_l", "Width_l"] and list(s.inputs[w:w + 2]) == [0.30864, 0.30646]
    assert list(s.priors[:2, w]) == [s.resol, 10.6795 / 3]
    # noise: mean of the Harvey profiles + white noise at the two ends of the slice, Uniform(0.5 min, 1.5 max) (:1160-1220)
    def bg(f):
        return 664.13440 / (1 + (1e-3 * 32.186165 * f) ** 4.0) + 355.78922 / (1 + (1e-3 * 13.739317 * f) ** 2.5) + 5.1845856
    lo, hi = bg(94.30), bg(102.20)
    assert s.inputs_names[12] == "White_Noise_N0" and s.priors_names[12] == "Uniform"
    assert s.inputs[12] == pytest.approx((lo + hi) / 2, rel=1e-14)
    assert s.priors[0, 12] == pytest.approx(0.5 * min(lo, hi), rel=1e-14) and s.priors[1, 12] == pytest.approx(1.5 * max(lo, hi), rel=1e-14)
    # no trunc_c keyword -> -1 -> 10000 (:1102-1106); extra priors of a local fit (:676-680)
    assert list(s.inputs[-2:]) == [10000.0, 0.0] and list(s.extra_priors) == [0, 0, 0.2, 0]
    # initial proposal errors err = A var + B by exact name match (MALA.cpp:246-257, errors_default.cfg); the
    # "sqrt(Splitting_a1).sini" entry of the file is spelt with a capital S and therefore never matches -> 1
    exp = [0.02 * 1199.06221 + 0.01, 0.02 * 635.50294 + 0.01, 0.07, 0.07, 0.1 * s.inputs[q + 3] + 0.05, 1.0,
           0.015 * 0.30864 + 0.005, 0.015 * 0.30646 + 0.005, 0.015 * s.inputs[12] + 0.0002]
    assert np.allclose(s.err, exp, rtol=1e-15)
    assert list(s.index_to_relax) == [0, 1, 2, 3, 7, 8, 10, 11, 12]


def test_local_every_slice_has_consistent_layout():
    for k in range(8):
        s = Setup(CFG).load(TF_MODEL, TF_DATA, k)
        pl = s.plength
        assert pl.sum() == s.Nparams and pl[0] == pl[7] == pl[2] + pl[3] + pl[4] + pl[5] and pl[1] == 0
        off = pl[0]
        f = s.inputs[off:off + pl[0]]
        assert np.all((f > s.fmin) & (f < s.fmax))
        assert s.Nvars == np.count_nonzero(s.relax) and len(s.err) == s.Nvars
    with pytest.raises(SetupError) as e:                        # a slice index beyond the file: no range, no modes
        Setup(CFG).load(TF_MODEL, TF_DATA, 8)
    assert e.value.code in (IO_E_RANGE, IO_E_SYNTAX)


def test_local_hnlm(tmp_path):
    s = load_global(KPLR_LOCAL, tmp_path, reader="io_local")
    assert (s.model_fullname, s.model_case) == ("model_MS_local_Hnlm", 14)
    # slice (2400, 2460): l=0 2442.8799 and l=2 2433.05 -> heights: 1 (l=0) + 3 (l=2, m=0..2)
    assert list(s.plength) == [4, 0, 1, 0, 1, 0, 6, 2, 1, 1, 2]
    assert s.inputs_names[:4] == ["Height_l", "H(0,2,0)", "H(0,2,1)", "H(0,2,2)"]
    i = math.radians(55.0)
    r = [0.25 * (3 * math.cos(i) ** 2 - 1) ** 2, 1.5 * math.cos(i) ** 2 * math.sin(i) ** 2, 0.375 * math.sin(i) ** 4]
    raw = [ln.split() for ln in open(KPLR_LOCAL) if len(ln.split()) == 6 and ln.split()[0] == "2" and ln.split()[1].startswith("2433.05")]
    h2 = float(raw[0][5])                                       # H of the l=2 mode in the eigen table
    assert np.allclose(s.inputs[1:4], [h2 * v for v in r], rtol=1e-13)
    assert list(s.priors[:2, 1]) == [1.0, 1000.0] and s.priors_names[1] == "Jeffreys"   # the Height keyword's prior
    assert s.inputs_names[6] == "Splitting_a1" and s.inputs[6] == 1.4 and s.relax[6] == 1   # not projected for Hnlm
    assert s.inputs_names[15] == "Empty" and s.extra_priors[3] == 2                       # inclination slot emptied


# ---------------------------------------------------------------- .model, global reader
def test_global_classic_v2(tmp_path):
    s = load_global(KPLR, tmp_path)
    assert (s.model_fullname, s.model_case, s.prior_case) == ("model_MS_Global_a1etaa3_HarveyLike_Classic_v2", 12, 2)
    assert list(s.plength) == [7, 2, 7, 7, 7, 0, 6, 7, 10, 9, 2] and s.Nparams == 64
    assert s.ID == "008379927" and s.Dnu == 120.186 and s.numax == -9999 and (s.fmin, s.fmax) == (2300, 3140)
    assert s.inputs_names[0] == "Height_l0" and list(s.priors[:, 0]) == [1, 1000, -9999, -9999]
    assert s.inputs_names[7:9] == ["Visibility_l1", "Visibility_l2"] and list(s.inputs[7:9]) == [1.5, 0.53]
    assert s.priors_names[7] == "Gaussian" and list(s.priors[:2, 7]) == [1.5, 0.15]
    # default GUG wings 0.01 Dnu (io_ms_global.cpp:562-570)
    assert s.priors[2, 9] == 0.01 * 120.186
    q = 9 + 21
    assert s.inputs_names[q] == "Splitting_a1" and s.inputs[q] == 1.4 and list(s.priors[:2, q]) == [0, 5]
    # noise block: first Harvey absent (H=0) -> (0, 0, 1) fixed; second fixed; third + white noise Gaussian
    z = q + 6 + 7
    assert list(s.inputs[z:z + 10]) == [0, 0, 1, 11.049588, 49.669854, 4.0, 0.93569041, 1.3516447, 2.0, 0.13392108]
    assert list(s.relax[z:z + 10]) == [0] * 6 + [1] * 4
    # sigma = 1.5 (err_m + err_p), floored at 5% / 0.5% / 5% of the value; p with zero errors -> 10%; N0 -> 10%
    assert s.priors[1, z + 6] == pytest.approx(0.05 * 0.93569041)
    assert s.priors[1, z + 7] == pytest.approx(1.5 * (0.0039247893 + 0.0039362189))
    assert s.priors[1, z + 8] == pytest.approx(0.2) and s.priors[1, z + 9] == pytest.approx(0.013392108)
    # Classic_v2: the inclination block becomes 9 m-height ratios, Uniform(0,1), from amplitude_ratio(l, 55 deg)
    k = z + 10
    i = math.radians(55.0)
    exp = [math.cos(i) ** 2, 0.5 * math.sin(i) ** 2,
           0.25 * (3 * math.cos(i) ** 2 - 1) ** 2, 1.5 * math.cos(i) ** 2 * math.sin(i) ** 2, 0.375 * math.sin(i) ** 4]
    assert s.inputs_names[k:k + 5] == ["Inc:H1,0", "Inc:H1,1", "Inc:H2,0", "Inc:H2,1", "Inc:H2,2"]
    assert np.allclose(s.inputs[k:k + 5], exp, rtol=1e-12) and s.inputs_names[k + 5:k + 9] == ["Empty"] * 4
    assert list(s.extra_priors) == [1, 2, 0.2, 1]
    assert "visibility_l3" in s.log


def test_global_id2_projection_and_defaults(tmp_path):
    s = load_global(M88, tmp_path)
    assert (s.model_fullname, s.model_case) == ("model_MS_Global_a1etaa3_HarveyLike", 2)
    assert list(s.plength) == [4, 1, 4, 4, 0, 0, 6, 4, 10, 1, 2] and s.inputs[-2] == 20.0
    q = 4 + 1 + 8
    assert s.inputs_names[q + 3:q + 5] == ["sqrt(splitting_a1).cosi", "sqrt(splitting_a1).sini"]
    assert s.inputs[q + 3] == pytest.approx(math.sqrt(1.4)) and s.inputs[q + 4] == 1e-2     # inclination 0 -> clipped
    assert s.priors[1, q + 3] == pytest.approx(math.sqrt(5.0))
    assert s.inputs_names[q] == "Empty" and s.inputs_names[-3] == "Empty"                    # a1 and inclination slots
    # 7 noise values are right-aligned into the 10 slots (io_ms_global.cpp:226-230): two absent Harveys + N0
    z = q + 6 + 4
    assert list(s.inputs[z:z + 10]) == [0, 0, 1, 0, 0, 1, 0, 0, 1, 0.3268] and list(s.relax[z:z + 10]) == [0] * 9 + [1]
    # quirk kept: without a Width keyword the width slots hold the l=0 HEIGHTS (io_ms_global.cpp:553-559)
    assert s.inputs_names[q + 6] == "Width_l0" and list(s.inputs[q + 6:q + 10]) == list(s.inputs[:4])


def test_model_file_errors(tmp_path):
    txt = open(KPLR).read()
    d = str(tmp_path / "s.data")
    synth_data(d, 2300, 3140)

    def load(text, reader="io_MS_Global"):
        m = str(tmp_path / "m.model")
        open(m, "w").write(text)
        s = Setup(CFG)
        s.set("Modeling", "prior_fct_name", reader)
        return s.load(m, d, 0)

    with pytest.raises(SetupError) as e:                        # two ranges with the global reader
        load(txt.replace("* 2300 3140.000", "* 2300 3140.000\n* 3200 3300"))
    assert e.value.code == IO_E_SYNTAX and "Multiple range" in str(e.value)
    with pytest.raises(SetupError) as e:                        # unknown model name
        load(txt.replace("model_MS_Global_a1etaa3_HarveyLike_Classic_v2", "model_MS_Global_unknown"))
    assert e.value.code == IO_E_NAME
    with pytest.raises(SetupError) as e:                        # duplicated frequency in the relax table
        load(txt.replace("p  0  2442.8799", "p  0  2324.4900"))
    assert e.value.code == IO_E_SYNTAX and "uniqueness" in str(e.value)
    with pytest.raises(SetupError) as e:                        # Fix_Auto is only defined for eta and Width
        load(txt.replace("Splitting_a3                 Fix", "Splitting_a3            Fix_Auto"))
    assert e.value.code == IO_E_SYNTAX
    with pytest.raises(SetupError) as e:
        load(txt, reader="io_nothing")
    assert e.value.code == IO_E_NAME
    with pytest.raises(SetupError) as e:
        Setup(CFG).load(str(tmp_path / "missing.model"), d, 0)
    assert e.value.code == IO_E_OPEN
    # the projected-splitting keywords must come as a pair
    with pytest.raises(SetupError):
        load(txt + "  sqrt(splitting_a1).cosi   Uniform   0.5   0.0   2.0\n")


def test_exports_match_header(accel_mod):
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "tamcmc_io.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tamcmc_[a-z_]+)\s*\(", txt)))
    assert len(names) >= 16
    lib = accel_mod.load_library()
    for n in names:
        assert hasattr(lib, n), n


def test_sampler_starts_from_reference_files(orc):
    """N3 -> N1/N2 on the CPU: the file's initial guesses lie inside the file's priors, and the adaptive sampler
    (oracle evaluator here; the HIP evaluator is compared with it in tests/test_real_inputs_gpu.py) moves."""
    from tamcmc_amd import sampler as S
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 0)
    s.set("MALA", "Nchains", 4)
    s.set("MALA", "Nt_learn", "20, 60, 100000")
    cfg = s.sampler_cfg(seed=99)

    def ev(P, T):
        return orc.generate_batch(s.model_case, s.plength, s.x, s.y, P, T, likelihood_p=s.likelihood_p)[:2]

    smp = S.Sampler(cfg, ev, s.plength, s.inputs, s.relax, s.err, s.priors_names_switch, s.priors, s.extra_priors)
    smp.init()
    lp = smp.get("logPrior")
    assert np.all(np.isfinite(lp)) and np.all(lp == lp[0])
    l0 = smp.get("logL") * smp.get("Tcoefs")
    moved, swaps = smp.run(200)
    assert 0.03 < moved.mean() < 0.9 and np.count_nonzero(swaps >= 0) == 199
    assert np.all(smp.get("logL") * smp.get("Tcoefs") > l0 - 50)      # chains do not wander off the mode
    smp.close()
