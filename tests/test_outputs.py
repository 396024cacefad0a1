"""N4: result / restore files in the reference's formats, restore round trip, the phase driver and the CLI's
configuration pass (include/tamcmc_outputs.h, tools/cpptamcmc_hip.cpp).  CPU only: the evaluator is the oracle
(the HIP evaluator drives the same code in tests/test_cli_gpu.py).  The reference ships no output file to compare
against; formats are checked against the layouts its own readers expect (Diagnostics::read_params_header,
tools/bin2txt_params.cpp, tools/read_stats.cpp, Config::read_restore_files)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tamcmc_amd import outputs as O
from tamcmc_amd import sampler as S
from tamcmc_amd.setup_io import Setup, SetupError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "ref_inputs")
CFG = os.path.join(G, "Config_default")
TF_MODEL = os.path.join(G, "TF_3443483_local-v3.model")
TF_DATA = os.path.join(G, "TF_3443483_local-v3.data")


def make_setup(out, Nsamples=90, Nbuffer=40, nchains=4, phase="Burn-in", fmt="binary", root="TF_B_"):
    s = Setup(CFG).load(TF_MODEL, TF_DATA, 0)
    s.set("MALA", "Nchains", nchains)
    s.set("Outputs", "output_dir", out)
    s.set("Outputs", "restore_dir", out)
    s.set("Outputs", "output_root_name", root)
    s.set("Outputs", "Nbuffer", Nbuffer)
    s.set("Outputs", "file_format", fmt)
    s.set("MALA", "Nt_learn", "10, 30, 100000")
    s.apply_phase(phase, Nsamples, 1.8)
    return s


def make_sampler(s, orc, seed=5):
    def ev(P, T):
        return orc.generate_batch(s.model_case, s.plength, s.x, s.y, P, T, likelihood_p=s.likelihood_p)[:2]
    return S.Sampler(s.sampler_cfg(seed=seed), ev, s.plength, s.inputs, s.relax, s.err, s.priors_names_switch, s.priors,
                     s.extra_priors)


def test_binary_outputs_equal_a_step_by_step_run(orc, tmp_path):
    out = str(tmp_path) + "/"
    s = make_setup(out)
    smp = make_sampler(s, orc)
    seen = []
    O.run_phase(s, smp, progress=lambda i, n, u: seen.append((i, n)))
    assert seen == [(0, 90), (40, 90), (80, 90), (90, 90)]
    # the same chain driven from Python, one call per step
    ref = make_sampler(s, orc)
    ref.init()
    V, L, PT, MV = [], [], [], []
    Pswap, swapped = 0.0, False
    for _ in range(90):
        ref.mh_step()
        att, A = 0, -1
        if ref.pt_due():
            A, u = ref.pt_draw()
            swapped, Pswap = ref.pt_local(A, u)
            att = 1
        V.append(ref.get("vars")); L.append(np.concatenate([ref.get("logL"), ref.get("logPrior"), ref.get("logPost")]))
        PT.append((att, A, Pswap, int(swapped)))
        ref.end_iteration()
    V, L = np.array(V), np.array(L)
    for c in range(4):
        v, h = O.read_params_bin(out + "TF_B_params", c)
        assert v.shape == (90, s.Nvars) and np.array_equal(v, V[:, c, :])
    # header fields the reference's readers parse (diagnostics.cpp:809-920)
    assert (int(h["Nsamples"]), int(h["Nchains"]), int(h["Nvars"]), int(h["Ncons"])) == (90, 4, s.Nvars, s.Nparams - s.Nvars)
    assert [int(v) for v in h["relax"].split()] == list(s.relax) and [int(v) for v in h["plength"].split()] == list(s.plength)
    assert h["variable_names"].split() == [n for n, r in zip(s.inputs_names, s.relax) if r == 1]
    cons = [float(v) for v in h["constant_values"].split()]
    assert np.allclose(cons, s.inputs[s.relax == 0], rtol=1e-5) and int(h["Nsamples_done"]) == 90
    a, b, c3, h2 = O.read_stat_criteria_bin(out + "TF_B_stat_criteria")
    assert a.shape == (90, 4) and np.array_equal(np.hstack([a, b, c3]), L)
    # logPost == logL + logPrior except where the reference's swap quirk applies (MALA.cpp:415-431, kept)
    assert np.mean(np.isclose(a + b, c3)) > 0.5
    assert h2["labels"].split()[:2] == ["logLikelihood[0]", "logLikelihood[1]"] and h2["labels"].split()[-1] == "logPosteriors[3]"
    pt, h3 = O.read_parallel_tempering_bin(out + "TF_B_parallel_tempering")
    assert os.path.getsize(out + "TF_B_parallel_tempering.bin") == 90 * 14              # bool,int,double,bool (outputs.cpp:1383)
    assert [(int(r["attempt"]), int(r["chain0"]), float(r["Pswitch"]), int(r["switched"])) for r in pt] == PT
    assert pt["attempt"][0] == 0 and pt["chain0"][0] == -1 and np.all(pt["attempt"][1:] == 1)   # i % dN_mixing == 0 && i != 0
    assert np.allclose([float(v) for v in h3["Tcoefs"].split()], 1.7 ** np.arange(4), rtol=1e-5)
    # acceptance.txt: one line per flushed block, x = (Ncopy + 0.5) Nbuffer (outputs.cpp:1836)
    rows = [ln.split() for ln in open(out + "TF_B_acceptance.txt") if ln[0] not in "#!"]
    assert [float(r[0]) for r in rows] == [20.0, 60.0, 100.0] and all(len(r) == 5 for r in rows)
    moved = np.abs(np.diff(np.concatenate([np.tile(s.inputs[s.index_to_relax], (1, 4, 1)), V]), axis=0)).sum(axis=2) > 0
    blocks = [moved[0:40], moved[40:80], moved[80:90]]
    # a swap also changes a chain's variables, so only bound the MH acceptance from above by the change rate
    for r, blk in zip(rows, blocks):
        assert np.all(np.array([float(v) for v in r[1:]]) <= blk.mean(axis=0) + 1e-12)


def test_text_and_debug_formats(orc, tmp_path):
    out = str(tmp_path) + "/"
    s = make_setup(out, Nsamples=30, Nbuffer=20, nchains=3, fmt="debug")
    O.run_phase(s, make_sampler(s, orc))
    vb, _ = O.read_params_bin(out + "TF_B_params", 1) if os.path.exists(out + "TF_B_params_chain-1.bin") else (None, None)
    vb = np.fromfile(out + "TF_B_params_chain-1.dbg", dtype="<f8").reshape(-1, s.Nvars)   # debug: binary with the .dbg extension
    txt = [ln for ln in open(out + "TF_B_params_chain-1.dbg.txt") if ln[0] not in "#!"]
    assert len(txt) == 30 and np.allclose(np.array([[float(v) for v in ln.split()] for ln in txt]), vb, rtol=1e-5)
    hdr = [ln for ln in open(out + "TF_B_params_chain-1.dbg.txt") if ln[0] == "!"]
    assert any(ln.startswith("! chain= 1") for ln in hdr)                              # outputs.cpp:553
    st = [ln.split() for ln in open(out + "TF_B_stat_criteria.dbg.txt") if ln[0] not in "#!"]
    assert len(st) == 30 and len(st[0]) == 9
    pt = [ln.split() for ln in open(out + "TF_B_parallel_tempering.dbg.txt") if ln[0] not in "#!"]
    assert len(pt) == 30 and pt[0][:2] == ["0", "-1"]
    s2 = make_setup(out, Nsamples=25, Nbuffer=50, nchains=3, fmt="text", root="TXT_")
    O.run_phase(s2, make_sampler(s2, orc))
    assert sum(1 for ln in open(out + "TXT_params_chain-0.txt") if ln[0] not in "#!") == 25
    s3 = make_setup(out, fmt="hdf5")
    with pytest.raises(SetupError):
        O.run_phase(s3, make_sampler(s3, orc))


def test_restore_round_trip_and_phase_chain(orc, tmp_path):
    out = str(tmp_path) + "/"
    # Burn-in, full precision restore files
    s = make_setup(out, Nsamples=60, Nbuffer=25)
    s.set("Outputs", "restore_file_out", "TF_restore_B_")
    smp = make_sampler(s, orc)
    O.run_phase(s, smp, restore_precision=17)
    state = {k: smp.get(k) for k in ("vars", "sigma", "mu", "covarmat")}
    txt = open(out + "TF_restore_B_1.dat").read()
    assert "! Nchains= 4" in txt and "! iteration=59" in txt and "! vars= " in txt and "! vars_mean= " in txt
    assert open(out + "TF_restore_B_3.dat").read().count("*0\n") == 2                  # covarmats + covarmats_mean
    # Learning: restore = 1 (variables only), new proposal
    s = make_setup(out, Nsamples=40, Nbuffer=25, phase="Learning", root="TF_L_")
    s.set("Outputs", "restore_file_in", "TF_restore_B_")
    s.set("Outputs", "restore_file_out", "TF_restore_L_")
    s.set("Outputs", "do_restore_variables", 1)
    smp = make_sampler(s, orc, seed=6)
    assert O.restore_apply(s, smp) == 0
    assert np.array_equal(smp.get("vars"), state["vars"]) and not np.array_equal(smp.get("sigma"), state["sigma"])
    assert np.array_equal(smp.get("params")[:, s.index_to_relax], state["vars"])
    O.run_phase(s, smp, restore_precision=17)
    pt, _ = O.read_parallel_tempering_bin(out + "TF_L_parallel_tempering")
    assert np.all(pt["attempt"] == 0)                                                   # Learning never mixes the chains
    # Acquire: restore = 2 (variables + proposal) from the Burn-in files, exact to the last bit at precision 17
    s = make_setup(out, Nsamples=30, Nbuffer=25, phase="Acquire", root="TF_A_")
    s.set("Outputs", "restore_file_in", "TF_restore_B_")
    s.set("Outputs", "do_restore_variables", 1)
    s.set("Outputs", "do_restore_proposal", 1)
    smp = make_sampler(s, orc, seed=7)
    assert O.restore_apply(s, smp) == 0
    for k in state:
        assert np.array_equal(smp.get(k), state[k]), k
    sig0 = smp.get("sigma").copy()
    O.run_phase(s, smp)
    assert np.array_equal(smp.get("sigma"), sig0)                                       # Acquire never learns
    # restore = 3: append to the Burn-in files from its last index
    s = make_setup(out, Nsamples=80, Nbuffer=25)
    s.set("Outputs", "restore_file_in", "TF_restore_B_")
    s.set("Outputs", "restore_file_out", "TF_restore_B_")
    for k in ("do_restore_variables", "do_restore_proposal", "do_restore_last_index"):
        s.set("Outputs", k, 1)
    s.set("Outputs", "erase_old_files", 0)
    before, _ = O.read_params_bin(out + "TF_B_params", 0)
    smp = make_sampler(s, orc, seed=8)
    O.run_phase(s, smp)
    after, _ = O.read_params_bin(out + "TF_B_params", 0)
    assert smp.iteration() == 80 and after.shape[0] == 60 + (80 - 59) and np.array_equal(after[:60], before)
    # default precision (6 digits, like the reference's streams) parses back to 6 digits
    s = make_setup(out, Nsamples=20, Nbuffer=25, root="P6_")
    s.set("Outputs", "restore_file_out", "P6_restore_")
    smp = make_sampler(s, orc)
    O.run_phase(s, smp)
    s.set("Outputs", "restore_file_in", "P6_restore_")
    s.set("Outputs", "do_restore_variables", 1)
    smp2 = make_sampler(s, orc)
    O.restore_apply(s, smp2)
    assert np.allclose(smp2.get("vars"), smp.get("vars"), rtol=1e-5) and not np.array_equal(smp2.get("vars"), smp.get("vars"))
    # inconsistent files are refused (model_def.cpp:103-116)
    s.set("MALA", "Nchains", 5)
    with pytest.raises(SetupError):
        O.restore_apply(s, make_sampler(s, orc))
    s.set("Outputs", "do_restore_last_index", 1)
    s.set("Outputs", "do_restore_proposal", 0)
    with pytest.raises(SetupError):                                                     # config.cpp:176-182
        O.restore_apply(s, smp2)


def test_cli_reads_the_reference_configuration(tmp_path):
    exe = os.path.join(ROOT, "bin", "cpptamcmc_hip")
    assert os.path.exists(exe), "build first: make -C tamcmc-c-_amd/csrc"
    root = tmp_path / "run"
    shutil.copytree(CFG, root / "Config" / "default")
    presets = open("/dev/null").read()
    presets = f"""# presets
   force_manual_config=0;
   manual_config_file=;
   cfg_models_dir={G}/;
   cfg_out_dir={tmp_path}/out;
   processing      = Burn-in  , Learning , Acquire;
   Nsamples        = 50     ,  50  , 50;
   c0              = 1.8      ,   1.7   ,    0;
   restore         =  0       ,    1    ,    2;
   core_out        =  B       ,    L    ,    A;
   core_in         =  B       ,    B    ,    L;
   start_index_processing=0;
   last_index_processing=2;
   table_ids=1, 2;
TF_3443483_local-v3   1;
/END;
"""
    open(root / "Config" / "config_presets.cfg", "w").write(presets)
    r = subprocess.run([exe, "execute", "0", "--root", str(root)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "execute=0" in r.stdout
    r = subprocess.run([exe, "version"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "tamcmc_accel" in r.stdout
    r = subprocess.run([exe, "frobnicate"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "To execute" in r.stderr
    open(root / "Config" / "config_presets.cfg", "w").write(presets.replace("   restore         =  0       ,    1    ,    2;\n", ""))
    r = subprocess.run([exe, "execute", "0", "--root", str(root)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "incorrect number of keywords" in r.stderr


def test_exports_match_header(accel_mod):
    import re
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tamcmc_outputs.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tamcmc_[a-z_]+)\s*\(", txt)) - {"tamcmc_progress_fn"})
    assert len(names) == 8
    lib = accel_mod.load_library()
    for n in names:
        assert hasattr(lib, n), n
