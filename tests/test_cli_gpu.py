"""The command-line tools on the GPU: cpptamcmc_hip runs the reference's three-phase recipe (Burn-in -> Learning ->
Acquire, Config/config_presets.cfg) on the reference's own spectrum through the HIP hot path; getmodel_hip rebuilds
model spectra on a .data grid."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tamcmc_amd import outputs as O
from tamcmc_amd.setup_io import Setup

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "ref_inputs")
CFG = os.path.join(G, "Config_default")


def test_three_phase_run_from_the_reference_files(tmp_path):
    exe = os.path.join(ROOT, "bin", "cpptamcmc_hip")
    root = tmp_path / "run"
    shutil.copytree(CFG, root / "Config" / "default")
    cfg = open(root / "Config" / "default" / "config_default.cfg").read()
    cfg = cfg.replace("Nchains=10;", "Nchains=6;").replace("Nbuffer=10000;", "Nbuffer=250;").replace("Nt_learn=500, 1500, 100000;", "Nt_learn=50, 150, 100000;")
    open(root / "Config" / "default" / "config_default.cfg", "w").write(cfg)
    open(root / "Config" / "config_presets.cfg", "w").write(f"""
   force_manual_config=0;
   manual_config_file=;
   cfg_models_dir={G}/;
   cfg_out_dir={tmp_path}/out;
   processing      = Burn-in  , Learning , Acquire;
   Nsamples        = 600     ,  400  , 500;
   c0              = 1.8      ,   1.7   ,    0;
   restore         =  0       ,    1    ,    2;
   core_out        =  B       ,    L    ,    A;
   core_in         =  B       ,    B    ,    L;
   start_index_processing=0;
   last_index_processing=2;
   table_ids=1, 2;
TF_3443483_local-v3   1;
/END;
""")
    # slice arguments as in main.cpp:56-61,115-126: first is 1-based inclusive, last is exclusive after the -1 shift,
    # so "2 4" runs the slices numbered 2 and 3
    r = subprocess.run([exe, "execute", "1", "1", "1", "2", "4", "--root", str(root), "--seed", "42", "--quiet"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    outd = tmp_path / "out" / "TF_3443483_local-v3"
    assert sorted(os.listdir(outd)) == ["diags", "outputs", "restore"]
    for sl in (2, 3):                                             # slices 2 and 3 (1-based), file names carry the slice
        s = Setup(CFG).load(os.path.join(G, "TF_3443483_local-v3.model"), os.path.join(G, "TF_3443483_local-v3.data"), sl - 1)
        for core, n in (("B", 600), ("L", 400), ("A", 500)):
            rootname = str(outd / "outputs" / f"TF_3443483_local-v3_{sl}_{core}_")
            v, h = O.read_params_bin(rootname + "params", 0)
            assert v.shape == (n, s.Nvars) and int(h["Nchains"]) == 6 and np.all(np.isfinite(v))
            logL, logP, logPost, _ = O.read_stat_criteria_bin(rootname + "stat_criteria")
            assert logL.shape == (n, 6) and np.all(np.isfinite(logL[:, 0]))
            pt, _ = O.read_parallel_tempering_bin(rootname + "parallel_tempering")
            assert pt.shape == (n,) and (np.all(pt["attempt"] == 0) if core == "L" else pt["attempt"].sum() == n - 1)
            for k in (1, 2, 3):
                assert os.path.exists(outd / "restore" / f"TF_3443483_local-v3_{sl}_restore_{core}_{k}.dat")
        # the cold chain of the Acquire phase stays inside the file's frequency windows and beats the starting point
        off = s.plength[0] + s.plength[1]
        nf = s.plength[2:6].sum()
        fidx = [int(np.flatnonzero(s.index_to_relax == off + k)[0]) for k in range(nf) if s.relax[off + k] == 1]
        lo, hi = s.priors[0, off:off + nf][s.relax[off:off + nf] == 1], s.priors[1, off:off + nf][s.relax[off:off + nf] == 1]
        assert np.all((v[:, fidx] > lo - 1.0) & (v[:, fidx] < hi + 1.0))
        first, _ = O.read_stat_criteria_bin(str(outd / "outputs" / f"TF_3443483_local-v3_{sl}_B_") + "stat_criteria")[:2]
        assert np.median(logL[:, 0]) > first[0, 0]


def test_getmodel_tool(tmp_path, orc):
    exe = os.path.join(ROOT, "bin", "getmodel_hip")
    s = Setup(CFG).load(os.path.join(G, "TF_3443483_local-v3.model"), os.path.join(G, "TF_3443483_local-v3.data"), 0)
    # a small .data file (the tool takes the whole file, no cropping): slice 1 of the spectrum
    d = tmp_path / "slice.data"
    with open(d, "w") as f:
        f.write("# slice 1\n! frequency power\n* (microHz) (ppm^2/microHz)\n")
        for a, b in zip(s.x, s.y):
            f.write("%.8f %.8f\n" % (a, b))
    rows = np.vstack([s.inputs, s.inputs * np.where(s.relax == 1, 1.01, 1.0)])
    p = tmp_path / "params.txt"
    with open(p, "w") as f:
        f.write("# plength then parameter rows\n" + " ".join(str(v) for v in s.plength) + "\n")
        for r in rows:
            f.write(" ".join("%.17g" % v for v in r) + "\n")
    out = tmp_path / "model.ascii"
    env = dict(os.environ, TAMCMC_MODELS_LIST=os.path.join(CFG, "models_ctrl.list"))
    r = subprocess.run([exe, str(d), str(p), "model_MS_local_basic", str(out)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    m = np.loadtxt(out)
    x = np.array([float("%.8f" % v) for v in s.x])
    assert m.shape == (s.Nx, 4) and np.allclose(m[:, 0], x, rtol=1e-11)
    for k in range(2):
        ref = orc.model(s.model_case, rows[k], s.plength, x)[0]
        assert np.allclose(m[:, 2 + k], ref, rtol=1e-10)           # setprecision(12) in the file
    r = subprocess.run([exe, str(d), str(p), "model_unknown", str(out)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "Unknown model name" in r.stderr
