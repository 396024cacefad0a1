"""The sharded phase driver with the HIP evaluator: two processes (both on the one GPU of the test box, gloo between
them; on the 8-GPU node each process has its own GPU and the exchange is RCCL) against the single-process driver.
Byte-identical files need, besides the replicated random stream, that a chain's log-likelihood does not depend on how
many chains share its batch: 3 chains per rank here, 6 in the single process."""
import filecmp
import os
import subprocess
import sys

import pytest

from tamcmc_amd import outputs as O
from tamcmc_amd import sampler as S
from tamcmc_amd.setup_io import Setup

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "ref_inputs")


def test_two_ranks_on_one_gpu_equal_single_process(accel_mod, tmp_path):
    a, b = str(tmp_path / "sharded") + "/", str(tmp_path / "single") + "/"
    os.makedirs(b)
    common = ["--config-dir", os.path.join(G, "Config_default"), "--model", os.path.join(G, "TF_3443483_local-v3.model"),
              "--data", os.path.join(G, "TF_3443483_local-v3.data"), "--slice", "3", "--nchains", "6", "--nbuffer", "50",
              "--seed", "21", "--root-name", "TF_", "--backend", "gloo", "--single-device"]
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29577", os.path.join(ROOT, "tools", "run_sharded.py"), "--out-dir", a]
    for extra in (["--phase", "Burn-in", "--nsamples", "120"], ["--phase", "Acquire", "--nsamples", "80", "--restore", "2", "--restore-from", "B"]):
        r = subprocess.run(launch + common + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    for phase, n, restore in (("Burn-in", 120, 0), ("Acquire", 80, 2)):
        s = Setup(os.path.join(G, "Config_default"))
        s.set("MALA", "Nchains", 6)
        s.set("Outputs", "Nbuffer", 50)
        s.load(os.path.join(G, "TF_3443483_local-v3.model"), os.path.join(G, "TF_3443483_local-v3.data"), 3)
        s.set("Outputs", "output_dir", b)
        s.set("Outputs", "restore_dir", b)
        s.set("Outputs", "output_root_name", f"TF_{phase[0]}_")
        s.set("Outputs", "restore_file_out", f"TF_restore_{phase[0]}_")
        s.apply_phase(phase, n, 1.8)
        if restore:
            s.set("Outputs", "restore_file_in", "TF_restore_B_")
            s.set("Outputs", "do_restore_variables", 1)
            s.set("Outputs", "do_restore_proposal", 1)
        with accel_mod.Accel(s.model_case, s.plength, s.x, s.y, sigma_y=s.sigma_y) as acc:
            smp = S.Sampler(s.sampler_cfg(seed=21), acc, s.plength, s.inputs, s.relax, s.err, s.priors_names_switch, s.priors,
                            s.extra_priors)
            O.run_phase(s, smp, restore_precision=17)
            smp.close()
    names = sorted(os.listdir(b))
    assert names == sorted(os.listdir(a)) and len(names) == 30
    for f in names:
        assert filecmp.cmp(a + f, b + f, shallow=False), f
