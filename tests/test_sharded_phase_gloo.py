"""Sharded phase driver on CPU (two gloo ranks, oracle evaluator): started from the reference's own input files, the
files rank 0 writes -- samples, statistics, parallel-tempering log, acceptance, restore files -- must be byte-identical
to the ones the single-process driver writes, for a fresh run and for a restored one (tamcmc-c-_amd/sharded.py;
RCCL replaces gloo on the GPU box)."""
import filecmp
import os
import socket
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden", "ref_inputs")


def _setup(out, phase, Nsamples, restore):
    from tamcmc_amd.setup_io import Setup
    s = Setup(os.path.join(G, "Config_default")).load(os.path.join(G, "TF_3443483_local-v3.model"),
                                                      os.path.join(G, "TF_3443483_local-v3.data"), 1)
    s.set("MALA", "Nchains", 6)
    s.set("MALA", "Nt_learn", "10, 30, 100000")
    s.set("Outputs", "output_dir", out)
    s.set("Outputs", "restore_dir", out)
    s.set("Outputs", "output_root_name", f"TF_{phase[0]}_")
    s.set("Outputs", "restore_file_out", f"TF_restore_{phase[0]}_")
    s.set("Outputs", "Nbuffer", 25)
    s.apply_phase(phase, Nsamples, 1.8)
    if restore:
        s.set("Outputs", "restore_file_in", "TF_restore_B_")
        s.set("Outputs", "do_restore_variables", 1)
        s.set("Outputs", "do_restore_proposal", 1)
    return s


def _evaluator(s):
    from oracle import pyoracle as orc

    def ev(P, T):
        return orc.generate_batch(s.model_case, s.plength, s.x, s.y, P, T, likelihood_p=s.likelihood_p)[:2]
    return ev


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tamcmc_amd import sharded
    for phase, n, restore in (("Burn-in", 60, False), ("Acquire", 40, True)):
        s = _setup(out, phase, n, restore)
        sharded.run_phase_sharded(s, _evaluator(s), dist, rank, world, seed=11, restore_precision=17)
        dist.barrier()
    dist.destroy_process_group()


def test_two_rank_phase_files_equal_single_process(tmp_path):
    from tamcmc_amd import outputs as O
    from tamcmc_amd import sampler as S
    a, b = str(tmp_path / "sharded") + "/", str(tmp_path / "single") + "/"
    os.makedirs(a); os.makedirs(b)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_worker, args=(2, port, a), nprocs=2, join=True)
    for phase, n, restore in (("Burn-in", 60, False), ("Acquire", 40, True)):
        s = _setup(b, phase, n, restore)
        smp = S.Sampler(s.sampler_cfg(seed=11), _evaluator(s), s.plength, s.inputs, s.relax, s.err, s.priors_names_switch, s.priors,
                        s.extra_priors)
        O.run_phase(s, smp, restore_precision=17)
    names = sorted(os.listdir(b))
    assert names == sorted(os.listdir(a)) and len(names) == 2 * (6 + 1 + 2 + 2 + 1 + 3)     # chains, hdr, stat, pt, acceptance, restore
    for f in names:
        assert filecmp.cmp(a + f, b + f, shallow=False), f
