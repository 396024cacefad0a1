"""Sharded sampler on CPU: two gloo ranks, each owning half of the temperature ladder, must reproduce the
single-process run bit for bit -- the random stream is replicated, chains are independent within an
iteration, and a parallel-tempering pair that straddles the ranks is exchanged with one send/recv each way
(sampler.run_sharded; RCCL on the GPU box)."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NIT = 60


def _setup():
    import test_priors_sampler as tps
    from oracle import pyoracle as orc
    from tamcmc_amd import synth
    w, sw, pp, b = tps.ms_global_prior_setup()
    w = dict(w); w["x"] = synth.grid(900, 2300.0, 840.0 / 900)
    m, _ = orc.model(3, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=5)
    return w, sw, pp, y, orc, tps


def _make(offset, nloc, NCH):
    from tamcmc_amd import sampler as S
    w, sw, pp, y, orc, tps = _setup()
    cfg = S.default_cfg(NCH, chain_offset=offset, Nchains_local=nloc, seed=99, Nt_learn=(10, 40, 100000),
                        periods_learn=(1, 1), prior_fct_switch=2, dN_mixing=1)
    smp = S.Sampler(cfg, tps.oracle_evaluator(orc, 3, w, y), w["plength"], w["params_true"], w["relax"], w["err"], sw, pp,
                    [1.0, 5.0, 0.5, 0.0])
    smp.init()
    return smp


def _worker(rank, world, port, out_dir, NCH):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tamcmc_amd import sampler as S
    per = NCH // world
    smp = _make(rank * per, per, NCH)
    swaps = S.run_sharded(smp, NIT, dist, rank, world, per)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), vars=smp.get("vars"), logL=smp.get("logL"), logPost=smp.get("logPost"),
             sigma=smp.get("sigma"), swaps=np.array([[a, int(s)] for a, s in swaps]).reshape(-1, 2))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("NCH", [6, 40])     # 40: the other rank's 20 chains are passed over by the generator's jump-ahead
def test_two_rank_sampler_equals_single_process(tmp_path, NCH):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), NCH), nprocs=2, join=True)
    single = _make(0, NCH, NCH)
    moved, swaps = single.run(NIT)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for key in ("vars", "logL", "logPost", "sigma"):
        assert np.array_equal(np.concatenate([r0[key], r1[key]]), single.get(key)), key
    # the boundary pair was attempted by both ranks with the same outcome (6 chains: and it did come up)
    bp = NCH // 2 - 1
    b0 = {tuple(r) for r in r0["swaps"] if r[0] == bp}
    b1 = {tuple(r) for r in r1["swaps"] if r[0] == bp}
    assert b0 == b1 and (len(b0) > 0 or NCH > 6)
    ref = [(int(v) // 2, int(v) % 2) for v in swaps if v >= 0]
    got = sorted({(int(a), int(sw)) for a, sw in np.concatenate([r0["swaps"], r1["swaps"]])})
    assert got == sorted(set(ref))
    assert any(sw for _, sw in ref)
