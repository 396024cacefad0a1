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


def _worker(rank, world, port, out_dir, NCH, NIT):
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


@pytest.mark.parametrize("world,NCH,NIT", [
    (2, 6, 60),        # the boundary pair comes up often
    (2, 40, 60),       # the other rank's 20 chains are passed over by the generator's jump-ahead
    (8, 256, 240)])    # BASELINE config C3's shape: 8 blocks of 32 chains, 7 boundary pairs among 255 (a handful of attempts
                       # in 240 iterations), interior ranks that own neither end of most attempts
def test_sharded_sampler_equals_single_process(tmp_path, world, NCH, NIT):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), NCH, NIT), nprocs=world, join=True)
    single = _make(0, NCH, NCH)
    moved, swaps = single.run(NIT)
    rr = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for key in ("vars", "logL", "logPost", "sigma"):
        assert np.array_equal(np.concatenate([r[key] for r in rr]), single.get(key)), key
    # every pair that straddles two ranks was attempted by both owners with the same outcome
    per = NCH // world
    seen = 0
    for r in range(world - 1):
        bp = per * (r + 1) - 1
        b0 = {tuple(v) for v in rr[r]["swaps"] if v[0] == bp}
        b1 = {tuple(v) for v in rr[r + 1]["swaps"] if v[0] == bp}
        assert b0 == b1
        seen += len(b0)
    assert seen > 0 or NCH == 40
    ref = [(int(v) // 2, int(v) % 2) for v in swaps if v >= 0]
    got = sorted({(int(a), int(sw)) for a, sw in np.concatenate([r["swaps"] for r in rr])})
    assert got == sorted(set(ref))
    assert any(sw for _, sw in ref)
