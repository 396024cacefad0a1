"""N>1 path on CPU: world_size-2 gloo processes.  Each rank evaluates its block of the temperature
ladder (the oracle stands in for the GPU evaluator -- this is a test), and the results, the bench
contract's max-over-ranks timing and the parallel-tempering boundary exchange are checked against the
single-process computation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import workloads as W
    from oracle import pyoracle as orc
    from tamcmc_amd import shard, synth

    n_per = 3
    w = W.make(2, Nx=1500)
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=31)
    P_all = W.perturbed(w, n_per * world, scale=0.003, seed=2)
    T_all = synth.temperatures(n_per * world, Tmax=20.0)
    sl = shard.chain_slice(rank, world, n_per)

    calls = {"n": 0}
    res = {}

    def step():
        res["logL"], res["st"] = orc.generate_batch(2, w["plength"], w["x"], y, P_all[sl], T_all[sl], nthreads=1)
        calls["n"] += 1

    dt = shard.timed_loop(step, 3, lambda: None, dist=dist)
    assert calls["n"] == 3 and dt > 0

    # gather the shards (test-only collective; the product path has none)
    mine = torch.from_numpy(res["logL"].copy())
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = torch.cat(parts).numpy()

    # PT swaps: one intra-rank pair, then the boundary pair, forced to be accepted (u = 0) and rejected (u = 2)
    rows = torch.from_numpy(P_all[sl].copy())
    logL = torch.from_numpy(res["logL"].copy())
    extras = torch.arange(n_per * 2, dtype=torch.float64).reshape(n_per, 2) + 100 * rank
    log = []
    for (A, u) in ((0, 0.0), (n_per - 1, 0.0), (n_per - 1, 2.0), (n_per, 0.0)):
        sw, r = shard.pt_swap_sharded(dist, rank, world, n_per, A, u, T_all, rows, logL, extras)
        log.append((A, u, sw, r))
        dist.barrier()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), full=full, rows=rows.numpy(), logL=logL.numpy(),
             extras=extras.numpy(), dt=dt, log=np.array([[a, u, -1 if s is None else int(s), -1.0 if r is None else r] for a, u, s, r in log]))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_pt_exchange(tmp_path, orc):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import workloads as W
    from tamcmc_amd import shard, synth
    n_per = 3
    w = W.make(2, Nx=1500)
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=31)
    P = W.perturbed(w, n_per * world, scale=0.003, seed=2)
    T = synth.temperatures(n_per * world, Tmax=20.0)
    ref, _ = orc.generate_batch(2, w["plength"], w["x"], y, P, T)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # 1. the union of the shards is the single-process result, on every rank; both ranks report one (max) time
    assert np.array_equal(r0["full"], ref) and np.array_equal(r1["full"], ref)
    assert r0["dt"] == r1["dt"]
    # 2. replay the four swap attempts in one process
    rows, L = P.copy(), ref.copy()
    ex = np.concatenate([np.arange(6.0).reshape(3, 2), np.arange(6.0).reshape(3, 2) + 100])
    for (A, u) in ((0, 0.0), (2, 0.0), (2, 2.0), (3, 0.0)):
        r = shard.pt_swap_probability(L[A], L[A + 1], T[A], T[A + 1])
        if u <= r:
            rows[[A, A + 1]] = rows[[A + 1, A]]
            ex[[A, A + 1]] = ex[[A + 1, A]]
            L[A], L[A + 1] = L[A + 1] * T[A + 1] / T[A], L[A] * T[A] / T[A + 1]
    got_rows = np.concatenate([r0["rows"], r1["rows"]])
    got_L = np.concatenate([r0["logL"], r1["logL"]])
    got_ex = np.concatenate([r0["extras"], r1["extras"]])
    assert np.array_equal(got_rows, rows) and np.array_equal(got_ex, ex)
    assert np.allclose(got_L, L, rtol=1e-15)
    # 3. who took part: pair (0,1) only rank 0; the boundary pair (2,3) both; pair (3,4) only rank 1
    assert r0["log"][0][2] == 1 and r1["log"][0][2] == -1
    assert r0["log"][1][2] == 1 and r1["log"][1][2] == 1 and r0["log"][1][3] == r1["log"][1][3]
    assert r0["log"][2][2] == 0 and r1["log"][2][2] == 0
    assert r0["log"][3][2] == -1 and r1["log"][3][2] == 1


def test_chain_slices_partition_the_ladder():
    from tamcmc_amd import shard
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            sl = shard.chain_slice(r, world, 32)
            seen += list(range(sl.start, sl.stop))
        assert seen == list(range(32 * world))
    with pytest.raises(ValueError):
        shard.chain_slice(8, 8, 32)
