"""One phase of MALA::execute (MALA.cpp:608-720) with the tempered chains sharded over processes, one process per
GPU: rank g owns the contiguous block [g n, (g+1) n) of the temperature ladder, evaluates it on its own
device, and the only exchange step is the parallel-tempering attempt on a pair that straddles two ranks (one record
each way between neighbours; RCCL on GPUs, gloo in the CPU tests).  Every rank consumes the whole random stream, so the
chains -- and therefore the files rank 0 writes -- are the ones a single process would have produced."""
import ctypes as C

import numpy as np

from . import outputs as O
from . import sampler as S
from .setup_io import IO_OK, SetupError


def _gather(dist, arr, rank, world, device):
    """All ranks' equally shaped float64 arrays, stacked on rank 0 ([world, ...]); None elsewhere."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    if rank != 0:
        return None
    return np.stack([o.cpu().numpy() for o in out])


def _abort_group(dist):
    """Best effort: close this rank's side of the process group at once (peers blocked on it then fail fast)."""
    try:
        pg = dist.distributed_c10d._get_default_group()
        if hasattr(pg, "abort"):
            pg.abort()
        elif hasattr(pg, "_shutdown"):
            pg._shutdown()
        else:
            dist.destroy_process_group()
    except Exception:       # noqa: BLE001 -- we are already on an error path
        pass


def run_phase_sharded(setup, evaluator, dist, rank, world, seed, device=None, restore_precision=6, progress=None):
    """Restore (if configured) -> init -> iterate to Outputs.Nsamples; rank 0 writes the result and restore files.
    `evaluator`: this rank's Accel (or a callable for CPU tests).  Returns the local Sampler (state after the run)."""
    lib = O._lib()
    cfg = setup.sampler_cfg(seed=seed)
    N = int(cfg.Nchains)
    if N % world != 0:
        raise ValueError(f"Nchains={N} must be a multiple of the number of processes ({world})")
    nloc = N // world
    cfg.chain_offset, cfg.Nchains_local = rank * nloc, nloc
    smp = S.Sampler(cfg, evaluator, setup.plength, setup.inputs, setup.relax, setup.err, setup.priors_names_switch,
                    setup.priors, setup.extra_priors)
    it0 = O.restore_apply(setup, smp)
    smp.init()
    if world > 1:
        # With the NCCL (= RCCL) backend the first call on a group has to involve every rank; the boundary exchange
        # involves two (torch.distributed.batch_isend_irecv's note).  One all-rank reduction opens the communicator.
        import torch
        t0 = torch.zeros(1, dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t0)
    nv = smp.Nvars
    Nsamples, Nbuffer = int(setup.get("Outputs", "Nsamples")), int(setup.get("Outputs", "Nbuffer"))
    out = C.c_void_p()
    if rank == 0:
        T_all = float(cfg.lambda_temp) ** np.arange(N)
        rc = lib.tamcmc_outputs_create(C.byref(out), setup._h, N, T_all.ctypes.data_as(C.POINTER(C.c_double)), it0, int(restore_precision))
        if rc != IO_OK:
            raise SetupError(rc, "tamcmc_outputs_create", "see stderr")
    lib.tamcmc_outputs_push_block.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 15
    cap = min(Nbuffer, max(1, Nsamples - it0))
    block = S.ShardBlock(smp, cap)                 # this rank's share of an output block, filled inside the library
    Pswap_last, swapped_last = 0.0, 0

    def flush():
        nonlocal Pswap_last, swapped_last
        n = block.count()
        g_vars = _gather(dist, block.data("vars"), rank, world, device)
        g_stat = _gather(dist, block.data("stat"), rank, world, device)
        g_moved = _gather(dist, block.data("moved"), rank, world, device)
        g_pt = _gather(dist, block.data("pt"), rank, world, device)
        last = [_gather(dist, smp.get(w), rank, world, device) for w in ("vars", "sigma", "mu", "covarmat")]
        g_sums = [_gather(dist, block.data(w), rank, world, device) for w in ("sum_sigma", "sum_mu", "sum_covar", "sum_vars")]
        block.reset()
        if rank != 0:
            return
        vars_all = np.ascontiguousarray(np.concatenate(list(g_vars), axis=1))                     # [n, N, nv]
        stat_all = np.ascontiguousarray(np.concatenate(list(g_stat), axis=2).reshape(n, 3 * N))   # [n, logL | logPrior | logPost]
        moved_all = np.ascontiguousarray(np.concatenate(list(g_moved), axis=1) != 0, dtype=np.uint8)
        att = np.ascontiguousarray(g_pt[0][:, 0] != 0, dtype=np.uint8)
        chain0 = np.ascontiguousarray(g_pt[0][:, 1], dtype=np.int32)
        Psw, sw = np.empty(n), np.empty(n, dtype=np.uint8)
        for i in range(n):                                     # outcome known to the owner(s) of the pair
            if att[i]:
                r = np.array([g[i, 2] for g in g_pt]); s2 = np.array([g[i, 3] for g in g_pt])
                Pswap_last, swapped_last = float(np.nanmax(r)), int(s2.max())
            Psw[i], sw[i] = Pswap_last, swapped_last          # Model_def::Pswap / ::swaped keep their last values
        cat = lambda g: np.ascontiguousarray(np.concatenate(list(g), axis=0))     # noqa: E731
        l_vars, l_sigma, l_mu, l_cov = (cat(x) for x in last)
        s_sigma, s_mu, s_cov, s_vars = (cat(x) for x in g_sums)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)                                # noqa: E731
        rc = lib.tamcmc_outputs_push_block(out, n, ptr(vars_all), ptr(stat_all), ptr(moved_all), ptr(att), ptr(chain0), ptr(Psw), ptr(sw),
                                           ptr(l_vars), ptr(l_sigma), ptr(l_mu), ptr(l_cov), ptr(s_vars), ptr(s_sigma), ptr(s_mu), ptr(s_cov))
        if rc != IO_OK:
            raise SetupError(rc, "tamcmc_outputs_push_block", lib.tamcmc_outputs_error(out).decode(errors="replace"))

    # the iteration loop runs inside the library (tamcmc_sampler_run_sharded); Python is entered once per boundary-pair
    # exchange and once per block of Nbuffer samples
    exchange = S.neighbour_exchange(dist, nloc, device)
    i = smp.iteration()
    while i < Nsamples:
        if progress is not None and rank == 0:
            progress(i, Nsamples)
        try:
            done, _, _ = smp.run_sharded(min(cap, Nsamples - i), exchange, block=block)
        except Exception:
            # a failed rank cannot answer its neighbours any more (include/tamcmc_sampler.h): take the group down so
            # that their pending send / recv fail and they raise too, instead of waiting for the backend's timeout
            _abort_group(dist)
            raise
        i += done
        flush()
    block.close()
    if rank == 0:
        if progress is not None:
            progress(Nsamples, Nsamples)
        lib.tamcmc_outputs_destroy(out)
    return smp

