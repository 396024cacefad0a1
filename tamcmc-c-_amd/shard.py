"""Sharding of the parallel-tempered chains across ranks (one process per GPU) -- SURVEY.md section 8e.

Chains are independent within an iteration, so the evaluation needs NO data-path collective: rank g owns
the contiguous block [g*n, (g+1)*n) of the temperature ladder (contiguous so that every PT pair (A, A+1)
is intra-rank except the world-1 boundary pairs), x/y are replicated.  The only exchange step of the
path's caller is the parallel-tempering swap of a boundary pair (MALA.cpp:381-445): a neighbour
send/recv of one params row + a few scalars, implemented here on torch.distributed (backend "nccl" is
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import math
import time

import numpy as np


def chain_slice(rank, world, chains_per_rank):
    """Contiguous block of the global temperature ladder owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return slice(rank * chains_per_rank, (rank + 1) * chains_per_rank)


def owner_of(chain, chains_per_rank):
    return chain // chains_per_rank


def timed_loop(step_fn, steps, sync_fn, dist=None, device=None):
    """The bench contract's timed region: barrier + sync on both sides, EXACTLY `steps` calls of step_fn,
    and the MAX over ranks of the elapsed time."""
    import torch
    sync_fn()
    if dist is not None:
        dist.barrier()
        sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
        sync_fn()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def pt_swap_probability(L_A, L_B, T_A, T_B):
    """MALA.cpp:393-397 on TEMPERED log-likelihoods: r_T = min(1, exp(L_A T_A/T_B + L_B T_B/T_A - L_A - L_B))."""
    e = L_A * T_A / T_B + L_B * T_B / T_A - L_A - L_B
    return 1.0 if e >= 0 else math.exp(e)


def pt_swap_sharded(dist, rank, world, chains_per_rank, A, u, T, rows, logL, extras=None, device=None):
    """Attempt the swap of global chains (A, A+1) given the uniform draw `u` every rank agrees on.

    rows   : torch tensor (chains_per_rank, Nparams) of this rank -- swapped in place
    logL   : torch tensor (chains_per_rank,) tempered log-likelihoods -- rescaled in place like
             MALA.cpp:416,426 (the chain keeps its slot's temperature, the state moves)
    extras : optional torch tensor (chains_per_rank, K) of per-chain scalars that travel with the state
             (logPrior, Pmove, ...)
    T      : the GLOBAL temperature ladder (numpy)
    Returns (swapped: bool, r_T: float) on the ranks that own A or A+1, (None, None) elsewhere.
    Intra-rank pairs need no communication; a boundary pair is one send/recv each way between neighbours.
    """
    import torch
    B = A + 1
    rA, rB = owner_of(A, chains_per_rank), owner_of(B, chains_per_rank)
    if rank != rA and rank != rB:
        return None, None
    T_A, T_B = float(T[A]), float(T[B])
    if rA == rB:
        a, b = A - rank * chains_per_rank, B - rank * chains_per_rank
        L_A, L_B = float(logL[a]), float(logL[b])
        r = pt_swap_probability(L_A, L_B, T_A, T_B)
        if u <= r:
            tmp = rows[a].clone(); rows[a] = rows[b]; rows[b] = tmp
            logL[a] = L_B * T_B / T_A
            logL[b] = L_A * T_A / T_B
            if extras is not None:
                tmp = extras[a].clone(); extras[a] = extras[b]; extras[b] = tmp
        return u <= r, r
    # boundary pair: I own one end, my neighbour the other
    mine = (A if rank == rA else B) - rank * chains_per_rank
    peer = rB if rank == rA else rA
    k = 0 if extras is None else extras.shape[1]
    send = torch.cat([logL[mine:mine + 1], rows[mine], extras[mine] if k else rows.new_zeros(0)]).contiguous()
    recv = torch.empty_like(send)
    ops = [dist.P2POp(dist.isend, send, peer), dist.P2POp(dist.irecv, recv, peer)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    L_mine, L_peer = float(logL[mine]), float(recv[0])
    L_A, L_B = (L_mine, L_peer) if rank == rA else (L_peer, L_mine)
    r = pt_swap_probability(L_A, L_B, T_A, T_B)
    if u <= r:
        rows[mine] = recv[1:1 + rows.shape[1]]
        logL[mine] = (L_B * T_B / T_A) if rank == rA else (L_A * T_A / T_B)
        if k:
            extras[mine] = recv[1 + rows.shape[1]:]
    return u <= r, r
