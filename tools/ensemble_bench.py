#!/usr/bin/env python3
"""Developer tool: BASELINE config C5 on one GPU -- several independent stars, 16 chains each, one context (own HIP
stream) per star, all driven from ONE host thread with device-resident inputs (launches are asynchronous, so the
streams overlap on the GPU).  Prints the aggregate rate against one context holding the same number of chains.

    python tools/ensemble_bench.py [stars=4] [chains_per_star=16] [steps=300]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tamcmc_amd
from tamcmc_amd import synth

nstars = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)


def make(n_chains, seed):
    w = synth.workload_c2()
    y = np.abs(np.sin(np.arange(w["x"].size) + seed)) + 0.5
    acc = tamcmc_amd.Accel(2, w["plength"], w["x"], y)
    acc.set_vars(w["index_to_relax"])
    P = torch.from_numpy(synth.chain_params(w, n_chains, seed=seed)).to(dev)
    T = torch.from_numpy(synth.temperatures(n_chains)).to(dev)
    L = torch.empty(n_chains, dtype=torch.float64, device=dev)
    G = torch.empty(n_chains, w["index_to_relax"].size, dtype=torch.float64, device=dev)
    S = torch.empty(n_chains, dtype=torch.int32, device=dev)
    return acc, (n_chains, P, T, L, G, S)


def step(acc, b, grad):
    n, P, T, L, G, S = b
    acc.eval_batch_device(n, P.data_ptr(), T.data_ptr(), L.data_ptr(), G.data_ptr() if grad else 0, S.data_ptr())


def rate(ctxs, grad):
    for _ in range(300):
        for acc, b in ctxs:
            step(acc, b, grad)
    for acc, _ in ctxs:
        acc.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for acc, b in ctxs:
            step(acc, b, grad)
    for acc, _ in ctxs:
        acc.synchronize()
    dt = time.perf_counter() - t0
    return sum(b[0] for _, b in ctxs) * steps / dt


def rate_threads(ctxs, grad):
    """one host thread per star (the GIL is released inside the C call)"""
    import threading
    def loop(acc, b, n):
        for _ in range(n):
            step(acc, b, grad)
        acc.synchronize()
    th = [threading.Thread(target=loop, args=(a, b, 300)) for a, b in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    th = [threading.Thread(target=loop, args=(a, b, steps)) for a, b in ctxs]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    return sum(b[0] for _, b in ctxs) * steps / dt


def make_multi(nstars, nch):
    """the same ensemble as ONE context holding all the spectra, every chain fitted to its own (tamcmc_ctx_set_spectra)"""
    w = synth.workload_c2()
    Y = np.stack([np.abs(np.sin(np.arange(w["x"].size) + 11 + k)) + 0.5 for k in range(nstars)])
    acc = tamcmc_amd.Accel(2, w["plength"], w["x"], Y[0])
    acc.set_vars(w["index_to_relax"])
    acc.set_spectra(Y)
    acc.set_chain_spectrum(np.repeat(np.arange(nstars, dtype=np.int32), nch))
    n = nstars * nch
    P = torch.from_numpy(np.concatenate([synth.chain_params(w, nch, seed=11 + k) for k in range(nstars)])).to(dev)
    T = torch.from_numpy(np.tile(synth.temperatures(nch), nstars)).to(dev)
    L = torch.empty(n, dtype=torch.float64, device=dev)
    G = torch.empty(n, w["index_to_relax"].size, dtype=torch.float64, device=dev)
    S = torch.empty(n, dtype=torch.int32, device=dev)
    return acc, (n, P, T, L, G, S)


ens = [make(nch, 11 + k) for k in range(nstars)]
multi = [make_multi(nstars, nch)]
one = [make(nch * nstars, 5)]
single = [ens[0]]
for grad in (True, False):
    r_e, r_1, r_s = rate(ens, grad), rate(one, grad), rate(single, grad)
    r_t = rate_threads(ens, grad)
    r_m = rate(multi, grad)
    print(f"{'logL+grad' if grad else 'logL only'}: {nstars} stars x {nch} chains on {nstars} streams {r_e:,.0f} chain-steps/s "
          f"(one host thread per star: {r_t:,.0f}); "
          f"the same ensemble as one context with {nstars} spectra {r_m:,.0f}; "
          f"one context x {nch * nstars} chains of one star {r_1:,.0f}; one star x {nch} chains alone {r_s:,.0f}")
