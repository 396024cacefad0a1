#!/usr/bin/env python3
"""Developer tool: per-kernel mean duration (us) of N evaluations, from HIP events around each launch...
uses rocprofv3-free timing: runs eval_batch_device in a loop under torch.profiler?  No -- simply prints
wall time per step for grad / logL paths.  Real per-kernel numbers come from rocprofv3 (profiles/)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tamcmc_amd
from tamcmc_amd import synth

# usage: kstats.py [c2|c4|c1] [chains]
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
kw = {"trunc_c": float(os.environ["TRUNC_C"])} if "TRUNC_C" in os.environ else {}     # TRUNC_C=10000: every window spans the grid
w = {"c2": synth.workload_c2, "c4": synth.workload_c4, "c1": synth.workload_c1}[which](**kw)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
acc = tamcmc_amd.Accel(w["model_case"], w["plength"], w["x"], y)
acc.set_vars(w["index_to_relax"])
dev = torch.device("cuda", 0)
acc.set_stream(torch.cuda.current_stream(dev).cuda_stream)
dP = torch.from_numpy(P).to(dev); dT = torch.from_numpy(T).to(dev)
dL = torch.empty(n, dtype=torch.float64, device=dev); dG = torch.empty(n, w["index_to_relax"].size, dtype=torch.float64, device=dev)
dS = torch.empty(n, dtype=torch.int32, device=dev)
for grad in (True, False):
    for _ in range(400):
        acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr() if grad else 0, dS.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr() if grad else 0, dS.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{which} {n} chains x {w['x'].size} bins, {w['plength'].sum()} params, {w['index_to_relax'].size} variables: grad={int(grad)} step {dt * 1e6:.1f} us = {n / dt:,.0f} chain-steps/s")
