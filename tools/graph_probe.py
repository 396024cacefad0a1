#!/usr/bin/env python3
"""Developer probe (GPU box): host-side cost of enqueueing one likelihood step (setup + eval kernels) directly vs as a
replayed HIP graph (captured through torch.cuda.CUDAGraph on the context's stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tamcmc_amd
from tamcmc_amd import synth

w = synth.workload_c2()
n = 64
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
acc = tamcmc_amd.Accel(w["model_case"], w["plength"], w["x"], y)
acc.set_vars(w["index_to_relax"])
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev)
acc.set_stream(s.cuda_stream)
dP = torch.from_numpy(P).to(dev); dT = torch.from_numpy(T).to(dev)
dL = torch.empty(n, dtype=torch.float64, device=dev); dS = torch.empty(n, dtype=torch.int32, device=dev)
dG = torch.empty(n, w["index_to_relax"].size, dtype=torch.float64, device=dev)

def step(grad):
    acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr() if grad else 0, dS.data_ptr())

for grad in (False, True):
    with torch.cuda.stream(s):
        for _ in range(50): step(grad)
        torch.cuda.synchronize()
        # direct: host time to enqueue, and wall per step
        N = 400
        t0 = time.perf_counter()
        for _ in range(N): step(grad)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"grad={int(grad)} direct : enqueue {1e6*(t1-t0)/N:6.2f} us/step, wall {1e6*(t2-t0)/N:6.2f} us/step")
        L0 = dL.clone()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                step(grad)
        except Exception as e:
            print("capture failed:", repr(e)[:300]); continue
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N): g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"grad={int(grad)} graph  : enqueue {1e6*(t1-t0)/N:6.2f} us/step, wall {1e6*(t2-t0)/N:6.2f} us/step, same results: {bool(torch.equal(L0, dL))}")
