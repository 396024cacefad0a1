#!/usr/bin/env python3
"""Developer tool: does replaying the step (setup -> eval -> backward) as a captured HIP graph beat launching the
three kernels one by one?  (64 chains x 1e5 bins, device-resident inputs.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tamcmc_amd
from tamcmc_amd import synth

dev = torch.device("cuda", 0)
w = synth.workload_c2()
n = 64
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
acc = tamcmc_amd.Accel(2, w["plength"], w["x"], y)
acc.set_vars(w["index_to_relax"])
P = torch.from_numpy(synth.chain_params(w, n)).to(dev); T = torch.from_numpy(synth.temperatures(n)).to(dev)
L = torch.empty(n, dtype=torch.float64, device=dev); G = torch.empty(n, 44, dtype=torch.float64, device=dev)
S = torch.empty(n, dtype=torch.int32, device=dev)


def step(grad):
    acc.eval_batch_device(n, P.data_ptr(), T.data_ptr(), L.data_ptr(), G.data_ptr() if grad else 0, S.data_ptr())


for grad in (True, False):
    side = torch.cuda.Stream(dev)
    acc.set_stream(side.cuda_stream)
    with torch.cuda.stream(side):
        for _ in range(600):
            step(grad)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            step(grad)
        side.synchronize()
        t_direct = (time.perf_counter() - t0) / 300
        L_direct = L.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(10):
                step(grad)
        for _ in range(30):
            g.replay()
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        side.synchronize()
        t_graph = (time.perf_counter() - t0) / 300
        assert torch.equal(L, L_direct)
    print(f"grad={int(grad)}: direct launches {t_direct * 1e6:.1f} us/step, graph of 10 steps replayed {t_graph * 1e6:.1f} us/step")
