#!/usr/bin/env python3
"""Developer probe: does the host-pointer path slow down after torch / the device-pointer path were used?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth

w = synth.workload_c2()
n = 64
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
acc = tamcmc_amd.Accel(2, w["plength"], w["x"], y)
acc.set_vars(w["index_to_relax"])

def host(tag):
    for grad in (False, True, False, True):
        for _ in range(5):
            acc.eval_batch(P, T, grad=grad)
        t0 = time.perf_counter()
        for _ in range(100):
            acc.eval_batch(P, T, grad=grad)
        print(f"{tag}: host path grad={int(grad)}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per call", flush=True)

host("torch imported, not initialised")
print("torch threads", torch.get_num_threads())
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
host("torch imported, cuda initialised")
acc.set_stream(torch.cuda.current_stream(dev).cuda_stream)
dP = torch.from_numpy(P).to(dev); dT = torch.from_numpy(T).to(dev)
dL = torch.empty(n, dtype=torch.float64, device=dev); dG = torch.empty(n, 44, dtype=torch.float64, device=dev)
dS = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(50):
    acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr(), dS.data_ptr())
torch.cuda.synchronize()
host("after device path on torch's stream (still set)")
acc.set_stream(0)
host("own stream again")
