#!/usr/bin/env python3
"""Developer probe: the host-pointer entry point (tamcmc_eval_batch) in a loop, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
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
for grad in (True, False):
    for _ in range(5):
        acc.eval_batch(P, T, grad=grad)
    t0 = time.perf_counter()
    for _ in range(100):
        acc.eval_batch(P, T, grad=grad)
    print(f"host path grad={int(grad)}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per call")
