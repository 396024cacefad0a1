#!/usr/bin/env python3
"""Developer tool (GPU box): cycle stamps of the setup kernel's phases (setup and backward kernels built with
-DTM_SU_TRACE: the stamps travel through the series table into the gradient rows).
Here with the stamps inside the multiplet derivation and the ratio tables (-DTM_SU_TRACE_FINE as well).
Usage: TAMCMC_ACCEL_LIB=gpurun_variants/lib_sufine.so python tools/su_trace_fine.py [c2|c4|c1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = {"c2": synth.workload_c2, "c4": synth.workload_c4, "c1": synth.workload_c1}[which]()
n = 64
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
names = ["start", "params row staged (barrier)", "chain scalars (barrier)", "(t0) multiplets up to the ratios", "(t64) cell polynomials",
         "(t128) m-ratios", "after barrier", "(t0) multiplet records stored", "after barrier", "(t0) tile lists", "(t0) launch ranks",
         "end", "mult: entry", "mult: frequency+splitting done", "mult: width done", "mult: heights done", "mult: components done", "mult: window done", "ratios: inclination known", "ratios: tables cleared, angle known"]
with tamcmc_amd.Accel(w["model_case"], w["plength"], w["x"], y) as acc:
    acc.set_vars(w["index_to_relax"])
    for _ in range(20):
        L, st, g = acc.eval_batch(P, T, grad=True)
    ts = np.median(g[:, :20], axis=0)
    for i, nm in enumerate(names):
        print(f"{i:2d} {nm:36s} {ts[i]:9.0f} cycles")
