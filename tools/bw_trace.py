#!/usr/bin/env python3
"""Developer tool (GPU box): cycle stamps of the backward kernel's phases (build with -DTM_BW_TRACE, which sends them out
through the gradient rows).  Usage: TAMCMC_ACCEL_LIB=gpurun_variants/lib_bwtrace.so python tools/bw_trace.py [c2|c4|c1]"""
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
names = ["start", "staged", "gathered tiles", "per-multiplet chain rule", "chain-level sums", "(t0) enter chain-level",
         "(t0) chain-level pairs done", "after barrier", "gather per variable done", "after barrier", "(t64) enter noise",
         "(t64) noise pairs done", "(t0) params + records staged", "(t64) noise partials summed", "(t192) logL finalized",
         "(t0) at the staging barrier", "(t0) kernel arguments arrived", "(t0) first load from memory back",
         "(t0) second load, same line", "(t0) third load, another buffer",
         "chain rule: entry", "chain rule: component sums done", "chain rule: asymmetry + splitting done", "chain rule: heights done",
         "chain rule: width done"]
with tamcmc_amd.Accel(w["model_case"], w["plength"], w["x"], y) as acc:
    acc.set_vars(w["index_to_relax"])
    for _ in range(20):
        L, st, g = acc.eval_batch(P, T, grad=True)
    ts = np.median(g[:, :25], axis=0)
    for i, nm in enumerate(names):
        print(f"{i:2d} {nm:32s} {ts[i]:9.0f} cycles")
