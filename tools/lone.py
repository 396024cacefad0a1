#!/usr/bin/env python3
"""Developer tool: eval-kernel time (HIP events) at small chain counts -- the run time of a workgroup that has its CU to itself."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth
w = synth.workload_c2()
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
q = 7 + 2 + 21 + 6 + 7 + 10
for n in [int(v) for v in (sys.argv[1:] or ["1", "4", "8", "16", "32", "64"])]:
    P = synth.chain_params(w, n); T = synth.temperatures(max(n, 2))[:n]
    for tag, PP in (("base", P), ("tiny windows", np.where(np.arange(P.shape[1]) == q + 1, 0.05, P))):
        with tamcmc_amd.Accel(2, w["plength"], w["x"], y) as acc:
            acc.set_vars(w["index_to_relax"])
            out = []
            for grad in (False, True):
                for _ in range(20): acc.eval_batch(PP, T, grad=grad)
                acc.profile(True)
                for _ in range(50): acc.eval_batch(PP, T, grad=grad)
                ms, k = acc.kernel_time(); acc.profile(False)
                out.append(ms / k * 1e3)
        print(f"{n:3d} chains {tag:13s}: eval<logL> {out[0]:6.1f} us   eval<grad> {out[1]:6.1f} us")
