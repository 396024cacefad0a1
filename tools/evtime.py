#!/usr/bin/env python3
"""Developer tool: eval-kernel time (HIP events around the eval launch) and step time at C2 (or c4 / c1) through the
device-pointer entry point, for the environment it is started in (TAMCMC_PRIO, TAMCMC_TILES, ... and TAMCMC_ACCEL_LIB
for variant builds; timing-only builds that skip the finalize are fine here: nothing waits on the results).
usage: evtime.py [c2|c4|c1] [chains] [repeats]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tamcmc_amd
from tamcmc_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = {"c2": synth.workload_c2, "c4": synth.workload_c4, "c1": synth.workload_c1}[which]()
P = synth.chain_params(w, n); T = synth.temperatures(max(n, 2))[:n]
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TAMCMC_"))
dev = torch.device("cuda", 0)
with tamcmc_amd.Accel(w["model_case"], w["plength"], w["x"], y) as acc:
    acc.set_vars(w["index_to_relax"])
    acc.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    dP = torch.from_numpy(P).to(dev); dT = torch.from_numpy(T).to(dev)
    dL = torch.empty(n, dtype=torch.float64, device=dev); dG = torch.empty(n, w["index_to_relax"].size, dtype=torch.float64, device=dev)
    dS = torch.empty(n, dtype=torch.int32, device=dev)
    def step(grad):
        acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr() if grad else 0, dS.data_ptr())
    for _ in range(500): step(True)            # clocks settle
    torch.cuda.synchronize()
    for r in range(reps):
        out = []
        for grad in (False, True):
            for _ in range(50): step(grad)
            torch.cuda.synchronize()
            acc.profile(4)
            t0 = time.perf_counter()
            for _ in range(400): step(grad)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 400
            ms, k = acc.kernel_time(); acc.profile(False)
            out.append((ms / max(k, 1) * 1e3, dt * 1e6))
        print(f"[{tag}] {which} {n} chains: eval<logL> {out[0][0]:6.2f} us (step {out[0][1]:6.1f})   eval<grad> {out[1][0]:6.2f} us (step {out[1][1]:6.1f})", flush=True)
