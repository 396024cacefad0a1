#!/usr/bin/env python3
"""Developer tool: iterations/s of the library's sampler loop at C2 (64 chains x 1e5 bins, PT every iteration) for the
environment it is started in (TAMCMC_SAMPLER_PIPELINE, TAMCMC_SAMPLER_THREADS, TAMCMC_SAMPLER_TIMING=1 ...)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth, sampler as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = synth.workload_c2()
with tamcmc_amd.Accel(2, w["plength"], w["x"], np.ones(w["x"].size)) as a0:
    m, _ = a0.model_explicit(w["params_true"])
y = synth.make_spectrum(m)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TAMCMC_SAMPLER"))
with tamcmc_amd.Accel(2, w["plength"], w["x"], y) as acc:
    for name, learn, err in (("acquire", (10 ** 9, 10 ** 9 + 1, 10 ** 9 + 2), 0.05 * w["err"]), ("learning", (20, 60, 10 ** 9), w["err"])):
        cfg = S.default_cfg(n, seed=7, Nt_learn=learn, periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], err)
        smp.init()
        smp.run(300, history=False)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            smp.run(2000, history=False)
            best = max(best, 2000 / (time.perf_counter() - t0))
        print(f"[{tag}] {name}: {best:8.0f} iterations/s ({1e6 / best:5.1f} us per iteration)", flush=True)
        smp.close()
