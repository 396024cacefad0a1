#!/usr/bin/env python3
"""Developer tool (GPU box): a long run of the library's sampler loop at C2 (64 chains x 1e5 bins, parallel tempering every
iteration) in its default mode -- armed batches, accept on arrival -- and the same run with both switched off: the final
state must be the same bit for bit, and nothing may hang.  usage: soak_sampler.py [iterations]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(n_it, out):
    import tamcmc_amd
    from tamcmc_amd import synth, sampler as S
    w = synth.workload_c2()
    with tamcmc_amd.Accel(2, w["plength"], w["x"], np.ones(w["x"].size)) as a0:
        m, _ = a0.model_explicit(w["params_true"])
    y = synth.make_spectrum(m)
    with tamcmc_amd.Accel(2, w["plength"], w["x"], y) as acc:
        cfg = S.default_cfg(64, seed=11, Nt_learn=(20, n_it // 2, 10 ** 9), periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], w["err"])
        smp.init()
        t0 = time.perf_counter()
        done = 0
        while done < n_it:
            k = min(5000, n_it - done)
            smp.run(k, history=False)
            done += k
            print(f"  {done} iterations, {done / (time.perf_counter() - t0):.0f} it/s", flush=True)
        np.savez(out, vars=smp.get("vars"), logL=smp.get("logL"), sigma=smp.get("sigma"), covar=smp.get("covarmat"))
        smp.close()

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        run(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    outs = []
    for tag, env in (("default", {}), ("plain", {"TAMCMC_SAMPLER_ARM": "0", "TAMCMC_SAMPLER_ARRIVE": "0"})):
        out = f"/tmp/soak_sampler_{tag}.npz"
        print(f"== {tag} {env}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n_it), out], env={**os.environ, **env}, check=True)
        outs.append(np.load(out))
    same = all(np.array_equal(outs[0][k], outs[1][k]) for k in ("vars", "logL", "sigma", "covar"))
    print("final state bit for bit the same:", same)
    sys.exit(0 if same else 1)
