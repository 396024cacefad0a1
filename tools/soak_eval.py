#!/usr/bin/env python3
"""Developer tool (GPU box): hammer the likelihood path -- in-launch finalize (ticket hand-over between workgroups), host
path that watches the results arrive, batches of changing size -- and check that every chain's logL is bit for bit the
value of the first evaluation of that chain, whatever batch it is evaluated in.  Usage: soak_eval.py [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd
from tamcmc_amd import synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = synth.workload_c2()
n = 256
P = synth.chain_params(w, n); T = synth.temperatures(n)
y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
rng = np.random.default_rng(1)
with tamcmc_amd.Accel(2, w["plength"], w["x"], y) as acc:
    acc.set_vars(w["index_to_relax"])
    L0, st0 = acc.eval_batch(P, T)
    Lg0, _, g0 = acc.eval_batch(P, T, grad=True)
    assert np.all(st0 == 0)
    t0 = time.time(); bad = 0; nev = 0
    for r in range(rounds):
        k = int(rng.integers(1, n + 1)); a = int(rng.integers(0, n - k + 1))
        if r % 10 == 9:
            Lg, _, g = acc.eval_batch(P[a:a + k], T[a:a + k], grad=True)
            ok = np.array_equal(Lg, Lg0[a:a + k]) and np.array_equal(g, g0[a:a + k])
        else:
            L, st = acc.eval_batch(P[a:a + k], T[a:a + k])
            ok = np.array_equal(L, L0[a:a + k]) and np.all(st == 0)
        nev += k
        if not ok:
            bad += 1
            print("MISMATCH at round", r, "chains", a, a + k)
        if r % 2000 == 1999:
            print(f"round {r + 1}: {nev} chain evaluations, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
print("soak done:", rounds, "batches,", nev, "chain evaluations,", bad, "mismatches")
sys.exit(1 if bad else 0)
