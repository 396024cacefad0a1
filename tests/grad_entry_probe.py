"""Developer probe: HIP gradient against the oracle's analytic gradient, entry by entry (window on), printing for every
case the worst |dg|/|g| and the worst |dg| / sum|terms| (the entry's conditioning).  Usage: python tests/grad_entry_probe.py
(kept under tests/: it uses the oracle, which only tests, smoke() and bench.py's cpu_baseline leg may touch)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import workloads as W          # noqa: E402
import tamcmc_amd              # noqa: E402
from tamcmc_amd import synth   # noqa: E402
from oracle import pyoracle as orc   # noqa: E402


def report(tag, g, ga, gr):
    ninf = np.max(np.abs(gr), axis=1, keepdims=True)
    e_rel = np.abs(g - gr) / (np.abs(gr) + 1e-8 * ninf)
    e_cond = np.abs(g - gr) / ga
    i = np.unravel_index(np.argmax(e_rel), e_rel.shape)
    print(f"{tag}: worst |dg|/(|g|+1e-8 ninf) = {e_rel.max():.2e} at chain {i[0]} var {i[1]} (g = {gr[i]:.3e}, ninf = {ninf[i[0], 0]:.2e}, "
          f"cond = {ga[i] / abs(gr[i]):.1e});  worst |dg|/sum|terms| = {e_cond.max():.2e};  "
          f"worst pure |dg|/|g| = {np.max(np.abs(g - gr) / np.abs(gr)):.2e}", flush=True)


def case(tag, mid, w, y, P, T, sigma=None, like=0):
    idx = w["index_to_relax"]
    with tamcmc_amd.Accel(mid, w["plength"], w["x"], y, sigma_y=sigma, likelihood_case=like) as acc:
        acc.set_vars(idx)
        L, st, g = acc.eval_batch(P, T, grad=True)
    t0 = time.time()
    gr, ga, rL, rst = orc.grad_analytic(mid, w["plength"], w["x"], y, P, T, idx, sigma_y=sigma, likelihood_case=like)
    dt = time.time() - t0
    assert np.array_equal(st, rst), (st, rst)
    report(f"{tag} (oracle {dt:.1f} s)", g, ga, gr)


def main():
    for mid in W.ALL_IDS:
        for kw in (dict(trunc_c=20.0), dict(trunc_c=7.0, asym=-40.0, do_amp=True)):
            if mid in (0, 1):
                w = W.make_gauss(mid, Nx=5000)
            else:
                w = W.any_model(mid, Nx=5000, **kw)
            m, _ = orc.model(mid, w["params_true"], w["plength"], w["x"])
            y = synth.make_spectrum(m, seed=17)
            P = W.perturbed(w, 4, scale=0.004)
            case(f"id {mid} {kw}", mid, w, y, P, synth.temperatures(4))
            if mid in (0, 1):
                break
    w = synth.workload_c2()
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m)
    case("C2 64 x 1e5", 2, w, y, synth.chain_params(w, 64), synth.temperatures(64))
    w = synth.workload_c4()
    m, _ = orc.model(2, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m)
    case("C4 16 x 1e5", 2, w, y, synth.chain_params(w, 16), synth.temperatures(16))


if __name__ == "__main__":
    main()
