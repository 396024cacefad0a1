#!/usr/bin/env python3
"""Kernel-time ablations on the C2 workload (developer tool): which part of the eval kernel costs what."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tamcmc_amd  # noqa: E402
from tamcmc_amd import synth  # noqa: E402


def run(tag, P, T, w, y, grad, reps=30):
    with tamcmc_amd.Accel(2, w["plength"], w["x"], y) as acc:
        acc.set_vars(w["index_to_relax"])
        for _ in range(3):
            acc.eval_batch(P, T, grad=grad)
        acc.profile(True)
        for _ in range(reps):
            acc.eval_batch(P, T, grad=grad)
        ms, n = acc.kernel_time()
    print(f"{tag:40s} grad={int(grad)} kernel {ms / n * 1e3:8.1f} us")


def main():
    w = synth.workload_c2()
    n = 64
    P = synth.chain_params(w, n)
    T = synth.temperatures(n)
    y = np.abs(np.sin(np.arange(w["x"].size))) + 0.5
    b = dict(Nmax=7, lmax=2, s=7 + 2 + 21, w=7 + 2 + 21 + 6, z=7 + 2 + 21 + 6 + 7, q=7 + 2 + 21 + 6 + 7 + 10)
    for grad in (False, True):
        run("base", P, T, w, y, grad)
        P1 = P.copy(); P1[:, b["z"] + 4] = 0.0; P1[:, b["z"] + 7] = 0.0
        run("no harvey (tau=0)", P1, T, w, y, grad)
        P2 = P.copy(); P2[:, b["q"] + 1] = 0.05
        run("tiny windows (trunc_c=0.05)", P2, T, w, y, grad)
        P3 = P2.copy(); P3[:, b["z"] + 4] = 0.0; P3[:, b["z"] + 7] = 0.0
        run("tiny windows + no harvey", P3, T, w, y, grad)
        P4 = P.copy(); P4[:, b["q"] + 1] = 10000.0
        run("no truncation (all 63 comps everywhere)", P4, T, w, y, grad)


if __name__ == "__main__":
    main()
