#!/usr/bin/env python3
"""Generates tests/golden/oracle_regression_v1.npz.

WHAT THESE VECTORS ARE: outputs of THIS repo's CPU oracle (oracle/tamcmc_oracle.c) on seeded synthetic
inputs.  They are NOT outputs of the reference (which cannot be built in this image and holds no golden
vectors for this path -- DESIGN.md section 2).  They pin the oracle and the HIP path against drift.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import workloads as W  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tamcmc_amd import synth  # noqa: E402

CASES = []
for mid in W.ALL_IDS:
    CASES.append((mid, dict()))
    CASES.append((mid, dict(asym=25.0, do_amp=True, trunc_c=7.0)))


def main():
    out = {}
    for ci, (mid, kw) in enumerate(CASES):
        w = W.any_model(mid, Nx=3001, **kw)
        m, st = orc.model(mid, w["params_true"], w["plength"], w["x"])
        assert st == 0
        y = synth.make_spectrum(m, seed=1000 + ci)
        P = W.perturbed(w, 3, scale=0.003, seed=ci)
        T = synth.temperatures(3, Tmax=9.0)
        logL, status, models = orc.generate_batch(mid, w["plength"], w["x"], y, P, T, want_models=True)
        probe = np.linspace(0, w["x"].size - 1, 48).astype(np.int64)
        k = f"c{ci:02d}"
        out[k + "_id"] = np.int32(mid)
        out[k + "_plength"] = w["plength"]
        out[k + "_x"] = w["x"]
        out[k + "_y"] = y
        out[k + "_params"] = P
        out[k + "_T"] = T
        out[k + "_logL"] = logL
        out[k + "_status"] = status
        out[k + "_probe"] = probe
        out[k + "_model_probe"] = models[:, probe]
    out["ncases"] = np.int32(len(CASES))
    np.savez_compressed(os.path.join(HERE, "oracle_regression_v1.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
