"""Golden regression vectors (tests/golden/oracle_regression_v1.npz, made by tests/golden/make_golden.py).
They are oracle outputs, not reference outputs (the reference holds none for this path): the CPU test
pins the oracle against drift, the GPU test pins the HIP path against the committed numbers."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    z = np.load(os.path.join(HERE, "golden", "oracle_regression_v1.npz"))
    for ci in range(int(z["ncases"])):
        k = f"c{ci:02d}"
        yield dict(id=int(z[k + "_id"]), plength=z[k + "_plength"], x=z[k + "_x"], y=z[k + "_y"], P=z[k + "_params"], T=z[k + "_T"],
                   logL=z[k + "_logL"], status=z[k + "_status"], probe=z[k + "_probe"], mprobe=z[k + "_model_probe"])


def test_oracle_reproduces_golden(orc):
    n = 0
    for c in cases():
        logL, st, models = orc.generate_batch(c["id"], c["plength"], c["x"], c["y"], c["P"], c["T"], want_models=True)
        assert np.array_equal(st, c["status"])
        # same code, same flags -> bitwise; allow 1e-14 for a different libm / compiler on another box
        assert np.allclose(logL, c["logL"], rtol=1e-14, atol=0)
        assert np.allclose(models[:, c["probe"]], c["mprobe"], rtol=1e-13, atol=0)
        n += 1
    assert n == 26


@pytest.mark.gpu
def test_hip_matches_golden(accel_mod):
    for c in cases():
        with accel_mod.Accel(c["id"], c["plength"], c["x"], c["y"]) as acc:
            logL, st, models = acc.eval_batch(c["P"], c["T"], model_rows=[0, 1, 2])
        assert np.array_equal(st, c["status"])
        assert np.max(np.abs(logL - c["logL"]) / np.abs(c["logL"])) <= 1e-10
        assert np.max(np.abs(models[:, c["probe"]] - c["mprobe"]) / c["mprobe"]) <= 1e-12
