"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/tamcmc_accel.h
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute is launched here."""
import os
import re

import numpy as np
import pytest

import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="tamcmc_accel.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tamcmc_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(accel_mod):
    lib = accel_mod.load_library()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(accel_mod.capi.EXPORTS)
    assert accel_mod.capi.version().startswith("tamcmc_accel")
    sampler_names = [n for n in declared_symbols("tamcmc_sampler.h") if n not in ("tamcmc_eval_fn", "tamcmc_eval_batch")]
    assert len(sampler_names) >= 18
    for n in sampler_names:
        assert hasattr(lib, n), n


def test_strerror_covers_codes(accel_mod):
    lib = accel_mod.load_library()
    seen = {lib.tamcmc_strerror(c).decode() for c in range(9)}
    assert len(seen) == 9 and "ok" in seen


def test_no_cpu_fallback(accel_mod):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    if accel_mod.capi.device_count() > 0:
        pytest.skip("a GPU is present; the fallback question does not arise")
    w = W.make(2, Nx=512)
    with pytest.raises(accel_mod.AccelError) as e:
        accel_mod.Accel(2, w["plength"], w["x"], np.ones_like(w["x"]))
    assert e.value.code == accel_mod.capi.E_NODEVICE
    from tamcmc_amd.model_def import ModelDef, Data
    with pytest.raises(accel_mod.AccelError):
        ModelDef(Data(w["x"], np.ones_like(w["x"])), 2, w["plength"], w["params_true"], w["relax"], np.ones(2))


def test_argument_validation_happens_before_device_use(accel_mod):
    lib = accel_mod.load_library()
    import ctypes as C
    ctx = C.c_void_p()
    x = np.linspace(1.0, 2.0, 16)
    pl = (C.c_int32 * 11)(*([0] * 11))
    dp = C.POINTER(C.c_double)
    xp = x.ctypes.data_as(dp)
    # disabled / unknown model ids are rejected whatever the device situation
    assert lib.tamcmc_ctx_create(C.byref(ctx), 0, 4, 0, 1.0, pl, 16, xp, xp, None) == accel_mod.capi.E_MODEL_DISABLED
    assert lib.tamcmc_ctx_create(C.byref(ctx), 0, 5, 0, 1.0, pl, 16, xp, xp, None) == accel_mod.capi.E_MODEL_DISABLED
    assert lib.tamcmc_ctx_create(C.byref(ctx), 0, 99, 0, 1.0, pl, 16, xp, xp, None) == accel_mod.capi.E_UNKNOWN_MODEL
    assert lib.tamcmc_ctx_create(C.byref(ctx), 0, 2, 7, 1.0, pl, 16, xp, xp, None) == accel_mod.capi.E_UNKNOWN_MODEL
    assert lib.tamcmc_ctx_create(C.byref(ctx), 0, 2, 0, 1.0, pl, 1, xp, xp, None) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_ctx_create(None, 0, 2, 0, 1.0, pl, 16, xp, xp, None) == accel_mod.capi.E_INVALID
    # the batch entry points without a context: an error code, not a crash (no device is touched)
    L, st = C.c_double(), C.c_int32()
    assert lib.tamcmc_eval_batch_arm(None, 4) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_eval_batch_fire(None, 4, 3, xp, xp) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_eval_batch_disarm(None) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_eval_batch_poll(None, 0, C.byref(L), C.byref(st)) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_eval_batch_begin(None, 4, 3, xp, xp) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_ctx_reserve(None, 4) == accel_mod.capi.E_INVALID
    A = np.array([[4.0, 2.0], [2.0, 3.0]]); Lc = np.zeros((2, 2))
    assert lib.tamcmc_host_cholesky(A.ctypes.data_as(dp), 2, Lc.ctypes.data_as(dp)) == 0 and np.allclose(Lc @ Lc.T, A)
    assert lib.tamcmc_host_cholesky(None, 2, Lc.ctypes.data_as(dp)) == accel_mod.capi.E_INVALID
    assert lib.tamcmc_ctx_destroy(None) == 0


def test_synthetic_generator_is_deterministic(accel_mod):
    from tamcmc_amd import synth
    w = synth.workload_c2()
    assert w["params_true"].size == 56 and list(w["plength"]) == [7, 2, 7, 7, 7, 0, 6, 7, 10, 1, 2]
    assert w["index_to_relax"].size == 44
    a = synth.chain_params(w, 4)
    b = synth.chain_params(w, 4)
    assert np.array_equal(a, b) and not np.array_equal(a[0], a[1])
    T = synth.temperatures(64)
    assert T[0] == 1.0 and T[-1] == pytest.approx(150.0)
    r = synth.XorShift64(88172645463325252)
    assert r.next_u64() == 8748534153485358512   # first xorshift64 output for this classic seed
    w4 = synth.workload_c4()
    assert w4["params_true"].size == 106
    w1 = synth.workload_c1()
    assert w1["model_case"] == 11 and w1["x"].size == 10000


def test_tile_geometry_partitions_the_grid():
    """tm_tiles / tm_tile_bound (tamcmc_dev.h; the setup kernel runs the same function): equal-cost boundaries cover the
    grid in order with tiles of at most TM_TILE_MAXU units, whatever the unit costs."""
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.run(["make", "-C", cpp, "-s", "geometry_check"], check=True)
    r = subprocess.run([os.path.join(cpp, "geometry_check")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_bench_refuses_a_rank_count_that_does_not_match():
    """`bench.py --gpus N` inside a launcher: the reported n_gpus is the process-group size, so a mismatch between
    --gpus and WORLD_SIZE is an error (round 1 silently benchmarked one GPU and printed n_gpus = 1).  Needs no GPU:
    the check comes before anything touches one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
