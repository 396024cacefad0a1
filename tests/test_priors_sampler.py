"""N1 (log-priors) and N2 (batched sampler) on the CPU.
Pins available from the reference: the IDL-derived known answers of stats_dictionary.cpp:252-326 for the
primitive priors, and libc's own rand() for the private copy of the generator.  The sampler itself has no
reference fixture ("sampler-level parity is otherwise unpinned"); it is checked through
invariants and, on the GPU, by running it twice -- oracle evaluator vs HIP evaluator -- under one seed."""
import ctypes
import math

import numpy as np
import pytest

import workloads as W
from tamcmc_amd import sampler as S
from tamcmc_amd import synth


# ---------------------------------------------------------------- N1
def test_primitive_priors_known_answers():
    # stats_dictionary.cpp:252-326 ("Expected values were calculated using the IDL parent program")
    inf = float("inf")
    assert S.logP_primitive(2, [1, 0.5], 4) == pytest.approx(-18.225791, abs=2e-6)
    assert S.logP_primitive(1, [0, 1], -2) == -inf
    assert S.logP_primitive(1, [0, 1], 0.3) == 0.0
    assert S.logP_primitive(4, [0.1, 1], -2) == -inf
    assert S.logP_primitive(4, [0.1, 1], 0.5) == pytest.approx(-0.363766, abs=2e-6)
    assert S.logP_primitive(5, [1, 2, 0.1], -1) == -inf
    assert S.logP_primitive(5, [1, 2, 0.1], 1.5) == pytest.approx(-0.11807759, abs=2e-7)
    assert S.logP_primitive(5, [1, 2, 0.1], 3) == pytest.approx(-50.118074, abs=1e-5)
    assert S.logP_primitive(6, [1, 2, 0.1], -1) == pytest.approx(-200.11806, abs=5e-5)
    assert S.logP_primitive(6, [1, 2, 0.1], 1.5) == pytest.approx(-0.11807759, abs=2e-7)
    assert S.logP_primitive(6, [1, 2, 0.1], 3) == -inf
    assert S.logP_primitive(7, [1, 2, 0.1, 0.4], -1) == pytest.approx(-200.48651, abs=5e-5)
    assert S.logP_primitive(7, [1, 2, 0.1, 0.4], 1.5) == pytest.approx(-0.48652704, abs=2e-7)
    assert S.logP_primitive(7, [1, 2, 0.1, 0.4], 3) == pytest.approx(-3.6115268, abs=1e-6)
    # the remaining primitives against their definitions
    assert S.logP_primitive(8, [1, 3], -2) == pytest.approx(-math.log(2))
    assert S.logP_primitive(8, [1, 3], 0.5) == -inf
    assert S.logP_primitive(10, [0.1, 1], -0.5) == pytest.approx(S.logP_primitive(4, [0.1, 1], 0.5))
    assert S.logP_primitive(0, [], 123.0) == 0.0 and S.logP_primitive(11, [], 123.0) == 0.0


def ms_global_prior_setup():
    w = synth.workload_c2(model_case=3)
    n = w["params_true"].size
    sw = np.zeros(n, dtype=np.int32)
    pp = np.zeros((4, n))
    b = W.split(w)
    sw[:b["Nmax"]] = 4; pp[0, :b["Nmax"]] = 1.0; pp[1, :b["Nmax"]] = 1000.0            # Jeffreys on heights
    f0 = b["Nmax"] + b["lmax"]
    for i in range(f0, f0 + b["Nf"]):
        sw[i] = 1; pp[0, i] = w["params_true"][i] - 5; pp[1, i] = w["params_true"][i] + 5   # uniform windows
    sw[b["Nmax"]] = 2; pp[0, b["Nmax"]] = 1.5; pp[1, b["Nmax"]] = 0.15
    return w, sw, pp, b


def test_priors_ms_global_pieces():
    w, sw, pp, b = ms_global_prior_setup()
    p = w["params_true"]
    extra = [1.0, 2.0, 0.2, 0.0]       # smoothness on, coefficient 2, a3/a1 limit 0.2
    v, err = S.log_prior(2, p, w["plength"], sw, pp, extra)
    assert err == 0 and np.isfinite(v)
    # rebuild the value from the definitions (priors_calc.cpp:24-172)
    ref = 0.0
    for i in range(p.size):
        if sw[i]:
            ref += S.logP_primitive(sw[i], pp[:, i], p[i])
    f0 = b["Nmax"] + b["lmax"]
    fl = [p[f0 + l * 7:f0 + (l + 1) * 7] for l in range(3)]
    d = np.empty(7); d[0] = fl[0][1] - fl[0][0]; d[-1] = fl[0][-1] - fl[0][-2]; d[1:-1] = (fl[0][2:] - fl[0][:-2]) / 2
    Dnu = d.sum()                       # the reference sums the derivative array (priors_calc.cpp:90)
    for i in range(7):
        ref += S.logP_primitive(6, [0, Dnu / 3, 0.015 * Dnu], fl[0][i] - fl[2][i])
    for l in range(3):
        y = fl[l]
        sd = np.empty(7); sd[0] = y[2] - 2 * y[1] + y[0]; sd[-1] = y[-1] - 2 * y[-2] + y[-3]; sd[1:-1] = y[2:] - 2 * y[1:-1] + y[:-2]
        for i in range(7):
            ref += S.logP_primitive(2, [0, 2.0], sd[i])
    assert v == pytest.approx(ref, rel=1e-13)
    # hard constraints
    q = p.copy(); q[b["Nmax"] + 1] = -0.1           # negative visibility
    assert S.log_prior(2, q, w["plength"], sw, pp, extra)[0] == -float("inf")
    q = p.copy(); q[b["s"] + 2] = 0.5               # |a3/a1| above the limit
    assert S.log_prior(2, q, w["plength"], sw, pp, extra)[0] == -float("inf")
    q = p.copy(); q[0] = 2000.0                     # outside the Jeffreys range
    assert S.log_prior(2, q, w["plength"], sw, pp, extra)[0] == -float("inf")
    # unsupported in the reference: multivariate Gaussian (exit) -> error flag
    sw2 = sw.copy(); sw2[3] = 3
    assert S.log_prior(2, p, w["plength"], sw2, pp, extra)[1] == 1


def test_priors_local():
    w = synth.workload_c1()
    n = w["params_true"].size
    sw = np.zeros(n, dtype=np.int32); pp = np.zeros((4, n))
    sw[0] = 1; pp[0, 0] = 0; pp[1, 0] = 100
    v, err = S.log_prior(3, w["params_true"], w["plength"], sw, pp, [0, 0, 0.2, 0])
    assert err == 0 and v == pytest.approx(-math.log(100))
    q = w["params_true"].copy()
    s = int(w["plength"][0] + w["plength"][1] + w["plength"][2:6].sum())
    q[s] = 0.0; q[s + 2] = 5.0                      # a1 slot 0 -> ratio against (sqrt a1 cos)^2 + (sqrt a1 sin)^2
    assert S.log_prior(3, q, w["plength"], sw, pp, [0, 0, 0.2, 0])[0] == -float("inf")


# ---------------------------------------------------------------- RNG
def test_rand_jump_ahead_equals_stepping():
    """GlibcRand::jump (x^k modulo the generator's characteristic polynomial, tamcmc_sampler.cpp): a sharded sampler
    passes over the draws of the chains other processes own without making them.  Every distance -- below the
    threshold where it just steps, powers of two, the distances a 45-variable 64-chain block gives -- must land on the
    value plain stepping reaches."""
    for seed in (1, 99, 2**31 + 7):
        ref = S.glibc_rand(seed, 70000)
        for skip in (0, 1, 30, 31, 63, 64, 65, 127, 1000, 2898, 2944, 4096, 12345, 65536):
            got = S.glibc_rand_jump(seed, skip, 200)
            assert np.array_equal(got, ref[skip:skip + 200]), (seed, skip)


def test_private_rand_is_glibc_rand():
    libc = ctypes.CDLL(None)
    for seed in (1, 12345, 2**31 + 7):
        libc.srand(ctypes.c_uint(seed))
        ref = [libc.rand() for _ in range(2000)]
        assert ref == list(S.glibc_rand(seed, 2000))


# ---------------------------------------------------------------- N2 with the oracle as evaluator
def oracle_evaluator(orc, mid, w, y):
    def f(P, T):
        return orc.generate_batch(mid, w["plength"], w["x"], y, P, T, nthreads=1)
    return f


def make_run(orc, evaluator_factory, nchains=6, n_iter=120, seed=4242, cfg_kw=None, Nx=1200):
    w, sw, pp, b = ms_global_prior_setup()
    w = dict(w); w["x"] = synth.grid(Nx, 2300.0, 840.0 / Nx)
    m, st = orc.model(3, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=77)
    kw = dict(dN_mixing=1)
    kw.update(cfg_kw or {})
    cfg = S.default_cfg(nchains, seed=seed, Nt_learn=(20, 60, 100000), periods_learn=(1, 2), prior_fct_switch=2, **kw)
    smp = S.Sampler(cfg, evaluator_factory(3, w, y), w["plength"], w["params_true"], w["relax"], w["err"], sw, pp,
                    [1.0, 5.0, 0.5, 0.0])
    smp.init()
    moved, swaps = smp.run(n_iter)
    return smp, moved, swaps, w, y


def test_sampler_invariants(orc):
    smp, moved, swaps, w, y = make_run(orc, lambda mid, w, y: oracle_evaluator(orc, mid, w, y))
    assert smp.iteration() == 120
    # reproducible under a fixed seed
    smp2, moved2, swaps2, _, _ = make_run(orc, lambda mid, w, y: oracle_evaluator(orc, mid, w, y))
    assert np.array_equal(moved, moved2) and np.array_equal(swaps, swaps2)
    assert np.array_equal(smp.get("vars"), smp2.get("vars"))
    # another seed gives another sequence
    _, moved3, _, _, _ = make_run(orc, lambda mid, w, y: oracle_evaluator(orc, mid, w, y), seed=7)
    assert not np.array_equal(moved, moved3)
    # swap attempts at every iteration but the first (dN_mixing = 1, i != 0)
    assert swaps[0] == -1 and np.all(swaps[1:] >= 0)
    # stored state is self-consistent: logL(T) of the stored params, posterior = likelihood + prior
    T = smp.get("Tcoefs")
    assert T[0] == 1.0 and T[-1] == pytest.approx(150.0)
    L, _ = orc.generate_batch(3, w["plength"], w["x"], y, smp.get("params"), T)
    assert np.allclose(smp.get("logL"), L, rtol=1e-12)
    P = smp.get("params"); V = smp.get("vars")
    assert np.array_equal(P[:, w["index_to_relax"]], V)
    # some moves accepted, not all, on every chain
    acc = moved.mean(axis=0)
    assert np.all(acc > 0.0) and np.all(acc < 1.0)
    # adaptation touched the proposal: sigma left its initial value 2.38^2 T^0.2 / Nvars, covariance is symmetric
    sig0 = 2.38 ** 2 * T ** 0.2 / smp.Nvars
    assert not np.allclose(smp.get("sigma"), sig0)
    C0 = smp.get("covarmat")[0]
    assert np.allclose(C0, C0.T) and np.all(np.linalg.eigvalsh(C0) > -1e-12)


def test_sharded_loop_in_the_library_single_rank(orc):
    """tamcmc_sampler_run_sharded with every chain owned here (no exchange callback needed) is tamcmc_sampler_run, and its
    block buffer holds what the single-process driver records per iteration: the samples, the statistics and the
    parallel-tempering attempts, plus running sums of the proposal parameters; it stops when the block is full; bad
    arguments are refused."""
    ev = lambda mid, w, y: oracle_evaluator(orc, mid, w, y)      # noqa: E731
    ref, moved, swaps, w, y = make_run(orc, ev, n_iter=50)
    smp, _, _, _, _ = make_run(orc, ev, n_iter=0)
    blk = S.ShardBlock(smp, 30)
    done, mv, sw = smp.run_sharded(50, None, block=blk, history=True)
    assert done == 30 and blk.count() == 30                     # the block is full: gather, reset, call again
    assert np.array_equal(mv[:30], moved[:30]) and np.array_equal(sw[:30], swaps[:30])
    vars30, stat30, pt30 = blk.data("vars"), blk.data("stat"), blk.data("pt")
    assert vars30.shape == (30, smp.nloc, smp.Nvars) and stat30.shape == (30, 3, smp.nloc) and pt30.shape == (30, 4)
    assert np.array_equal(pt30[0], [0.0, -1.0, np.nan, -1.0], equal_nan=True)       # iteration 0: no attempt
    assert np.array_equal(pt30[1:, 0], np.ones(29)) and np.array_equal(pt30[1:, 1] * 2 + pt30[1:, 3], swaps[1:30])
    assert np.allclose(stat30[0, 2], stat30[0, 0] + stat30[0, 1])     # logPost = logL + logPrior (before any swap: after one the
    #                                                                 reference's stale-prior quirk applies)
    sum_sigma = blk.data("sum_sigma")
    blk.reset()
    assert blk.count() == 0 and np.all(blk.data("sum_sigma") == 0.0) and np.all(sum_sigma > 0.0)
    done2, mv2, sw2 = smp.run_sharded(20, None, block=blk, history=True)
    assert done2 == 20 and np.array_equal(mv2, moved[30:]) and np.array_equal(sw2, swaps[30:])
    assert np.array_equal(blk.data("vars")[-1], ref.get("vars")) and np.array_equal(smp.get("logPost"), ref.get("logPost"))
    assert smp.iteration() == ref.iteration() == 50
    sec, its = smp.timing()
    assert all(v == 0.0 for v in sec.values())                   # timing is off unless asked for
    smp.set_timing(True)
    smp.run_sharded(5, None)
    sec, its = smp.timing()
    assert its == 5 and sec["proposals"] > 0.0 and sec["foreign_draws"] == 0.0 and sec["exchange"] == 0.0
    lib = S._lib()
    assert lib.tamcmc_sampler_run_sharded(smp._h, -1, S.C.cast(None, S.EXCHANGE_FN), None, None, None, None, None) != 0
    other, _, _, _, _ = make_run(orc, ev, nchains=4, n_iter=0)
    assert lib.tamcmc_sampler_run_sharded(other._h, 1, S.C.cast(None, S.EXCHANGE_FN), None, blk._h, None, None, None) != 0    # a block of another shape
    blk.close()


def test_pt_swap_bookkeeping(orc):
    smp, _, _, w, y = make_run(orc, lambda mid, w, y: oracle_evaluator(orc, mid, w, y), n_iter=5)
    T = smp.get("Tcoefs")
    L0, P0, V0, pr0 = smp.get("logL"), smp.get("params"), smp.get("vars"), smp.get("logPrior")
    sw, r = smp.pt_local(2, 0.0)        # u = 0 -> always swapped
    assert sw and 0.0 < r <= 1.0
    L1, P1, pr1, po1 = smp.get("logL"), smp.get("params"), smp.get("logPrior"), smp.get("logPost")
    assert np.array_equal(P1[2], P0[3]) and np.array_equal(P1[3], P0[2])
    assert L1[2] == pytest.approx(L0[3] * T[3] / T[2]) and L1[3] == pytest.approx(L0[2] * T[2] / T[3])   # MALA.cpp:393-394
    assert pr1[2] == pr0[3] and pr1[3] == pr0[2]
    assert po1[2] == pytest.approx(L1[2] + pr0[3])
    assert po1[3] == pytest.approx(L1[3] + pr0[3])      # quirk: B's posterior uses A's ALREADY OVERWRITTEN prior (MALA.cpp:428)
    sw, _ = smp.pt_local(0, 2.0)        # u > 1 -> never
    assert not sw and np.array_equal(smp.get("params")[0], P1[0])


def test_sampler_rejects_nan_and_minus_infinity(orc):
    calls = {"n": 0}

    def make(mid, w, y):
        inner = oracle_evaluator(orc, mid, w, y)

        def f(P, T):
            calls["n"] += 1
            L, st = inner(P, T)
            if calls["n"] > 1:
                L = L.copy(); L[1] = np.nan; st = st.copy(); st[1] = 1     # chain 1 always proposes a NaN model
            return L, st
        return f
    smp, moved, _, _, _ = make_run(orc, make, n_iter=30, cfg_kw=dict(dN_mixing=10**9), nchains=4)   # no PT: states stay in their slot
    assert moved[:, 1].sum() == 0 and moved[:, 0].sum() > 0
    assert np.all(smp.get("Pmove")[1] == 0.0)


def test_split_normal_generator_equals_the_one_piece_routine():
    """The sampler draws the random stream sequentially and leaves the Box-Muller transforms to worker threads;
    that pair must reproduce r8vec_normal_01 (random_JB.cpp:22-213) value for value, carried sine included."""
    sizes = [3, 4, 1, 5, 2, 2, 7, 1, 1, 44, 9, 10]
    a = S.normals(123, sizes, split=False)
    b = S.normals(123, sizes, split=True)
    assert np.array_equal(a, b) and np.all(np.isfinite(a)) and abs(a.mean()) < 0.5
    # many odd counts, several seeds: the cosine of a call's last pair has to come out of the same sin / cos pair as in
    # the one-piece routine (a lone cos() differed from it in the last bit once in ~9000 values: seed 5 below)
    for seed in (5, 77, 2024):
        sizes = [44, 7, 1, 3, 2, 44, 5, 9, 1, 1, 12] * 50
        assert np.array_equal(S.normals(seed, sizes, split=False), S.normals(seed, sizes, split=True)), seed
    # calls longer than one block of the raw generator (GlibcRand::fill unrolls the ring 480 values at a time)
    sizes = [2000, 481, 480, 479, 961, 1, 31, 62, 1443]
    assert np.array_equal(S.normals(9, sizes, split=False), S.normals(9, sizes, split=True))


def test_thread_count_does_not_change_the_chains(orc, monkeypatch):
    w, sw, pp, b = ms_global_prior_setup()
    w = dict(w); w["x"] = synth.grid(1500, 2300.0, 840.0 / 1500)
    m, _ = orc.model(3, w["params_true"], w["plength"], w["x"])
    y = synth.make_spectrum(m, seed=3)
    out = []
    for nt in ("1", "3", "8"):
        monkeypatch.setenv("TAMCMC_SAMPLER_THREADS", nt)
        cfg = S.default_cfg(6, seed=77, Nt_learn=(5, 30, 100000), periods_learn=(1, 2), prior_fct_switch=2, dN_mixing=3)
        smp = S.Sampler(cfg, oracle_evaluator(orc, 3, w, y), w["plength"], w["params_true"], w["relax"], w["err"], sw, pp,
                        [1.0, 5.0, 0.5, 0.0])
        smp.init()
        mv, sws = smp.run(60)
        out.append((mv, sws, smp.get("vars"), smp.get("covarmat"), smp.get("sigma")))
        smp.close()
    for o in out[1:]:
        for a, b2 in zip(out[0], o):
            assert np.array_equal(a, b2)
