#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on BASELINE.json's metric/config.

One "step" = one pass of the hot path over one batch: every chain of this rank gets its model
spectrum, tempered log-likelihood AND gradient (setup -> eval -> backward kernels) with params already
resident in HBM.  Workload: config C2 of BASELINE.json -- model_MS_Global_a1etaa3_HarveyLike (id 2),
1e5 bins, 64 chains per GPU (weak scaling: N GPUs carry 64*N chains of one temperature ladder; the
evaluation itself needs no collective).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank processes
itself (fresh children, before anything touches the GPU), one per GPU, backend "nccl" (= RCCL);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with, besides the contract keys:
  roofline        -- dominant kernel (tamcmc_eval_kernel<grad>): algorithmic bytes (16 B x Nx x chains per launch)
                     over its mean duration from HIP events on the launch stream, against 8 TB/s; `bound` names the
                     roof that actually binds (fp64 VALU issue) and `valu` the fraction of it in use, with the core
                     clock measured in this run beside the timed kernels (tamcmc_ctx_clock_probe_*).
  logL_only       -- the likelihood-only step (what the reference's sampler evaluates per iteration).
  sampler         -- end-to-end loop on one GPU (N = 1 only).
  sampler_sharded -- the same loop with 64*N chains sharded over the N ranks, parallel tempering every iteration,
                     boundary pairs exchanged between neighbours (RCCL send/recv), loop in C++
                     (tamcmc_sampler_run_sharded); reports the replicated random-stream term per iteration.
  ensemble        -- BASELINE config C5: 32 independent synthetic stars x 16 chains, 32/N stars per rank held in ONE
                     multi-spectrum context (tamcmc_ctx_set_spectra), no communication in the timed region, one
                     max-over-ranks time (what the reference does with a Slurm array over stars,
                     scripts/slurm/job_array.sh:8,24) -- strong scaling: the 32 stars are fixed, N divides them.
  cpu_baseline    -- the CPU oracle (OpenMP over chains like MALA.cpp:632) on this box's host cores, logL only
                     (the reference has no gradient), rank 0 / N=1 only, bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "MALA steps/sec (all chains) + achieved HBM GB/s, 64 chains × 1e5-bin spectrum"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs; one wave64 fp64 VALU instruction occupies a SIMD for 4 cycles
PREWARM_MS = 80.0              # an idle MI355X needs ~50 ms of load before its clocks settle (profiles/README.md)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--chains", type=int, default=64, help="chains per GPU")
    ap.add_argument("--nx", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-sampler", action="store_true", help="skip the end-to-end sampler legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise torch.distributed even for one rank (checks the RCCL path on a one-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--no-ensemble", action="store_true", help="skip the C5 ensemble leg")
    ap.add_argument("--ensemble-stars", type=int, default=32, help="stars of the ensemble leg (BASELINE C5: 32), spread over the ranks")
    ap.add_argument("--strict-exit", action="store_true", help="exit with code 3 (after printing the line) when the sharded sampler leg hit its deadline; "
                    "by default the line carries the error and the exit code stays 0, so that a stuck exchange cannot cost the measurement")
    ap.add_argument("--sharded-seconds", type=float, default=150.0, help="deadline of the sharded sampler leg (a hung exchange must not cost the line)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside a launcher: N fresh children, one per GPU (rank r -> cuda:r), rendezvous on
    127.0.0.1.  This process never touches the GPU; it forwards rank 0's line and the first failure."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(len(procs)))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the others", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    run_rank(args)


def run_rank(args):
    import numpy as np
    import torch
    import tamcmc_amd
    from tamcmc_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or "
                 f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.single_device:
            local = 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus
    else:
        dist = None
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local)
    comm_dev = dev if args.backend == "nccl" else None      # tensors handed to the collectives

    # ---- workload (identical on every rank; each rank takes its slice of the temperature ladder)
    w = synth.workload_c2(model_case=2, Nx=args.nx)
    nchains = args.chains
    total_chains = nchains * world
    P_all = synth.chain_params(w, total_chains)
    T_all = synth.temperatures(total_chains)
    sl = shard.chain_slice(rank, world, nchains)           # contiguous in temperature
    with tamcmc_amd.Accel(2, w["plength"], w["x"], np.ones(args.nx), device_id=local) as a0:
        m_true, st = a0.model_explicit(w["params_true"])   # product path builds the synthetic truth
    assert st == 0
    y = synth.make_spectrum(m_true)

    acc = tamcmc_amd.Accel(2, w["plength"], w["x"], y, device_id=local)
    acc.set_vars(w["index_to_relax"])
    nvars = int(w["index_to_relax"].size)
    stream = torch.cuda.current_stream(dev)
    acc.set_stream(stream.cuda_stream)
    d_params = torch.from_numpy(P_all[sl].copy()).to(dev)
    d_T = torch.from_numpy(T_all[sl].copy()).to(dev)
    d_logL = torch.empty(nchains, dtype=torch.float64, device=dev)
    d_grad = torch.empty(nchains, nvars, dtype=torch.float64, device=dev)
    d_status = torch.empty(nchains, dtype=torch.int32, device=dev)

    def step(grad=True):
        acc.eval_batch_device(nchains, d_params.data_ptr(), d_T.data_ptr(), d_logL.data_ptr(),
                              d_grad.data_ptr() if grad else 0, d_status.data_ptr())

    def sync():
        torch.cuda.synchronize(dev)

    def timed(n, grad):
        # barrier + synchronize on both sides, exactly n steps, MAX over ranks (tests/test_shard_gloo.py)
        return shard.timed_loop(lambda: step(grad), n, sync, dist=dist, device=comm_dev)

    def prewarm(grad):
        # untimed: load the GPU for a fixed WALL time so that a short --warmup still measures settled clocks
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < PREWARM_MS:
            for _ in range(20):
                step(grad)
            sync()

    def clock_under_load(n, grad):
        # the same steps again, untimed, with the one-wave clock probe beside them on its own stream
        est_ms = max(1.0, min(200.0, 0.8 * n * (0.13 if grad else 0.07)))
        sync()
        acc.clock_probe_begin(est_ms)
        for _ in range(n):
            step(grad)
        sync()
        ghz, sec = acc.clock_probe_end()
        return ghz

    # HIP events around every 4th eval launch of the timed region (an event pair costs ~3 us of stream time: around every
    # launch it took 5 % off `value`)
    ev_stride = 4 if args.steps >= 8 else 1
    prewarm(True)
    for _ in range(args.warmup):
        step(True)
    acc.profile(ev_stride)
    dt = timed(args.steps, True)
    k_ms, k_n = acc.kernel_time()
    acc.profile(False)
    assert int(d_status.abs().sum().item()) == 0, "a chain reported a non-zero status"
    assert bool(torch.isfinite(d_logL).all()) and bool(torch.isfinite(d_grad).all())
    ghz_g = clock_under_load(max(args.steps, 100), True)

    # secondary: likelihood only (what the reference's sampler actually evaluates per step)
    prewarm(False)
    for _ in range(max(2, args.warmup // 4)):
        step(False)
    acc.profile(ev_stride)
    dt_l = timed(args.steps, False)
    kl_ms, kl_n = acc.kernel_time()
    acc.profile(False)
    ghz_l = clock_under_load(max(args.steps, 100), False)

    value = total_chains * args.steps / dt
    value_l = total_chains * args.steps / dt_l
    geo = acc.geometry()
    bytes_per_launch = 16.0 * args.nx * nchains               # B_alg = 16 Nx per chain-step
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    kl_avg_s = (kl_ms / max(kl_n, 1)) * 1e-3
    achieved = bytes_per_launch / k_avg_s / 1e9
    # Instruction counts and HBM traffic need the PMC counters, which cannot be collected from inside this process:
    # they come from the committed rocprofv3 passes of this same command (tools/profile_round.sh ->
    # profiles/hbm_traffic.json).  Kernel time and core clock are this run's.
    traffic, traffic_l, valu, pmc_note = None, None, None, None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            pm = json.load(open(tfile))
            if pm.get("library") != tamcmc_amd.capi.version():
                # counters of another build say nothing about this one's kernels: no instruction counts, no traffic
                raise LookupError(f"profiles/hbm_traffic.json is of {pm.get('library')!r}, the loaded library is {tamcmc_amd.capi.version()!r}")
            scale = nchains / 64.0 * args.nx / 100000.0          # the PMC passes ran 64 chains x 1e5 bins
            traffic = pm.get("eval_grad_bytes_per_launch") * scale
            traffic_l = pm.get("eval_logL_bytes_per_launch") * scale

            def busy(insts, t_s, ghz):
                return insts * scale * 4.0 / (t_s * SIMDS * ghz * 1e9)
            valu = {"grad_kernel_issue_frac": round(busy(pm["eval_grad_valu_insts_per_launch"], k_avg_s, ghz_g), 3),
                    "logL_kernel_issue_frac": round(busy(pm["eval_logL_valu_insts_per_launch"], kl_avg_s, ghz_l), 3),
                    "clock_GHz": {"grad": round(ghz_g, 3), "logL": round(ghz_l, 3),
                                  "source": "this run: s_memtime / s_memrealtime of a one-wave probe beside the same steps (tamcmc_ctx_clock_probe_*)"},
                    "kernel_time_source": "this run (HIP events on the launch stream)",
                    "insts_per_launch": {"grad": pm["eval_grad_valu_insts_per_launch"] * scale, "logL": pm["eval_logL_valu_insts_per_launch"] * scale,
                                         "source": "committed rocprofv3 --pmc SQ_INSTS_VALU pass of this command (profiles/hbm_traffic.json, build "
                                                   + str(pm.get("build", "?")) + ")"},
                    "model": "wave-level VALU instructions x 4 cycles / (1024 SIMDs x clock x kernel time): an upper bound, not every VALU instruction is a 4-cycle fp64 one"}
        except Exception as e:           # noqa: BLE001
            traffic, traffic_l, valu = None, None, None
            pmc_note = f"{type(e).__name__}: {e}"
    roofline = {
        "bound": "fp64-valu", "kernel": "tamcmc_eval_kernel<grad>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel_ms": round(k_avg_s * 1e3, 4), "launches": int(k_n), "launches_timed_every": ev_stride,
        "note": "achieved / peak / frac are the algorithmic bytes (16 B x Nx x chains) against the HBM peak, as the north-star "
                "asks; the kernel is bound by fp64 VALU issue (SURVEY.md F6): x / y are shared by all chains through L2, so measured "
                "HBM traffic (FETCH_SIZE doubled per the gfx950 calibration, + WRITE_SIZE) is ~4x below the algorithmic bytes",
        "valu": valu, "pmc_note": pmc_note, "library": tamcmc_amd.capi.version(),
        "logL_only": {"achieved": round(bytes_per_launch / kl_avg_s / 1e9, 2),
                      "frac": round(bytes_per_launch / kl_avg_s / 1e9 / HBM_PEAK_GBS, 5),
                      "kernel_ms": round(kl_avg_s * 1e3, 4), "traffic": traffic_l},
    }

    # host-pointer entry point (what a host-resident sampler calls): includes the PCIe copies of params / results and
    # the wait per call; reported beside `value`, never as `value`
    host_path = None
    if rank == 0:
        Ph, Th = np.ascontiguousarray(P_all[sl]), np.ascontiguousarray(T_all[sl])
        acc.set_stream(0)
        n_h = max(10, min(args.steps, 200))
        for g in (False, True):
            acc.eval_batch(Ph, Th, grad=g)
            dts = np.empty(n_h)
            for k in range(n_h):
                t0 = time.perf_counter()
                acc.eval_batch(Ph, Th, grad=g)
                dts[k] = time.perf_counter() - t0
            key = "logL_grad" if g else "logL_only"
            host_path = dict(host_path or {}, **{key: round(nchains * n_h / dts.sum(), 1),
                                                 key + "_median_call_us": round(float(np.median(dts)) * 1e6, 1),
                                                 key + "_max_call_us": round(float(dts.max()) * 1e6, 1)})
        host_path["unit"] = "chain-steps/s through tamcmc_eval_batch (host pointers, PCIe copies + wait included)"
        acc.set_stream(stream.cuda_stream)

    # ---- BASELINE config C5: the ensemble.  Every rank holds its share of the stars in one multi-spectrum context;
    # nothing is exchanged in the timed region ("replicas only": independent stars).
    ensemble = None
    if not args.no_ensemble:
        try:
            ensemble = ensemble_leg(args, w, m_true, dist, rank, world, local, dev, comm_dev, stream)
        except Exception as e:          # noqa: BLE001 -- report, keep the line
            ensemble = {"error": f"{type(e).__name__}: {e}"}

    # the whole sampler loop (SURVEY.md 8f N1+N2: proposals, priors, accept/reject, adaptation and parallel tempering in
    # host C++; one likelihood batch per iteration): iterations/s of MALA::execute's loop body, all chains
    sampler_rate = None
    if rank == 0 and world == 1 and not args.no_sampler:
        from tamcmc_amd import sampler as S
        acc.set_stream(0)      # the context's own (non-blocking) stream: the loop keeps launches armed behind a gate kernel
        cfg = S.default_cfg(nchains, seed=7, Nt_learn=(20, 60, 10 ** 9), periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], w["err"])
        smp.init()
        smp.run(200, history=False)
        n_s = max(50, min(10 * args.steps, 2000))      # ~0.2 s per segment: shorter ones scatter by 20 %
        # three segments, the median reported (the host threads share a many-tenant machine: single segments scatter by +-5 %)
        seg = []
        for _ in range(3):
            t0 = time.perf_counter()
            moved, _ = smp.run(n_s)
            seg.append(time.perf_counter() - t0)
        el = sorted(seg)[1]
        sampler_rate = {"iterations_per_s": round(n_s / el, 1), "chain_steps_per_s": round(nchains * n_s / el, 1),
                        "segments_iterations_per_s": [round(n_s / e, 1) for e in seg],
                        "acceptance_cold_chain": round(float(moved[:, 0].mean()), 3),
                        "what": "adaptive Metropolis + parallel tempering (the reference's 'MALA' has no gradient), host C++ "
                                "sampler, likelihood on the GPU through tamcmc_eval_batch_begin/_end (host pointers); proposal adapted "
                                "every iteration (Burn-in / Learning phases)"}
        smp.close()
        # Acquire phase: the proposal is frozen (config_presets.cpp:87-92), so no Cholesky per iteration
        cfg = S.default_cfg(nchains, seed=7, Nt_learn=(10 ** 9, 10 ** 9 + 1, 10 ** 9 + 2), periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], 0.05 * w["err"])
        smp.init()
        smp.run(200, history=False)
        seg = []
        for _ in range(3):
            t0 = time.perf_counter()
            smp.run(n_s, history=False)
            seg.append(time.perf_counter() - t0)
        el = sorted(seg)[1]
        sampler_rate["acquire_phase_segments_iterations_per_s"] = [round(n_s / e, 1) for e in seg]
        sampler_rate["acquire_phase_iterations_per_s"] = round(n_s / el, 1)
        sampler_rate["acquire_phase_chain_steps_per_s"] = round(nchains * n_s / el, 1)
        smp.close()

    cpu = None
    out = None
    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "chain-steps/s (model+logL+grad)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: model_MS_Global_a1etaa3_HarveyLike (id 2), 21 modes l=0..2, 56 params, "
                                   f"{args.nx} bins, {nchains} chains per GPU, trunc_c=20, logL + gradient over "
                                   f"{nvars} variables, params resident in HBM",
                       "chains_total": total_chains, "geometry_logL": geo,
                       "prewarm_ms": PREWARM_MS},
            "logL_only": {"value": round(value_l, 1), "unit": "chain-steps/s (model+logL)",
                          "ms_per_step": round(dt_l / args.steps * 1e3, 4)},
            "host_path": host_path, "ensemble": ensemble, "sampler": sampler_rate, "sampler_sharded": None, "roofline": roofline,
            "cpu_baseline": cpu,
        }

    # ---- sharded sampler: 64 N chains over the N ranks, PT every iteration, boundary pairs between neighbours.
    # Guarded by a deadline: a stuck exchange must not cost the line above (rank 0 prints it and every rank leaves).
    printed = threading.Lock()

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            if note is not None:
                out["sampler_sharded"] = {"error": note}
            print(json.dumps(out), flush=True)

    if not args.no_sampler:
        def bail():
            emit(f"the sharded sampler leg did not finish within {args.sharded_seconds:.0f} s")
            os._exit(3 if args.strict_exit else 0)
        dog = threading.Timer(args.sharded_seconds, bail)
        dog.daemon = True
        dog.start()
        try:
            res = sharded_sampler_leg(args, acc, w, dist, rank, world, nchains, dev, comm_dev)
            if rank == 0:
                out["sampler_sharded"] = res
        except Exception as e:          # noqa: BLE001 -- report, keep the line
            if rank == 0:
                out["sampler_sharded"] = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()

    # the CPU baseline last: its OpenMP threads (one per host core) would otherwise still be spinning beside the sampler legs
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc       # reported baseline only; never on the product path
        cores = orc.max_threads()
        Pc, Tc = P_all[sl], T_all[sl]
        orc.generate_batch(2, w["plength"], w["x"], y, Pc, Tc)          # warm-up
        n_it, t0 = 0, time.perf_counter()
        while True:
            orc.generate_batch(2, w["plength"], w["x"], y, Pc, Tc)
            n_it += 1
            el = time.perf_counter() - t0
            if el >= args.cpu_seconds or n_it >= 2000:
                break
        out["cpu_baseline"] = cpu = {"value": round(nchains * n_it / el, 2), "unit": "chain-steps/s", "cores": int(cores),
               "threads_busy": int(min(cores, nchains)), "kind": "port",
               "sample": f"{n_it} iterations x {nchains} chains x {args.nx} bins, logL only (the reference has no "
                         f"gradient), OpenMP over chains (one thread per chain at most: threads_busy), {el:.1f} s"}

    emit()
    acc.close()
    if dist is not None:
        dist.destroy_process_group()


def ensemble_leg(args, w, m_true, dist, rank, world, local, dev, comm_dev, stream):
    """BASELINE.json config C5: `--ensemble-stars` (32) independent synthetic stars x 16 tempered chains each, the stars
    dealt to the ranks in contiguous blocks.  A rank's stars share one context (same grid and layout, one spectrum each:
    tamcmc_ctx_set_spectra / _set_chain_spectrum), so its 16 * stars_per_rank chains are ONE batch per step; a chain's
    result is bit for bit what a context of its own would return (tests/test_baseline_configs_gpu.py).  Timed like the
    main leg: barrier + synchronize on both sides, exactly --steps steps, MAX over ranks; no communication inside."""
    import numpy as np
    import torch
    import tamcmc_amd
    from tamcmc_amd import shard, synth
    stars, nch = int(args.ensemble_stars), 16
    if stars % world != 0:
        return {"skipped": f"{stars} stars do not divide over {world} ranks"}
    ns = stars // world
    mine = range(rank * ns, (rank + 1) * ns)
    Y = np.stack([synth.make_spectrum(m_true, seed=1000 + k) for k in mine])
    P = synth.chain_params(w, stars * nch, seed=11)[rank * ns * nch:(rank + 1) * ns * nch]
    T = np.tile(synth.temperatures(nch), ns)
    n = ns * nch
    nvars = int(w["index_to_relax"].size)
    with tamcmc_amd.Accel(2, w["plength"], w["x"], Y[0], device_id=local) as acc:
        acc.set_vars(w["index_to_relax"])
        acc.set_spectra(Y)
        acc.set_chain_spectrum(np.repeat(np.arange(ns, dtype=np.int32), nch))
        acc.set_stream(stream.cuda_stream)
        dP = torch.from_numpy(np.ascontiguousarray(P)).to(dev)
        dT = torch.from_numpy(np.ascontiguousarray(T)).to(dev)
        dL = torch.empty(n, dtype=torch.float64, device=dev)
        dG = torch.empty(n, nvars, dtype=torch.float64, device=dev)
        dS = torch.empty(n, dtype=torch.int32, device=dev)

        def step(grad):
            acc.eval_batch_device(n, dP.data_ptr(), dT.data_ptr(), dL.data_ptr(), dG.data_ptr() if grad else 0, dS.data_ptr())

        def sync():
            torch.cuda.synchronize(dev)

        out = {"stars_total": stars, "chains_per_star": nch, "stars_per_rank": ns, "chains_per_rank": n, "scaling": "strong",
               "what": "32/N stars per rank in one multi-spectrum context, one batch per step, no communication in the timed region"}
        steps = max(10, min(args.steps, 200))
        for grad, key in ((True, "chain_steps_per_s"), (False, "logL_only_chain_steps_per_s")):
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < PREWARM_MS:
                for _ in range(5):
                    step(grad)
                sync()
            dt = shard.timed_loop(lambda: step(grad), steps, sync, dist=dist, device=comm_dev)
            out[key] = round(stars * nch * steps / dt, 1)
            out[("ms_per_step" if grad else "logL_only_ms_per_step")] = round(dt / steps * 1e3, 4)
        assert int(dS.abs().sum().item()) == 0 and bool(torch.isfinite(dL).all()) and bool(torch.isfinite(dG).all())
        out["steps"] = steps
        acc.set_stream(0)
    return out


def sharded_sampler_leg(args, acc, w, dist, rank, world, nchains, dev, comm_dev):
    """64*world chains of ONE temperature ladder, 64 per rank; every iteration: one likelihood batch on this rank's GPU,
    then the parallel-tempering attempt (MALA.cpp:381-445) -- a pair that straddles two ranks is one record each way
    between neighbours.  The loop runs in C++ (tamcmc_sampler_run_sharded); Python is entered per boundary exchange."""
    import numpy as np
    import torch
    from tamcmc_amd import sampler as S
    total = nchains * world
    acc.set_stream(0)
    exchange = S.neighbour_exchange(dist, nchains, comm_dev) if world > 1 else None
    if world > 1:
        # first use of a neighbour link sets the connection up (RCCL: lazily, ~100 ms): do it outside the timed region,
        # pairs (0,1),(2,3).. first, then (1,2),(3,4)..
        dummy = np.zeros(8)
        for phase in (0, 1):
            if rank % 2 == phase and rank + 1 < world:
                exchange(rank * nchains, (rank + 1) * nchains, dummy)
            elif rank % 2 != phase and rank - 1 >= 0:
                exchange(rank * nchains, (rank - 1) * nchains, dummy)
    res = {"chains_total": total, "chains_per_rank": nchains, "dN_mixing": 1,
           "what": "adaptive Metropolis + parallel tempering, chains sharded over the ranks (contiguous temperature blocks), "
                   "likelihood on each rank's GPU, boundary pairs exchanged with torch.distributed send/recv "
                   f"(backend {args.backend if world > 1 else 'none: one rank'}), loop in C++ (tamcmc_sampler_run_sharded)"}
    n_s = max(50, min(10 * args.steps, 2000))
    for name, learn, err in (("acquire_phase", (10 ** 9, 10 ** 9 + 1, 10 ** 9 + 2), 0.05 * w["err"]), ("learning_phase", (20, 60, 10 ** 9), w["err"])):
        cfg = S.default_cfg(total, chain_offset=rank * nchains, Nchains_local=nchains, seed=7, Nt_learn=learn, periods_learn=(1, 1),
                            prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], err)
        smp.init()
        smp.run_sharded(100, exchange)
        smp.set_timing(True)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        _, _, swaps = smp.run_sharded(n_s, exchange, history=True)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        sec, its = smp.timing()
        us = {k: round(v / max(its, 1) * 1e6, 2) for k, v in sec.items()}
        res[name] = {"iterations_per_s": round(n_s / el, 1), "chain_steps_per_s": round(total * n_s / el, 1), "iterations": n_s,
                     "rank0_us_per_iteration": us,
                     "rank0_boundary_exchanges": int(np.count_nonzero((swaps >= 0) & ((swaps // 2 + 1) % nchains == 0))) if world > 1 else 0,
                     "replicated_stream_us_per_iteration": us["foreign_draws"]}
        smp.close()
    return res


if __name__ == "__main__":
    main()
