#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on BASELINE.json's metric/config.

One "step" = one pass of the hot path over one batch: every chain of this rank gets its model
spectrum, tempered log-likelihood AND gradient (setup -> eval -> finalize -> backward kernels) with
params already resident in HBM.  Workload: config C2 of BASELINE.json -- model_MS_Global_a1etaa3_
HarveyLike (id 2), 1e5 bins, 64 chains per GPU (weak scaling: N GPUs carry 64*N chains of one
temperature ladder; the evaluation itself needs no collective).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (tamcmc_eval_kernel): algorithmic bytes (16 B x Nx x chains per
                  launch) / its mean duration from HIP events on the launch stream, against 8 TB/s.
                  The kernel is fp64-VALU bound, not HBM bound; "valu_frac" says how
                  close it is to the roof that actually binds.
  cpu_baseline -- the CPU oracle (OpenMP over chains like MALA.cpp:632) on this box's host cores,
                  logL only (the reference has no gradient), rank 0 / N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "MALA steps/sec (all chains) + achieved HBM GB/s, 64 chains × 1e5-bin spectrum"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=500)   # ~65 ms: the clocks of an idle GPU take ~50 ms of load to settle
    ap.add_argument("--chains", type=int, default=64, help="chains per GPU")
    ap.add_argument("--nx", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-sampler", action="store_true", help="skip the end-to-end sampler rate")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise torch.distributed even for one rank (checks the RCCL path on a one-GPU box)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    import torch
    import tamcmc_amd
    from tamcmc_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.single_device:
            local = 0
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    else:
        dist = None
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local)

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

    def timed(n, grad):
        # barrier + synchronize on both sides, exactly n steps, MAX over ranks (tests/test_shard_gloo.py)
        return shard.timed_loop(lambda: step(grad), n, lambda: torch.cuda.synchronize(dev), dist=dist,
                                device=dev if args.backend == "nccl" else None)

    for _ in range(args.warmup):
        step(True)
    acc.profile(True)
    dt = timed(args.steps, True)
    k_ms, k_n = acc.kernel_time()
    acc.profile(False)
    assert int(d_status.abs().sum().item()) == 0, "a chain reported a non-zero status"
    assert bool(torch.isfinite(d_logL).all()) and bool(torch.isfinite(d_grad).all())

    # secondary: likelihood only (what the reference's sampler actually evaluates per step)
    for _ in range(max(2, args.warmup // 4)):
        step(False)
    acc.profile(True)
    dt_l = timed(args.steps, False)
    kl_ms, kl_n = acc.kernel_time()
    acc.profile(False)

    value = total_chains * args.steps / dt
    value_l = total_chains * args.steps / dt_l
    geo = acc.geometry()
    bytes_per_launch = 16.0 * args.nx * nchains               # B_alg = 16 Nx per chain-step
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    kl_avg_s = (kl_ms / max(kl_n, 1)) * 1e-3
    achieved = bytes_per_launch / k_avg_s / 1e9
    # PMC-derived numbers cannot be collected from inside this process: they come from the committed rocprofv3
    # passes of this same command (tools/profile_round.sh -> profiles/hbm_traffic.json)
    traffic, traffic_l, valu = None, None, None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            pm = json.load(open(tfile))
            scale = nchains / 64.0 * args.nx / 100000.0          # the PMC passes ran 64 chains x 1e5 bins
            traffic = pm.get("eval_grad_bytes_per_launch") * scale
            traffic_l = pm.get("eval_logL_bytes_per_launch") * scale
            clk = pm.get("clock_GHz", {})
            simds = 256 * 4

            def busy(insts, t_s, ghz):
                # fp64 VALU instruction = 4 cycles on a SIMD-32 with 16 fp64 lanes/clk; fraction of all issue slots
                return insts * scale * 4.0 / (t_s * simds * ghz * 1e9)
            valu = {"source": "profiles/hbm_traffic.json (rocprofv3 SQ_INSTS_VALU, GRBM_GUI_ACTIVE clock estimate)",
                    "grad_kernel_issue_frac": round(busy(pm["eval_grad_valu_insts_per_launch"], k_avg_s,
                                                         clk.get("grad") or 2.4), 3),
                    "logL_kernel_issue_frac": round(busy(pm["eval_logL_valu_insts_per_launch"], kl_avg_s,
                                                         clk.get("logL") or 2.4), 3)}
        except Exception:
            traffic, traffic_l, valu = None, None, None
    roofline = {
        "bound": "hbm", "kernel": "tamcmc_eval_kernel<grad>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel_ms": round(k_avg_s * 1e3, 4), "launches": int(k_n),
        "binding_roof": "fp64 VALU, not HBM (SURVEY.md F6): x/y are shared by all chains through L2, so measured HBM "
                        "traffic (FETCH_SIZE doubled per the gfx950 calibration, + WRITE_SIZE) is ~4x below the algorithmic bytes; see 'valu' for the fraction of fp64 issue slots used",
        "valu": valu,
        "logL_only": {"achieved": round(bytes_per_launch / kl_avg_s / 1e9, 2),
                      "frac": round(bytes_per_launch / kl_avg_s / 1e9 / HBM_PEAK_GBS, 5),
                      "kernel_ms": round(kl_avg_s * 1e3, 4), "traffic": traffic_l},
    }

    # host-pointer entry point (what a host-resident sampler calls): includes the PCIe copies of params / results and
    # a stream synchronize per call; reported beside `value`, never as `value`
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
        host_path["unit"] = "chain-steps/s through tamcmc_eval_batch (host pointers, PCIe copies + sync included)"
        acc.set_stream(stream.cuda_stream)

    # the whole sampler loop (SURVEY.md 8f N1+N2: proposals, priors, accept/reject, adaptation and parallel tempering in
    # host C++; one tamcmc_eval_batch per iteration): iterations/s of MALA::execute's loop body, all chains
    sampler_rate = None
    if rank == 0 and world == 1 and not args.no_sampler:
        from tamcmc_amd import sampler as S
        cfg = S.default_cfg(nchains, seed=7, Nt_learn=(20, 60, 10 ** 9), periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], w["err"])
        smp.init()
        smp.run(200, history=False)
        n_s = max(50, min(10 * args.steps, 2000))      # ~0.25 s per leg: shorter legs scatter by 20 %
        t0 = time.perf_counter()
        moved, _ = smp.run(n_s)
        el = time.perf_counter() - t0
        sampler_rate = {"iterations_per_s": round(n_s / el, 1), "chain_steps_per_s": round(nchains * n_s / el, 1),
                        "acceptance_cold_chain": round(float(moved[:, 0].mean()), 3),
                        "what": "adaptive Metropolis + parallel tempering (the reference's 'MALA' has no gradient), host C++ "
                                "sampler, likelihood on the GPU through tamcmc_eval_batch (host pointers); proposal adapted "
                                "every iteration (Burn-in / Learning phases)"}
        smp.close()
        # Acquire phase: the proposal is frozen (config_presets.cpp:87-92), so no Cholesky per iteration
        cfg = S.default_cfg(nchains, seed=7, Nt_learn=(10 ** 9, 10 ** 9 + 1, 10 ** 9 + 2), periods_learn=(1, 1), prior_fct_switch=0, dN_mixing=1)
        smp = S.Sampler(cfg, acc, w["plength"], w["params_true"], w["relax"], 0.05 * w["err"])
        smp.init()
        smp.run(200, history=False)
        t0 = time.perf_counter()
        smp.run(n_s, history=False)
        el = time.perf_counter() - t0
        sampler_rate["acquire_phase_iterations_per_s"] = round(n_s / el, 1)
        sampler_rate["acquire_phase_chain_steps_per_s"] = round(nchains * n_s / el, 1)
        smp.close()

    cpu = None
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
        cpu = {"value": round(nchains * n_it / el, 2), "unit": "chain-steps/s", "cores": int(cores), "kind": "port",
               "sample": f"{n_it} iterations x {nchains} chains x {args.nx} bins, logL only (the reference has no "
                         f"gradient), OpenMP over chains, {el:.1f} s"}

    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 1), "unit": "chain-steps/s (model+logL+grad)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: model_MS_Global_a1etaa3_HarveyLike (id 2), 21 modes l=0..2, 56 params, "
                                   f"{args.nx} bins, {nchains} chains per GPU, trunc_c=20, logL + gradient over "
                                   f"{nvars} variables, params resident in HBM",
                       "chains_total": total_chains, "geometry_logL": geo},
            "logL_only": {"value": round(value_l, 1), "unit": "chain-steps/s (model+logL)",
                          "ms_per_step": round(dt_l / args.steps * 1e3, 4)},
            "host_path": host_path, "sampler": sampler_rate, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    acc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
