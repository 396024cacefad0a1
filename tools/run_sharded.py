#!/usr/bin/env python3
"""One phase of a TAMCMC run with the tempered chains sharded over GPUs, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        tools/run_sharded.py --config-dir Config/default --model star.model --data star.data --out-dir out/ \\
        --phase Burn-in --nsamples 5000 --c0 1.8 [--slice 0] [--seed 1] [--restore-from B --restore 2]

Each rank evaluates its block of the temperature ladder on its own device; the parallel-tempering attempt on a pair that
straddles two ranks is a neighbour send/recv (RCCL over xGMI with --backend nccl); rank 0 writes the files, which are
the ones the single-process driver (bin/cpptamcmc_hip) writes for the same seed."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", required=True)
    ap.add_argument("--model", required=True)
    ap.add_argument("--data", required=True)
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--root-name", default="run_")
    ap.add_argument("--reader", default=None, help="io_MS_Global or io_local (default: config_default.cfg)")
    ap.add_argument("--slice", type=int, default=0)
    ap.add_argument("--phase", default="Burn-in")
    ap.add_argument("--nsamples", type=int, required=True)
    ap.add_argument("--c0", type=float, default=1.8)
    ap.add_argument("--nchains", type=int, default=None)
    ap.add_argument("--nbuffer", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--restore", type=int, default=0, help="0 none, 1 variables, 2 variables + proposal (config_presets.cfg)")
    ap.add_argument("--restore-from", default=None, help="core name of the restore files to read (e.g. B)")
    ap.add_argument("--core", default=None, help="core name of the files to write (default: first letter of the phase)")
    ap.add_argument("--restore-precision", type=int, default=17)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tamcmc_amd
    from tamcmc_amd import sharded
    from tamcmc_amd.setup_io import Setup

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dev = torch.device("cuda", local)
    else:
        dist.init_process_group(args.backend)
        dev = None

    s = Setup(args.config_dir)
    if args.reader:
        s.set("Modeling", "prior_fct_name", args.reader)
    if args.nchains:
        s.set("MALA", "Nchains", args.nchains)
    if args.nbuffer:
        s.set("Outputs", "Nbuffer", args.nbuffer)
    s.load(args.model, args.data, args.slice)
    core = args.core or args.phase[0]
    out = args.out_dir if args.out_dir.endswith("/") else args.out_dir + "/"
    if rank == 0:
        os.makedirs(out, exist_ok=True)
    dist.barrier()
    s.set("Outputs", "output_dir", out)
    s.set("Outputs", "restore_dir", out)
    s.set("Outputs", "output_root_name", f"{args.root_name}{core}_")
    s.set("Outputs", "restore_file_out", f"{args.root_name}restore_{core}_")
    s.apply_phase(args.phase, args.nsamples, args.c0)
    if args.restore >= 1:
        s.set("Outputs", "restore_file_in", f"{args.root_name}restore_{args.restore_from or core}_")
        s.set("Outputs", "do_restore_variables", 1)
        s.set("Outputs", "do_restore_proposal", 1 if args.restore >= 2 else 0)
    acc = tamcmc_amd.Accel(s.model_case, s.plength, s.x, s.y, sigma_y=s.sigma_y, likelihood_case=s.likelihood_case,
                           likelihood_p=s.likelihood_p, device_id=local)
    prog = (lambda i, n: print(f"[{i}] of {n}", flush=True)) if rank == 0 else None
    sharded.run_phase_sharded(s, acc, dist, rank, world, seed=args.seed, device=dev, restore_precision=args.restore_precision,
                              progress=prog)
    dist.barrier()
    acc.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
