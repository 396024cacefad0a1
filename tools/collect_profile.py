#!/usr/bin/env python3
"""Copies what is worth keeping of a tools/profile_round.sh output directory (gpurun_out/<tag>) into profiles/<tag>:
bench lines, test log, rocprofv3 kernel stats, condensed PMC numbers, and the raw PMC passes trimmed to the last 120
dispatches per kernel (the condensed numbers in pmc_summary.json are means over all launches of the pass)."""
import csv
import glob
import json
import os
import shutil
import sys


def trimmed(src_glob, dst, keep=120):
    files = sorted(glob.glob(src_glob), key=os.path.getmtime)      # a directory used twice holds two runs: the newest
    if not files:
        return
    rows = list(csv.DictReader(open(files[-1])))
    if not rows:
        return
    by = {}
    for r in rows:
        by.setdefault((r.get("Kernel_Name", ""), r.get("Counter_Name", "")), []).append(r)
    out = [r for v in by.values() for r in v[-keep:]]
    with open(dst, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(out)


def main(tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for f in ("bench.json", "bench_2rank_gloo_rehearsal.json", "pytest_gpu.log", "pmc_summary.json", "hbm_traffic.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    # (the gloo backend greets on stdout before bench.py's line: keep the JSON line only)
    reh = os.path.join(dst, "bench_2rank_gloo_rehearsal.json")
    if os.path.exists(reh):
        lines = [l for l in open(reh) if l.startswith("{")]
        open(reh, "w").writelines(lines)
    ks = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(dst, "kernel_stats.csv"))
    for name in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_clock", "calib_fetch", "calib_write"):
        trimmed(os.path.join(src, name, "*", "*_counter_collection.csv"), os.path.join(dst, name + ".csv"))
    hb = os.path.join(dst, "hbm_traffic.json")
    if os.path.exists(hb):
        d = json.load(open(hb))
        d["build"] = tag
        json.dump(d, open(hb, "w"), indent=1)
        shutil.copy(hb, os.path.join(root, "profiles", "hbm_traffic.json"))
    print("collected", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1])
