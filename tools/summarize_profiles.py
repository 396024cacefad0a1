#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into pmc_summary.json (per-kernel means over the
64-chain launches): instruction counts, HBM traffic from FETCH_SIZE / WRITE_SIZE, estimated shader clock."""
import collections
import csv
import glob
import json
import os
import sys


def rows(d):
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        yield from csv.DictReader(open(f))


def library_version():
    """tamcmc_version() of the library the passes ran with (it carries a hash of the kernel sources): bench.py uses the
    instruction counts only while this is the library it has loaded."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tamcmc_amd
    return tamcmc_amd.capi.version()


def main(out_dir):
    res = collections.defaultdict(dict)
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows(os.path.join(out_dir, sub)):
            if "tamcmc" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, cs in agg.items():
            for c, v in cs.items():
                gmax = max(g for g, _, _ in v)
                full = [(val, dur) for g, val, dur in v if g == gmax]     # drop the 1-chain model_explicit launch
                res[k][c] = sum(x for x, _ in full) / len(full)
                res[k].setdefault("launches", len(full))
                res[k]["mean_ns_profiled"] = sum(d for _, d in full) / len(full)
    clock = {}
    for r in rows(os.path.join(out_dir, "pmc_clock")):
        if "tamcmc_eval_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if dur > 1_000_000:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                clock.setdefault(k, []).append(float(r["Counter_Value"]) / 8.0 / dur)   # cycles per ns = GHz
    out = {"kernels": res, "clock_GHz_long_kernels": {k: sum(v) / len(v) for k, v in clock.items()},
           "note": "FETCH_SIZE/WRITE_SIZE are in KB (x1024 for bytes).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
                   "reports half the bytes of coalesced streaming reads; calibrated for this code's 8-byte-per-lane loads with "
                   "tools/fetch_calib.hip (ratio 0.50001, profiles/r01e/fetch_calibration.txt), WRITE_SIZE exact (1.0): "
                   "hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) x 1024"}
    for k, d in res.items():
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
    json.dump(out, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    # the condensed file bench.py reads for roofline.traffic / roofline.valu
    def pick(prefix):
        # tamcmc_eval_kernel<GRAD, GEN>: the variant with the most launches (the bench's; a one-off model_explicit launch
        # uses the generic variant)
        c = [k for k in res if k.startswith(prefix)]
        return max(c, key=lambda k: res[k].get("launches", 0)) if c else None
    kg, kf = pick("tamcmc_eval_kernel<true"), pick("tamcmc_eval_kernel<false")
    if kg and kf and "hbm_bytes_per_launch" in res[kg] and "SQ_INSTS_VALU" in res[kg]:
        hb = {"source": "%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, separate passes, "
                        "64 chains x 1e5 bins)" % os.path.basename(os.path.normpath(out_dir)),
              "eval_grad_kernel": kg, "eval_logL_kernel": kf,
              "eval_grad_bytes_per_launch": res[kg]["hbm_bytes_per_launch"],
              "eval_logL_bytes_per_launch": res[kf]["hbm_bytes_per_launch"],
              "eval_grad_valu_insts_per_launch": res[kg]["SQ_INSTS_VALU"],
              "eval_logL_valu_insts_per_launch": res[kf]["SQ_INSTS_VALU"],
              "clock_GHz": {"grad": out["clock_GHz_long_kernels"].get(kg), "logL": out["clock_GHz_long_kernels"].get(kf)},
              "library": library_version(), "note": out["note"]}
        json.dump(hb, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main(sys.argv[1])
