#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass over tools/evtime.py and the per-kernel means of every counter.
# usage: bash tools/pmc_pass.sh <outdir-under-gpurun_out> COUNTER [COUNTER ...]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; shift
mkdir -p $O && rm -rf $O/pmc_tmp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_tmp -- python3 $R/tools/evtime.py c2 64 1 > $O/pmc_tmp.log 2>&1 || { tail -5 $O/pmc_tmp.log; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/pmc_tmp/*/*_counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, n) in sorted(acc.items()):
    if "eval" in kn or "setup" in kn or "backward" in kn:
        print(f"{kn:42s} {cn:28s} {s / n:16.1f}  ({n} launches)")
PY
