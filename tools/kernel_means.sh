#!/bin/bash
# Developer tool (GPU box): rocprofv3 mean duration of every tamcmc kernel over tools/kstats.py runs.
# Usage: bash tools/kernel_means.sh [c2|c4|c1] [chains]   (TAMCMC_ACCEL_LIB selects a variant build)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/km; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km -- python3 $R/tools/kstats.py ${1:-c2} ${2:-64} > /tmp/km.log 2>&1
grep step /tmp/km.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/km/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "tamcmc" in r["Name"]: print("   %-30s %6s calls %8.2f us" % (r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
