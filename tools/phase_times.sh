#!/bin/bash
# Developer tool (GPU box): setup / backward kernel time of early-return builds (gpurun_variants/lib_{su,bw}N.so)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-su1 su3 su2 su4 bw1 bw2 bw3 bw4 bw5 full}; do
  if [ $v = full ]; then unset TAMCMC_ACCEL_LIB; else export TAMCMC_ACCEL_LIB=$R/gpurun_variants/lib_$v.so; fi
  rm -rf /tmp/ph_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph_$v -- python3 $R/tools/kstats.py c2 64 > /tmp/ph_$v.log 2>&1
  python3 - <<PY
import csv, glob
out=[]
for f in glob.glob("/tmp/ph_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "setup" in r["Name"] or "backward" in r["Name"]: out.append(r["Name"][:22]+" %.2f" % (float(r["AverageNs"])/1e3))
print("$v", out)
PY
done
