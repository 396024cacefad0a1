#!/bin/bash
# Developer tool (GPU box): eval-kernel time of the timing-only builds gpurun_variants/lib_ab<N>.so (TM_ABLATE bits)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in full ${VARIANTS:-ab1 ab2 ab64 ab256}; do
  if [ $v = full ]; then unset TAMCMC_ACCEL_LIB; else export TAMCMC_ACCEL_LIB=$R/gpurun_variants/lib_$v.so; fi
  rm -rf /tmp/ab_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/tools/kstats.py ${WL:-c2} 64 > /tmp/ab_$v.log 2>&1
  python3 - <<PY
import csv, glob
out=[]
for f in glob.glob("/tmp/ab_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "eval" in r["Name"]: out.append(r["Name"][5:30]+" %.2f" % (float(r["AverageNs"])/1e3))
print("$v", sorted(out))
PY
done
