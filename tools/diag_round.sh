#!/bin/bash
# Developer tool (GPU box): kernel stats and SQ counters of the C2 step (tools/kstats.py), optional per-workgroup traces.
# Usage: [ENVS="A=1 B=2"] [TRACE=1] bash tools/diag_round.sh <tag>
TAG=${1:-diag}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in $ENVS; do export $e; done
K="python3 $R/tools/kstats.py ${WL:-c2} 64"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $K > $O/stats.log 2>&1 && echo "stats ok"
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "tamcmc" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2))
PY
grep step $O/stats.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $K > $O/pmc_sq.log 2>&1 && echo "pmc sq ok"
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_sq/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tamcmc" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
if [ -n "$TRACE" ]; then cd $R
for g in 0 1; do
  if [ $g = 1 ]; then export TAMCMC_TRACE_GRAD=1; else unset TAMCMC_TRACE_GRAD; fi
  echo "== trace grad=$g"; TAMCMC_ACCEL_LIB=$R/gpurun_variants/lib_trace.so TAMCMC_TRACE_FILE=/tmp/t.bin python3 tools/block_trace.py 2>&1 | grep -v amdgpu.ids
done; fi
