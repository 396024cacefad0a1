cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base; do
  lib=$R/tamcmc-c-_amd/libtamcmc_accel.so
  for wl in c2 c4 c1; do
  rm -rf /tmp/su_$v; TAMCMC_ACCEL_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/su_$v -- python3 $R/tools/kstats.py $wl 64 > /tmp/su_$v.log 2>&1
  echo "== $v $wl"; grep step /tmp/su_$v.log; f=$(find /tmp/su_$v -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tamcmc' in r['Name']: print("   %-28s %5s calls  %8.2f us" % (r['Name'][:28], r['Calls'], float(r['AverageNs'])/1e3))
PY
  done
done
