#!/bin/bash
# Developer tool: soak run of bin/cpptamcmc_hip on the reference's own example (tests/golden/ref_inputs), three phases
# with the reference's default chain count; prints wall time per phase set.  Usage: tools/soak_cli.sh [Nsamples per phase]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-20000}
W=$(mktemp -d /tmp/soak.XXXXXX)
G=$R/tests/golden/ref_inputs
mkdir -p $W/run/Config && cp -r $G/Config_default $W/run/Config/default
cat > $W/run/Config/config_presets.cfg <<CFG
   force_manual_config=0;
   manual_config_file=;
   cfg_models_dir=$G/;
   cfg_out_dir=$W/out;
   processing      = Burn-in  , Learning , Acquire;
   Nsamples        = $N     ,  $N  , $N;
   c0              = 1.8      ,   1.7   ,    0;
   restore         =  0       ,    1    ,    2;
   core_out        =  B       ,    L    ,    A;
   core_in         =  B       ,    B    ,    L;
   start_index_processing=0;
   last_index_processing=2;
   table_ids=1, 2;
TF_3443483_local-v3   1;
/END;
CFG
t0=$(date +%s.%N)
timeout -k 10 900 $R/bin/cpptamcmc_hip execute 1 1 1 1 2 --root $W/run --seed 42 --quiet; rc=$?
t1=$(date +%s.%N)
python3 - <<PY
n=3*$N; dt=$t1-$t0
print(f"cpptamcmc_hip rc=$rc: 3 phases x $N samples, 10 chains, slice 1 of TF_3443483_local-v3: {dt:.1f} s wall = {n/dt:,.0f} iterations/s incl. start-up and file output")
PY
ls -la $W/out/*/outputs | head -8
rm -rf $W
