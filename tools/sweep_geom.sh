#!/bin/bash
# Developer tool (GPU box): C2 step time for a few tile counts / settings.  CFGS="A=1;B=2 C=3" overrides the list.
cd ${GRAFT_REPO_ROOT:-$PWD}
IFS=';' read -ra L <<< "${CFGS:-;TAMCMC_TILES=14;TAMCMC_TILES=17;TAMCMC_TILES=20;TAMCMC_TILES=22;TAMCMC_TILES=24;TAMCMC_TILES=25;TAMCMC_TILES_GRAD=28;TAMCMC_TILES_GRAD=36}"
for cfg in "${L[@]}"; do
  echo "== $cfg"; env $cfg python3 tools/kstats.py ${WL:-c2} 64 2>&1 | grep step
done
