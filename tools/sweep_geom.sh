#!/bin/bash
# Developer tool (GPU box): C2 step time for a few tile counts / priority settings.
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in "" "TAMCMC_PRIO=1" "TAMCMC_EQUAL_COST=1" "TAMCMC_TILES=26 TAMCMC_TILES_GRAD=26" "TAMCMC_TILES=33 TAMCMC_TILES_GRAD=40" "TAMCMC_TILES=40 TAMCMC_TILES_GRAD=48" "TAMCMC_ORDER=1" "TAMCMC_ORDER=0" $EXTRA; do
  echo "== $cfg"; env $cfg python3 tools/kstats.py c2 64 2>&1 | grep step
done
