#!/bin/bash
# The GPU suite once per developer switch (include/tamcmc_accel.h lists them):
# every non-default tiling, launch order and fused/split choice has to give the
# same answers as the default.  Run on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/modes_check.sh > gpurun_out/modes.log 2>&1'
set -o pipefail
rc=0
for cfg in "TAMCMC_EQUAL_COST=1" "TAMCMC_ORDER=0" "TAMCMC_ORDER=1 TAMCMC_PRIO=1" \
           "TAMCMC_FUSED=0" "TAMCMC_BG_EXACT=1" \
           "TAMCMC_TILES=31 TAMCMC_TILES_GRAD=37 TAMCMC_EQUAL_COST=1" \
           "TAMCMC_TAIL=0" "TAMCMC_TAIL=60,3 TAMCMC_TAIL_L=80,5" "TAMCMC_SAMPLER_PIPELINE=2" "TAMCMC_SAMPLER_PIPELINE=3" \
           "TAMCMC_SAMPLER_ARM=0" "TAMCMC_SAMPLER_ARRIVE=0" "TAMCMC_SAMPLER_ARM=0 TAMCMC_SAMPLER_ARRIVE=0"; do
    echo "== $cfg"
    # the resource test reads the code object, not the switches
    env $cfg timeout -k 10 300 python -m pytest tests -m gpu -q -x \
        --deselect tests/test_kernel_resources.py 2>&1 | tail -3 | cut -c1-300 || rc=1
done
exit $rc
