#!/bin/bash
# Runs on the GPU box (via gpurun): tests, bench, rocprofv3 kernel stats and PMC passes.
# Usage: bash tools/profile_round.sh <tag>     -> everything under gpurun_out/<tag>/
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
rm -rf $O/stats $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_clock $O/calib_fetch $O/calib_write     # a tag used twice: no mixing of runs
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
# the bench at its defaults (200 timed steps after 500 warm-up steps: settled clocks), minus the CPU and sampler legs
B="python3 $R/bench.py --no-cpu-baseline --no-sampler --no-ensemble"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 && echo "stats ok"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1 && echo "pmc sq ok"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1 && echo "pmc fetch ok"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1 && echo "pmc write ok"
# clock probe: long kernels (4096 chains) so that GRBM_GUI_ACTIVE / 8 / duration is a meaningful clock estimate
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clock -- python3 $R/bench.py --steps 3 --warmup 1 --chains 4096 --no-cpu-baseline --no-ensemble --no-sampler > $O/pmc_clock.log 2>&1 && echo "pmc clock ok"
# FETCH_SIZE / WRITE_SIZE calibration for 8-byte-per-lane streaming accesses (MI355X_MICROARCH.md: other widths are uncalibrated)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o $O/fetch_calib 2> /dev/null && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $O/fetch_calib > $O/calib_fetch.log 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- $O/fetch_calib > $O/calib_write.log 2>&1 && echo "calibration ok"; rm -f $O/fetch_calib
cd $R && python tools/summarize_profiles.py $O
# rehearsal of the N>1 bench path on this one GPU: `bench.py --gpus 2` spawns its two ranks itself (gloo, both on cuda:0)
cd $R && timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --backend gloo --single-device > $O/bench_2rank_gloo_rehearsal.json 2> $O/bench_2rank.err; echo "2-rank rehearsal rc=$?"; cat $O/bench_2rank_gloo_rehearsal.json
