#!/bin/bash
# SQ counters and HBM traffic of the precision-high run (one --pmc set per pass, kernel trace only): gpurun_out/r4h/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
P4="--steps 4 --warmup 1 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --precision high"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- $B $P4 > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B $P4 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B $P4 > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*counter_collection.csv"
echo done
