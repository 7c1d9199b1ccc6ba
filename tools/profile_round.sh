#!/bin/bash
# Collect the round's bench lines and rocprofv3 summaries on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
# Counters are collected in their own runs (one --pmc set per pass, no trace domains besides the kernel trace).
set -u
TAG=${1:-round}
PART=${2:-all}        # a: bench lines + kernel stats; b: counters, files -> .lab, recurrence micro-benchmarks (two gpurun calls of < 20 min)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
if [ $PART != b ]; then
echo "== bench lines"
timeout -k 10 400 $B > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err || exit 1
timeout -k 10 300 $B --full-head --steps 24 --warmup 3 --no-cpu-baseline > $OUT/bench_fullhead_b16.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 $B --full-head --batch 64 --steps 12 --warmup 3 --no-cpu-baseline > $OUT/bench_fullhead_b64.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 $B --config-index 2 --steps 6 --warmup 2 --cpu-clips 2 --cpu-calls 3 > $OUT/bench_cfg3.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 $B --config-index 3 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg4.json 2>> $OUT/bench.err || exit 1
timeout -k 10 300 $B --config-index 4 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_fp8.json 2>> $OUT/bench.err || exit 1
# round 4: the activation formats of the fp8 config beside its default (bf16 activations), same box
for act in fp8 fp8_pair fp8_nonscaled; do
  timeout -k 10 300 $B --config-index 4 --activation-dtype $act --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/bench_cfg5_act_$act.json 2>> $OUT/bench.err || exit 1
done
timeout -k 10 300 $B --precision high --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg2_precision_high.json 2>> $OUT/bench.err || exit 1
WFL_BENCH_FAKE_WORLD=1 timeout -k 10 200 $B --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-h2d > $OUT/bench_fake_world.json 2>> $OUT/bench.err || exit 1
echo "== kernel stats"
P="--steps 10 --warmup 2 --no-kernel-events --no-cpu-baseline --no-h2d --no-precision-high"     # (one model per profile: the precision-high leg has its own run)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks1 -o cfg2_inflight1 -- $B $P --inflight 1 > $OUT/ks1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks2 -o cfg2_inflight2 -- $B $P --inflight 2 > $OUT/ks2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ksf -o fullhead_inflight1 -- $B $P --inflight 1 --full-head > $OUT/ksf.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks5 -o cfg5_inflight1 -- $B --steps 4 --warmup 1 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --config-index 4 > $OUT/ks5.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks5a -o cfg5_act_fp8_inflight1 -- $B --steps 4 --warmup 1 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --config-index 4 --activation-dtype fp8 > $OUT/ks5a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ksh -o cfg2_precision_high -- $B --steps 6 --warmup 2 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --precision high > $OUT/ksh.log 2>&1 || exit 1
fi
if [ $PART = a ]; then echo done; exit 0; fi
echo "== pmc"
P="--steps 10 --warmup 2 --no-kernel-events --no-cpu-baseline --no-h2d --no-precision-high"
if [ $PART = b ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks1 -o cfg2_inflight1 -- $B $P --inflight 1 > $OUT/ks1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks2 -o cfg2_inflight2 -- $B $P --inflight 2 > $OUT/ks2.log 2>&1 || exit 1
fi
P4="--steps 4 --warmup 1 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --no-precision-high"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B $P4 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B $P4 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- $B $P4 > $OUT/pmc_sq.log 2>&1 || exit 1
echo "== files -> .lab end to end"
timeout -k 10 300 python3 $ROOT/tools/e2e_bench.py --files 512 2> /dev/null | tail -1 > $OUT/e2e_label_files.txt || exit 1
timeout -k 10 300 python3 $ROOT/tools/e2e_bench.py --full-head --files 768 2> /dev/null | tail -1 >> $OUT/e2e_label_files.txt || exit 1
timeout -k 10 300 python3 $ROOT/tools/e2e_bench.py --files 512 --rate 44100 2> /dev/null | tail -1 >> $OUT/e2e_label_files.txt || exit 1
WFL_GPU_INGEST=0 timeout -k 10 300 python3 $ROOT/tools/e2e_bench.py --files 256 --rate 44100 2> /dev/null | tail -1 | sed 's/^/(WFL_GPU_INGEST=0: host resampler) /' >> $OUT/e2e_label_files.txt || exit 1
timeout -k 10 300 python3 $ROOT/tools/e2e_bench.py --wavlm --files 512 2> /dev/null | tail -2 >> $OUT/e2e_label_files.txt || exit 1
echo "== lstm micro"
if [ $ROOT/wfl-asr_amd/csrc/lstm.hip -nt $ROOT/tools/micro/lstm_bench_x ] || [ ! -x $ROOT/tools/micro/lstm_bench_x ]; then
  echo "micro-benchmarks older than lstm.hip: rebuilding"; bash $ROOT/tools/micro/build.sh || exit 1
fi
(cd $ROOT/tools/micro && ./lstm_bench_stamps > $OUT/lstm_step_breakdown.txt 2>&1; ./lstm_bench_x 512 64 499 >> $OUT/lstm_step_breakdown.txt 2>&1; ./lstm_bench_x 384 64 1500 >> $OUT/lstm_step_breakdown.txt 2>&1)
find $OUT -name "*.csv" | head -30
echo done
