set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
timeout -k 10 400 $B > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err || exit 1
timeout -k 10 300 $B --precision high --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg2_precision_high.json 2>> $OUT/bench.err || exit 1
WFL_TRI=0 timeout -k 10 300 $B --precision high --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-h2d > $OUT/bench_cfg2_precision_high_tri0.json 2>> $OUT/bench.err || exit 1
P="--steps 10 --warmup 2 --no-kernel-events --no-cpu-baseline --no-h2d --no-precision-high"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks1 -o cfg2_inflight1 -- $B $P --inflight 1 > $OUT/ks1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ksh -o cfg2_precision_high -- $B --steps 6 --warmup 2 --no-kernel-events --no-cpu-baseline --no-h2d --inflight 1 --precision high > $OUT/ksh.log 2>&1 || exit 1
find $OUT -name "*kernel_stats.csv"
echo done
