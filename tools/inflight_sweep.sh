#!/bin/bash
# bench.py at several batches in flight, all on one box (box-to-box variance is a few percent, so only numbers from one call
# compare).  Usage: tools/inflight_sweep.sh [fullhead|configs]   ->  gpurun_out/inflight_sweep_<what>.jsonl
set -e
what=${1:-fullhead}
out=gpurun_out/inflight_sweep_$what.jsonl
mkdir -p gpurun_out
: > $out
common="--steps 24 --warmup 3 --no-cpu-baseline --no-kernel-events --no-h2d"
if [ $what = fullhead ]; then
  for b in 16 64; do
    for n in 1 2 3 4 5 6; do
      timeout -k 10 240 python3 bench.py --full-head --batch $b --inflight $n $common >> $out
      echo "full head batch $b inflight $n done"
    done
  done
else
  for ci in 1 2 3; do
    for n in 2 3 4; do
      timeout -k 10 300 python3 bench.py --config-index $ci --inflight $n $common >> $out
      echo "config $ci inflight $n done"
    done
  done
fi
