#!/bin/bash
# Build the LSTM micro-benchmarks from the CURRENT wfl-asr_amd/csrc/lstm.hip (they include the kernel source; a stale binary
# times an old kernel -- this hid a regression of the in-forward recurrence once).  Run before tools/profile_round.sh.
set -e
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -I wfl-asr_amd/csrc"
/opt/rocm/bin/hipcc $F -DWFL_LSTM_STAMPS tools/micro/lstm_bench.hip -o tools/micro/lstm_bench_stamps
/opt/rocm/bin/hipcc $F tools/micro/lstm_bench.hip -o tools/micro/lstm_bench_x
echo built tools/micro/lstm_bench_stamps tools/micro/lstm_bench_x
