#!/bin/bash
# Build the LSTM micro-benchmarks from the CURRENT wfl-asr_amd/csrc/lstm.hip (they include the kernel source; a stale binary
# times an old kernel -- this hid a regression of the in-forward recurrence once).  Run before tools/profile_round.sh.
set -e
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -I wfl-asr_amd/csrc"
/opt/rocm/bin/hipcc $F -DWFL_LSTM_STAMPS tools/micro/lstm_bench.hip -o tools/micro/lstm_bench_stamps
/opt/rocm/bin/hipcc $F tools/micro/lstm_bench.hip -o tools/micro/lstm_bench_x
echo built tools/micro/lstm_bench_stamps tools/micro/lstm_bench_x
# the conv0 / packed-f32 probes: conv0 compiled as round 2 did (SLP vectoriser on), as the library does now, and
# with a neighbour attention kernel that holds no MFMA instruction
P="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result -I wfl-asr_amd/csrc -I include"
/opt/rocm/bin/hipcc $P tools/micro/conv0_probe.hip -o tools/micro/conv0_probe
/opt/rocm/bin/hipcc $P -fno-slp-vectorize tools/micro/conv0_probe.hip -o tools/micro/conv0_probe_noslp
/opt/rocm/bin/hipcc $P -DWFL_ABL_ATTN=2 tools/micro/conv0_probe.hip -o tools/micro/conv0_probe_nomfma
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result tools/micro/trans_war_probe.hip -o tools/micro/trans_war_probe
echo built tools/micro/conv0_probe tools/micro/conv0_probe_noslp tools/micro/conv0_probe_nomfma tools/micro/trans_war_probe
# round 4: the block-scaled fp8 MFMA's operand / scale layout and issue rate
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result tools/micro/mx_probe.hip -o tools/micro/mx_probe
echo built tools/micro/mx_probe
# round 4: the four-wave / 512-register form of the 192 x 256 GEMM tile (lab only)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result tools/micro/gemm4w.hip -o tools/micro/gemm4w
echo built tools/micro/gemm4w
