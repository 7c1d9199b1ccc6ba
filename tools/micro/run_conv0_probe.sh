#!/bin/bash
# Runs the conv0 / packed-f32 probes on the GPU box (after tools/micro/build.sh) and writes profiles/round3_conv0_probe.txt.
cd "$(dirname "$0")/../.."
O=${1:-gpurun_out/r3/round3_conv0_probe.txt}
mkdir -p "$(dirname $O)"; : > $O
P=tools/micro
run() { echo "\$ $*" >> $O; timeout -k 10 150 "$@" 2>&1 | grep -v amdgpu.ids >> $O || exit 1; }
echo "### the product's conv0 launches (form 2), compiled as round 2 did (SLP vectoriser on)" >> $O
for nb in none attn64 attn256 hammer_tr; do run $P/conv0_probe $nb 100 2; done
echo "### the same, the neighbour attention built without MFMA instructions" >> $O
run $P/conv0_probe_nomfma attn64 100 2
echo "### the same, conv0 compiled as the library compiles it now (-fno-slp-vectorize: no packed-f32 instruction)" >> $O
run $P/conv0_probe_noslp attn64 100 2
echo "### LDS-only victims (forms 0, 1): every value read is checked" >> $O
run $P/conv0_probe attn64 100 0
run $P/conv0_probe attn64 100 1
echo "### pinned packed-f32 instructions (form 3), beside attn64 / nothing / attention without MFMA / an LDS hammer" >> $O
run $P/conv0_probe attn64 30 3
PK_FROM=9 run $P/conv0_probe none 30 3
PK_FROM=9 run $P/conv0_probe_nomfma attn64 30 3
PK_FROM=9 run $P/conv0_probe hammer_tr 30 3
sed -i -e '/^launch [0-9]* differs/,+12{/^launch \([2-9]\|[1-9][0-9]\) differs/,+12d}' $O
