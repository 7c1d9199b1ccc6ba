#!/bin/bash
# Diagnostic (not product): attention_pp.hip with -DPP_ABL=n linked against the product build's other objects -> tools/_diag/libwfl_attnpp_abl<n>.so
set -e
cd "$(dirname "$0")/.."
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -I include"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DPP_ABL=$n -c wfl-asr_amd/csrc/attention_pp.hip -o tools/_diag/attention_pp_abl$n.o &&
    objs=$(ls wfl-asr_amd/csrc/build/*.o | grep -v "/attention_pp.o") &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_diag/attention_pp_abl$n.o -o tools/_diag/libwfl_attnpp_abl$n.so && echo built $n ) &
done
wait
