#!/bin/bash
# Diagnostic (not product): build the whole library a second time with extra hipcc flags into tools/_diag/libwfl_<name>.so, for A/B runs
# in which WFL_LIB_PATH selects the build (wfl-asr_amd/_lib.py).   usage: tools/build_variant.sh <name> <extra flags ...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/_diag/var_$name
mkdir -p $out
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -I include"
pids=()
for src in wfl-asr_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc $F "$@" -c $src -o $out/$(basename ${src%.hip}).o &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $out/*.o -o tools/_diag/libwfl_$name.so
echo tools/_diag/libwfl_$name.so
