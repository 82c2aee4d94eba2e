#!/bin/bash
# Experiment builds: compile ONE kernel file with extra -D flags and link it with the other (default) objects into
# ark_amd/lib/variants/<name>/libark_amd.so; run with ARK_AMD_LIB=<that path> (honoured by ark_amd/_lib.py).
#   bash tools/build_variant.sh g16n4 gemm16.hip -DARK_G16_NBUF=4
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
out=$R/ark_amd/lib/variants/$name
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c "$R/ark_amd/csrc/$src" -o "$out/${src%.hip}.o"
objs=""
for o in "$R"/ark_amd/lib/obj/*.o; do
  b=$(basename "$o")
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs $out/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libark_amd.so" $objs
echo "$out/libark_amd.so"
