#!/bin/bash
# A/B of library variants on one box: bash tools/ab_variants.sh <out-dir> <rounds> <bench args...> -- <variant names...>
# ("default" = the in-tree library).  Prints ms/step per variant and round (rule: same box, interleaved rounds).
out=$1; rounds=$2; shift 2
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset ARK_AMD_LIB; else export ARK_AMD_LIB=$PWD/ark_amd/lib/variants/$v/libark_amd.so; fi
    python bench.py --no-cpu-baseline --no-other "${args[@]}" > "$out/$v.$r.json" 2> "$out/$v.$r.err" || { echo "$v failed"; tail -3 "$out/$v.$r.err"; }
    echo "$v round $r: $(grep -o '"ms_per_step": [0-9.]*' "$out/$v.$r.json")"
  done
done
