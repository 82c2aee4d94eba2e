#!/bin/bash
# diagonal-kernel timings (tools/diag_sweep.py) of library variants on one box, alternating rounds:
#   bash tools/ab_sweep_variants.sh <rounds> "<diag_sweep settings>" <variant names... | default>
rounds=$1; setting=$2; shift 2
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset ARK_AMD_LIB; else export ARK_AMD_LIB=$PWD/ark_amd/lib/variants/$v/libark_amd.so; fi
    echo "$v round $r: $(python tools/diag_sweep.py --rounds 2 --steps 200 "$setting" 2>/dev/null | tail -1)"
  done
done
