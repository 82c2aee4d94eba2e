#!/bin/bash
# A/B of bench.py argument sets on one box, interleaved rounds:  bash tools/ab_cfg.sh <out-dir> <rounds> "<args A>" "<args B>" ...
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  i=0
  for a in "$@"; do
    i=$((i + 1))
    python bench.py --no-cpu-baseline --no-other $a > "$out/cfg$i.$r.json" 2> "$out/cfg$i.$r.err" || { echo "[$a] failed"; tail -3 "$out/cfg$i.$r.err"; }
    echo "[$a] round $r: $(grep -o '"ms_per_step": [0-9.]*' "$out/cfg$i.$r.json")"
  done
done
