#!/bin/bash
# same-box A/B of two builds of the library:  bash tools/exp_ab_lib.sh <variant-name> <workload> <steps> [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
V=$R/ark_amd/lib/variants/$1/libark_amd.so; W=$2; S=$3; N=${4:-2}
for i in $(seq 1 $N); do
  for lib in default "$V"; do
    if [ "$lib" = default ]; then unset ARK_AMD_LIB; else export ARK_AMD_LIB=$lib; fi
    timeout -k 10 300 python bench.py --workload $W --no-other --no-cpu-baseline --steps $S --warmup 20 --settle 200 > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$W', '$(basename $(dirname $lib))', round(d['ms_per_step'],4))"
  done
done
