#!/bin/bash
# PMC passes of a small tool on the GPU box:  bash tools/pmc_tool.sh <out-name> <counters...> -- <python script and args>
# (one rocprofv3 --pmc pass with --kernel-trace only; prints the per-kernel averages)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; shift
C=""
while [ "$1" != "--" ]; do C="$C $1"; shift; done
shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
S=$1; shift
rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O" -o p -- python3 "$R/$S" "$@" > "$O/run.log" 2>&1
python3 - "$O" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY
