#!/bin/bash
# One gpurun call of the development loop:  bash tools/gpu_call.sh <tag> [pytest-args...]
# GPU tests (a failing assertion does not stop the call; a timeout / kill does), then short bench runs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-call}; shift
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > "$O/${TAG}_tests.log" 2>&1
rc=$?
tail -15 "$O/${TAG}_tests.log"
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-other --no-cpu-baseline --steps 500 > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err" || { echo "bench failed"; tail -5 "$O/${TAG}_bench.err"; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${TAG}_bench.json"))
k = d["roofline"]["kernels"]
print("syn-paths ms/step", round(d["ms_per_step"], 4), {n: round(v["kernel_avg_us"], 2) for n, v in k.items()})
PY
for w in syn-types wd-movies wd-articles; do
  timeout -k 10 300 python bench.py --workload $w --no-other --no-cpu-baseline --steps 60 --warmup 10 --settle 40 > "$O/${TAG}_$w.json" 2> "$O/${TAG}_$w.err" || { echo "$w failed"; tail -5 "$O/${TAG}_$w.err"; exit 1; }
  python -c "import json; d=json.load(open('$O/${TAG}_$w.json')); print('$w ms/step', round(d['ms_per_step'],4))"
done
