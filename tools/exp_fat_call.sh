set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 200 > $O/r4z_bench.json 2> $O/r4z_bench.err; rc=$?
grep "bench\]" $O/r4z_bench.err | tail -16
echo "rc=$rc"
python - <<PY
import json
d = json.load(open("$O/r4z_bench.json"))
print(d["ms_per_step"], json.dumps(d.get("transformer_variants", d.get("other_workloads", {}).get("transformer_variants")), indent=0)[:1500])
PY
