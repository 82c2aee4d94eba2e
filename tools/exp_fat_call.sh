set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r4z_all_tests.log 2>&1; rc=$?
tail -3 $O/r4z_all_tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
bash tools/profile_round.sh r04b_wd-movies wd-movies stats-only && bash tools/profile_round.sh r04b_wd-articles wd-articles stats-only
for w in wd-movies wd-articles; do timeout -k 10 200 python bench.py --workload $w --no-other --no-cpu-baseline --steps 100 --warmup 20 --settle 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4))"; done
