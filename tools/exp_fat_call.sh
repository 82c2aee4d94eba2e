set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r4x_tests.log 2>&1; rc=$?
tail -6 $O/r4x_tests.log
echo "pytest rc=$rc"
