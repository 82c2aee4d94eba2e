set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_attn_flash_gpu.py -q -p no:cacheprovider -x > $O/r4y_flash_tests.log 2>&1; rc=$?
tail -40 $O/r4y_flash_tests.log
echo "pytest rc=$rc"
