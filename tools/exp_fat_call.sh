set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_attn_flash_gpu.py tests/test_dp_gpu.py -q -p no:cacheprovider -x -k "flash or transformer" > $O/r4z_dp_tests.log 2>&1; rc=$?
tail -25 $O/r4z_dp_tests.log
echo "pytest rc=$rc"
