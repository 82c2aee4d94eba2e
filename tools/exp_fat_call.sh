set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_attn_flash_gpu.py tests/test_txf_gpu.py -q -p no:cacheprovider -x > $O/r4z_tests.log 2>&1; rc=$?
tail -25 $O/r4z_tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python tools/txf_wd_time.py t-SAIL wd-articles > $O/r4z_txf_wd2.log 2>&1; tail -2 $O/r4z_txf_wd2.log
timeout -k 10 300 python tools/txf_wd_time.py t-ARK wd-articles >> $O/r4z_txf_wd2.log 2>&1; tail -2 $O/r4z_txf_wd2.log
