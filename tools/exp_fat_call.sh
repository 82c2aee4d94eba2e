set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_fat_gpu.py -x -q -p no:cacheprovider > $O/r4w_fat_tests.log 2>&1; rc=$?
tail -8 $O/r4w_fat_tests.log
if [ $rc -ge 124 ]; then echo TIMEOUT; exit $rc; fi
timeout -k 10 200 python tools/fat_time.py 1024 > $O/r4w_fat_time.log 2>&1; rc=$?; tail -6 $O/r4w_fat_time.log
