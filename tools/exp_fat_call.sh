set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -p no:cacheprovider -x -k "vocab" > $O/r4z_vc_tests.log 2>&1; rc=$?
tail -5 $O/r4z_vc_tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
for v in default vc2bar; do
  if [ $v = default ]; then unset ARK_AMD_LIB; else export ARK_AMD_LIB=$PWD/ark_amd/lib/variants/$v/libark_amd.so; fi
  echo "== $v movies"; timeout -k 10 120 python tools/vc_time.py 2>&1 | grep " us"
  echo "== $v articles"; timeout -k 10 120 python tools/vc_time.py articles 2>&1 | grep " us"
done
done
