#!/bin/bash
# Timing ablations of the layer-diagonal GRU kernels (results are INVALID, only the clock matters).
# Builds ark_amd/csrc/gru_diag.hip with -DARK_ABL=<mask> into a side library and runs bench.py on it:
#   1 no A-fragment LDS reads   2 no B-fragment LDS reads   4 no MFMA   8 no LDS-DMA
#   16 no epilogue stores (forward kernel)   32 forward kernel returns at once (launch floor)
#   64 forward kernel skips its whole ring loop (waits + barriers)   128 forward kernel returns before its epilogue
# usage (on a GPU box, after __graft_entry__.build()):  tools/ablate_diag.sh 0 3 7 15 16 31 32
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ark_amd/lib/obj
for m in "$@"; do
  if [ "$m" = "0" ]; then lib=$ROOT/ark_amd/lib/libark_amd.so; else
    lib=/tmp/libark_abl$m.so
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DARK_ABL=$m -c $ROOT/ark_amd/csrc/gru_diag.hip -o /tmp/gru_diag_abl$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $(ls $OBJ/*.o | grep -v gru_diag.o) /tmp/gru_diag_abl$m.o
  fi
  echo -n "ARK_ABL=$m: "
  ARK_AMD_LIB=$lib python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step', round(d['ms_per_step'],4), ' fwd diagonal launch us', round(d['roofline']['kernel_avg_us'],2))"
done
