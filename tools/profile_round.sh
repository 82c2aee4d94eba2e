#!/bin/bash
# Collect the round's profiles on the GPU box:  bash tools/profile_round.sh <out-name> [workload] [stats-only]
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2. three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; MFMA/GRBM) -- never combined with other trace domains
#   3. one SQ instruction-mix pass for the two diagonal kernels
# Outputs land in gpurun_out/<out-name>/; tools/summarize_profile.py reduces them to the files kept under profiles/.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-prof}
W=${2:-syn-paths}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o st -- python3 "$R/bench.py" --workload "$W" --steps 40 --warmup 10 --settle 0 --no-cpu-baseline --no-other > "$O/stats.log" 2>&1
echo "stats done"
if [ "${3:-}" = "stats-only" ]; then exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o p -- python3 "$R/bench.py" --workload "$W" --steps 5 --warmup 2 --no-graph --settle 0 --no-cpu-baseline --no-other > "$O/pmc_$c.log" 2>&1
  echo "$c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_SQ" -o p -- python3 "$R/bench.py" --workload "$W" --steps 5 --warmup 2 --no-graph --settle 0 --no-cpu-baseline --no-other > "$O/pmc_SQ.log" 2>&1
echo "SQ busy done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_BF16 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$O/pmc_mix" -o p -- python3 "$R/bench.py" --workload "$W" --steps 3 --warmup 1 --no-graph --settle 0 --no-cpu-baseline --no-other > "$O/pmc_mix.log" 2>&1 || echo "instruction-mix pass failed (see pmc_mix.log)"
echo "all passes done"
