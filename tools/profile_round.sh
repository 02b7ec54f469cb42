#!/bin/bash
# Round profile of the default bench (both modes): kernel-trace stats + three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ).
# Run on the GPU box from the repo root:  bash tools/profile_round.sh r02
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-prof --no-cpu-baseline --no-parity-check"
# --no-overlap: every kernel alone on the chip, as in bench.py's HIP-event pass, so the per-kernel averages are comparable
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-parity-check > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_overlap -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-check > $OUT/stats_overlap.log 2>&1 || echo "stats (overlap) pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1 || echo "write pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || echo "sq pass failed"
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json | tail -14
python3 tools/sq_counters.py $OUT/sq $OUT/sq_counters.json | tail -18
find $OUT/stats -name "*kernel_stats.csv" | head -2
# keep only the summaries small enough to merge back
find $OUT -name "*counter_collection.csv" -size +20M -delete
