#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's names.
# usage: bash tools/collect_profiles.sh r02d r02
T=gpurun_out/$1; R=$2
cp $(find $T/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_no_overlap_kernel_stats.csv
cp $(find $T/stats_overlap -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_kernel_stats.csv
cp $T/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $T/sq_counters.json profiles/${R}_sq_counters.json
grep '^{"metric"' $T/stats.log | tail -1 > profiles/${R}_bench_no_overlap.json
[ -f $T/bench_default.json ] && cp $T/bench_default.json profiles/${R}_bench_default.json
ls -la profiles/${R}_*
