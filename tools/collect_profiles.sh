#!/bin/bash
# after tools/final_measure.sh <tag> (on the GPU box): copy what is to be judged from gpurun_out/ into profiles/ (tracked).  Usage: tools/collect_profiles.sh r04
T=${1:-r04}; G=gpurun_out; P=profiles
cp $G/${T}_bench_full.json $P/ && cp $G/${T}_bench_cfg4.json $G/${T}_bench_cfg5.json $P/
cp $G/prof_$T/bench_under_profiler.json $P/${T}_bench_under_profiler.json
cp $G/prof_$T/summary.txt $P/${T}_rocprof_summary.txt; cp $G/prof_${T}_cfg4/summary.txt $P/${T}_rocprof_summary_cfg4.txt; cp $G/prof_${T}_cfg5/summary.txt $P/${T}_rocprof_summary_cfg5.txt
cp $G/pmcb_$T/summary.txt $P/${T}_pmc_sq_bound.txt; cp $G/pmcta_$T/summary.txt $P/${T}_pmc_ta_tcp_td.txt
cp $G/${T}_shard_ceiling.log $P/${T}_shard_ceiling.txt; cp $G/${T}_visits.log $P/${T}_visit_counts.txt
cp $G/prof_$T/latest_profile.json $P/latest_profile.json; cp $G/prof_${T}_cfg4/latest_profile_cfg4.json $G/prof_${T}_cfg5/latest_profile_cfg5.json $P/
{ echo "driver form (--steps 20 --warmup 5) against the default (--steps 60 --warmup 6), alternating, one gpurun call (tools/driver_form.sh):"; cat $G/${T}_driver_form.log; echo
  echo "PCIe-inclusive rates of the C++ host (every frame copied to pinned host memory; ./rt_headless --width 1920 --height 1080 --frames 120 --spp 4 --bounce 3, animated cfg3 scene;"
  echo "with one frame in flight rt_headless prints every frame and no summary line):"; cat $G/${T}_headless.log; } > $P/${T}_driver_form_and_headless.txt
