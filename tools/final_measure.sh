#!/bin/bash
# everything the committed profiles/ files of a round come from, in one gpurun call: tools/final_measure.sh <tag>
TAG=${1:-r03}
python3 bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; head -c 250 gpurun_out/${TAG}_bench_full.json; echo
bash tools/profile_gpu.sh $TAG --steps 24 --warmup 6 > gpurun_out/${TAG}_profile.log 2>&1
bash tools/pmc_bound.sh $TAG > gpurun_out/${TAG}_pmcb.log 2>&1
for w in cfg4 cfg5; do
  bash tools/profile_gpu.sh ${TAG}_$w --workload $w --steps 12 --warmup 4 > gpurun_out/${TAG}_${w}_profile.log 2>&1
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; head -c 220 gpurun_out/${TAG}_bench_$w.json; echo
done
bash tools/shard_ceiling.sh > gpurun_out/${TAG}_shard_ceiling.log 2>&1
python3 tools/visit_counts.py 2>&1 | grep workload > gpurun_out/${TAG}_visits.log; WORKLOADS=cfg4,cfg5 MESHES=standin python3 tools/visit_counts.py 2>&1 | grep workload >> gpurun_out/${TAG}_visits.log
grep "closest-hit traversal" gpurun_out/prof_${TAG}_cfg5/summary.txt | cut -c1-90
