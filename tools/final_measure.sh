#!/bin/bash
# everything the committed profiles/ files of a round come from, in one gpurun call: tools/final_measure.sh <tag>
TAG=${1:-r04}
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
# driver form (--steps 20 --warmup 5) beside the default 60-step form, alternating; PCIe-inclusive rates of the C++ host
bash tools/driver_form.sh > gpurun_out/${TAG}_driver_form.log 2>&1
python3 -c "from vulkan_raytracing_amd import host; host.armadillo_path('resources')" > /dev/null 2>&1    # (the stand-in mesh rt_headless looks for)
for cfg in "--frames-in-flight 1" "--frames-in-flight 4" "--frames-in-flight 6"; do
  echo "rt_headless $cfg: $(./rt_headless --width 1920 --height 1080 --frames 120 --spp 4 --bounce 3 $cfg 2>&1 | grep "ms per frame" | tail -1 | cut -c1-200)" >> gpurun_out/${TAG}_headless.log
done
GPU_MAX_HW_QUEUES=8 ./rt_headless --width 1920 --height 1080 --frames 120 --spp 4 --bounce 3 --frames-in-flight 6 2>&1 | grep "ms per frame" | tail -1 | cut -c1-200 | sed 's/^/rt_headless 6 slots, 8 hardware queues: /' >> gpurun_out/${TAG}_headless.log
bash tools/pmc_ta.sh $TAG > gpurun_out/${TAG}_pmcta.log 2>&1
