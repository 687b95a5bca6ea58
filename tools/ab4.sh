#!/bin/bash
# round-4 A/B helper: one line per (label, bench args) in the driver's form and, optionally, the lone-frame breakdown — all inside ONE gpurun
# call so that the box (clocks differ by up to 10 % between boxes) is the same.  Usage: tools/ab4.sh <outfile> "label|bench args[|ENV=value ...]" ...
out=$1; shift
for spec in "$@"; do
  label=${spec%%|*}; rest=${spec#*|}; args=${rest%%|*}; envs=""
  if [ "$rest" != "$args" ]; then envs=${rest#*|}; fi      # optional third field: environment assignments (e.g. RT_LIB_VARIANT=name)
  env $envs python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $args > gpurun_out/ab4_tmp.json 2> gpurun_out/ab4_tmp.err || { echo "$label: bench FAILED"; tail -5 gpurun_out/ab4_tmp.err; continue; }
  python3 - "$label" <<'PY' | tee -a "$out"
import json, sys
d = json.loads(open("gpurun_out/ab4_tmp.json").read().strip().splitlines()[-1])
k = (d.get("roofline") or {}).get("frame_kernel_ms") or {}
k = {a: b for a, b in k.items() if isinstance(b, float)}
print("%-34s ms/step %.4f  anim %.4f  lone %.4f  value %.0f | kernels %s | visits c %.2f s %.2f" % (
    sys.argv[1], d["ms_per_step"], d.get("animated_ms_per_step") or 0, d.get("ms_per_frame_single") or 0, d["value"],
    " ".join("%s %.3f" % (a, b) for a, b in k.items()), d.get("mean_node_visits_per_closest_ray") or 0, d.get("mean_node_visits_per_shadow_ray") or 0))
PY
done
