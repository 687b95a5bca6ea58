for L in 1 2 4; do
  RT_LBVH_MAX_LEAF=$L python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/r3i_leaf$L.json 2> gpurun_out/r3i_leaf$L.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r3i_leaf$L.json")); r=d["roofline"]
print("leaf $L: ms/step %.4f closest live %.3f iso %.3f nodes %.2f tris %.2f | kernels %s" % (d["ms_per_step"], r["avg_launch_ms"], r["isolated"]["avg_launch_ms"], r["mean_node_visits_per_ray"], r["mean_tri_tests_per_ray"], {k: round(v,3) for k,v in r["frame_kernel_ms"].items() if k!="timing"}))
PY
done
