#!/bin/bash
# What bounds the traversal kernels?  SQ counter passes for k_trace<closest> / k_trace<shadow>, run on the GPU box:
#   tools/pmc_bound.sh <tag> [bench args]        -> gpurun_out/pmcb_<tag>/{summary.txt,summary.json}
# One rocprofv3 pass per counter set (8 SQ slots per pass), each with its own timeout; no tracing domain besides
# --kernel-trace (gpurun refuses --pmc together with sys/hip/hsa traces).  The program follows `--` directly.
TAG=${1:-x}; shift
OUT=gpurun_out/pmcb_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras $@"
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
 "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_FLAT SQ_INST_LEVEL_LDS"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  echo "pass $i: $set" >> $OUT/progress.log
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set" >> $OUT/progress.log
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections, json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("rt::","")
        if "k_trace" in k or "k_beam" in k or "k_shade" in k or "k_raygen" in k or "k_tail" in k or "k_resolve" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={}
for k,cs in sorted(agg.items()):
    # the big launches only (bounce 0): a kernel's first-frame small launches would dilute the averages
    print(k)
    out[k]={}
    for c,v in sorted(cs.items()):
        big=[x for x in v if x>=0.5*max(v)] if max(v)>0 else v
        out[k][c]={"avg_big":sum(big)/len(big),"n_big":len(big),"n":len(v),"max":max(v)}
        print("   %-26s avg(big) %.6g  (n=%d of %d)  max %.6g"%(c,sum(big)/len(big),len(big),len(v),max(v)))
json.dump(out,open("$OUT/summary.json","w"),indent=1)
PY
cat $OUT/progress.log
tail -60 $OUT/summary.txt
