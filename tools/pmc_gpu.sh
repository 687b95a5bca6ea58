#!/bin/bash
# PMC passes for the traversal kernels (run on the GPU box). Usage: tools/pmc_gpu.sh <tag> [bench args]
# Every pass has its own timeout and writes a log under gpurun_out/ (a hung pass must not stall the call).
TAG=${1:-x}; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline $@"
i=0
# PMC_SETS="A B C;D E" overrides the counter sets (one rocprofv3 pass per ';'-separated set)
DEFAULT_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VMEM_WR;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU;GRBM_GUI_ACTIVE"
IFS=';' read -ra SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for set in "${SETS[@]}"; do
  i=$((i+1))
  echo "pass $i: $set" >> $OUT/progress.log
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set" >> $OUT/progress.log
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_trace" in k or "k_beam" in k or "k_shade" in k or "k_raygen" in k:
            agg[k.replace("rt::","")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(agg.items()):
    print(k)
    for c,v in sorted(cs.items()):
        print("   %-28s avg %.5g  max %.5g (n=%d)"%(c,sum(v)/len(v),max(v),len(v)))
PY
cat $OUT/summary.txt
