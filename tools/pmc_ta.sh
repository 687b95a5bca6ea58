#!/bin/bash
# Is the vector-memory address path (TA/TCP) what bounds the traversal kernels?  tools/pmc_ta.sh <tag> [bench args]
TAG=${1:-x}; shift
OUT=gpurun_out/pmcta_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" $OUT/counters_list.txt | sort -u > $OUT/ta_tcp_names.txt
CMD="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --frames-in-flight 1 $@"
# (round 3: the four-counter sets of the TA_FLAT_* + TA_*_STALLED_BY_TC and of the TA_BUFFER_* counters did not fit one pass on gfx950 —
# rocprofv3 aborted with "Request exceeds the capabilities of the hardware to collect" — so those go two per pass now; a pass that
# still fails is reported at the END of the summary, not only in progress.log)
SETS=(
 "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE"
 "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum"
 "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum"
 "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum"
)
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  echo "pass $i: $set" >> $OUT/progress.log
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $set" >> $OUT/progress.log; tail -3 $OUT/p$i.log >> $OUT/progress.log; }
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("rt::","")
        if "k_trace" in k or "k_beam" in k or "k_shade" in k or "k_raygen" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(agg.items()):
    print(k)
    for c,v in sorted(cs.items()):
        big=[x for x in v if x>=0.5*max(v)] if max(v)>0 else v
        print("   %-44s avg(big) %.6g  (n=%d of %d)"%(c,sum(big)/len(big),len(big),len(v)))
PY
cat $OUT/progress.log; cat $OUT/summary.txt
if grep -q "failed" $OUT/progress.log; then echo "!! counter passes FAILED (their counters are missing above):"; grep "failed" $OUT/progress.log; fi
