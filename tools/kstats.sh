#!/bin/bash
# per-kernel time table (rocprofv3 --kernel-trace --stats) of a python command.  Usage: tools/kstats.sh <tag> <script.py> [args]
# (the program follows `--` directly)
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/stdout.log 2> $OUT/stderr.log || { echo "rocprofv3 failed"; tail -5 $OUT/stderr.log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-100s %8s %12s %10s %10s %10s" % ("kernel", "calls", "total us", "avg us", "min us", "max us"))
for r in rows[:24]:
    print("%-100s %8s %12.1f %10.1f %10.1f %10.1f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
