#!/bin/bash
# tools/uat_trace.sh [lib.so] : rocprofv3 kernel trace + stats over tools/uat_rate.py (1 GiB, serial calls) -> gpurun_out/uat_trace[_tag]/stats.txt
export TMPDIR=/tmp
tag=${2:-new}
out=$PWD/gpurun_out/uat_trace_$tag; rm -rf $out; mkdir -p $out
[ -n "$1" ] && export ADSB_AMD_LIB=$PWD/$1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --reps 5 --no-cpu > $out/log_trace 2>&1)
python3 - <<PY > $out/stats.txt
import csv,glob
print("== rocprofv3 --kernel-trace --stats -- python3 tools/uat_rate.py --reps 5 --no-cpu   (1 GiB synthetic UAT IQ, device resident, calls back to back) lib=$1")
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  %-78s calls=%s avg_ns=%s min=%s max=%s pct=%s" % (r["Name"][:78], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
cat $out/stats.txt; tail -3 $out/log_trace
