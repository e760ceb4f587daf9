#!/bin/bash
# tools/uat_pmc2.sh : further rocprofv3 --pmc passes (each in its own run) over tools/uat_rate.py for the UAT demodulation kernel:
# instruction fetch, LDS queueing.  Prints per-dispatch means for uat_demod_kernel only.
export TMPDIR=/tmp
out=$PWD/gpurun_out/uat_pmc2; rm -rf $out; mkdir -p $out
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH" "InstrFetchLatency" "LdsLatency" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --reps 2 --no-cpu > $out/log_pmc$i 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uat_demod_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()): print("   %-30s n=%d mean %.6g" % (c, len(v), sum(v)/len(v)))
PY
