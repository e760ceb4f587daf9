#!/bin/bash
# tools/uat_pmc.sh : kernel trace + a few PMC passes over tools/uat_rate.py (256 MiB), per-kernel means for the UAT kernels
export TMPDIR=/tmp
out=$PWD/gpurun_out/uat_pmc; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --mib 256 --reps 3 --no-cpu > $out/log_trace 2>&1)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --mib 256 --reps 2 --no-cpu > $out/log_pmc$i 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  %-70s calls=%s avg_ns=%s pct=%s" % (r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"]))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uat_" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    print(k)
    for c,v in sorted(cs.items()): print("   %-26s n=%d mean %.6g" % (c, len(v), sum(v)/len(v)))
PY
