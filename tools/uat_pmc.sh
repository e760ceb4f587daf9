#!/bin/bash
# tools/uat_pmc.sh : rocprofv3 kernel trace + PMC passes (each in its own run) over tools/uat_rate.py (1 GiB), UAT kernels
export TMPDIR=/tmp
out=$PWD/gpurun_out/uat_pmc; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --reps 5 --no-cpu > $out/log_trace 2>&1)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --reps 2 --no-cpu > $out/log_pmc$i 2>&1) || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
print("== rocprofv3 --kernel-trace --stats -- python3 tools/uat_rate.py --reps 5 --no-cpu   (1 GiB synthetic UAT IQ, device resident)")
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("  %-78s calls=%s avg_ns=%s min=%s max=%s pct=%s" % (r["Name"][:78], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
for f in glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "uat_" in k and k not in seen:
            seen.add(k)
            print("  dispatch %-50s grid=%s wg=%s vgpr=%s sgpr=%s lds=%s scratch=%s" % (k, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uat_" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (separate passes, per-dispatch means)")
for k,cs in acc.items():
    print(k)
    for c,v in sorted(cs.items()): print("   %-26s n=%d mean %.6g" % (c, len(v), sum(v)/len(v)))
PY
grep -h "^{" $out/log_trace | tail -1
