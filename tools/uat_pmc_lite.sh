#!/bin/bash
# tools/uat_pmc_lite.sh <lib.so> <tag> : one PMC pass (LDS conflicts, LDS-active cycles, vector instructions, busy cycles) over tools/uat_rate.py for the UAT kernels
export TMPDIR=/tmp
out=$PWD/gpurun_out/uat_pmcl_$2; rm -rf $out; mkdir -p $out
export ADSB_AMD_LIB=$PWD/$1
(cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $out/pmc -- python3 $GRAFT_REPO_ROOT/tools/uat_rate.py --reps 2 --no-cpu > $out/log 2>&1)
python3 - <<PY > $out/pmc.txt
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "uat_" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC, one pass, per-dispatch means, lib=$1")
for k,cs in acc.items():
    print("  %-40s %s" % (k, "  ".join("%s %.4g" % (c, sum(v)/len(v)) for c,v in sorted(cs.items()))))
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
cat $out/pmc.txt
