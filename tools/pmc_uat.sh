#!/bin/bash
# tools/pmc_uat.sh : SQ counters of the UAT kernels on the bench workload (bench.py --workload uat978, 3 steps), means per launch
export TMPDIR=/tmp
root=$PWD
for pass in 1 2; do
  if [ $pass = 1 ]; then ctr="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
  else ctr="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"; fi
  out=$root/gpurun_out/pmc_uat/$pass; rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $out/p -- python3 $root/bench.py --workload uat978 --steps 3 --warmup 1 > $out/log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.getcwd()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/gpurun_out/pmc_uat/*/p/**/*counter_collection.csv" % root, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "demod" if "uat_demod" in k else "scan" if "uat_scan" in k else None
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    print("== %s" % name)
    for k, v in sorted(d.items()):
        v = [x for x in v if x > 0] or [0]
        print("   %-24s launches %d  mean %.5g   max %.5g" % (k, len(v), sum(v) / len(v), max(v)))
PY
find $root/gpurun_out/pmc_uat -name "*.csv" -size +1M -delete
