#!/bin/bash
# SQ instruction counters for env-selected variants: tools/pmc_ab.sh "ADSB_AMD_TUNE=0" "ADSB_AMD_TUNE=1"
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_ab; rm -rf $out; mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  (cd /tmp && env $v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR --output-format csv -d $out/v$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-buffers 0 > $out/v$i.log 2>&1)
  echo "== $v"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/v$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan1090" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("   %-22s per-chunk %.1f" % (k, sum(v)/len(v)/131072))
PY
done
