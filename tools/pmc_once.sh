#!/bin/bash
# tools/pmc_once.sh "<counters...>"  : one PMC pass over bench --serial (3 steps), per-kernel means for scan1090
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_once; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --pmc $1 --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial > $out/log 2>&1)
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan1090" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("   %-26s mean %.5g" % (k, sum(v)/len(v)))
PY
grep -o 'kernel_ms": [0-9.]*' $out/log | tail -1
