#!/bin/bash
# dynamic instruction counts per phase (profiling aid): SQ_INSTS_* with the chunk pipeline cut after phase N
export TMPDIR=/tmp
out=$PWD/gpurun_out/phase_pmc; rm -rf $out; mkdir -p $out
for p in 1 2 3 0; do
  (cd /tmp && ADSB_AMD_PHASE_LIMIT=$p rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p$p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-buffers 0 > $out/p$p.log 2>&1)
  echo "== phase_limit=$p"
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/p$p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan1090" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("   %-22s %.4g  per-chunk %.1f" % (k, sum(v)/len(v), sum(v)/len(v)/131072))
PY
done
