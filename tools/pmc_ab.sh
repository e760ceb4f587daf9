#!/bin/bash
# tools/pmc_ab.sh <name> [<name> ...] : SQ counters of scan1090_kernel for ab_ship/<name>.so (bench --serial, 3 steps, means per launch)
# AB_BENCH_ARGS="--rate 24" AB_KERNEL=scan2400 for the 2.4 MS/s mode; AB_PASSES="1 2 3 4" adds FETCH_SIZE and WRITE_SIZE passes
export TMPDIR=/tmp
root=$PWD
for name in "$@"; do
  for pass in ${AB_PASSES:-1 2}; do
    if [ $pass = 1 ]; then ctr="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
    elif [ $pass = 3 ]; then ctr="FETCH_SIZE";
    elif [ $pass = 4 ]; then ctr="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";
    elif [ $pass = 5 ]; then ctr="SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES";
    else ctr="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"; fi
    out=$root/gpurun_out/pmc_ab/$name.$pass; rm -rf $out; mkdir -p $out
    (cd /tmp && ADSB_AMD_LIB=$root/${AB_DIR:-ab_ship}/$name.so rocprofv3 --pmc $ctr --output-format csv -d $out/p -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial --no-extras $AB_BENCH_ARGS > $out/log 2>&1)
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys, os
root = os.getcwd()
for name in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/gpurun_out/pmc_ab/%s.*/p/**/*counter_collection.csv" % (root, name), recursive=True):
        for r in csv.DictReader(open(f)):
            if os.environ.get("AB_KERNEL", "scan1090") in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s" % name)
    for k, v in sorted(acc.items()):
        print("   %-24s mean %.5g   per chunk %.1f" % (k, sum(v) / len(v), sum(v) / len(v) / 131072))
PY
find $root/gpurun_out/pmc_ab -name "*.csv" -size +1M -delete
grep -ho 'kernel_ms": [0-9.]*' $root/gpurun_out/pmc_ab/*/log
