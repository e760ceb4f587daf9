#!/bin/bash
# tools/rotate_ab.sh -- does the bench's rescanning of ONE resident GiB flatter the scan kernel (L2s, 256 MiB memory-side cache)?
# A/B of bench.py --rotate 1 against --rotate 4 (four distinct device-resident GiB, step i scans input i mod 4): kernel_ms by HIP events in the
# pipelined and the serial loop, and FETCH_SIZE / WRITE_SIZE per launch by rocprofv3 (separate --pmc passes).  Output: gpurun_out/rotate/summary.txt
set -e
out=$PWD/gpurun_out/rotate
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for k in 1 4 1 4; do
  python3 bench.py --steps 200 --warmup 5 --no-extras --cpu-buffers 0 --rotate $k >> $out/pipelined_k$k.json
  python3 bench.py --steps 200 --warmup 5 --no-extras --cpu-buffers 0 --rotate $k --serial >> $out/serial_k$k.json
done
for k in 1 4; do
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_k$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --cpu-buffers 0 --serial --no-extras --rotate $k > $out/fetch_k$k.log 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write_k$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --cpu-buffers 0 --serial --no-extras --rotate $k > $out/write_k$k.log 2>&1)
done
python3 - <<PY > $out/summary.txt
import csv, glob, json, collections
out = "$out"
print("bench.py --rotate K: step i scans device-resident input i mod K (K distinct GiB of the generator); --steps 200 --warmup 5, two runs each")
for mode in ("pipelined", "serial"):
    for k in (1, 4):
        for line in open("%s/%s_k%d.json" % (out, mode, k)):
            line = line.strip()
            if not line.startswith("{"): continue
            j = json.loads(line)
            r = j["roofline"]
            print("  %-9s K=%d  kernel_ms %.4f  frac %.4f  ms_per_step %.4f  first_100 %.4f  records/step %s" % (mode, k, r["kernel_ms"], r["frac"], j["ms_per_step"], r["kernel_ms_first_100"], (j["config"].get("rotate") or {}).get("records_per_step_mean", j["records_per_step"])))
print("rocprofv3 --pmc (separate passes), scan1090_kernel, mean per launch over the run's dispatches (KB as rocprofv3 reports them; FETCH_SIZE is to be doubled on gfx950):")
for k in (1, 4):
    acc = collections.defaultdict(list)
    for d in ("fetch", "write"):
        for f in glob.glob("%s/%s_k%d/**/*counter_collection.csv" % (out, d, k), recursive=True):
            for r in csv.DictReader(open(f)):
                if "scan1090" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("  K=%d  " % k + "  ".join("%s n=%d mean %.6g" % (c, len(v), sum(v) / len(v)) for c, v in sorted(acc.items())))
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        f = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]); w = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
        print("        traffic = 2 x FETCH_SIZE + WRITE_SIZE = %.0f bytes per launch" % ((2 * f + w) * 1024))
PY
find $out -name "*.csv" -size +2M -delete
find $out -name "*.db" -delete
cat $out/summary.txt
