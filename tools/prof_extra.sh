#!/bin/bash
# tools/prof_extra.sh <tag> : kernel-trace statistics of the two secondary workloads (UAT 978 bench, 2.4 MS/s mode bench) into gpurun_out/prof_<tag>_{uat978,mode2400}/
set -e
tag=$1
export TMPDIR=/tmp
root=$PWD
for w in uat978 mode2400; do
  out=$root/gpurun_out/prof_${tag}_$w; rm -rf $out; mkdir -p $out
  if [ $w = uat978 ]; then args="--workload uat978 --steps 10 --warmup 2 --serial --cpu-buffers 0"; else args="--rate 24 --no-extras --steps 10 --warmup 2 --serial --cpu-buffers 0"; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py $args > $out/bench.log 2>&1)
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  cp $f $out/kernel_stats.csv
  find $out/trace -name "*.csv" -size +1M -delete
  head -12 $out/kernel_stats.csv
done
