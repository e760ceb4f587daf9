#!/bin/bash
# usage: ab1090.sh out lib1 lib2 ... : bench kernel_ms at noise 3 / 10 / 20 for each lib (two rounds)
out=$1; shift
for rnd in 1 2; do for lib in "$@"; do for noise in 3 20; do
  extra=""; [ $noise != 3 ] && extra="--noise $noise"
  ADSB_AMD_LIB=$PWD/$lib python3 bench.py --steps 100 --warmup 5 --no-extras --cpu-buffers 0 $extra 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib noise $noise  kernel_ms %.4f  step %.4f  records %d' % (j['roofline']['kernel_ms'], j['ms_per_step'], j['records_per_step']))" >> $out
done; done; done
cat $out
