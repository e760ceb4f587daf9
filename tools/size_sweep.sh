#!/bin/bash
for mib in 32 64 128 256 512 1024 2048; do
  echo -n "mib=$mib : "
  python bench.py --mib $mib --steps 100 --warmup 5 --cpu-buffers 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('kernel_ms', r['kernel_ms'], 'GB/s', r['achieved'], 'ms/step', d['ms_per_step'], 'per-GiB-ms', round(r['kernel_ms']*1024/$mib,4))"
done
