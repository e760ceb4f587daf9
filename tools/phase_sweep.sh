#!/bin/bash
# kernel time with the chunk pipeline cut after phase N (profiling aid)
for p in 1 2 3 0; do
  echo -n "phase_limit=$p : "
  ADSB_AMD_PHASE_LIMIT=$p python bench.py --steps 10 --warmup 2 --cpu-buffers 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel_ms', d['roofline']['kernel_ms'], 'GB/s', d['roofline']['achieved'], 'ms/step', d['ms_per_step'])"
done
