#!/bin/bash
for st in 0 1 2 3 4 6; do
  echo -n "stagger=$st : "
  ADSB_AMD_STAGGER=$st python bench.py --steps 10 --warmup 2 --cpu-buffers 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel_ms', d['roofline']['kernel_ms'], 'GB/s', d['roofline']['achieved'], 'ms/step', d['ms_per_step'])"
done
