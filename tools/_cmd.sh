python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_t29.log 2>&1; tail -2 gpurun_out/r04_t29.log
python3 bench.py --steps 20 --warmup 5 --no-extras --cpu-buffers 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
