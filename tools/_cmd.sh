set -o pipefail
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; echo "rc=$?"; tail -1 gpurun_out/r04_final_bench.json | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['uat978']['ms_per_step'] if 'uat978' in d else None, d.get('mode_2400',{}).get('kernel_ms'))"
