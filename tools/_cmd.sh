set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_t5.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r04_full_t5.log
timeout -k 10 400 python tools/fuzz_parts.py 300 > gpurun_out/r04_fuzz_uat2.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r04_fuzz_uat2.txt
timeout -k 10 300 python bench.py --workload uat978 --steps 20 --warmup 5 > gpurun_out/r04_uat_b2.json 2> gpurun_out/r04_uat_b2.err; echo "bench rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_uat_b2.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['ms_per_step_serial'],d['demod_kernel_ms'],d['roofline']['kernel_ms'])"
