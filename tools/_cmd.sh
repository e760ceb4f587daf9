set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_t8.log 2>&1; echo "tests rc=$?" ; tail -2 gpurun_out/r04_full_t8.log
timeout -k 10 400 python tools/fuzz_parts.py 300 > gpurun_out/r04_fuzz_uat3.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04_fuzz_uat3.txt
timeout -k 10 800 bash tools/uat_pmc.sh > gpurun_out/r04c_uat978_rocprof_summary.txt 2>&1; echo "pmc rc=$?"; grep -n "uat_demod_kernel" gpurun_out/r04c_uat978_rocprof_summary.txt | head -3
timeout -k 10 300 python bench.py --workload uat978 --steps 20 --warmup 5 > gpurun_out/r04_uat_b3.json 2> gpurun_out/r04_uat_b3.err; echo "bench rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/r04_uat_b3.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['ms_per_step_serial'],d['demod_kernel_ms'],d['roofline']['kernel_ms'])"
