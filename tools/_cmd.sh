set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_t10.log 2>&1; echo "tests rc=$?" ; tail -2 gpurun_out/r04_full_t10.log
(timeout -k 10 900 python tools/fuzz_many.py 3000 1500 800 ; timeout -k 10 400 python tools/fuzz_parts.py 300) > gpurun_out/r04_fuzz_final.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu.ids gpurun_out/r04_fuzz_final.txt | tail -7
