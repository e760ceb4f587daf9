set -o pipefail
timeout -k 10 500 python -m pytest tests/test_uat978_gpu.py -x -q -m gpu > gpurun_out/r04_uat_t2.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r04_uat_t2.log
timeout -k 10 500 python tools/uat_ab.py ab_libs/u_base.so ab_libs/u_t2.so ab_libs/u_base.so ab_libs/u_t2.so > gpurun_out/r04_uat_ab2.txt 2>&1; cat gpurun_out/r04_uat_ab2.txt
