set -o pipefail
export AB_STEPS=40
timeout -k 10 900 python tools/uat_ab.py ab_libs/u_m7.so ab_libs/u_g7.so ab_libs/u_m7.so ab_libs/u_g7.so > gpurun_out/r04_uat_ab10.txt 2>&1; cat gpurun_out/r04_uat_ab10.txt
