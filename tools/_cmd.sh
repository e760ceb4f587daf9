set -o pipefail
timeout -k 10 600 python tools/uat_ab.py ab_libs/u_h8.so ab_libs/u_n8.so ab_libs/u_h8.so ab_libs/u_n8.so > gpurun_out/r04_uat_ab6.txt 2>&1; cat gpurun_out/r04_uat_ab6.txt
