python3 -m pytest tests/test_gpu_parity.py tests/test_mode2400_gpu.py -x -q -m gpu > gpurun_out/r04_t9.log 2>&1; tail -3 gpurun_out/r04_t9.log
for p in host ctl stream; do for r in 20 24; do echo "count path $p rate $r"; ADSB_AMD_COUNT_PATH=$p AB_RATE=$r python3 tools/sustained_ab.py ab_libs/cnt.so 2>&1 | grep -v amdgpu.ids | tail -1; done; done | tee gpurun_out/r04_ab8.txt
for p in host stream; do echo "count path $p"; ADSB_AMD_COUNT_PATH=$p python3 tools/stamps.py ab_libs/stamps.so 4 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r04_ab8.txt
