python3 -m pytest tests/test_gpu_parity.py tests/test_mode2400_gpu.py -x -q -m gpu > gpurun_out/r04_t23.log 2>&1; tail -2 gpurun_out/r04_t23.log
python3 tools/stamps_waves.py ab_libs/stamps.so 2>&1 | grep -v amdgpu.ids | grep -A1 launch | grep -v "^--" | head -6 | tee gpurun_out/r04_waves8.txt
AB_CONTEXTS=3 python3 tools/ab_dense.py ab_libs/cur4.so ab_libs/tailfast.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ab30.txt
AB_RATE=24 AB_CONTEXTS=2 python3 tools/ab_dense.py ab_libs/cur4.so ab_libs/tailfast.so 2>&1 | grep -v amdgpu.ids | grep baseline | tee -a gpurun_out/r04_ab30.txt
