python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_t11.log 2>&1; tail -3 gpurun_out/r04_t11.log
python3 tools/sustained_ab.py ab_libs/cnt.so ab_libs/defer.so ab_libs/refill.so ab_libs/alias3.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ab11.txt
