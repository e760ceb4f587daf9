python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04_t25.log 2>&1; tail -2 gpurun_out/r04_t25.log
AB_CONTEXTS=3 python3 tools/ab_dense.py ab_libs/tailfast.so ab_libs/pf.so ab_libs/fulla.so ab_libs/pfa.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_ab34.txt
