tools/prof.sh r04_final > gpurun_out/r04_prof_final.log 2>&1
tools/prof.sh r04_mode2400 --rate 24 > gpurun_out/r04_prof_2400.log 2>&1
python3 tools/stamps.py ab_libs/stamps.so 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_stamps_final.txt
python3 tools/stamps_waves.py ab_libs/stamps.so 2>&1 | grep -v amdgpu.ids | head -33 | tee gpurun_out/r04_waves_final.txt
