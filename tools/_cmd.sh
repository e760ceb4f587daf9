python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_t22.log 2>&1; tail -2 gpurun_out/r04_t22.log
python3 tools/fuzz_many.py > gpurun_out/r04_fuzz.txt 2>&1; tail -3 gpurun_out/r04_fuzz.txt
