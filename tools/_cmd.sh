python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_t24.log 2>&1; tail -2 gpurun_out/r04_t24.log
python3 tools/soak.py 40 > gpurun_out/r04_soak.txt 2>&1; tail -4 gpurun_out/r04_soak.txt
python3 __graft_entry__.py --smoke 2>&1 | tail -3
