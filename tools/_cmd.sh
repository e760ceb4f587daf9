set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04_full_t9.log 2>&1; echo "tests rc=$?" ; tail -2 gpurun_out/r04_full_t9.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
