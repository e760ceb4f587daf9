#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- kernel trace + stats, then two PMC passes, summaries into gpurun_out/prof_<tag>/
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-buffers 0 --serial --no-extras "$@" > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial --no-extras "$@" > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial --no-extras "$@" > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial --no-extras "$@" > $out/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-buffers 0 --serial --no-extras "$@" > $out/bench_pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1 || true
# keep only small files for the merge back
find $out -name "*.csv" -size +2M -delete
tail -60 $out/summary.txt
