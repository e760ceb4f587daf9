#!/bin/bash
# usage: tools/timeline.sh <tag> [bench args...]  -- kernel + memory-copy trace of the PIPELINED bench loop, then the per-step gaps (tools/step_timeline.py)
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/tl_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --cpu-buffers 0 --no-extras "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
k=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $k | tee $out/timeline.txt
m=$(find $out/trace -name "*memory_copy_trace.csv" | head -1)
[ -n "$m" ] && python3 - "$m" <<'PY' | tee -a $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[len(rows) // 2:]
import collections
d = collections.defaultdict(list)
for r in rows:
    d[(r.get("Direction", "?"), )].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v.sort()
    print("copies %s: n %d median %.1f us max %.1f us" % (k, len(v), v[len(v) // 2] / 1e3, v[-1] / 1e3))
PY
find $out -name "*.csv" -size +2M -delete
tail -3 $out/bench.log
