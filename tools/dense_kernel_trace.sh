#!/bin/bash
# per-kernel times of one dense (or bench) 64 x 1080p step with every launch alone on the GPU (tools/dense_stage_times.py: direct
# launches, one context): rocprofv3 --kernel-trace reduced to one line per kernel.  usage: bash tools/dense_kernel_trace.sh [dense|bench] [lib]
R=${GRAFT_REPO_ROOT:-$PWD}
KIND=${1:-dense}
[ -n "$2" ] && export SIFTMI_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/dense_trace_$KIND
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/dense_stage_times.py 3 $KIND > $OUT/run.log 2>&1
tail -n 1 $OUT/run.log
python3 - <<PY
import csv, glob, collections
kt = sorted(glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(kt[-1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void siftmi::", "").replace("siftmi::", "")[:70]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = 5.0          # 2 warm-up + 3 timed steps of the script
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("%-72s calls %4d  avg %9.1f us  per step %8.3f ms" % (n, len(v), sum(v) / len(v), sum(v) / steps / 1e3))
PY
