#!/bin/bash
# Per-launch durations of the extrema scan by octave (grid size), serialised launches: rocprofv3 kernel trace of tools/dense_stage_times.py
# usage (on the GPU box): bash tools/extrema_trace.sh [dense|bench]
KIND=${1:-dense}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/extrema_trace_$KIND
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/dense_stage_times.py 3 $KIND > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "extrema_kernel" in n or "refine_kernel" in n or "orientation_kernel" in n or "descriptor_kernel" in n:
        key = (n.split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]          # the timed steps
    print("%-42s grid %7s x %5s x %4s  wg %4s: %3d launches, avg %9.1f us" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v)))
PY
