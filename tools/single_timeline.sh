#!/bin/bash
# Kernel timeline of one single-frame call (BASELINE configs[1]) from rocprofv3 --kernel-trace: durations and gaps.
# usage: bash tools/single_timeline.sh [tag]      (environment knobs such as GPU_MAX_HW_QUEUES / SIFTMI_NO_GRAPH pass through)
R=$PWD; cd /tmp && export TMPDIR=/tmp
TAG=${1:-sf}
rm -rf /tmp/$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/$TAG -- python3 $R/tools/prof_single.py > /tmp/$TAG.log 2>&1
python3 - <<PY
import csv,glob
fs=glob.glob("/tmp/$TAG/**/*kernel_trace.csv",recursive=True)
rows=sorted(csv.DictReader(open(fs[0])), key=lambda r:int(r["Start_Timestamp"]))
# last call: find the last seed kernel (SEED=true blur2_kernel) and take everything from there
idx=[i for i,r in enumerate(rows) if "blur2_kernel<5, 32, 256, 4, 4, true" in r["Kernel_Name"]]
seg=rows[idx[-2]:idx[-1]]
t0=int(seg[0]["Start_Timestamp"])
busy=0; prev_end=t0
for r in seg:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"]; n=n[n.find("siftmi::")+8:][:46] if "siftmi::" in n else n[:46]
    print("%8.1f us  +%6.1f  dur %6.1f  end %6.1f  q %s  %s  grid %s" % ((s-t0)/1e3, (s-prev_end)/1e3, (e-s)/1e3, (e-t0)/1e3, r.get("Queue_Id","?"), n, r["Grid_Size_X"]))
    busy+=e-s; prev_end=max(prev_end,e)
print("span %.1f us, sum of kernel durations %.1f us, kernels %d" % ((prev_end-t0)/1e3, busy/1e3, len(seg)))
PY
