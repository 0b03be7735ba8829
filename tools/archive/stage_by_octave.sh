#!/bin/bash
# Per-launch-shape durations of the dense kernels (blur, extrema) in the bench workload (run on the GPU box).
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bo
rocprofv3 --kernel-trace --output-format csv -d /tmp/bo -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras --no-roofline > /tmp/bo.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/bo/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    short=n[:n.index("(")] if "(" in n else n
    short=short.replace("void siftmi::","")[:70]
    d[(short, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if sum(v) < 1e5: continue
    print("%-72s blocks %6d x %4d x %3d  n=%4d avg %8.1f us  total %8.2f ms"%(k[0],k[1],k[2],k[3],len(v),sum(v)/len(v)/1e3,sum(v)/1e6))
PY
