#!/bin/bash
# Per-launch-shape durations of the blur kernels in the bench workload (run on the GPU box).
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bo
rocprofv3 --kernel-trace --output-format csv -d /tmp/bo -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > /tmp/bo.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/bo/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "blur" not in n: continue
    short=n[n.index("blur"):n.index("(")][:64]
    d[(short, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
W,H,F=3840,2160,32
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    print("%-66s blocks %7d  n=%4d avg %8.1f us total %9.1f us"%(k[0],k[1],len(v),sum(v)/len(v)/1e3,sum(v)/1e3))
PY
