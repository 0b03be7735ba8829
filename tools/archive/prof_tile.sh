R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $R/tools/prof_tile.py > /tmp/pt.log 2>&1
tail -2 /tmp/pt.log
python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pt/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    d[(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=0
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    tot+=sum(v)
    if sum(v)/4 > 30e3: print("%-72s grid %8s %5s %3s  n=%3d  avg %9.1f us"%(k[0],k[1],k[2],k[3],len(v),sum(v)/len(v)/1e3))
print("total per call %.2f ms"%(tot/4/1e6))
PY
