export PYTHONPATH=$PWD; R=$PWD
python -m pytest tests -m gpu -q -k match 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for B in 0; do
unset SIFTMI_MATCH_SPLITS
rm -rf /tmp/mp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- python3 $R/tools/bench_match.py > /tmp/mp.log 2>&1
echo "== target blocks $B"; grep " x " /tmp/mp.log
python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/mp/**/*kernel_trace.csv",recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "mfma" not in r["Kernel_Name"]: continue
    d[(r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items(): print("  mfma grid", k, len(v), round(sum(v)/len(v)/1e3,1), "us")
PY
done
