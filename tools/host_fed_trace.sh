#!/bin/bash
# timeline of the host-fed stream: kernel + memory-copy trace, reduced to a per-step summary (copies, gaps, kernel time under copies)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/host_fed_trace
rm -rf $OUT; mkdir -p $OUT
for MODE in hostfed resident; do
  ARG=""; [ $MODE = resident ] && ARG=resident
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$MODE -- python3 $R/tools/host_fed_trace.py $ARG > $OUT/$MODE.log 2>&1
  tail -n 1 $OUT/$MODE.log
done
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for mode in ("hostfed", "resident"):
    kt = sorted(glob.glob(out + "/%s/**/*kernel_trace.csv" % mode, recursive=True))
    mt = sorted(glob.glob(out + "/%s/**/*memory_copy_trace.csv" % mode, recursive=True))
    ks = list(csv.DictReader(open(kt[-1]))) if kt else []
    ms = list(csv.DictReader(open(mt[-1]))) if mt else []
    print(mode, "kernels", len(ks), "copies", len(ms))
    if ms:
        print(" copy columns:", list(ms[0].keys()))
        big = [m for m in ms if int(m["End_Timestamp"]) - int(m["Start_Timestamp"]) > 1000000]
        for m in big[-8:]:
            print("  copy %s dur %.3f ms start %.3f" % (m.get("Direction", "?"), (int(m["End_Timestamp"]) - int(m["Start_Timestamp"])) / 1e6, int(m["Start_Timestamp"]) / 1e6))
        if len(big) > 3:
            gaps = [(int(big[i + 1]["Start_Timestamp"]) - int(big[i]["End_Timestamp"])) / 1e6 for i in range(len(big) - 1)]
            print("  gaps between big copies (ms):", [round(g, 3) for g in gaps[-10:]])
    names = collections.Counter()
    dur = collections.defaultdict(float)
    for k in ks:
        n = k["Kernel_Name"].split("(")[0].replace("void siftmi::", "")[:60]
        names[n] += 1; dur[n] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e6
    for n, c in names.most_common(14):
        print("  %-62s calls %5d total %.2f ms avg %.1f us" % (n, c, dur[n], dur[n] / c * 1e3))
PY
