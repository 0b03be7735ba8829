#!/bin/bash
# Where the time of a small siftmi_match_descriptors call goes: kernel + copy trace of ten 20k x 20k calls.
# usage (GPU box): bash tools/match_call_trace.sh [n]
N=${1:-20000}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/match_call_trace
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "$R")
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
eng = sm.Engine(64, 64, n_octaves=1)
rng = np.random.default_rng(0)
n = $N
tgt = np.zeros(n, sm.descriptor_dtype); tgt["features"] = np.clip(np.abs(rng.normal(0, 40, (n, 128))), 0, 255)
src = np.zeros(n, sm.descriptor_dtype); src["features"] = tgt["features"][rng.integers(0, n, n)]
d_src, d_tgt = smstream.DeviceFrames(src.view(np.uint8), 0), smstream.DeviceFrames(tgt.view(np.uint8), 0)
res, cnt = C.c_void_p(), C.c_int64()
for k in range(12):
    t0 = time.perf_counter()
    _capi.check(eng.L.siftmi_match_descriptors(eng.h, d_src.ptr, n, d_tgt.ptr, n, 1, 1.176, 0.6, C.byref(res), C.byref(cnt)))
    print("call %d: %.1f us" % (k, (time.perf_counter() - t0) * 1e6))
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/run.log 2>&1
tail -4 $OUT/run.log
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
# the last call: events after the last-but-one finalize kernel
fin = [i for i, e in enumerate(ev) if "match_finalize" in e[2]]
lo = fin[-2] + 1
t0 = ev[lo][0]
for s, e, n in ev[lo:]:
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
