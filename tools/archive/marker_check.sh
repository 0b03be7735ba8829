#!/bin/bash
# The roctx stage ranges of the library in a rocprofv3 marker trace (run on the GPU box): prints the distinct range names.
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mk
rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/mk -- python3 $R/tools/prof_pipeline.py 4 2 1 > /tmp/mk.log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/mk/**/*marker*trace*.csv", recursive=True)
print("marker trace files:", [f.split("/")[-1] for f in fs])
c = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        c[r.get("Function") or r.get("Name") or str(r)] += 1
for k, v in c.most_common(12): print("%5d  %s" % (v, k))
PY
