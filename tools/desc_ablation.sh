#!/bin/bash
# descriptor-kernel ablations on dense frames (tools/build_variant.sh abl1|abl2|abl3, -DSIFTMI_DESC_ABL=n): stage times
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for V in "" $(ls tools/tmp_variants/*.so 2>/dev/null); do
  if [ -z "$V" ]; then python3 tools/dense_stage_times.py 5 dense; else SIFTMI_LIB=$R/$V python3 tools/dense_stage_times.py 5 dense; fi
done
