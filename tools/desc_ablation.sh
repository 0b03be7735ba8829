#!/bin/bash
# descriptor-kernel ablations on dense frames: stage times of every variant library under tools/tmp_variants (tools/build_variant.sh).
# The -DSIFTMI_DESC_ABL=n / -DSIFTMI_ORI_ABL=n hooks are kept as tools/experiments/sample_loop_ablations_r06.diff (apply with patch -p0).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for V in "" $(ls tools/tmp_variants/*.so 2>/dev/null); do
  if [ -z "$V" ]; then python3 tools/dense_stage_times.py 5 dense; else SIFTMI_LIB=$R/$V python3 tools/dense_stage_times.py 5 dense; fi
done
