#!/bin/bash
# usage: tools/pmc_pipeline.sh <kernel-name substring> [frames] [batch] [dense]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; F=${2:-16}; B=${3:-8}; KIND=${4:-}
OUT=$R/gpurun_out/pmc_pipe
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/prof_pipeline.py $F $B 1 $KIND > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$K" not in k: continue
        acc[k[:70] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%4d mean=%.5g max=%.5g" % (c, len(v), sum(v)/len(v), max(v)))
PY
