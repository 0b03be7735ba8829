#!/bin/bash
# Split count x pre-pass sweep of the matcher kernel (tools/ubench/match_variants) for the launch plan in siftmi_api.hip (match_plan).
# usage (GPU box): bash tools/match_plan_sweep.sh
cd $(dirname $0)/ubench
for n in 20000 30000 50000 70000 100000 200000; do
  g=$(( (n + 511) / 512 ))
  for blocks in 256 400 512 768 1000 1024 1500 2000 2048 2500 3000; do
    k=$(( blocks / g )); [ $k -lt 2 ] && continue
    for p in 0 512; do
      ./match_variants $n $n $k $p | head -1
    done
  done
done
