#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats of the bench command and
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the Gaussian-layer blur kernel.
# usage: bash tools/profile_round.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT
# 1. per-kernel time of exactly the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2. HBM bytes of the blur kernel: octave 0, all five layers, 8 frames per launch
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/prof_blur.py 0 0 5 8 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/prof_blur.py 0 0 5 8 > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
# --- kernel stats
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
rows = []
if stats:
    rows = list(csv.DictReader(open(stats[0])))
with open(out + "/kernel_stats_$TAG.csv", "w") as f:
    if rows:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:14]:
    print("%-70s calls %6s total_ns %12s avg_ns %10s pct %s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
# --- traffic: dispatches of the full-size octave-0 launches = those with the largest grid
def collect(d, name):
    vals = collections.defaultdict(list)
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"]
            if "blur_march_kernel" not in k: continue        # the octave-0 launches of tools/prof_blur.py use the marching kernel
            vals[(k.split("(")[0][-90:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return vals
fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
res = []
W, H, NF = 3840, 2160, 8
alg = 8 * W * H * NF
for key in sorted(fetch, key=lambda k: -k[1]):
    kname, grid = key
    if "true" in kname.split("blur2_kernel")[-1].split(",")[5:6]: pass
    f = sorted(fetch[key]); wv = sorted(write.get(key, [0]))
    fm, wm = f[len(f) // 2], wv[len(wv) // 2]
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
    # streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact for 16-B stores
    hbm = 2 * fm * 1024 + wm * 1024
    res.append({"kernel": kname, "grid": grid, "launches_sampled": len(f), "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm,
                "hbm_bytes_per_launch_corrected": hbm})
res = [r for r in res if r["grid"] == max(x["grid"] for x in res)]     # octave-0 launches only
json.dump({"workload": "octave 0 (3840x2160), 8 frames per launch", "algorithmic_bytes_per_launch": alg, "kernels": res,
           "correction": "hbm = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"},
          open(out + "/blur_hbm_traffic_$TAG.json", "w"), indent=1)
for r in res: print(r["kernel"][-50:], r["grid"], "traffic/algorithmic = %.2f" % (r["hbm_bytes_per_launch_corrected"] / alg))
PY
