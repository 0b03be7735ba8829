#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats of the bench command and
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the Gaussian-layer blur kernel.
# usage: bash tools/profile_round.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/profile_$TAG
rm -rf $OUT; mkdir -p $OUT
# 1. per-kernel time of the bench command.  With two steps in flight (the default) kernels of the two contexts run
#    concurrently and the trace's durations include the time they spend sharing the GPU; the per-kernel figures that
#    bench.py's roofline object reports come from its one-step-at-a-time pass, so the trace they are compared with is
#    taken with --pipeline 1 --serial-graph (one step at a time, the captured launch sequence not forked into per-octave
#    chains: every kernel alone on the GPU) and the trace of the default command is kept beside it.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras --pipeline 1 --serial-graph > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/bench_under_rocprof_pipelined.json 2> $OUT/bench_under_rocprof_pipelined.err
# 2. HBM bytes of the blur kernel: the five octave-0 layer launches of the pipeline itself (8 frames per launch, one call),
#    plus the calibration copy with the same access shapes and a known byte count (tools/ubench/pmc_calib.hip)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/tools/prof_pipeline.py 8 8 2 > $OUT/pmc_$C.log 2>&1
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- $R/tools/ubench/pmc_calib 3840 2176 8 5 > $OUT/calib_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, os, re
out, tag = "$OUT", "$TAG"
# --- kernel stats
stats = sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(stats[-1]))) if stats else []
with open(out + "/rocprof_kernel_stats_%s.csv" % tag, "w") as f:
    if rows:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
# per launch shape (kernel name x grid): the kernel-trace rows of the same run grouped like bench.py's roofline.by_launch_shape
trace = sorted(glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
if trace:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[-1])):
        n = r["Kernel_Name"]
        short = (n[:n.index("(")] if "(" in n else n).replace("void siftmi::", "").replace("siftmi::", "")
        d[(short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out + "/rocprof_kernel_shapes_%s.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroups_x", "grid_y", "grid_z", "calls", "avg_us", "min_us", "max_us", "total_ms"])
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k[0], k[1], k[2], k[3], len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2), round(sum(v) / 1e6, 3)])
    # roofline.frac as the kernel trace gives it (bench.py carries it as roofline.frac_rocprof): one Gaussian-layer launch per (octave, layer)
    # and step; a launch shape = one kernel instantiation (radius, decimating, flags) x grid.  Seed launches (SEEDF >= 0) and the
    # memory-only ablation (DBG != 0) are left out; the radius names the layer.
    OW = [(3840, 2160), (1920, 1080), (960, 540), (480, 270)]
    FR = 64
    by_r = collections.defaultdict(list)
    for (short, gx, gy, gz), v in d.items():
        m = re.match(r"blur_ring_kernel<(\d+), \d+, \d+, (?:true|false), (?:true|false), (\d+), (-?\d+)", short)
        m2 = re.match(r"blur2_kernel<(\d+), \d+, \d+, \d+, \d+, (true|false)", short)
        if m and int(m.group(2)) == 0 and int(m.group(3)) < 0: by_r[int(m.group(1))].append((gx * gy * gz, sum(v) / len(v), len(v), short))
        elif m2 and m2.group(2) == "false": by_r[int(m2.group(1))].append((gx * gy * gz, sum(v) / len(v), len(v), short))
    radii = {5: 1, 7: 2, 8: 3, 10: 4, 13: 5}
    tot_b = tot_ns = o0_b = o0_ns = 0.0
    rows_r = []
    ok = True
    for R, layer in radii.items():
        # octaves differ 4x in pixels, so the launch shapes of one radius in descending DURATION are octaves 0, 1, 2, ... (not in descending
        # grid size: the chunk height differs by octave and radius, R = 10 runs octave 1 on 1920 workgroups and octave 2 on 2048)
        ordered = sorted(by_r.get(R, []), key=lambda t: -t[1])
        if len(ordered) != len(OW): ok = False
        for o, t in enumerate(ordered[:len(OW)]):
            nb = 8.0 * OW[o][0] * OW[o][1] * FR
            tot_b += nb; tot_ns += t[1]
            if o == 0: o0_b += nb; o0_ns += t[1]
            rows_r.append({"octave": o, "layer": layer, "radius": R, "kernel": t[3][:60], "workgroups": t[0], "launches_in_trace": t[2], "avg_us": round(t[1] / 1e3, 2),
                           "GBps": round(nb / t[1], 1)})
    if ok and tot_ns > 0:
        json.dump({"source": "rocprofv3 --kernel-trace of 'bench.py --steps 3 --warmup 1 --no-cpu --no-extras --pipeline 1 --serial-graph' (rocprof_kernel_shapes_%s.csv)" % tag,
                   "algorithmic_bytes_per_step": tot_b, "blur_ms_per_step": round(tot_ns / 1e6, 4), "GBps_all_layers": round(tot_b / tot_ns, 1),
                   "frac_all_layers": round(tot_b / tot_ns / 8000.0, 4), "GBps_octave0": round(o0_b / o0_ns, 1), "frac_octave0": round(o0_b / o0_ns / 8000.0, 4),
                   "peak_GBps": 8000.0, "launch_shapes": rows_r}, open(out + "/roofline_rocprof_%s.json" % tag, "w"), indent=1)
        print("roofline from the kernel trace: all layers %.4f, octave 0 %.4f of 8 TB/s" % (tot_b / tot_ns / 8000.0, o0_b / o0_ns / 8000.0))
    else:
        print("roofline from the kernel trace: launch shapes did not map onto 4 octaves x 5 layers:", {R: len(v) for R, v in by_r.items()})
pst = sorted(glob.glob(out + "/trace_pipelined/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
if pst:
    prow = list(csv.DictReader(open(pst[-1])))
    with open(out + "/rocprof_kernel_stats_%s_pipelined.csv" % tag, "w") as f:
        if prow:
            w = csv.DictWriter(f, fieldnames=list(prow[0].keys())); w.writeheader(); w.writerows(prow)
for r in rows[:16]:
    print("%-74s calls %6s total_ns %12s avg_ns %10s pct %s" % (r.get("Name", "")[:74], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

def collect(d, counter, want):
    vals = collections.defaultdict(list)
    files = sorted(glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                                    # the newest pass only (a merged copy of this directory can hold older ones)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or want not in r["Kernel_Name"]: continue
            vals[(r["Kernel_Name"].split("(")[0][-100:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return vals
W, H, NF = 3840, 2160, 8
alg = 8 * W * H * NF
# calibration: known 4 B/px read + 4 B/px written with the ring kernel's access shapes
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    v = collect("calib_" + C, C, "calib_copy_kernel")
    xs = sorted(sum(v.values(), []))
    cal[C] = (4.0 * 3840 * 2176 * 8) / (xs[len(xs) // 2] * 1024.0) if xs else None    # the calibration image is 3840 x 2176 (whole 32-row steps)
print("calibration (known bytes / counter KiB x 1024):", cal)
fetch, write = collect("pmc_FETCH_SIZE", "FETCH_SIZE", "blur_ring_kernel"), collect("pmc_WRITE_SIZE", "WRITE_SIZE", "blur_ring_kernel")
res = []
for key in sorted(fetch, key=lambda k: -k[1]):
    kname, grid = key
    f, wv = sorted(fetch[key]), sorted(write.get(key, [0.0]))
    fm, wm = f[len(f) // 2], wv[len(wv) // 2]
    m = re.search(r"blur_ring_kernel<(\d+), \d+, \d+, (true|false), (true|false), \d+, (-?\d+)(?:, (?:true|false)){0,2}(?:, \d+)?>", kname)
    hbm = (cal["FETCH_SIZE"] or 2.0) * fm * 1024 + (cal["WRITE_SIZE"] or 1.0) * wm * 1024
    res.append({"kernel": kname.replace("void siftmi::", ""), "grid": grid, "launches_sampled": len(f), "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm,
                "hbm_bytes_per_launch_corrected": hbm, "radius": int(m.group(1)) if m else None, "seed": bool(m and int(m.group(4)) >= 0),
                "decimating": bool(m and m.group(2) == "true"), "activity_flags": bool(m and m.group(3) == "true")})
top = max((x["grid"] for x in res if not x["seed"]), default=0)
for x in res:
    x["pipeline_layer"] = (not x["seed"]) and x["grid"] == top          # the five octave-0 layer launches
json.dump({"workload": "octave 0 (3840x2160) of tools/prof_pipeline.py, 8 frames per launch", "algorithmic_bytes_per_launch": alg, "kernels": res,
           "calibration": {"kernel": "tools/ubench/pmc_calib.hip (same load / store shapes, 4 B read + 4 B written per pixel)",
                           "bytes_per_FETCH_SIZE_KiB": None if cal["FETCH_SIZE"] is None else cal["FETCH_SIZE"] * 1024,
                           "bytes_per_WRITE_SIZE_KiB": None if cal["WRITE_SIZE"] is None else cal["WRITE_SIZE"] * 1024},
           "correction": "hbm = cal_fetch * FETCH_SIZE * 1024 + cal_write * WRITE_SIZE * 1024 with the factors measured by the calibration copy "
                         "(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-B-per-lane read -> ~2; WRITE_SIZE is "
                         "specified exact for 16-B stores only, the ring kernel stores 8 B per lane)"},
          open(out + "/blur_hbm_traffic_%s.json" % tag, "w"), indent=1)
for x in res:
    if x["pipeline_layer"]: print("R=%2d dec=%d act=%d grid %d traffic/algorithmic = %.3f" % (x["radius"], x["decimating"], x["activity_flags"], x["grid"], x["hbm_bytes_per_launch_corrected"] / alg))
PY
