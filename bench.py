#!/usr/bin/env python3
"""bench.py -- SIFT detect+describe throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (gray -> 2x bilinear -> Gaussian pyramid -> DoG extrema ->
refinement -> orientation -> 128-D descriptors, packed results) over a batch of synthetic
1920x1080 frames per GPU (BASELINE.json configs[2]/[3]: 64 frames per GPU, 4 octaves x 3 scales per
octave), frames already resident in HBM, results left in HBM; with N > 1 every rank processes its own
64 frames (frame-per-GPU sharding, weak scaling) and the step ends with the RCCL all-gather of the
descriptor buffers.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     Gaussian-layer blur kernel (the "pyramid kernel"): algorithmic bytes (8 B per octave
               pixel per layer) / average launch duration measured with hipEvents on the
               launch stream in a second, identical, event-instrumented pass of K steps.
  cpu_baseline the CPU oracle (a port of the reference's algorithm; kind "port") timed on the host
               cores on a bounded sample of the same frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, N_OCT, NSPO = 1920, 1080, 4, 3
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_frames(n, distinct):
    from tests.synth import blob_frame
    base = [blob_frame(W, H, i) for i in range(min(n, distinct))]
    return np.stack([base[i % len(base)] for i in range(n)])


def cpu_baseline(frames, seconds_budget=20.0):
    """Times the oracle (oracle/sift_oracle.c, OpenMP) on a bounded sample of the same frames."""
    from oracle import pyoracle
    orc = pyoracle.Oracle(W, H, n_octaves=N_OCT, nspo=NSPO)
    t0 = time.time()
    done = 0
    n_desc = 0
    while done < len(frames) and (done == 0 or (time.time() - t0) * (done + 1) / done < seconds_budget):
        tot, _ = orc.detect_describe_counts(frames[done])
        n_desc += tot
        done += 1
    dt = time.time() - t0
    return {"value": round(done * W * H / dt / 1e6, 4), "unit": "Mpixels/s", "cores": pyoracle.num_threads(), "kind": "port",
            "sample": "%d x %dx%d synthetic frames (same generator), %d octaves, %.1f s, %d descriptors" % (done, W, H, N_OCT, dt, n_desc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SIFTMI_BATCH", "32")), help="frames processed in lock-step per launch")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames (cycled to fill the batch)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # Library chatter (e.g. RCCL's version banner at communicator init) must not reach stdout: the
    # contract is ONE JSON line.  Point fd 1 at stderr for the run and restore it for the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        log("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SIFTMI_FORCE_GATHER=1 exercises the RCCL exchange with a single rank (smoke test on a 1-GPU box)
    force_gather = os.environ.get("SIFTMI_FORCE_GATHER") == "1" and "RANK" in os.environ
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as ge
    ge.build()
    import siftmetal_amd as sm
    from siftmetal_amd import stream as smstream

    F = args.frames
    frames_np = make_frames(F, args.distinct)
    # every rank gets different frames (rotate) so the gathered descriptors are not copies
    frames_np = np.roll(frames_np, rank, axis=0)
    d_frames = torch.from_numpy(frames_np).to(dev)
    eng = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=min(args.batch, F))
    runner = smstream.FrameStream(eng, F, device=dev, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        runner.run(d_frames)
        if world > 1 or force_gather:
            runner.all_gather()

    # reference counts from one synchronised step; every later step (graph replays included) must reproduce them
    step()
    barrier()
    first = runner.results_host()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * F * W * H * args.steps / dt / 1e6

    res = runner.results_host()
    if (res["n_keypoints"], res["n_descriptors"]) != (first["n_keypoints"], first["n_descriptors"]) or \
            not np.array_equal(res["counts"], first["counts"]):
        raise SystemExit("bench: results of the last timed step differ from the first step (%d/%d vs %d/%d keypoints/descriptors): "
                         "the timed region did not compute the workload" %
                         (res["n_keypoints"], res["n_descriptors"], first["n_keypoints"], first["n_descriptors"]))
    log("rank0 per-step: %d keypoints, %d descriptors over %d frames; %.3f ms/step, %.3f ms/frame" %
        (res["n_keypoints"], res["n_descriptors"], F, ms_per_step, ms_per_step / F))

    out = {"metric": "Mpixels/sec detect+describe (1920x1080, 4 octaves)", "value": round(value, 2), "unit": "Mpixels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d x 1920x1080 BGRA8 frames per GPU per step (BASELINE configs[2]/[3]), %d octaves x %d scales/octave, "
                                  "detect+describe, frames and results resident in HBM%s" %
                                  (F, N_OCT, NSPO, ", RCCL all-gather of descriptors" if world > 1 else ""),
                      "frames_per_gpu": F, "lockstep_batch": eng.max_batch, "parallelism": "frame-per-GPU x%d" % world,
                      "keypoints_per_step_rank0": res["n_keypoints"], "descriptors_per_step_rank0": res["n_descriptors"]}}

    if rank == 0 and not args.no_roofline:
        # second, identical pass with per-launch hipEvents on the launch stream
        eng.enable_timings(True)
        eng.reset_timings()
        for _ in range(args.steps):
            runner.run(d_frames)
        torch.cuda.synchronize()
        tm = eng.timings()
        eng.enable_timings(False)
        blur_ms, blur_n = tm["blur"]
        # one blur launch of octave o moves 8 B x N_o x (frames in the launch); per step every octave
        # gets nspo+2 launches per sub-batch
        bytes_per_step = sum(eng.blur_algorithmic_bytes(o) for o in range(N_OCT)) * (NSPO + 2) * F
        total_bytes = bytes_per_step * args.steps
        achieved = total_bytes / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0
        stage_ms = {k: round(v[0] / args.steps, 4) for k, v in tm.items()}
        log("stage ms/step:", stage_ms)
        per_layer = {}
        for layer in range(1, NSPO + 3):
            ms = eng.time_blur(0, layer, 10)
            per_layer["o0_l%d_taps%d" % (layer, len(eng.weights(layer)))] = round(eng.blur_algorithmic_bytes(0) * eng.max_batch / (ms * 1e-3) / 1e9, 1)
        log("octave-0 blur GB/s by layer:", per_layer)
        # HBM bytes per launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; tools/profile_round.sh)
        traffic, traffic_src = None, None
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("blur_hbm_traffic_")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if prof:
            pj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
            import re
            # the five octave-0 layer launches of this pipeline: radius from the schedule; layer nspo carries the fused
            # decimation output (DEC), layers 2 ... nspo+1 write the extrema activity flags (ACT, re-reads its input layer)
            def find(radius, dec, act):
                for k in pj["kernels"]:
                    m = re.search(r"march_kernel<(\d+), \d+, \d+, (true|false), (?:true|false), \d+, \d+, \d+, \d+, (true|false)>", k["kernel"])
                    if m and int(m.group(1)) == radius and (m.group(2) == "true") == dec and (m.group(3) == "true") == act:
                        return k
                return None
            ks = []
            for layer in range(1, NSPO + 3):
                k = find(len(eng.weights(layer)) // 2, layer == NSPO, 2 <= layer <= NSPO + 1)
                if k:
                    ks.append(k)
            if len(ks) == NSPO + 2:
                ratio = sum(k["hbm_bytes_per_launch_corrected"] for k in ks) / (pj["algorithmic_bytes_per_launch"] * len(ks))
                traffic = int(ratio * total_bytes / max(blur_n, 1))
                traffic_src = ("profiles/%s: PMC bytes / algorithmic bytes = %.3f averaged over the five octave-0 layer launches "
                               "(the one with the fused decimation output writes 1.06x more), scaled to the average launch" % (prof[-1], ratio))
        out["roofline"] = {"bound": "hbm", "kernel": "blur2_kernel<R> / blur_march_kernel<R> (one Gaussian layer, fused X+Y)", "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_source": traffic_src,
                           "launches": blur_n, "avg_launch_ms": round(blur_ms / max(blur_n, 1), 5),
                           "algorithmic_bytes_per_launch_avg": int(total_bytes / max(blur_n, 1)),
                           "octave0_GBps_by_layer": per_layer, "stage_ms_per_step": stage_ms,
                           "measured_in": "second identical pass of K steps, hipEvents around every launch on the launch stream"}
    if rank == 0:
        # BASELINE configs[1]: ONE 1920x1080 frame per call (lock-step batch 1, hipGraph replay), frame in HBM
        e1 = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=1)
        r1 = smstream.FrameStream(e1, 1, device=dev)
        one = d_frames[:1].contiguous()
        for _ in range(3):
            r1.run(one)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 30
        for _ in range(n1):
            r1.run(one)
        torch.cuda.synchronize()
        ms1 = (time.perf_counter() - t1) / n1 * 1e3
        out["config"]["single_frame"] = {"workload": "BASELINE configs[1]: one 1920x1080 frame per call, 4 octaves", "ms_per_frame": round(ms1, 4),
                                         "Mpixels_per_s": round(W * H / ms1 / 1e3, 1)}
        log("single frame: %.3f ms (%.0f Mpixels/s)" % (ms1, W * H / ms1 / 1e3))
        del r1, e1
    if rank == 0 and not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(frames_np)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force_gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
