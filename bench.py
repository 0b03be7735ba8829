#!/usr/bin/env python3
"""bench.py -- SIFT detect+describe throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path (gray -> 2x bilinear -> Gaussian pyramid -> DoG extrema ->
refinement -> orientation -> 128-D descriptors, packed results) over a batch of synthetic
1920x1080 frames per GPU (BASELINE.json configs[2]/[3]: 64 frames per GPU, 4 octaves x 3 scales per
octave).  `value` is the metric as SURVEY.md 8d words it: the frames start in page-locked HOST memory and cross PCIe
(H2D) inside the timed region, and every step's packed keypoints + descriptors are copied back to page-locked host memory
(D2H) inside it; with N > 1 every rank processes its own 64 frames (frame-per-GPU sharding, weak scaling) and every step's
results are also all-gathered over RCCL.  The same step with the frames already resident in HBM and the results left there
is reported beside it (`resident_Mpixels_per_s`), with `h2d_floor_ms` = the step's upload at the synchronous H2D rate measured
in the run, so that the line itself says when a step is PCIe-bound.  Rank 0 prints ONE JSON line.

The timed loop calls only the C ABI's frame stream (siftmi_stream_submit_host / siftmi_stream_result_host /
siftmi_exchange_gather through the ctypes binding siftmetal_amd/stream.py): uploads on a copy stream into rotating staging
buffers, two steps in flight, copy-back on a third stream and the RCCL exchange are all inside libsiftmi.so.  torch is used for
the contract's synchronize() and, with N > 1, as the control plane only (gloo: hands rank 0's ncclUniqueId to the other ranks,
barrier, max over ranks).  A rank that dies or hangs ends the job: the exchange's waits are bounded (SIFTMI_EXCHANGE_TIMEOUT_S),
the error propagates as an exception, the process exits non-zero; nothing is restarted or re-executed.

`--gpus N` with N > 1 and no torchrun environment: this process spawns the N ranks itself (before any
GPU call) and relays rank 0's line; it exits non-zero if fewer than N devices are visible or the
process group does not come up with N ranks.

Extra objects in the line:
  roofline     Gaussian-layer blur kernel (the "pyramid kernel"): algorithmic bytes (8 B per octave
               pixel per layer) / average launch duration measured with hipEvents on the
               launch stream in a second, identical, event-instrumented pass of K steps.
  cpu_baseline the CPU oracle (a port of the reference's algorithm; kind "port") timed on the host
               cores on a bounded sample of the same frames: all cores, and one thread.
  config.single_frame / config.host_io / config.dense
               BASELINE configs[1] (one frame per call), the same 64-frame step through the SYNCHRONOUS
               host-buffer call (siftmi_detect_describe_batch), and the step on 64 dense natural-texture
               frames (mirror-tiled butterfly, ~7x the keypoints), resident and host-fed.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, N_OCT, NSPO = 1920, 1080, 4, 3
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (the copy rate is MEASURED in the run: roofline.peak_measured)
MFMA_I8_DENSE_PEAK_TFLOPS = 5000.0      # same guide: int8 dense = 2x bf16 (2.5 PFLOP/s)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_frames(n, distinct, indices=None):
    """n synthetic frames, `distinct` of them different (default: all) -- generated on a few host threads (0.25 s per frame).
    indices: positions in the (endless) synthetic stream, frame g = blob field g mod distinct; default 0 ... n-1."""
    from concurrent.futures import ThreadPoolExecutor
    from tests.synth import blob_frame
    idx = list(range(n)) if indices is None else list(indices)
    k = max(min(n, distinct) if indices is None else distinct, 1)
    need = sorted(set(g % k for g in idx))
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        base = dict(zip(need, ex.map(lambda i: blob_frame(W, H, i), need)))
    return np.stack([base[g % k] for g in idx])


def match_extra(eng, local_rank):
    """SURVEY.md 8f row f1 in the driver's line: siftmi_match_descriptors (SIFTDescriptor.match, SIFT/SIFTDescriptor.swift:298-361;
    int8 MFMA, exact int32 accumulation) on descriptors resident in HBM, whole call including the 12 B/source result copy."""
    import ctypes as C
    from siftmetal_amd import _capi, stream as smstream
    import siftmetal_amd as sm
    out = {}
    rng = np.random.default_rng(0)
    for ns, nt in ((2500, 2300), (20000, 20000), (100000, 100000), (200000, 200000)):
        tgt = np.zeros(nt, sm.descriptor_dtype)
        tgt["features"] = np.clip(np.abs(rng.normal(0, 40, (nt, 128))), 0, 255)
        src = np.zeros(ns, sm.descriptor_dtype)
        src["features"] = np.clip(tgt["features"][rng.integers(0, nt, ns)].astype(np.int32) + rng.integers(-12, 13, (ns, 128)), 0, 255)
        d_src, d_tgt = smstream.DeviceFrames(src.view(np.uint8), local_rank), smstream.DeviceFrames(tgt.view(np.uint8), local_rank)
        res, n = C.c_void_p(), C.c_int64()

        def call():
            _capi.check(eng.L.siftmi_match_descriptors(eng.h, d_src.ptr, ns, d_tgt.ptr, nt, 1, 1.176, 0.6, C.byref(res), C.byref(n)))

        call()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        tf = ns * nt * 256 / dt / 1e12
        # the device-resident variant: matches packed in source order + their count stay in HBM, calls queued without a host synchronisation
        d_out = smstream.DeviceFrames(np.zeros(ns * 12 + 4, np.uint8), local_rank)

        def call_dev():
            eng.match_device(d_src.ptr, ns, d_tgt.ptr, nt, d_out.ptr + 4, d_out.ptr)

        call_dev()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call_dev()
        eng.synchronize()
        dtd = (time.perf_counter() - t0) / reps
        cnt = np.zeros(1, np.int32)
        _capi.check(eng.L.siftmi_memcpy(cnt.ctypes.data, d_out.ptr, 4, 1))
        if int(cnt[0]) != int(n.value):
            raise SystemExit("bench: the device-resident matcher found %d matches, the host call %d" % (int(cnt[0]), int(n.value)))
        tfd = ns * nt * 256 / dtd / 1e12
        key = "%dk_x_%dk" % (ns // 1000, nt // 1000) if ns >= 10000 else "%d_x_%d" % (ns, nt)
        out[key] = {"ms_per_call": round(dt * 1e3, 4), "Gpairs_per_s": round(ns * nt / dt / 1e9, 1), "matches": int(n.value),
                    "int8_TFLOPs": round(tf, 1), "frac_of_int8_mfma_peak": round(tf / MFMA_I8_DENSE_PEAK_TFLOPS, 4),
                    "device_output_ms_per_call": round(dtd * 1e3, 4), "device_output_frac_of_int8_mfma_peak": round(tfd / MFMA_I8_DENSE_PEAK_TFLOPS, 4)}
        d_src.close(); d_tgt.close(); d_out.close()
    out["workload"] = ("siftmi_match_descriptors (brute force + ratio test, SIFTDescriptor.match), descriptors resident in HBM, whole call incl. the host copy of "
                       "the match records; device_output_*: siftmi_match_descriptors_device, matches packed in source order + count left in HBM, %d calls queued and "
                       "one synchronisation; bound = int8 MFMA, peak %.0f TFLOP/s dense" % (10, MFMA_I8_DENSE_PEAK_TFLOPS))
    return out


def make_dense_frames(n):
    """SURVEY.md 8d 'dense stress variant': the reference's test image mirror-tiled to 1920x1080; frame i is the mosaic
    rolled by 16 i columns so that the frames differ."""
    from PIL import Image
    im = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "butterfly.png")))
    b = np.ascontiguousarray(im[..., [2, 1, 0, 3]])
    row = np.concatenate([b, b[:, ::-1], b, b[:, ::-1]], axis=1)
    full = np.ascontiguousarray(np.concatenate([row, row[::-1], row, row[::-1]], axis=0)[:H, :W])
    return np.stack([np.roll(full, 16 * (i % 8), axis=1) for i in range(n)])


def cpu_baseline(frames, seconds_budget=16.0):
    """Times the CPU restatement of the reference's algorithm (oracle/sift_oracle.c) on a bounded sample of the same frames, on every core.
    Frames are independent, so the all-cores figure is one single-threaded worker PROCESS per physical core, each on its own frames
    (oracle/cpu_worker.py) -- the way a CPU deployment of this algorithm would run -- with the restatement built for speed (-O3, AVX2 +
    FMA: oracle/libsift_oracle_tuned.so; the IEEE-strict build stays the checker).  Rounds 4-5 ran threads inside this process: 128 of
    them reached 9-11x one thread, because the page faults of every frame's stacks serialise on one process's memory-map lock.
    Then one thread of the same build, and one thread of the checker build for reference."""
    import tempfile
    from oracle import pyoracle
    odir = os.path.join(ROOT, "oracle")
    tuned = os.path.join(odir, "libsift_oracle_tuned.so")
    lib = tuned if os.path.exists(tuned) else os.path.join(odir, "libsift_oracle.so")
    threads_hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, threads_hw // 2) if threads_hw >= 16 else threads_hw      # physical cores of an SMT host
    # ... of which a container may only get a share: the pool's boxes show 256 hardware threads under a cgroup quota of 16 CPUs (cpu.max
    # "1600000 100000"; tools/cpu_quota_probe.py: 128 busy loops run 12.5x one).  More workers than the quota only throttle each other
    # (rounds 3-5: "9-11 effective cores of 128").
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    if quota:
        workers = max(1, min(workers, int(quota)))
    workers = min(workers, int(os.environ.get("SIFTMI_BENCH_CPU_WORKERS", "100000")))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(suffix=".npy", prefix="siftmi_cpu_baseline_", dir=shm)
    os.close(fd)
    try:
        np.save(path, frames[:min(len(frames), 32)])

        def run(n_workers, budget, which):
            env = dict(os.environ, SIFT_ORACLE_LIB=which, OMP_NUM_THREADS="1")
            cmd = lambda wi: [sys.executable, os.path.join(odir, "cpu_worker.py"), path, str(wi), str(n_workers), "1", str(budget),
                              str(W), str(H), str(N_OCT), str(NSPO)]
            t0 = time.time()
            procs = [subprocess.Popen(cmd(wi), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for wi in range(n_workers)]
            res = []
            for p in procs:
                o, _ = p.communicate()
                lines = [ln for ln in o.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
                if p.returncode == 0 and lines:
                    res.append(json.loads(lines[-1]))
            return res, time.time() - t0

        res, wall = run(workers, seconds_budget, lib)
        if not res:
            raise RuntimeError("no cpu_baseline worker finished")
        done, n_desc = sum(r["frames"] for r in res), sum(r["descriptors"] for r in res)
        span = max(r["seconds"] for r in res)                 # the workers start within a fraction of a second of each other
        out = {"value": round(done * W * H / span / 1e6, 3), "unit": "Mpixels/s", "cores": len(res), "kind": "port",
               "host": {"hardware_threads": threads_hw, "cgroup_cpu_quota": quota},
               "implementation": "oracle/sift_oracle.c (CPU restatement of the reference's algorithm, stage by stage) built -O3 -march=x86-64-v3 "
                                 "(%s); %d single-threaded worker processes, one frame at a time each" % (os.path.basename(lib), len(res)),
               "sample": "%d x %dx%d synthetic frames (same generator), %d octaves, %.1f s (%.1f s with process start-up), %d descriptors" %
                         (done, W, H, N_OCT, span, wall, n_desc)}
        one, _ = run(1, 4.0, lib)
        if one:
            out["single_thread"] = {"value": round(one[0]["frames"] * W * H / one[0]["seconds"] / 1e6, 4), "unit": "Mpixels/s", "cores": 1,
                                    "sample": "%d frame(s), %.1f s, %d descriptors" % (one[0]["frames"], one[0]["seconds"], one[0]["descriptors"])}
            out["cores_effective"] = round(out["value"] / out["single_thread"]["value"], 1) if out["single_thread"]["value"] > 0 else None
        chk, _ = run(1, 4.0, os.path.join(odir, "libsift_oracle.so"))
        if chk:
            out["single_thread_checker_build"] = {"value": round(chk[0]["frames"] * W * H / chk[0]["seconds"] / 1e6, 4), "unit": "Mpixels/s",
                                                  "what": "the IEEE-strict -O2 build the parity tests check against (oracle/libsift_oracle.so), one thread"}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    return out


def bind_to_gpu_numa_node(props):
    """Pins this rank's threads (and, by first touch, its page-locked frame buffers) to the NUMA node its GPU hangs off: with one rank
    per GPU every rank streams 57 GB/s out of host DRAM, and a buffer on the other socket crosses the inter-socket links.  Returns
    (node, previous affinity mask) or (None, None) when the topology cannot be read (then nothing is changed)."""
    try:
        if os.environ.get("SIFTMI_BENCH_NO_AFFINITY") == "1":
            return None, None
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None, None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        prev = os.sched_getaffinity(0)
        cpus &= prev
        if not cpus:
            return None, None
        os.sched_setaffinity(0, cpus)
        return node, prev
    except Exception:
        return None, None


def self_launch(args):
    """--gpus N > 1 without a torchrun environment: spawn the ranks (no GPU call has happened in this process)."""
    import torch
    n_dev = torch.cuda.device_count()                      # does not initialise the GPU runtime
    if n_dev < (1 if args.share_gpu else args.gpus):
        print("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, n_dev), file=sys.stderr)
        return 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print("bench.py: the %d-rank launch failed (exit code %d, %d JSON lines)" % (args.gpus, p.returncode, len(lines)), file=sys.stderr)
        return p.returncode or 4
    print(lines[-1], flush=True)
    return 0


def timed_steps(step, barrier, steps, warmup):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SIFTMI_BATCH", "64")), help="frames processed in lock-step per launch")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames (cycled to fill the batch when fewer than --frames)")
    ap.add_argument("--march-min-blocks", type=int, default=0, help="siftmi_config.blur_march_min_blocks (0 = library default)")
    ap.add_argument("--serial-graph", action="store_true", help="siftmi_config.graph_fork = -1: the captured launch sequence stays one chain, "
                    "so that every kernel runs alone (per-kernel traces: tools/profile_round.sh)")
    ap.add_argument("--pipeline", type=int, default=2, choices=(1, 2),
                    help="steps in flight: 2 alternates consecutive steps between two contexts (two pyramids) on two streams")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: let the N ranks share the visible device(s) (rank r on device r mod n_devices).  Real RCCL refuses two ranks "
                         "on one GPU; with SIFTMI_RCCL_LIB=tests/c/libfake_rccl.so the exchange runs over shared memory, which exercises "
                         "self-launch -> unique-id broadcast -> siftmi_exchange_* with N ranks -> one JSON line on a 1-GPU box")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip single_frame / host_io / dense")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # Library chatter (e.g. RCCL's version banner at communicator init) must not reach stdout: the
    # contract is ONE JSON line.  Point fd 1 at stderr for the run and restore it for the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = node_rank = int(os.environ.get("LOCAL_RANK", "0"))      # node_rank: position on the node (who builds); local_rank: HIP device index
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if args.share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif torch.cuda.device_count() == 1 and local_rank > 0:
        local_rank = 0                                      # a launcher that shows every rank its own GPU only (HIP_VISIBLE_DEVICES per rank)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: local rank %d but only %d HIP device(s) visible" % (local_rank, torch.cuda.device_count()))
    numa_node, prev_affinity = bind_to_gpu_numa_node(torch.cuda.get_device_properties(local_rank))     # before any page-locked allocation
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SIFTMI_FORCE_GATHER=1 exercises the RCCL exchange with a single rank (smoke test on a 1-GPU box)
    force_gather = os.environ.get("SIFTMI_FORCE_GATHER") == "1"
    use_dist = world > 1 or force_gather
    if world > 1:
        # control plane only (unique id, barrier, max over ranks): the data path's RCCL communicator lives in libsiftmi.so
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # one node: the loopback interface (the box's hostname may not resolve)
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=20))   # (rank 0's extra measurements run while the others wait at the last barrier)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: process group has %d ranks, expected %d" % (dist.get_world_size(), world))

    import __graft_entry__ as ge
    if node_rank == 0:                                      # one build per node; the others load the finished library
        ge.build()
    if world > 1:
        dist.barrier()
    import siftmetal_amd as sm
    from siftmetal_amd import _capi, stream as smstream

    F = args.frames
    # the stream of world x F frames per step, sharded frame-per-GPU by the rule the library's hosts use (stream.shard_frames: frame g -> rank
    # g mod world): this rank takes frames rank, rank + world, ... (one rank: frames 0 ... F-1)
    my_frames = smstream.shard_frames(world * F, world, rank)
    frames_np = make_frames(F, args.distinct, my_frames)
    hpin = sm.pinned_empty(frames_np.shape, np.uint8)            # the frames as a capture loop delivers them: page-locked host memory
    hpin[...] = frames_np
    d_frames = smstream.DeviceFrames(frames_np, local_rank)      # HBM through siftmi_device_alloc / siftmi_memcpy (the resident figure, roofline pass)
    tune = {"blur_march_min_blocks": args.march_min_blocks} if args.march_min_blocks > 0 else {}
    if args.serial_graph:
        tune["graph_fork"] = -1
    eng = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=min(args.batch, F), **tune)
    uid = None
    transport = None
    if use_dist:
        origin = _capi.load().siftmi_exchange_transport().decode("utf-8", "replace")
        transport = ("test transport %s (shared memory between ranks that share a GPU; NOT RCCL)" % origin) if "fake_rccl" in origin else "librccl: %s" % origin
        box = [smstream.Exchange.make_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid = box[0]
    # the timed stream: host-fed (upload of step k+1 under the kernels of step k), consecutive steps alternate between two contexts
    # (step k+1's HBM-bound dense stages run under step k's VALU-bound keypoint stages), every step's packed results copied back to
    # page-locked host memory and read `back` steps late; `plain` (one context, one step at a time, resident frames) is what the
    # per-kernel measurements use
    runner = smstream.FrameStream(eng, F, device=dev, world_size=world, overlap_gather=use_dist, pipeline=args.pipeline,
                                  result_sets=2 * args.pipeline, rank=rank, unique_id=uid)
    plain = smstream.FrameStream(eng, F, device=dev)
    back = args.pipeline
    rccl_ranks = runner.exchange.ranks()[0] if use_dist else 0       # what the communicator itself reports, not the environment
    if use_dist and rccl_ranks != world:
        raise SystemExit("bench.py: the communicator has %d ranks, WORLD_SIZE is %d" % (rccl_ranks, world))

    def dev_sync():
        torch.cuda.synchronize()
        _capi.check(_capi.load().siftmi_device_synchronize(local_rank))

    def barrier():
        if use_dist:
            runner.exchange.wait()   # bounded: a collective that cannot complete (a rank is gone) raises here instead of hanging the device sync
        if world > 1:
            dist.barrier()
        dev_sync()

    def step():
        runner.run_host(hpin)
        if use_dist:
            runner.all_gather()      # on a side stream, from one of the rotating result sets: overlaps the next step's kernels
        if runner.step_no >= back:   # views of the stream's page-locked buffers: nothing is allocated in the timed loop
            return runner.results_host(back=back, copy=False)
        return None

    # reference counts from one synchronised step; every later step (graph replays included) must reproduce them
    step()
    barrier()
    first = runner.results_host()
    # warm-up: every (staging buffer, result set, context) pairing seen twice (captured, then replayed), every result set's
    # page-locked host buffers allocated
    for _ in range(2 * runner.n_sets + 2):
        step()
    barrier()
    g0 = runner.exchange.stats() if use_dist else None
    dt = timed_steps(step, barrier, args.steps, args.warmup)
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        gathered_dt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered_dt, t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in gathered_dt]
        dt = max(float(x.item()) for x in gathered_dt)
    ms_per_step = dt / args.steps * 1e3
    value = world * F * W * H * args.steps / dt / 1e6
    gather_ms = gather_bytes = None
    regathered = overflowed = 0
    checksum = None
    if use_dist:
        regathered, overflowed = runner.exchange.finish()
        if overflowed:
            raise SystemExit("bench: list overflow in %d steps" % overflowed)
        g1 = runner.exchange.stats()
        gather_ms = (g1["ms"] - g0["ms"]) / max(g1["gathers"] - g0["gathers"], 1)
        gather_bytes = g1["bytes_last"]
        # what the collective delivered: a checksum of every rank's row of the last gathered step, equal on every rank or the line is not printed
        import zlib
        gh = runner.exchange.result_host(0)
        crc = 0
        for r in range(world):
            crc = zlib.crc32(gh["descriptors"][r].tobytes(), zlib.crc32(gh["keypoints"][r].tobytes(), crc))
        crcs = [crc]
        if world > 1:
            tc = torch.tensor([crc], dtype=torch.int64)
            got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(got, tc)
            crcs = [int(x.item()) for x in got]
        if len(set(crcs)) != 1 or not gh["complete"]:
            raise SystemExit("bench: the ranks hold different gathered results (crc32 by rank %s, complete %s)" % (crcs, gh["complete"]))
        checksum = {"crc32_of_gathered_keypoints_and_descriptors": "%08x" % crc, "equal_on_all_ranks": True, "ranks_compared": len(crcs),
                    "records_gathered": [int(sum(len(x) for x in gh["keypoints"])), int(sum(len(x) for x in gh["descriptors"]))]}

    res = runner.results_host()
    by_rank = None
    if use_dist:
        # What every rank did, gathered over the control plane (gloo), so that the first real multi-GPU record shows WHERE weak scaling goes if
        # it goes: per-rank step time, device, NUMA node, frames, and the page-locked -> HBM copy rate with every rank copying AT THE SAME TIME
        # (N x 57 GB/s is ~460 GB/s of host-memory reads on an 8-GPU node).  And the row check: row r of the gathered step is rank r's own
        # packed result, byte for byte (crc32), or the line is not printed.
        import zlib
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            _capi.check(_capi.load().siftmi_memcpy(d_frames.ptr, hpin.ctypes.data, hpin.nbytes, 0))
        my_h2d = 3 * hpin.nbytes / (time.perf_counter() - t1) / 1e9
        own_crc = zlib.crc32(res["descriptors"].tobytes(), zlib.crc32(res["keypoints"].tobytes(), 0))
        mine = {"rank": rank, "device": local_rank, "host_numa_node": numa_node, "ms_per_step": round(rank_ms[rank] if world > 1 else rank_ms[0], 4),
                "concurrent_h2d_GBps": round(my_h2d, 1), "frames": [int(g) for g in my_frames[:4]] + (["..."] if len(my_frames) > 4 else []),
                "own_results_crc32": "%08x" % own_crc, "keypoints": int(res["n_keypoints"]), "descriptors": int(res["n_descriptors"])}
        everyone = [mine]
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
        gh = runner.exchange.result_host(0)
        rows = ["%08x" % zlib.crc32(gh["descriptors"][r].tobytes(), zlib.crc32(gh["keypoints"][r].tobytes(), 0)) for r in range(world)]
        if rows != [e["own_results_crc32"] for e in everyone]:
            raise SystemExit("bench: a row of the gathered step is not that rank's own result (rows %s, own %s)" % (rows, [e["own_results_crc32"] for e in everyone]))
        by_rank = {k: [e[k] for e in everyone] for k in mine}
        by_rank["gathered_row_equals_own_result"] = True
        by_rank["sharding"] = "stream.shard_frames: frame g of the %d-frame step -> rank g mod %d" % (world * F, world)
    if (res["n_keypoints"], res["n_descriptors"]) != (first["n_keypoints"], first["n_descriptors"]) or \
            not np.array_equal(res["counts"], first["counts"]):
        raise SystemExit("bench: results of the last timed step differ from the first step (%d/%d vs %d/%d keypoints/descriptors): "
                         "the timed region did not compute the workload" %
                         (res["n_keypoints"], res["n_descriptors"], first["n_keypoints"], first["n_descriptors"]))
    d2h_bytes = int(res["keypoints"].nbytes + res["descriptors"].nbytes + res["counts"].nbytes)
    log("rank0 per-step: %d keypoints, %d descriptors over %d frames; %.3f ms/step, %.3f ms/frame (host-fed, results to host)" %
        (res["n_keypoints"], res["n_descriptors"], F, ms_per_step, ms_per_step / F))

    out = {"metric": "Mpixels/sec detect+describe (1920x1080, 4 octaves)", "value": round(value, 2), "unit": "Mpixels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d x 1920x1080 BGRA8 frames per GPU per step (BASELINE configs[2]/[3]; %d distinct synthetic frames), %d octaves x %d scales/octave, "
                                  "detect+describe as SURVEY.md 8d words the metric: frames uploaded from page-locked host memory (H2D) and every step's packed "
                                  "keypoints + descriptors copied back to page-locked host memory (D2H) inside the timed region%s" %
                                  (F, min(F, args.distinct), N_OCT, NSPO, ", plus the RCCL all-gather of every rank's results" if world > 1 else ""),
                      "frames_per_gpu": F, "lockstep_batch": eng.max_batch, "parallelism": "frame-per-GPU x%d" % world,
                      "steps_in_flight": args.pipeline,
                      "pipelining": ("siftmi_stream_submit_host / siftmi_stream_result_host (C ABI): the upload of step k+1 runs under the kernels of step k (copy "
                                     "stream, rotating staging buffers); consecutive steps alternate between two contexts (two pyramids, two streams); every "
                                     "step's results are copied to page-locked host memory (copy started at submit time) and read %d steps late; every step is "
                                     "computed in full, the timed region ends with a device synchronisation" % back),
                      "h2d_bytes_per_step": int(hpin.nbytes), "d2h_bytes_per_step": d2h_bytes,
                      "host_numa_node": numa_node,            # this rank's threads and frame buffers sit on its GPU's NUMA node (None: topology unreadable, nothing pinned)
                      "rccl_ranks": rccl_ranks,
                      "rccl_ranks_source": "ncclCommCount of the exchange's communicator (checked against WORLD_SIZE)" if use_dist else None,
                      "ranks_share_one_gpu": bool(args.share_gpu),
                      "all_gather_transport": transport,
                      "ms_per_step_by_rank": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
                      "by_rank": by_rank,
                      "all_gather_ms_per_step": None if gather_ms is None else round(gather_ms, 4),
                      "all_gather_bytes_received_per_rank_per_step": gather_bytes,
                      "all_gather_steps_regathered": regathered, "all_gather_steps_overflowed": overflowed,
                      "all_gather_checksum": checksum,
                      "all_gather": ("siftmi_exchange_gather (librccl inside libsiftmi.so) on a side stream, rotating result sets: step k's exchange "
                                     "runs under step k+1's kernels; payload sizes from step k-1's totals; every host wait bounded "
                                     "(SIFTMI_EXCHANGE_TIMEOUT_S), expiry aborts the communicator and ends the job") if use_dist else None,
                      "host_binding": "ctypes -> siftmi_stream_* / siftmi_exchange_* (C ABI); no torch tensors or torch.distributed in the data path",
                      "keypoints_per_step_rank0": res["n_keypoints"], "descriptors_per_step_rank0": res["n_descriptors"]}}

    if rank == 0:
        # the upload alone at the synchronous H2D rate of this box, measured now: when ms_per_step is close to it the step is PCIe-bound
        t1 = time.perf_counter()
        for _ in range(3):
            _capi.check(_capi.load().siftmi_memcpy(d_frames.ptr, hpin.ctypes.data, hpin.nbytes, 0))
        h2d_gbs = 3 * hpin.nbytes / (time.perf_counter() - t1) / 1e9
        out["h2d_floor_ms"] = round(hpin.nbytes / h2d_gbs / 1e6, 4)
        out["config"]["synchronous_h2d_GBps"] = round(h2d_gbs, 1)
        out["config"]["h2d_floor_how"] = "h2d_bytes_per_step / the rate of three synchronous siftmi_memcpy calls from the same page-locked frames (this box, this run)"
        # the same K steps with the frames resident in HBM and the results left there (rounds 1-4 quoted this as `value`)
        rrun = smstream.FrameStream(eng, F, device=dev, pipeline=args.pipeline, result_sets=2 * args.pipeline)
        for _ in range(2 * rrun.n_sets + 1):
            rrun.run(d_frames)
        dev_sync()
        dtr = timed_steps(lambda: rrun.run(d_frames), dev_sync, args.steps, 2)
        rres = rrun.results_host()
        if (rres["n_keypoints"], rres["n_descriptors"]) != (first["n_keypoints"], first["n_descriptors"]):
            raise SystemExit("bench: resident-frame results differ from the host-fed ones")
        out["resident_ms_per_step"] = round(dtr / args.steps * 1e3, 4)
        out["resident_Mpixels_per_s"] = round(F * W * H * args.steps / dtr / 1e6, 1)
        out["config"]["resident"] = ("the same step through siftmi_stream_submit_device: frames already in HBM, results left in HBM, %d steps in flight, "
                                     "this rank alone (no exchange)" % args.pipeline)
        rrun.close()
        log("resident frames: %.3f ms/step (%.0f Mpixels/s); synchronous H2D %.1f GB/s -> upload floor %.3f ms/step" %
            (out["resident_ms_per_step"], out["resident_Mpixels_per_s"], h2d_gbs, out["h2d_floor_ms"]))
        if not args.no_extras and not use_dist:
            # `value` times EXACTLY K steps between device synchronisations, so it pays one pipeline fill (nothing runs under the first
            # upload) and one drain (the last step's kernels run after the last upload) per K steps: ~one step's kernels / K.  The same
            # host-fed loop over 10 K steps shows the steady rate of the stream, which is what a capture loop sees.
            ks = 10 * args.steps
            dts = timed_steps(step, barrier, ks, 0)
            out["config"]["steady_state"] = {"steps": ks, "ms_per_step": round(dts / ks * 1e3, 4), "Mpixels_per_s": round(F * W * H * ks / dts / 1e6, 1),
                                             "what": "the host-fed loop of `value` over %d steps in one timed region: the fill / drain of a region amortised "
                                                     "over 10x as many steps" % ks}
            log("host-fed, %d steps in one region: %.3f ms/step (%.0f Mpixels/s)" % (ks, dts / ks * 1e3, out["config"]["steady_state"]["Mpixels_per_s"]))
        # the same K resident steps one at a time on one context: what the pipelining buys
        dt1 = timed_steps(lambda: plain.run(d_frames), dev_sync, args.steps, 3)
        out["resident_ms_per_step_one_in_flight"] = round(dt1 / args.steps * 1e3, 4)
        log("resident, one step in flight: %.3f ms/step" % (dt1 / args.steps * 1e3))
    if rank == 0 and not args.no_roofline:
        # second, identical pass with per-launch hipEvents on the launch stream
        eng.enable_timings(True)
        eng.reset_timings()
        for _ in range(args.steps):
            plain.run(d_frames)
        dev_sync()
        tm = eng.timings()
        eng.enable_timings(False)
        blur_ms, blur_n = tm["blur"]
        # one blur launch of octave o moves 8 B x N_o x (frames in the launch); per step every octave
        # gets nspo+2 launches per sub-batch
        bytes_per_step = sum(eng.blur_algorithmic_bytes(o) for o in range(N_OCT)) * (NSPO + 2) * F
        total_bytes = bytes_per_step * args.steps
        achieved = total_bytes / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0
        # the layer-nspo launch of octave o also writes octave o + 1's first layer (the reference's nearestNeighborDownScale pass, fused):
        # 4 B per pixel of the next octave that the 8 B-per-layer-pixel figure above does not credit
        dec_bytes = sum(eng.blur_algorithmic_bytes(o) // 2 for o in range(1, N_OCT)) * F * args.steps
        achieved_incl = (total_bytes + dec_bytes) / (blur_ms * 1e-3) / 1e9 if blur_ms > 0 else 0.0
        stage_ms = {k: round(v[0] / args.steps, 4) for k, v in tm.items()}
        log("stage ms/step:", stage_ms)
        # the same time split by launch shape = (octave, layer): one kernel instantiation and grid each, comparable row by
        # row with profiles/rocprof_kernel_shapes_rNN.csv (the kernel trace of this command, tools/profile_round.sh)
        shapes = {}
        for o in range(N_OCT):
            for layer in range(1, NSPO + 3):
                ms, n, marching = eng.blur_layer_timings(o, layer)
                if n:
                    radius = len(eng.weights(layer)) // 2
                    nbytes = eng.blur_algorithmic_bytes(o) * eng.max_batch
                    kind = eng.blur_layer_kind(o, layer)
                    shapes["o%d_l%d" % (o, layer)] = {"kernel": "%s<%d, ...>" % (kind["kernel"], radius),
                                                     "decimating": layer == NSPO and o + 1 < N_OCT, "activity_flags": kind["activity_flags"],
                                                     "launches": n, "avg_launch_us": round(ms / n * 1e3, 2),
                                                     "GBps": round(nbytes / (ms / n * 1e-3) / 1e9, 1)}
        per_layer = {}
        for layer in range(1, NSPO + 3):
            ms = eng.time_blur(0, layer, 10)
            per_layer["o0_l%d_taps%d" % (layer, len(eng.weights(layer)))] = round(eng.blur_algorithmic_bytes(0) * eng.max_batch / (ms * 1e-3) / 1e9, 1)
        log("octave-0 blur GB/s by layer:", per_layer)
        # HBM bytes per launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; tools/profile_round.sh)
        traffic, traffic_src = None, None
        pdir = os.path.join(ROOT, "profiles")
        prof = sorted(f for f in os.listdir(pdir) if f.startswith("blur_hbm_traffic_")) if os.path.isdir(pdir) else []
        if prof:
            pj = json.load(open(os.path.join(pdir, prof[-1])))
            ks = [k for k in pj.get("kernels", []) if k.get("pipeline_layer")]
            if len(ks) == NSPO + 2:
                ratio = sum(k["hbm_bytes_per_launch_corrected"] for k in ks) / (pj["algorithmic_bytes_per_launch"] * len(ks))
                traffic = int(ratio * total_bytes / max(blur_n, 1))
                traffic_src = ("profiles/%s: PMC bytes / algorithmic bytes = %.3f averaged over the five octave-0 layer launches of the "
                               "pipeline, scaled to the average launch" % (prof[-1], ratio))
        # the same fraction recomputed from the committed rocprofv3 kernel trace of this command (tools/profile_round.sh, serial graph: every
        # kernel alone on the GPU, under the profiler's clocks, on whatever box that run got): a number a reader can re-derive from profiles/
        rp, rp_src = None, None
        prof_r = sorted(f for f in os.listdir(pdir) if f.startswith("roofline_rocprof_")) if os.path.isdir(pdir) else []
        if prof_r:
            rp = json.load(open(os.path.join(pdir, prof_r[-1])))
            rp_src = "profiles/%s (from %s)" % (prof_r[-1], rp.get("source"))
        # the ceilings, measured on THIS device in THIS run: a plain float4 copy of one octave-0 layer launch's bytes
        # inside the same pyramid memory, and the ring kernel itself with its arithmetic compiled out (same loads, LDS staging,
        # barriers and stores).  Both overwrite the pyramid; every later measurement recomputes it.
        launch_bytes = eng.blur_algorithmic_bytes(0) * eng.max_batch
        copy_ms, copy_moved = eng.time_copy(launch_bytes // 2, 10)
        copy_gbs = copy_moved / (copy_ms * 1e-3) / 1e9
        mem_only = {}
        for layer in range(1, NSPO + 3):
            try:
                ms = eng.time_blur_memory(0, layer, 10)
                mem_only["o0_l%d_taps%d" % (layer, len(eng.weights(layer)))] = round(launch_bytes / (ms * 1e-3) / 1e9, 1)
            except sm.SiftmiError:
                pass
        log("measured float4 copy: %.0f GB/s (%.2f GB per launch); ring kernel without arithmetic, octave 0 by layer: %s" % (copy_gbs, copy_moved / 1e9, mem_only))
        # the seed launch (luma + 2x bilinear + blur, one kernel): 4 B read per input pixel, 16 B (four octave-0 floats) written
        seed_ms, seed_n = tm["seed"]
        seed_bytes = 20 * W * H * eng.max_batch
        seed_gbs = seed_bytes / (seed_ms / max(seed_n, 1) * 1e-3) / 1e9 if seed_ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "kernel": "blur_ring_kernel<R> (large launches) / blur2_kernel<R> (small octaves): one Gaussian layer, fused X+Y",
                           "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_source": traffic_src,
                           "achieved_incl_decimated_output": round(achieved_incl, 1), "frac_incl_decimated_output": round(achieved_incl / HBM_PEAK_GBS, 4),
                           "frac_rocprof": None if rp is None else rp.get("frac_all_layers"),
                           "frac_rocprof_octave0": None if rp is None else rp.get("frac_octave0"),
                           "frac_rocprof_source": rp_src,
                           "peak_measured": round(copy_gbs, 1), "frac_of_measured": round(achieved / copy_gbs, 4) if copy_gbs > 0 else None,
                           "peak_measured_how": "siftmi_time_copy: one-pass float4 streaming copy of %.2f GB (read + written) inside this context's pyramid memory, the faster "
                                                "of plain and non-temporal loads / stores, mean of 10 launches after a warm-up, hipEvents, this device, this run" % (copy_moved / 1e9),
                           "memory_only_GBps_by_layer": mem_only,
                           "memory_only_how": "siftmi_time_blur_memory: the layer's ring kernel with both passes' arithmetic compiled out (same loads, LDS staging, barriers, stores)",
                           "seed": {"kernel": "blur_ring_kernel<5, ..., SEEDF> (luma + 2x bilinear + seed blur, one launch)", "launches": seed_n,
                                    "algorithmic_bytes_per_launch": seed_bytes, "avg_launch_us": round(seed_ms / max(seed_n, 1) * 1e3, 2),
                                    "achieved": round(seed_gbs, 1), "frac": round(seed_gbs / HBM_PEAK_GBS, 4),
                                    "frac_of_measured": round(seed_gbs / copy_gbs, 4) if copy_gbs > 0 else None},
                           "launches": blur_n, "avg_launch_ms": round(blur_ms / max(blur_n, 1), 5),
                           "algorithmic_bytes_per_launch_avg": int(total_bytes / max(blur_n, 1)),
                           "octave0_GBps_by_layer": per_layer, "by_launch_shape": shapes, "stage_ms_per_step": stage_ms,
                           "measured_in": "second identical pass of K steps on one context, one step at a time, hipEvents around every launch on the launch stream"}
    if rank == 0 and not args.no_extras:
        out["extras"] = {"match": match_extra(eng, local_rank)}
        log("matcher:", out["extras"]["match"])
        # BASELINE configs[1]: ONE 1920x1080 frame per call (lock-step batch 1, hipGraph replay), frame in HBM
        e1 = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=1)
        r1 = smstream.FrameStream(e1, 1, device=dev)
        one = smstream.DeviceFrames(frames_np[:1], local_rank)
        for _ in range(5):
            r1.run(one)
        dev_sync()
        batches = []
        n1 = 20
        for _ in range(7):                                  # one call at a time is as much a host-side latency as a GPU one: median of 7 x 20 calls
            t1 = time.perf_counter()
            for _ in range(n1):
                r1.run(one)
            dev_sync()
            batches.append((time.perf_counter() - t1) / n1 * 1e3)
        batches.sort()
        ms1 = batches[len(batches) // 2]
        out["config"]["single_frame"] = {"workload": "BASELINE configs[1]: one 1920x1080 frame per call, 4 octaves", "ms_per_frame": round(ms1, 4),
                                         "ms_per_frame_best_of_7": round(batches[0], 4), "Mpixels_per_s": round(W * H / ms1 / 1e3, 1)}
        r2 = smstream.FrameStream(e1, 1, device=dev, pipeline=2)     # two calls in flight (two contexts)
        for _ in range(8):
            r2.run(one)
        dev_sync()
        t1 = time.perf_counter()
        for _ in range(100):
            r2.run(one)
        dev_sync()
        ms2 = (time.perf_counter() - t1) / 100 * 1e3
        out["config"]["single_frame"]["ms_per_frame_two_calls_in_flight"] = round(ms2, 4)
        log("single frame: %.3f ms (%.0f Mpixels/s); two calls in flight: %.3f ms per frame" % (ms1, W * H / ms1 / 1e3, ms2))
        r1.close(); r2.close(); e1.close(); one.close()
        del r1, r2, e1
        # The SYNCHRONOUS host-buffer entry (siftmi_detect_describe_batch) on the same page-locked frames: its own context with 8-frame
        # sub-batches (the first one 2 frames); the H2D copy of sub-batch i+1 runs under the kernels of sub-batch i, the packed results of
        # finished sub-batches are copied back under the later ones.  Warm-up until every sub-batch's launch sequence is replayed from its
        # captured graph (call 1 direct launches, call 2 captures; siftmi_graph_stats), then timed.
        eio = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=8)
        for _ in range(3):
            eio.detect_describe_batch(hpin, copy=False)
        gs0 = eio.graph_stats()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            k, kc, d, dc = eio.detect_describe_batch(hpin, copy=False)
        ms_io = (time.perf_counter() - t1) / reps * 1e3
        gs1 = eio.graph_stats()
        out["config"]["host_io"] = {"workload": "the same %d-frame step through the synchronous call siftmi_detect_describe_batch: BGRA8 frames in pinned host memory "
                                                "(8-frame sub-batches, one captured launch sequence each: the H2D copy of sub-batch i+1 runs under the kernels of sub-batch i, results of finished "
                                                "sub-batches are copied back under the later ones); a synchronous call cannot hide its first upload or its last sub-batch: "
                                                "time >= max(upload + last sub-batch, first upload + all kernels)" % F,
                                    "ms_per_step": round(ms_io, 4), "Mpixels_per_s": round(F * W * H / ms_io / 1e3, 1),
                                    "h2d_bytes_per_step": int(hpin.nbytes), "d2h_bytes_per_step": int(k.nbytes + d.nbytes),
                                    "keypoints": int(len(k)), "descriptors": int(len(d)),
                                    "launch_sequences_in_the_timed_calls": {"graph_replays": gs1["replays"] - gs0["replays"], "captures": gs1["captures"] - gs0["captures"],
                                                                            "direct": gs1["direct"] - gs0["direct"]}}
        log("host i/o, synchronous call: %.3f ms (%.0f Mpixels/s) %s" % (ms_io, F * W * H / ms_io / 1e3, out["config"]["host_io"]["launch_sequences_in_the_timed_calls"]))
        del k, d
        eio.close()
        # dense natural texture: the same step on 64 mirror-tiled butterfly frames (not sparse synthetic blobs), resident and host-fed
        dense_np = make_dense_frames(F)
        d_dense = smstream.DeviceFrames(dense_np, local_rank)
        drun = smstream.FrameStream(eng, F, device=dev, pipeline=args.pipeline, result_sets=2 * args.pipeline)
        drun.run(d_dense)
        dev_sync()
        dres = drun.results_host()
        dt_d = timed_steps(lambda: drun.run(d_dense), dev_sync, args.steps, 10)  # warm-up: the density hint settles and both contexts capture this input's (one-chain) launch sequence
        ms_d = dt_d / args.steps * 1e3
        out["config"]["dense"] = {"workload": "%d x 1920x1080 mirror-tiled butterfly frames (SURVEY.md 8d dense variant), resident in HBM" % F,
                                  "ms_per_step": round(ms_d, 4), "Mpixels_per_s": round(F * W * H / ms_d / 1e3, 1),
                                  "keypoints_per_step": dres["n_keypoints"], "descriptors_per_step": dres["n_descriptors"]}
        log("dense step: %.3f ms (%.0f Mpixels/s), %d keypoints, %d descriptors" % (ms_d, F * W * H / ms_d / 1e3, dres["n_keypoints"], dres["n_descriptors"]))
        # ... and as the headline is measured: host-fed, every step's results (27 -> ~190 MB) back to host memory
        dpin = sm.pinned_empty(dense_np.shape, np.uint8)
        dpin[...] = dense_np

        def dstep():
            drun.run_host(dpin)
            drun.results_host(back=back, copy=False)

        for _ in range(2 * drun.n_sets + 2):
            dstep()
        dev_sync()
        dt_dh = timed_steps(dstep, dev_sync, args.steps, 2)
        dres_h = drun.results_host()
        if (dres_h["n_keypoints"], dres_h["n_descriptors"]) != (dres["n_keypoints"], dres["n_descriptors"]):
            raise SystemExit("bench: host-fed dense results differ from the resident ones")
        out["config"]["dense"]["host_fed_ms_per_step"] = round(dt_dh / args.steps * 1e3, 4)
        out["config"]["dense"]["host_fed_Mpixels_per_s"] = round(F * W * H * args.steps / dt_dh / 1e6, 1)
        out["config"]["dense"]["host_fed_d2h_bytes_per_step"] = int(dres_h["keypoints"].nbytes + dres_h["descriptors"].nbytes)
        log("dense step, host-fed with results to host: %.3f ms" % (dt_dh / args.steps * 1e3))
        drun.close()
        sm.pinned_release(dpin)
        if not args.no_roofline:
            eng.enable_timings(True)
            eng.reset_timings()
            for _ in range(args.steps):
                plain.run(d_dense)
            dev_sync()
            out["config"]["dense"]["stage_ms_per_step"] = {k: round(v[0] / args.steps, 4) for k, v in eng.timings().items()}
            eng.enable_timings(False)
            # the north-star's "LDS tile staging for 16x16 descriptor patches" is a configuration of the descriptor kernel, not its default:
            # the same dense step with it on, so that every round's line says what the choice costs (identical record counts checked)
            e_p = sm.Engine(W, H, device=local_rank, n_octaves=N_OCT, nspo=NSPO, max_batch=min(args.batch, F), descriptor_patch_lds=1, **tune)
            r_p = smstream.FrameStream(e_p, F, device=dev)
            for _ in range(2):
                r_p.run(d_dense)
            dev_sync()
            e_p.enable_timings(True)
            e_p.reset_timings()
            n_p = max(2, args.steps // 4)
            for _ in range(n_p):
                r_p.run(d_dense)
            dev_sync()
            pres = r_p.results_host()
            if (pres["n_keypoints"], pres["n_descriptors"]) != (dres["n_keypoints"], dres["n_descriptors"]):
                raise SystemExit("bench: descriptor_patch_lds results differ")
            out["config"]["dense"]["descriptor_patch_lds"] = {
                "what": "siftmi_config.descriptor_patch_lds = 1: 18x18-texel tiles of every descriptor window staged in LDS (byte-identical records: tests)",
                "describe_ms_per_step": round(e_p.timings()["describe"][0] / n_p, 4),
                "describe_ms_per_step_default": out["config"]["dense"]["stage_ms_per_step"]["describe"]}
            log("dense step, descriptor stage with LDS-staged patches: %.3f ms (default %.3f)" %
                (out["config"]["dense"]["descriptor_patch_lds"]["describe_ms_per_step"], out["config"]["dense"]["stage_ms_per_step"]["describe"]))
            r_p.close()
            e_p.close()
        dev_sync()
        d_dense.close()
    if rank == 0 and not args.no_cpu and world == 1:
        if prev_affinity is not None:
            os.sched_setaffinity(0, prev_affinity)           # the CPU baseline runs on every core of the host
        out["cpu_baseline"] = cpu_baseline(frames_np)
    # everything that can still write to fd 1 (RCCL prints its version banner when the communicator goes) is torn down BEFORE
    # stdout is restored: the contract is ONE JSON line
    if world > 1:
        dist.barrier()                                      # rank 0's extra measurements are done: all ranks leave together (the exchange is collective)
    plain.close()
    runner.close()
    eng.close()
    sm.pinned_release(hpin)
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)                                           # and whatever library teardown prints at exit goes to stderr


if __name__ == "__main__":
    main()
