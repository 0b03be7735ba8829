"""world_size-2 gloo test (CPU) of the N>1 path: frame sharding + count exchange + padded all-gather
of packed keypoint/descriptor buffers (tests/gloo_exchange.py, a test-only mirror of the protocol; the C exchange itself runs multi-rank in tests/test_exchange_ranks.py).  The per-rank buffers are produced by
the CPU oracle here (checker role) because the product path needs a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_results(rank, world, n_frames):
    """Packed siftmi-format records for this rank's frames, computed with the oracle."""
    from oracle import pyoracle
    from siftmetal_amd import _capi
    from tests import gloo_exchange as smdist
    from tests.synth import blob_frame
    mine = smdist.shard_frames(n_frames, world, rank)
    kps, descs = [], []
    counts = np.zeros((2, (n_frames + world - 1) // world, 2), np.int32)   # equal shape on every rank
    for fi, f in enumerate(mine):
        img = blob_frame(96, 80, f, n_blobs=30 + 10 * f)
        res = pyoracle.Oracle(96, 80, n_octaves=2).run(img)
        for o, r in enumerate(res):
            k = np.zeros(len(r["keypoints"]), _capi.keypoint_dtype)
            for a, b in [("octave", "octave"), ("scale", "scale"), ("sub_scale", "subScale"), ("x", "x"), ("y", "y"), ("abs_x", "absX"),
                         ("abs_y", "absY"), ("norm_x", "normX"), ("norm_y", "normY"), ("sigma", "sigma"), ("value", "value")]:
                k[a] = r["keypoints"][b]
            d = np.zeros(len(r["descriptors"]), _capi.descriptor_dtype)
            d["keypoint"] = r["orientations"]["keypoint"][r["descriptors"]["keypoint"]] if len(d) else 0
            d["theta"] = r["descriptors"]["theta"]
            d["features"] = r["descriptors"]["features"].astype(np.uint8)
            kps.append(k); descs.append(d)
            counts[0, fi, o], counts[1, fi, o] = len(k), len(d)
    return mine, np.concatenate(kps), np.concatenate(descs), counts


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from siftmetal_amd import _capi
        from tests import gloo_exchange as smdist
        mine, kp, ds, counts = _rank_results(rank, world, n_frames)
        cap_kp, cap_ds = 4096, 4096
        kp_b = torch.zeros(cap_kp * smdist.KP_BYTES, dtype=torch.uint8)
        ds_b = torch.zeros(cap_ds * smdist.DESC_BYTES, dtype=torch.uint8)
        kp_b[:kp.nbytes] = torch.from_numpy(kp.view(np.uint8).copy())
        ds_b[:ds.nbytes] = torch.from_numpy(ds.view(np.uint8).copy())
        totals = torch.tensor([len(kp), len(ds)], dtype=torch.int32)
        g = smdist.gather_results(kp_b, ds_b, torch.from_numpy(counts), totals)
        out = {"rank": rank, "totals": g["totals"].numpy().copy(), "counts": g["counts"].numpy().copy()}
        rows = []
        for r in range(world):
            nk, nd = int(g["totals"][r, 0]), int(g["totals"][r, 1])
            rows.append((g["keypoints"][r, :nk * 44].numpy().tobytes(), g["descriptors"][r, :nd * 136].numpy().tobytes()))
        out["rows"] = rows
        out["own"] = (kp.tobytes(), ds.tobytes(), mine)
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_shard_frames():
    from tests import gloo_exchange as smdist
    assert smdist.shard_frames(5, 2, 0) == [0, 2, 4] and smdist.shard_frames(5, 2, 1) == [1, 3]
    got = sorted(sum((smdist.shard_frames(512, 8, r) for r in range(8)), []))
    assert got == list(range(512)) and all(len(smdist.shard_frames(512, 8, r)) == 64 for r in range(8))
    assert smdist.shard_frames(3, 8, 5) == []


@pytest.mark.timeout(300)
def test_gather_results_world2_gloo():
    world, n_frames = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    outs.sort(key=lambda o: o["rank"])
    # every rank sees every rank's exact bytes, ragged sizes included
    assert np.array_equal(outs[0]["totals"], outs[1]["totals"])
    assert (outs[0]["totals"][0] != outs[0]["totals"][1]).any()        # ragged on purpose (3 vs 2 frames)
    for viewer in outs:
        for r in range(world):
            assert viewer["rows"][r][0] == outs[r]["own"][0] and viewer["rows"][r][1] == outs[r]["own"][1]
    assert outs[0]["own"][2] == [0, 2, 4] and outs[1]["own"][2] == [1, 3]
    tot = outs[0]["totals"]
    assert tot[:, 1].min() > 10


def _exchange_worker(rank, world, port, q):
    """ResultExchange (the protocol of siftmi_exchange_gather over gloo, sized by the library's siftmi_gather_plan_*) over four
    steps: payload gathers of step k sized from step k-1; a step whose counts outgrow that (a >25 % jump at the default headroom)
    is gathered again in full by the next call, from buffers the caller has kept intact (the stream's rotating result sets)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import gloo_exchange as smdist
        cap = 4096
        ex = smdist.ResultExchange(cap, cap, headroom=1.25, quantum=64)
        rng = np.random.default_rng(100 + rank)
        got, results = [], []
        sizes = [(100 + 10 * rank, 120 + 7 * rank), (110 + 3 * rank, 90), (900 + rank, 1000), (50, 60 + rank)]
        for step, (nk, nd) in enumerate(sizes):
            kp = torch.from_numpy(rng.integers(0, 256, cap * smdist.KP_BYTES, dtype=np.uint8))      # a fresh buffer per step = a rotating result set
            ds = torch.from_numpy(rng.integers(0, 256, cap * smdist.DESC_BYTES, dtype=np.uint8))
            counts = torch.full((2, 3, 2), step + rank, dtype=torch.int32)
            totals = torch.tensor([nk, nd, 0, 0], dtype=torch.int32)
            g = ex.gather(kp, ds, counts, totals)
            results.append(g)
            got.append({"sent_first": g["records_per_rank"], "complete_at_once": bool(g["complete"]), "totals": g["totals_device"].numpy().copy(),
                        "own_kp": kp[:nk * smdist.KP_BYTES].numpy().tobytes(), "own_desc": ds[:nd * smdist.DESC_BYTES].numpy().tobytes()})
        regathered, overflow = ex.finish()
        for step, g in enumerate(results):               # after finish() every step is complete, re-gathered ones included
            tot = g["totals_device"].numpy()
            got[step]["complete"] = bool(g["complete"])
            got[step]["sent"] = g["records_per_rank"]
            got[step]["rows_kp"] = [g["keypoints"][r, :int(tot[r, 0]) * smdist.KP_BYTES].numpy().tobytes() for r in range(world)]
            got[step]["rows_desc"] = [g["descriptors"][r, :int(tot[r, 1]) * smdist.DESC_BYTES].numpy().tobytes() for r in range(world)]
        q.put({"rank": rank, "steps": got, "regathered": regathered, "overflow": overflow,
               "plan": (int(ex.plan.steps_resolved), int(ex.plan.steps_incomplete), int(ex.plan.send_kp), int(ex.plan.send_desc))})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_result_exchange_world2_gloo_sizes_from_previous_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for o in outs:
        # step 0 sizes from its own counts (first step), step 1 from step 0, step 2 from step 1 -> too small for 900 / 1000 records:
        # gathered again in full by the call for step 3; step 3 is sized from step 2
        assert o["regathered"] == [2] and o["overflow"] == []
        assert o["plan"][0] == 4 and o["plan"][1] == 1
        assert o["steps"][0]["sent_first"] == (110, 127)                 # exact: the maxima over the two ranks
        assert o["steps"][1]["sent_first"] == (192, 192)                 # 1.25 x 110 + 1 -> 138 -> next multiple of 64; 1.25 x 127 + 1 = 159 -> 192
        assert o["steps"][2]["sent_first"] == (192, 128) and o["steps"][2]["sent"] == (901, 1000)
        assert o["steps"][3]["sent_first"] == (1152, 1280)
        assert all(st["complete"] for st in o["steps"])
        assert np.array_equal(o["steps"][1]["totals"][:, 0], [110, 113])
    for step in range(4):                      # every rank holds every rank's exact keypoint and descriptor bytes, the re-gathered step included
        for viewer in outs:
            for r in range(world):
                assert viewer["steps"][step]["rows_kp"][r] == outs[r]["steps"][step]["own_kp"], (step, r)
                assert viewer["steps"][step]["rows_desc"][r] == outs[r]["steps"][step]["own_desc"], (step, r)


def test_gather_plan_rule_matches_its_statement():
    """siftmi_gather_plan_* (the sizing rule of siftmi_exchange_gather): next size = (1 + headroom) x the largest count of the
    resolved step + 1, rounded up to the quantum, clamped to the capacity; a step is incomplete when any rank held more than
    was sent; overflow flags are counted."""
    import ctypes as C
    from siftmetal_amd import _capi
    L = _capi.load()
    p = _capi.GatherPlan()
    assert L.siftmi_gather_plan_init(C.byref(p), 5000, 300) == 0
    assert (p.send_kp, p.send_desc, p.quantum, p.headroom_percent) == (-1, -1, 1024, 25)
    t = np.array([[1000, 100, 0, 0], [2000, 250, 0, 0], [5, 5, 4, 0]], np.int32)
    assert L.siftmi_gather_plan_resolve(C.byref(p), t.ctypes.data, 3, 2000, 250) == 0
    assert (p.send_kp, p.send_desc) == (3072, 300)                       # 2501 -> 3072; 313 -> clamped to the capacity 300
    assert (p.steps_resolved, p.steps_incomplete, p.steps_overflowed) == (1, 0, 1)
    assert L.siftmi_gather_plan_resolve(C.byref(p), t.ctypes.data, 3, 1999, 250) == 1
    assert L.siftmi_gather_plan_resolve(C.byref(p), t.ctypes.data, 3, 2000, 249) == 1
    assert p.steps_incomplete == 2
    assert L.siftmi_gather_plan_resolve(None, t.ctypes.data, 3, 0, 0) == _capi.E_BADARG


@pytest.mark.timeout(300)
def test_bench_control_plane_under_torchrun(tmp_path):
    """bench.py --gpus N uses torch.distributed only as its control plane (gloo on the loopback interface: rank 0's ncclUniqueId
    to the other ranks, barrier, per-rank times); the data path's RCCL communicator lives in libsiftmi.so.  The same calls under
    the launcher the driver uses, two ranks on CPU."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, torch, torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
        r, n = dist.get_rank(), dist.get_world_size()
        box = [bytes(range(128)) if r == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == bytes(range(128)) and n == int(os.environ["WORLD_SIZE"])
        t = torch.tensor([1.0 + r], dtype=torch.float64)
        g = [torch.zeros(1, dtype=torch.float64) for _ in range(n)]
        dist.all_gather(g, t)
        assert [float(x) for x in g] == [1.0 + i for i in range(n)]
        dist.barrier()
        dist.destroy_process_group()
        print("rank %d of %d ok" % (r, n), flush=True)
    """)
    script = tmp_path / "control_plane.py"
    script.write_text(code)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    out = p.stdout.decode()
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert "rank 0 of 2 ok" in out and "rank 1 of 2 ok" in out
