"""The library's result exchange (siftmi_exchange_*, siftmetal_amd/csrc/exchange_api.hip.h) with MORE THAN ONE RANK.

One process per rank (tests/c/exchange_ranks.c, plain C against include/siftmi.h), all on device 0, the collectives carried by
the test transport tests/c/libfake_rccl.so (librccl's entry points over shared memory; SIFTMI_RCCL_LIB) because real RCCL
refuses two ranks on one GPU -- the same test is also parametrised on the real librccl and skips when it refuses.  What is
under test is the C exchange itself, unchanged: the totals all-gather, step-late payload sizing, the re-gather of a step that
outgrew its sizes, the rotation of the two gathered sets against the stream's result sets, event ordering between the launch
streams and the gather stream, overflow propagation.  (BASELINE configs[3] is this path on 8 GPUs.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.synth import blob_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c")
FAKE = os.path.join(CDIR, "libfake_rccl.so")
EXE = os.path.join(CDIR, "exchange_ranks")
W, H, N_OCT, F = 640, 480, 3, 2

RCCL_SYMBOLS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString",
                "ncclCommCount", "ncclCommUserRank", "ncclCommGetAsyncError", "ncclCommAbort")


def test_test_transport_exports_what_the_exchange_resolves():
    """CPU: libfake_rccl.so is built by __graft_entry__.build() and carries the eleven symbols exchange_api.hip.h looks up."""
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(FAKE) and os.path.exists(EXE)
    out = subprocess.run(["nm", "-D", "--defined-only", FAKE], stdout=subprocess.PIPE, check=True).stdout.decode()
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert set(RCCL_SYMBOLS) <= names, set(RCCL_SYMBOLS) - names
    src = open(os.path.join(ROOT, "siftmetal_amd", "csrc", "exchange_api.hip.h")).read()
    for s in RCCL_SYMBOLS:
        assert '"%s"' % s in src, s                              # the transport implements exactly the library's dlsym list


def rank_frames(rank):
    """This rank's small and large frame set (different content on every rank; the large one holds > 1.5x the keypoints)."""
    small = np.stack([blob_frame(W, H, 100 * rank + i, n_blobs=220) for i in range(F)])
    large = np.stack([blob_frame(W, H, 100 * rank + 50 + i, n_blobs=800) for i in range(F)])
    return small, large


def parse_rank_output(path, world, steps):
    from siftmetal_amd import _capi
    raw = open(path, "rb").read()
    n_counts = 2 * F * N_OCT
    pos, out = 0, []
    for _ in range(steps):
        step = int(np.frombuffer(raw, np.int64, 1, pos)[0]); pos += 8
        nk, nd, flags, nc = (int(v) for v in np.frombuffer(raw, np.int32, 4, pos)); pos += 16
        assert nc == n_counts
        own = {"counts": np.frombuffer(raw, np.int32, nc, pos).copy()}; pos += 4 * nc
        own["kp"] = raw[pos:pos + 44 * nk]; pos += 44 * nk
        own["desc"] = raw[pos:pos + 136 * nd]; pos += 136 * nd
        own.update(nk=nk, nd=nd, flags=flags)
        gw, first_look, complete, _ = (int(v) for v in np.frombuffer(raw, np.int32, 4, pos)); pos += 16
        kp_rec, desc_rec = (int(v) for v in np.frombuffer(raw, np.int64, 2, pos)); pos += 16
        assert gw == world
        rows = []
        for _r in range(world):
            tot = np.frombuffer(raw, np.int32, 4, pos).copy(); pos += 16
            cnt = np.frombuffer(raw, np.int32, nc, pos).copy(); pos += 4 * nc
            k = min(int(tot[0]), kp_rec); d = min(int(tot[1]), desc_rec)
            rows.append({"totals": tot, "counts": cnt, "kp": raw[pos:pos + 44 * k], "desc": raw[pos + 44 * k:pos + 44 * k + 136 * d]})
            pos += 44 * k + 136 * d
        out.append({"step": step, "own": own, "first_look": first_look, "complete": complete, "records": (kp_rec, desc_rec), "rows": rows})
    trailer = np.frombuffer(raw, np.int64, 5, pos); pos += 40
    assert trailer[0] == -1 and pos == len(raw)
    assert _capi.keypoint_dtype.itemsize == 44 and _capi.descriptor_dtype.itemsize == 136
    return out, {"regathered": int(trailer[1]), "overflowed": int(trailer[2]), "gathers": int(trailer[3]), "bytes_last": int(trailer[4])}


def run_ranks(tmp_path, world, scenario, pipeline, synchronous, transport, steps=8, tag=""):
    """Spawns `world` fresh processes on device 0; returns (per-rank parsed output, per-rank summary) or skips."""
    import __graft_entry__ as ge
    ge.build()
    d = tmp_path / ("run_%s_%d_%s_p%d_s%d%s" % (transport, world, scenario.replace(":", "_"), pipeline, synchronous, tag))
    d.mkdir()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FAKE_RCCL_TIMEOUT_S"] = "90"
    if transport == "fake":
        env["SIFTMI_RCCL_LIB"] = FAKE
    else:
        env.pop("SIFTMI_RCCL_LIB", None)
    procs = []
    for r in range(world):
        small, large = rank_frames(r)
        np.concatenate([small, large]).tofile(str(d / ("frames%d.bin" % r)))
        cmd = [EXE, str(W), str(H), str(N_OCT), str(F), str(r), str(world), str(steps), scenario, str(pipeline), str(synchronous),
               str(d / "unique_id"), str(d / ("frames%d.bin" % r)), str(d / ("out%d.bin" % r))]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    outs, timed_out = [], False
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            timed_out = True
            for q in procs:                                       # exactly the processes started here
                if q.poll() is None:
                    q.kill()
            o, e = p.communicate()
        outs.append((p.returncode, o.decode("utf-8", "replace"), e.decode("utf-8", "replace")))
    if transport == "rccl" and (timed_out or any(rc == 77 for rc, _, _ in outs)):
        pytest.skip("real RCCL does not run %d ranks on one device here (%s): %s" %
                    (world, "timed out" if timed_out else "ncclCommInitRank refused", outs[0][2][-300:].strip()))
    assert not timed_out, [(rc, e[-1500:]) for rc, _, e in outs]
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, (r, rc, e[-2000:])
        assert "rank %d ok %d steps" % (r, steps) in o
    parsed = [parse_rank_output(str(d / ("out%d.bin" % r)), world, steps) for r in range(world)]
    return [p[0] for p in parsed], [p[1] for p in parsed]


def check_every_view_equals_every_ranks_own_results(per_rank, world, steps):
    for q in range(world):
        assert [s["step"] for s in per_rank[q]] == list(range(steps))
        for k in range(steps):
            view = per_rank[q][k]
            assert view["complete"] == 1, (q, k)                 # read one step late: complete whatever the step's size was
            for r in range(world):
                own, row = per_rank[r][k]["own"], view["rows"][r]
                assert tuple(row["totals"][:3]) == (own["nk"], own["nd"], own["flags"]), (q, k, r)
                assert np.array_equal(row["counts"], own["counts"]), (q, k, r)
                assert row["kp"] == own["kp"] and row["desc"] == own["desc"], (q, k, r)


@pytest.mark.gpu
@pytest.mark.parametrize("world,pipeline,transport", [(2, 2, "fake"), (3, 2, "fake"), (4, 2, "fake"), (2, 1, "fake"), (2, 2, "rccl"), (3, 2, "rccl")])
def test_exchange_ranks_jump_on_one_rank_is_regathered(tmp_path, world, pipeline, transport):
    """2, 3 and 4 ranks, two steps in flight (and one: two result sets against the exchange's two gathered sets), sizes from the previous step.  Rank world-1 alone runs its large frame set at step 3
    (> 1.5x the records): every rank's first look at step 3 is cut short (complete = 0) -- all ranks see the same totals and
    take the same decision --, the next gather call repeats that step in full from its intact result set, and every rank's view
    of every step, read one step late, holds every rank's own packed results byte for byte.  regathered_steps == 1."""
    import siftmetal_amd as sm
    steps, K, R = 8, 3, world - 1
    per_rank, summary = run_ranks(tmp_path, world, "jump:%d:%d" % (K, R), pipeline, 0, transport, steps)
    check_every_view_equals_every_ranks_own_results(per_rank, world, steps)
    # the C hosts' own results are the host API's results for the same frames
    eng = sm.Engine(W, H, n_octaves=N_OCT, max_batch=F)
    for r in range(world):
        small, large = rank_frames(r)
        ks, _, ds, _ = eng.detect_describe_batch(small)
        kl, _, dl, _ = eng.detect_describe_batch(large)
        if r == R:
            assert len(kl) > 1.5 * len(ks) > 150
        for k in range(steps):
            own = per_rank[r][k]["own"]
            wk, wd = (kl, dl) if (k == K and r == R) else (ks, ds)
            assert own["kp"] == wk.tobytes() and own["desc"] == wd.tobytes() and own["flags"] == 0, (r, k)
    eng.close()
    for q in range(world):
        looks = [s["first_look"] for s in per_rank[q]]
        assert looks == [0 if k == K else 1 for k in range(steps)], (q, looks)
        assert summary[q]["regathered"] == 1 and summary[q]["overflowed"] == 0 and summary[q]["gathers"] == steps, summary[q]
        # step K+1 was sized from step K's totals (+25 %): more records per rank than the steps before the jump
        assert per_rank[q][K + 1]["records"][0] > per_rank[q][K - 1]["records"][0]
    assert len({s["bytes_last"] for s in summary}) == 1          # every rank moved the same bytes


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_exchange_ranks_overlapped_equals_synchronous(tmp_path, world):
    """pipeline = 2 with step-late sizing on the side stream against one step at a time with every gather sized from its own
    totals on the host (siftmi_exchange_gather(x, 1)): the same gathered bytes on every rank, step for step."""
    steps = 6
    a, sa = run_ranks(tmp_path, world, "jump:2:0", 2, 0, "fake", steps)
    b, sb = run_ranks(tmp_path, world, "jump:2:0", 1, 1, "fake", steps)
    check_every_view_equals_every_ranks_own_results(a, world, steps)
    check_every_view_equals_every_ranks_own_results(b, world, steps)
    for q in range(world):
        for k in range(steps):
            assert a[q][k]["own"]["kp"] == b[q][k]["own"]["kp"] and a[q][k]["own"]["desc"] == b[q][k]["own"]["desc"]
            for r in range(world):
                assert a[q][k]["rows"][r]["kp"] == b[q][k]["rows"][r]["kp"] and a[q][k]["rows"][r]["desc"] == b[q][k]["rows"][r]["desc"]
        assert sb[q]["regathered"] == 0 and all(s["first_look"] == 1 for s in b[q])      # synchronous sizing is never cut short
        assert sa[q]["regathered"] == 1


@pytest.mark.gpu
def test_exchange_ranks_overflow_on_one_rank_reaches_every_rank(tmp_path):
    """Rank 1's context has 8-entry keypoint lists: its steps come back truncated with overflow flags; every rank counts every
    step as overflowed (siftmi_exchange_finish) and still holds rank 1's (truncated) records exactly."""
    world, steps = 2, 4
    per_rank, summary = run_ranks(tmp_path, world, "overflow:1", 2, 0, "fake", steps)
    check_every_view_equals_every_ranks_own_results(per_rank, world, steps)
    for q in range(world):
        assert summary[q]["overflowed"] == steps and summary[q]["regathered"] == 0, summary[q]
        for k in range(steps):
            assert per_rank[q][k]["rows"][1]["totals"][2] & 2 and per_rank[q][k]["rows"][0]["totals"][2] == 0
    assert per_rank[1][0]["own"]["flags"] & 2 and 0 < per_rank[1][0]["own"]["nk"] <= 8 * F * N_OCT


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline,synchronous", [(2, 0), (1, 1)])
def test_exchange_rank_dies_and_the_others_abort_within_the_deadline(tmp_path, pipeline, synchronous):
    """Three ranks; rank 1 kills itself (SIGKILL) after submitting step 3, before its gather.  A collective completes only if every
    rank takes part, so the other two would wait for ever inside a host wait of the exchange; with SIFTMI_EXCHANGE_TIMEOUT_S = 3
    every such wait is a bounded poll that aborts the communicator on expiry: both survivors return SIFTMI_E_HIP naming their rank
    and step 3, in well under 10 s after the kill, and exit by themselves (nobody is restarted)."""
    import time
    import __graft_entry__ as ge
    ge.build()
    world, steps, K, R = 3, 8, 3, 1
    d = tmp_path / ("die_p%d_s%d" % (pipeline, synchronous))
    d.mkdir()
    env = dict(os.environ, SIFTMI_RCCL_LIB=FAKE, FAKE_RCCL_TIMEOUT_S="60", SIFTMI_EXCHANGE_TIMEOUT_S="3")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(world):
        small, large = rank_frames(r)
        np.concatenate([small, large]).tofile(str(d / ("frames%d.bin" % r)))
        cmd = [EXE, str(W), str(H), str(N_OCT), str(F), str(r), str(world), str(steps), "die:%d:%d" % (K, R), str(pipeline), str(synchronous),
               str(d / "unique_id"), str(d / ("frames%d.bin" % r)), str(d / ("out%d.bin" % r))]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
    procs[R].wait(timeout=120)                                    # the moment of the kill (start-up and the first steps are not counted)
    t_dead = time.monotonic()
    assert procs[R].returncode == -9, procs[R].returncode
    ends = {}
    try:
        for r in range(world):
            if r != R:
                procs[r].wait(timeout=30)
                ends[r] = time.monotonic() - t_dead
    finally:
        for q in procs:                                           # exactly the processes started here
            if q.poll() is None:
                q.kill()
    for r in range(world):
        o, e = procs[r].communicate()
        e = e.decode("utf-8", "replace")
        if r == R:
            assert "dying at step %d" % K in e
            continue
        assert procs[r].returncode == 3, (r, procs[r].returncode, e[-1500:])
        assert ends[r] < 10.0, ends
        assert "rank %d of %d" % (r, world) in e and "timed out" in e and ("step %d" % K) in e and "communicator aborted" in e, e[-1500:]
        assert "rank %d ok" % r not in o.decode()


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_through_the_test_transport():
    """bench.py --gpus 2 --share-gpu end to end on a 1-GPU box: self-launch -> gloo broadcast of rank 0's unique id -> two ranks,
    each with its own stream + exchange on device 0, the collectives through the test transport -> ONE JSON line from rank 0."""
    import json
    env = dict(os.environ, SIFTMI_RCCL_LIB=FAKE, FAKE_RCCL_TIMEOUT_S="120")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--frames", "8", "--steps", "4", "--warmup", "1",
                        "--no-cpu", "--no-extras", "--no-roofline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["rccl_ranks"] == 2 and c["ranks_share_one_gpu"] is True
    assert c["all_gather_checksum"]["equal_on_all_ranks"] and c["all_gather_checksum"]["ranks_compared"] == 2
    assert abs(d["value"] - 2 * 8 * 1920 * 1080 / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    assert c["all_gather_ms_per_step"] > 0 and c["all_gather_steps_overflowed"] == 0
    # world x (counts + keypoint bytes + descriptor bytes + totals) received per rank and step
    assert c["all_gather_bytes_received_per_rank_per_step"] > 2 * 10 ** 5
    assert 0 < c["ms_per_step_by_rank"]["min"] <= c["ms_per_step_by_rank"]["max"]
    assert "test transport" in c["all_gather_transport"]
