"""Randomised sweeps, shared by the -m gpu tests (a fixed-seed slice: tests/test_gpu_parity.py::test_seeded_parity_sweep,
::test_seeded_api_sweep) and the command-line tools (tools/fuzz_parity.py, tools/fuzz_api.py: any length, any seed).

parity sweep: random sizes (incl. odd ones and thin strips) / octave counts / scales per octave / image contents / pixel formats /
blur + extrema launch forms, every stage of the HIP path against the oracle (tests/parity.py::check_full_path).
api sweep: one long-lived context driven through random sequences of host batches, device-resident (hipGraph-replayed) batches,
single-frame detect + describe and matcher calls, every result compared bit for bit with a fresh lock-step-1 context's."""
import numpy as np

from tests import parity
from tests.synth import blob_frame

# blur / extrema code paths: default (tile blur or, where it applies, the multi-layer chain kernel; full scan), the chain kernel off,
# marching blur + flagged-row extrema scan, marching blur with the exact raw count
LAUNCH_MODES = [{}, {}, {"blur_chain_max_tiles": -1}, {"blur_march_min_blocks": 1}, {"blur_march_min_blocks": 1, "count_raw_extrema": 1}]
KINDS = ["blobs", "noise", "smooth", "checker", "constant", "steps", "blobs_f32", "blobs_bgra"]


def make_image(rng, w, h):
    kind = rng.choice(KINDS)
    if kind in ("blobs", "blobs_f32", "blobs_bgra"):
        img = blob_frame(w, h, int(rng.integers(0, 1000)), n_blobs=int(rng.integers(3, 200)), gray=(kind != "blobs_bgra"))
        if kind == "blobs_f32":
            img = (img.astype(np.float32) / np.float32(255)).astype(np.float32)
    elif kind == "noise":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "smooth":
        yy, xx = np.mgrid[0:h, 0:w]
        img = (127 + 100 * np.sin(xx / rng.uniform(3, 40)) * np.cos(yy / rng.uniform(3, 40))).astype(np.uint8)
    elif kind == "checker":
        q = int(rng.integers(2, 24))
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((((xx // q) + (yy // q)) & 1) * int(rng.integers(40, 255))).astype(np.uint8)
    elif kind == "constant":
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    else:
        img = np.zeros((h, w), np.uint8)
        img[:, w // 2:] = 200
        img[h // 3:, :] //= 2
    return str(kind), np.ascontiguousarray(img)


def parity_case(rng, nspo_choices=(3, 3, 3, 4, 5), large=True):
    """One random case: {"w", "h", "octaves", "nspo", "kind", "img", "mode"}.  large=False (the test suite's slice) leaves out the
    single 1-2 Mpixel frames, whose oracle run takes most of a minute each."""
    nspo = int(rng.choice(nspo_choices))
    w = int(rng.integers(24, 700))
    h = int(rng.integers(24, 500))
    if rng.random() < 0.15:
        w, h = int(rng.integers(700, 2100)), int(rng.integers(24, 160))         # wide strips
    elif rng.random() < 0.15:
        w, h = int(rng.integers(24, 160)), int(rng.integers(700, 2100))         # tall strips
    elif rng.random() < 0.12 and large:
        w, h = int(rng.integers(1100, 2000)), int(rng.integers(700, 1100))      # a single large frame: tile blur with activity flags, flagged-row scan
    max_oct = 1
    while max_oct < 7 and min(2 * w, 2 * h) >> max_oct >= 12:
        max_oct += 1
    no = int(rng.integers(1, max_oct + 1))
    kind, img = make_image(rng, w, h)
    mode = LAUNCH_MODES[int(rng.integers(0, len(LAUNCH_MODES)))]
    return {"w": w, "h": h, "octaves": no, "nspo": nspo, "kind": kind, "img": img, "mode": dict(mode)}


def parity_cases(seed, n, **kw):
    rng = np.random.default_rng(seed)
    return [parity_case(rng, **kw) for _ in range(n)]


def describe_case(c):
    return "%dx%d octaves %d nspo %d %s %s" % (c["w"], c["h"], c["octaves"], c["nspo"], c["kind"], c["mode"] or "default")


def run_parity_case(sm, c):
    """-> report of parity.check_full_path (+ "raised_capacities", "symmetric_pattern").  Raises AssertionError (a stage disagrees with the
    oracle) or whatever the library raised.
    strict_theta=False: check_full_path lets 2 % of a case's angles pass TOL_THETA (none by more than 0.05 rad); the report says how far
    the case went into that allowance ("angles_over_tol" of "angles_compared", "max_dtheta").
    Checkerboards are exactly symmetric: check_full_path(symmetric_pattern=True) keeps only the 0.05 rad limit for their angles and reports
    their orientation counts without asserting them (why: its docstring); every other stage is checked as usual."""
    kw = dict(strict_theta=False, symmetric_pattern=(c["kind"] == "checker"))
    raised = False
    try:
        r = parity.check_full_path(sm, c["img"], c["octaves"], c["nspo"], **kw, **c["mode"])
    except sm.SiftmiError as e:
        if "capacity" not in str(e):
            raise
        # dense synthetic patterns (checkerboards: 4 orientations per corner) overflow the default lists, which is a
        # reported, recoverable condition: retry with explicit capacities
        r = parity.check_full_path(sm, c["img"], c["octaves"], c["nspo"], max_extrema=1 << 18, max_keypoints=1 << 17, max_descriptors=1 << 19, **kw, **c["mode"])
        raised = True
    r = dict(r)
    r["raised_capacities"] = raised
    r["symmetric_pattern"] = kw["symmetric_pattern"]
    return r


# ------------------------------------------------------------------------------------------------------------------------------------
def api_sweep(sm, n_ops, seed, log=print):
    """-> number of failures.  n_ops operations split over three (size, octaves, lock-step) rounds."""
    from siftmetal_amd import stream as smstream
    from oracle import pyoracle
    rng = np.random.default_rng(seed)
    fails = 0
    for round_ in range(3):
        w, h = [(640, 480), (1280, 960), (1920, 1080)][round_]
        n_oct = int(rng.integers(2, 5))
        B = int(rng.choice([1, 2, 3, 4, 8]))
        pool = [blob_frame(w, h, 100 * round_ + i, n_blobs=int(rng.integers(20, 400))) for i in range(6)]
        ref_eng = sm.Engine(w, h, n_octaves=n_oct, max_batch=1)
        ref = []
        for f in pool:
            k, kc, d, dc = ref_eng.detect_describe_batch(f[None])
            ref.append((k, kc[0], d, dc[0]))
        ref_eng.close()
        eng = sm.Engine(w, h, n_octaves=n_oct, max_batch=B)
        streams = {}

        def expect(ids):
            return (np.concatenate([ref[i][0] for i in ids]), np.stack([ref[i][1] for i in ids]),
                    np.concatenate([ref[i][2] for i in ids]), np.stack([ref[i][3] for i in ids]))

        def check(tag, got, ids):
            nonlocal fails
            ek, ekc, ed, edc = expect(ids)
            ok = (np.array_equal(got[1], ekc) and np.array_equal(got[3], edc) and got[0].tobytes() == ek.tobytes()
                  and got[2].tobytes() == ed.tobytes())
            log("%s %dx%d oct %d B %d: %s frames %s" % ("ok  " if ok else "FAIL", w, h, n_oct, B, tag, ids))
            fails += 0 if ok else 1

        for op in range(n_ops // 3):
            kind = rng.choice(["host", "device", "device", "single", "match", "approx", "geometry"])
            ids = [int(i) for i in rng.integers(0, len(pool), int(rng.integers(1, 2 * B + 2)))]
            log("next: %s %s" % (kind, ids))
            if kind == "host":
                check("host batch", eng.detect_describe_batch(np.stack([pool[i] for i in ids])), ids)
            elif kind == "device":
                F = len(ids)
                if F not in streams:
                    streams[F] = smstream.FrameStream(eng, F)
                fs = streams[F]
                reps = int(rng.integers(1, 5))
                d = smstream.DeviceFrames(np.stack([pool[i] for i in ids]))
                for _ in range(reps):
                    fs.run(d)
                    if rng.random() < 0.5:
                        fs.synchronize()
                r = fs.results_host()
                fs.synchronize()
                d.close()
                check("device batch x%d" % reps, (r["keypoints"], r["counts"][0], r["descriptors"], r["counts"][1]), ids)
            elif kind == "single":
                i = ids[0]
                kps, counts = eng.detect(pool[i])
                ds, dc = eng.describe(kps, counts)
                check("detect+describe", (kps, counts[None], ds, dc[None]), [i])
            elif kind == "approx":
                i, j = ids[0], ids[-1]
                m = eng.approximate_match(ref[i][2], ref[j][2])
                want = pyoracle.approximate_match(ref[i][2]["features"].astype(np.int32), ref[j][2]["features"].astype(np.int32))
                ok = np.array_equal(m["source"], want["source"]) and np.array_equal(m["target"], want["target"]) and \
                    np.array_equal(m["distance"], want["distance"])
                log("%s approximateMatch %d vs %d: %d matches" % ("ok  " if ok else "FAIL", i, j, len(m)))
                fails += 0 if ok else 1
            elif kind == "geometry":
                i, j = ids[0], ids[-1]

                def xy(rec):            # absolute coordinates of each descriptor's keypoint (octave groups are concatenated)
                    k, kc, d, dc = rec
                    out, kp0, d0 = np.zeros((len(d), 2), np.float32), 0, 0
                    for o in range(len(kc)):
                        kk = k[kp0:kp0 + kc[o]]; dd = d[d0:d0 + dc[o]]
                        out[d0:d0 + dc[o], 0] = kk["abs_x"][dd["keypoint"]]; out[d0:d0 + dc[o], 1] = kk["abs_y"][dd["keypoint"]]
                        kp0 += kc[o]; d0 += dc[o]
                    return out
                axy, bxy = xy(ref[i]), xy(ref[j])
                score, n = eng.match_geometry(ref[i][2], axy, ref[j][2], bxy)
                mm = eng.match(ref[i][2], ref[j][2])
                want = pyoracle.compare_geometry(mm[:80], axy, bxy) if len(mm) >= 7 else 0.0
                ok = n == len(mm) and (score == want or (np.isnan(score) and np.isnan(want)) or abs(score - want) <= 1e-6 * abs(want))
                log("%s matchGeometry %d vs %d: %d matches, score %.6f" % ("ok  " if ok else "FAIL", i, j, n, score))
                fails += 0 if ok else 1
            else:
                i, j = ids[0], ids[-1]
                m = eng.match(ref[i][2], ref[j][2])
                want = pyoracle.match(ref[i][2]["features"].astype(np.int32), ref[j][2]["features"].astype(np.int32))
                ok = np.array_equal(m["source"], want["source"]) and np.array_equal(m["target"], want["target"])
                if not ok:
                    # The product forms exact integer distances, the oracle a sequential f32 sum; they can only disagree where a
                    # threshold test is decided in the last ulp (best ~ second * 0.6) or two targets are exactly equidistant.  Verify with exact arithmetic.
                    a = ref[i][2]["features"].astype(np.int64); b = ref[j][2]["features"].astype(np.int64)
                    got_s, want_s = dict(zip(m["source"], m["target"])), dict(zip(want["source"], want["target"]))
                    knife = True
                    for src_i in set(got_s) ^ set(want_s) | {k for k in set(got_s) & set(want_s) if got_s[k] != want_s[k]}:
                        d = np.sqrt(((b - a[src_i]) ** 2).sum(axis=1).astype(np.float64)) / 255.0
                        bi = int(np.argmin(d)); sec = d[:bi].min() if bi else np.inf
                        margin = abs(d[bi] - sec * 0.6) / max(d[bi], 1e-12)
                        tie = int((np.abs(d - d[bi]) <= 1e-6 * max(d[bi], 1e-12)).sum()) > 1     # equal exact distances to different targets:
                        knife = knife and (margin < 1e-5 or abs(d[bi] - 1.176) < 1e-5 or tie)    # the f32 sums order them by rounding noise
                    ok = knife
                    if ok:
                        log("knife-edge threshold decision(s) differ from the f32 oracle")
                log("%s match %d vs %d: %d matches" % ("ok  " if ok else "FAIL", i, j, len(m)))
                fails += 0 if ok else 1
        for fs in streams.values():
            fs.close()
        eng.close()
    return fails
