"""The oracle's restatement of SIFTDescriptor.match (SIFT/SIFTDescriptor.swift:298-361) against a literal
per-source scan written the way the reference writes it (small sizes: pure Python loop)."""
import numpy as np

from oracle import pyoracle


def sift_like(rng, n, spread=40.0):
    """Descriptor-shaped integer vectors: sparse-ish, 0..255."""
    f = np.abs(rng.normal(0.0, spread, (n, 128)))
    return np.clip(f, 0, 255).astype(np.int32)


def literal_scan(src, tgt, abs_thr, rel_thr):
    out = []
    for i, a in enumerate(src):
        best_i, best, second = None, None, None
        fa = a.astype(np.float32) / np.float32(255.0)
        for t, b in enumerate(tgt):
            fb = b.astype(np.float32) / np.float32(255.0)
            d = fa - fb
            acc = np.float32(0.0)
            for v in d:                                  # f32 sequential sum, as the oracle does
                acc = np.float32(acc + np.float32(v * v))
            dist = np.sqrt(acc, dtype=np.float32)
            if t == 0:                                   # bestMatch = first; second = .greatestFiniteMagnitude (:320-332)
                best_i, best, second = 0, dist, np.float32(np.finfo(np.float32).max)
            elif dist < best:                            # :333-338 -- second only moves when best moves
                second, best, best_i = best, dist, t
        if best_i is None:
            continue
        if not best < np.float32(abs_thr):               # :349-351
            continue
        with np.errstate(over="ignore"):
            limit = np.float32(second * np.float32(rel_thr))
        if not best < limit:                             # :353-355
            continue
        out.append((i, best_i, float(best)))
    return out


def test_match_follows_reference_scan_including_second_best_quirk():
    rng = np.random.default_rng(11)
    tgt = sift_like(rng, 40)
    src = np.clip(tgt[rng.permutation(40)[:25]] + rng.integers(-6, 7, (25, 128)), 0, 255).astype(np.int32)
    src = np.concatenate([src, sift_like(rng, 8)])
    for abs_thr, rel_thr in ((1.176, 0.6), (1.176, 0.95), (0.3, 0.8), (10.0, 10.0)):
        want = literal_scan(src, tgt, abs_thr, rel_thr)
        got = pyoracle.match(src, tgt, abs_thr, rel_thr)
        assert [(int(m["source"]), int(m["target"])) for m in got] == [(s, t) for s, t, _ in want]
        np.testing.assert_allclose(got["distance"], [d for _, _, d in want], rtol=2e-6)
    # the quirk itself: the true second-nearest sits AFTER the nearest, so the ratio test does not see it
    a = np.zeros((1, 128), np.int32)
    far, near, near2 = np.full(128, 100, np.int32), np.full(128, 2, np.int32), np.full(128, 3, np.int32)
    got = pyoracle.match(a, np.stack([far, near, near2]), 1.176, 0.6)
    assert len(got) == 1 and got[0]["target"] == 1            # second = dist(far): accepted
    got = pyoracle.match(a, np.stack([far, near2, near]), 1.176, 0.6)
    assert len(got) == 0                                       # second = dist(near2): ratio 2/3 > 0.6, rejected
    # first target is the best: second stays FLT_MAX and only the absolute threshold applies
    got = pyoracle.match(a, np.stack([near, near2, far]), 1.176, 0.6)
    assert len(got) == 1 and got[0]["target"] == 0


def test_match_edges():
    rng = np.random.default_rng(3)
    f = sift_like(rng, 5)
    assert len(pyoracle.match(f, np.zeros((0, 128), np.int32))) == 0        # no targets: nil for every source
    assert len(pyoracle.match(np.zeros((0, 128), np.int32), f)) == 0
    m = pyoracle.match(f, f, 1.176, 0.6)                                    # identity: distance 0, always first strict minimum
    assert list(m["source"]) == list(m["target"]) == [0, 1, 2, 3, 4] and np.all(m["distance"] == 0)
    dup = np.concatenate([f[2:3], f[2:3], f])                               # exact ties: the first index wins (strict <)
    m = pyoracle.match(f[2:3], dup, 1.176, 0.6)
    assert len(m) == 1 and m[0]["target"] == 0
