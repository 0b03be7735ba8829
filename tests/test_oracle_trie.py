"""The oracle's restatement of the reference's ANN trie (Utilities/Trie.swift) and SIFTDescriptor.approximateMatch
(SIFT/SIFTDescriptor.swift:362-417): known answers from the reference's own TrieTests.swift, and a literal Python
transcription of the Swift classes as an independent cross-check."""
import numpy as np

from oracle import pyoracle


def sift_like(rng, n, spread=40.0):
    return np.clip(np.abs(rng.normal(0.0, spread, (n, 128))), 0, 255).astype(np.int32)


def test_known_answers_from_reference_trie_tests():
    # TrieTests.testContains_* (Tests/SIFTMetalTests/TrieTests.swift:36-64), numberOfBins 3
    t = pyoracle.Trie(3)
    t.insert([1.0, 1.0, 1.0], 0)
    assert not t.contains([0.0, 0.0, 0.0])
    t = pyoracle.Trie(3)
    t.insert([0.1, 0.2, 0.3], 0)
    assert t.contains([0.1, 0.2, 0.3])
    assert t.contains([0.1, 0.2])                                   # partial match
    t = pyoracle.Trie(3)
    t.insert([0, 0.5, 1.0], 0)                                      # bins 0, 1, 2
    assert t.contains([0.1, 0.5, 1.0]) and t.contains([0.1, 0.6, 1.0]) and t.contains([0.1, 0.6, 0.9])
    # testInsert_shouldMatchStructure (:14-34): key [0, 0.5, 1] occupies exactly the path 0 -> 1 -> 2
    assert t.capacity() == 1 and t.link() == 1
    for key, present in (([0.0], True), ([0.5], False), ([1.0], False), ([0.0, 0.5], True), ([0.0, 0.0], False), ([0.0, 0.5, 1.0], True),
                         ([0.0, 0.5, 0.5], False)):
        assert t.contains(key) == present
    # testNearest_shouldReturnNearestValue_whenTrieContainsSimilarValues (:66-72): radius 0, k 1
    f = np.zeros((1, 128), np.int32)
    t = pyoracle.Trie(3, f)
    t.insert([0, 0.5, 1.0], 0)
    t.link()
    assert [i for i, _ in t.nearest([0.1, 0.6, 0.9], f[0], 0, 1)] == [0]


# ---- literal transcription of Trie.swift / approximateMatch, for cross-checking the C oracle on small inputs

class PyTrie:
    def __init__(self, nb):
        self.nb, self.nodes, self.hasNodes, self.values, self.left, self.right = nb, [None] * nb, False, [], None, None

    def binIndex(self, v):
        x = np.float32(v) * np.float32(self.nb - 1)
        return int(np.floor(x + np.float32(0.5))) if x >= 0 else int(np.ceil(x - np.float32(0.5)))     # .rounded()

    def wrap(self, i):
        n = self.nb - 1
        return i + n if i < 0 else (i - n if i >= n else i)

    def insert(self, key, value):
        if len(key) == 0:
            self.values.append(value)
            return
        b = self.binIndex(key[0])
        if self.nodes[b] is None:
            self.nodes[b] = PyTrie(self.nb)
            self.hasNodes = True
        self.nodes[b].insert(key[1:], value)

    def leaves(self):
        if not self.hasNodes:
            return [self]
        return [l for n in self.nodes if n is not None for l in n.leaves()]

    def link(self):
        ls = self.leaves()
        for i, n in enumerate(ls):
            nxt = ls[(i + 1) % len(ls)]
            n.right, nxt.left = nxt, n

    def closestNode(self, b):
        if self.nodes[b] is not None:
            return self.nodes[b]
        best, node = None, None
        for j in range(self.nb):
            if self.nodes[j] is None:
                continue
            d = self.wrap(abs(j - b))
            if best is None or d < best:
                best, node = d, self.nodes[j]
        return node

    def nearestNode(self, key):
        cur = self
        for v in key:
            if not cur.hasNodes:
                return cur
            n = cur.closestNode(cur.binIndex(v))
            if n is None:
                return cur
            cur = n
        return cur

    def nearestValue(self, query, feats, q, cap):
        best = q[0][1] if q else np.float32(np.finfo(np.float32).max)
        for v in self.values:
            d = np.sqrt(np.float32(int(np.sum((feats[v].astype(np.int64) - query.astype(np.int64)) ** 2))), dtype=np.float32)
            if d < best:
                best = d
                q.insert(0, (v, d))
                if len(q) > cap:
                    q.pop()

    def nearest(self, key, query, feats, radius, k):
        q = []
        b = self.nearestNode(key)
        b.nearestValue(query, feats, q, k)
        n = b
        for _ in range(radius):
            n = n.left
            n.nearestValue(query, feats, q, k)
        n = b
        for _ in range(radius):
            n = n.right
            n.nearestValue(query, feats, q, k)
        return q


def literal_approximate_match(src, tgt, abs_thr, rel_thr):
    _, _, tkey = pyoracle.descriptor_index(tgt)
    _, _, skey = pyoracle.descriptor_index(src)
    trie = PyTrie(8)
    for i in range(len(tgt)):
        trie.insert(list(tkey[i]), i)
    trie.link()
    out = []
    for s in range(len(src)):
        q = trie.nearest(list(skey[s]), src[s], tgt, 10, 2)
        if len(q) != 2:
            continue
        if not q[0][1] < np.float32(abs_thr):
            continue
        if not q[0][1] < np.float32(q[1][1] * np.float32(rel_thr)):
            continue
        out.append((s, q[0][0], float(q[0][1])))
    return out


def test_approximate_match_oracle_vs_literal_python():
    rng = np.random.default_rng(8)
    for n_tgt, n_src in ((400, 120), (15, 30), (3, 10), (1, 4)):
        tgt = sift_like(rng, n_tgt)
        src = np.clip(tgt[rng.integers(0, n_tgt, n_src)] + rng.integers(-4, 5, (n_src, 128)), 0, 255).astype(np.int32)
        src[::4] = sift_like(rng, len(src[::4]))
        for abs_thr, rel_thr in ((300.0, 0.6), (1e9, 2.0)):
            want = literal_approximate_match(src, tgt, abs_thr, rel_thr)
            got = pyoracle.approximate_match(src, tgt, abs_thr, rel_thr)
            assert [(int(m["source"]), int(m["target"])) for m in got] == [(s, t) for s, t, _ in want]
            np.testing.assert_array_equal(got["distance"], np.array([d for _, _, d in want], np.float32))
    assert len(pyoracle.approximate_match(src, src[:0])) == 0
    # the wrap quirk of binDifference: |j - bin| = 7 counts as 0, so bin 0 prefers child 7 over child 1
    f = np.zeros((3, 128), np.int32)
    f[1, 40:48] = 255 // 7 + 1          # first key component (cell 5) -> bin 1
    f[2, 40:48] = 255                   # -> bin 7
    t = pyoracle.Trie(8, f)
    _, _, key = pyoracle.descriptor_index(f)
    t.insert(key[1], 1)
    t.insert(key[2], 2)
    t.link()
    assert [i for i, _ in t.nearest(key[0], f[0], 0, 1)] == [2]
