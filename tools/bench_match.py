"""Times siftmi_match_descriptors (device-resident inputs excluded: host API incl. H2D/D2H) and the bare
kernels via rocprofv3 when run under it.  Usage: python tools/bench_match.py [n_src n_tgt]..."""
import sys
import time

import numpy as np

import siftmetal_amd as sm


def main():
    sizes = [(2500, 2300), (20000, 20000), (100000, 100000)]
    if len(sys.argv) > 2:
        a = list(map(int, sys.argv[1:]))
        sizes = list(zip(a[::2], a[1::2]))
    eng = sm.Engine(64, 64, n_octaves=1)
    rng = np.random.default_rng(0)
    for ns, nt in sizes:
        tgt = np.zeros(nt, sm.descriptor_dtype)
        tgt["features"] = np.clip(np.abs(rng.normal(0, 40, (nt, 128))), 0, 255)
        src = np.zeros(ns, sm.descriptor_dtype)
        src["features"] = np.clip(tgt["features"][rng.integers(0, nt, ns)].astype(np.int32) + rng.integers(-12, 13, (ns, 128)), 0, 255)
        eng.match(src, tgt)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            m = eng.match(src, tgt)
        dt = (time.perf_counter() - t0) / reps
        print(f"{ns} x {nt}: {dt * 1e3:.3f} ms per call (host API, incl. copies), {ns * nt / dt / 1e9:.2f} Gpairs/s, {len(m)} matches", flush=True)


if __name__ == "__main__":
    main()
