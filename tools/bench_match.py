"""Measurement of the matcher row (SIFTDescriptor.match -> siftmi_match_descriptors).
Descriptors are resident in HBM when the timed region starts (on_device = 1); the call still returns the
compacted matches to the host (12 B per source).  Prints one JSON line per size:
  pairs/s, the int8 MAC rate against the dense MFMA peak, and -- on rank 0 only, bounded -- the oracle's
  CPU matcher on a sample of the sources.
usage: python tools/bench_match.py [--cpu] [--approx] [n_src n_tgt]..."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import siftmetal_amd as sm
from siftmetal_amd import _capi

MFMA_I8_DENSE_PEAK_TFLOPS = 5000.0      # MI355X_MICROARCH.md: int8 dense = 2x bf16 (2.5 PFLOP/s); measured issue-loop rate 4.4


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    sizes = [(2500, 2300), (20000, 20000), (100000, 100000)]
    if len(args) >= 2:
        a = list(map(int, args))
        sizes = list(zip(a[::2], a[1::2]))
    eng = sm.Engine(64, 64, n_octaves=1)
    rng = np.random.default_rng(0)
    for ns, nt in sizes:
        tgt = np.zeros(nt, sm.descriptor_dtype)
        tgt["features"] = np.clip(np.abs(rng.normal(0, 40, (nt, 128))), 0, 255)
        src = np.zeros(ns, sm.descriptor_dtype)
        src["features"] = np.clip(tgt["features"][rng.integers(0, nt, ns)].astype(np.int32) + rng.integers(-12, 13, (ns, 128)), 0, 255)
        d_src = torch.from_numpy(src.view(np.uint8).copy()).cuda()
        d_tgt = torch.from_numpy(tgt.view(np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        out, n = C.c_void_p(), C.c_int64()

        def call():
            _capi.check(eng.L.siftmi_match_descriptors(eng.h, d_src.data_ptr(), ns, d_tgt.data_ptr(), nt, 1, 1.176, 0.6, C.byref(out), C.byref(n)))

        call()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        line = {"metric": "descriptor pairs/s (brute-force match + ratio test)", "n_source": ns, "n_target": nt, "ms_per_call": round(dt * 1e3, 4),
                "value": round(ns * nt / dt / 1e9, 2), "unit": "Gpairs/s", "dtype": "i8 (exact int32 accumulation)", "matches": int(n.value),
                "roofline": {"bound": "mfma", "achieved": round(ns * nt * 256 / dt / 1e12, 1), "peak": MFMA_I8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(ns * nt * 256 / dt / 1e12 / MFMA_I8_DENSE_PEAK_TFLOPS, 4)}}
        if "--cpu" in sys.argv:
            from oracle import pyoracle
            sample = max(1, min(ns, int(4e9 / (nt * 128))))          # a few seconds of CPU work
            t0 = time.perf_counter()
            pyoracle.match(src["features"][:sample].astype(np.int32), tgt["features"].astype(np.int32))
            ct = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(sample * nt / ct / 1e9, 3), "unit": "Gpairs/s", "cores": pyoracle.num_threads(), "kind": "port",
                                    "sample": "%d of %d sources against all targets" % (sample, ns)}
        print(json.dumps(line), flush=True)


def approx_main():
    """SIFTDescriptor.approximateMatch (ANN trie: build + query) on REAL descriptors: the trie's leaf sizes depend on how the
    16 cell means spread, and i.i.d. random features all share one key.  Source = descriptors of blob frames, target =
    descriptors of the same frames shifted by 2 px."""
    from tests.synth import blob_frame
    from siftmetal_amd import stream as smstream
    dev = torch.device("cuda", 0)
    eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=8)
    for frames in (1, 8, 40):
        fs = smstream.FrameStream(eng, frames, device=dev)
        a = np.stack([blob_frame(1920, 1080, i) for i in range(frames)])
        fs.run(torch.from_numpy(a).to(dev))
        src = fs.results_host()["descriptors"].copy()
        fs.run(torch.from_numpy(np.roll(a, (2, 2), axis=(1, 2))).to(dev))
        tgt = fs.results_host()["descriptors"].copy()
        ns, nt = len(src), len(tgt)
        d_src = torch.from_numpy(src.view(np.uint8).copy()).cuda()
        d_tgt = torch.from_numpy(tgt.view(np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        out, n = C.c_void_p(), C.c_int64()
        res = {}
        for name, fn, thr in (("approximateMatch (ANN trie: build + query)", eng.L.siftmi_approximate_match, 300.0),
                              ("match (exact, same inputs)", eng.L.siftmi_match_descriptors, 300.0 / 255.0)):
            def call():
                _capi.check(fn(eng.h, d_src.data_ptr(), ns, d_tgt.data_ptr(), nt, 1, thr, 0.6, C.byref(out), C.byref(n)))
            call()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            dt = (time.perf_counter() - t0) / reps
            res[name] = {"ms_per_call": round(dt * 1e3, 4), "matches": int(n.value)}
        line = {"metric": "approximateMatch vs match on real descriptors", "frames": frames, "n_source": ns, "n_target": nt, **res}
        if "--cpu" in sys.argv and ns * nt < 5e8:
            from oracle import pyoracle
            t0 = time.perf_counter()
            m = pyoracle.approximate_match(src["features"].astype(np.int32), tgt["features"].astype(np.int32))
            line["cpu_baseline"] = {"ms_per_call": round((time.perf_counter() - t0) * 1e3, 2), "matches": len(m), "cores": 1, "kind": "port"}
        print(json.dumps(line), flush=True)
        del fs


if __name__ == "__main__":
    if "--approx" in sys.argv:
        approx_main()
    else:
        main()
