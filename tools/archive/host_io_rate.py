"""PCIe-inclusive rate of the host-facing batch API (frames in host memory, results copied back).
Never the bench `value`; quoted in DESIGN.md section 6."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from tests.synth import blob_frame

F = 64
base = [blob_frame(1920, 1080, i) for i in range(8)]
bgra = np.stack([base[i % 8] for i in range(F)])
gray = np.ascontiguousarray(bgra[..., 0])
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=MB)
print('lock-step sub-batch', MB)
pin_bgra = sm.pinned_empty(bgra.shape, np.uint8)
pin_bgra[...] = bgra
pin_gray = sm.pinned_empty(gray.shape, np.uint8)
pin_gray[...] = gray
for name, frames in (("BGRA8 pageable", bgra), ("GRAY8 pageable", gray), ("BGRA8 pinned", pin_bgra), ("GRAY8 pinned", pin_gray)):
    eng.detect_describe_batch(frames)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        k, kc, d, dc = eng.detect_describe_batch(frames, copy=False)
    dt = (time.perf_counter() - t0) / reps
    print("%s host->results: %.2f ms per %d frames, %.0f Mpixels/s (%d keypoints, %d descriptors)" %
          (name, dt * 1e3, F, F * 1920 * 1080 / dt / 1e6, len(k), len(d)))
