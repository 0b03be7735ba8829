"""Randomised API-sequence sweep of any length and seed (tests/sweep.py::api_sweep: one long-lived context driven through random sequences
of host batches, device-resident (hipGraph-replayed) batches, single-frame detect + describe and matcher calls, every result compared bit
for bit with per-frame results from a fresh lock-step-1 context).  Catches state leaking between calls (stale counters, graph replays,
stream ordering).  A fixed-seed slice runs inside `pytest -m gpu` (tests/test_gpu_parity.py::test_seeded_api_sweep).
Run on the GPU box:  python tools/fuzz_api.py [n_ops] [seed] [--torch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if "--torch" in sys.argv:            # torch first: the process then runs on the HIP runtime torch bundles (ROCm 7.0 build)
    sys.argv.remove("--torch")
    import torch
    torch.cuda.init()

import siftmetal_amd as sm
from tests import sweep


def main():
    n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    fails = sweep.api_sweep(sm, n_ops, seed, log=lambda s: print(s, flush=True))
    print("%d failures, %.0f s" % (fails, time.time() - t0), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
