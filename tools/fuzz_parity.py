"""Randomised parity sweep of any length and seed (tests/sweep.py: random sizes / octave counts / scales per octave / image contents /
pixel formats / launch forms, every stage of the HIP path against the oracle).  A fixed-seed slice of it runs inside `pytest -m gpu`
(tests/test_gpu_parity.py::test_seeded_parity_sweep); this tool is for longer runs on the GPU box:
    python tools/fuzz_parity.py [n_cases] [seed] [nspo=3..7]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import siftmetal_amd as sm
from tests import sweep


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kw = {"nspo_choices": (3, 3, 4, 5, 6, 7)} if (len(sys.argv) > 3 and sys.argv[3].startswith("nspo")) else {}
    rng = np.random.default_rng(seed)
    fails = ties = 0
    worst_l2 = worst_theta = 0.0
    t0 = time.time()
    for case in range(n):
        c = sweep.parity_case(rng, **kw)
        tag = "case %d: %s" % (case, sweep.describe_case(c))
        try:
            r = sweep.run_parity_case(sm, c)
            ties += 1 if (r["symmetric_pattern"] and r["max_dtheta"] > 2e-3) else 0
            worst_l2 = max(worst_l2, r["max_l2_float"])
            if not r["symmetric_pattern"]:
                worst_theta = max(worst_theta, r["max_dtheta"])
            print("ok   %s%s -> %d keypoints, %d of %d angles over 2e-3 rad (max %.2e)%s, descriptor L2 %.2e, %d bins differ" %
                  (tag, " (raised capacities)" if r["raised_capacities"] else "", r["keypoints"], r["angles_over_tol"], r["angles_compared"], r["max_dtheta"],
                   (" [symmetric pattern; %d orientation counts differ]" % r["orientation_count_mismatch"]) if r["symmetric_pattern"] else "", r["max_l2_float"],
                   r["bins_differing"]), flush=True)
        except AssertionError:
            fails += 1
            print("FAIL %s" % tag, flush=True)
            traceback.print_exc(limit=3)
        except Exception as e:
            fails += 1
            print("ERR  %s: %r" % (tag, e), flush=True)
    print("%d cases, %d failures, %d symmetric patterns with angles past 2e-3 rad, worst descriptor L2 %.2e, worst dtheta elsewhere %.2e, %.0f s" %
          (n, fails, ties, worst_l2, worst_theta, time.time() - t0), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
