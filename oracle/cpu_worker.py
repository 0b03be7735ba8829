"""One worker process of bench.py's cpu_baseline (test / measurement infrastructure, like everything under oracle/): runs the CPU
restatement on its share of a frame file until a time budget is spent and prints one JSON line.
usage: SIFT_ORACLE_LIB=<build> python oracle/cpu_worker.py <frames.npy> <worker> <n_workers> <threads> <seconds> <w> <h> <octaves> <nspo>
Processes, not threads: every worker has its own address space, so the page faults of its ~650 MB of stacks do not serialise on one
process's memory-map lock (128 OpenMP-less threads inside ONE process reached 11x a single thread)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    path, wi, nw, threads, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    w, h, no, nspo = (int(v) for v in sys.argv[6:10])
    from oracle import pyoracle
    pyoracle.set_num_threads(threads)
    frames = np.load(path, mmap_mode="r")
    orc = pyoracle.Oracle(w, h, n_octaves=no, nspo=nspo)
    done = desc = 0
    t0 = time.time()
    i = wi % len(frames)
    while done == 0 or (time.time() - t0) * (done + 1) / done < budget:
        tot, _ = orc.detect_describe_counts(np.ascontiguousarray(frames[i]))
        desc += int(tot)
        done += 1
        i = (i + nw) % len(frames)
    print(json.dumps({"worker": wi, "frames": done, "descriptors": desc, "seconds": time.time() - t0}), flush=True)


if __name__ == "__main__":
    main()
