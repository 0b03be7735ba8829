"""How many cores a job really gets on this host: cgroup CPU quota, scheduler affinity, and the aggregate rate of N busy-loop processes
against one (bench.py's cpu_baseline reaches ~11 effective cores with 128 single-threaded workers: a quota, or the restatement's memory traffic?)."""
import multiprocessing as mp
import os
import time


def burn(_):
    t = time.time(); x = 0
    while time.time() - t < 3:
        x += 1
    return x


if __name__ == "__main__":
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(p, open(p).read().strip())
        except OSError:
            pass
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
    one = None
    for n in (1, 16, 64, 128):
        with mp.Pool(n) as pool:
            r = pool.map(burn, range(n))
        one = one or r[0]
        print("%d busy-loop processes: %.1f x one process" % (n, sum(r) / one), flush=True)
