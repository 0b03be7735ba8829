"""Back-to-back uploads of a 64-frame step's bytes (530 MB of page-locked memory) on one stream, nothing else running: the time per
copy in a queue of 1, 2, 4, 8, i.e. what the copy engine itself leaves between two large copies -- the floor under the host-fed
stream's `ms_per_step - h2d_floor_ms`.  Also the same bytes as 2 / 4 / 8 smaller copies.
usage: python tools/h2d_gap_probe.py"""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
N = 64 * 1920 * 1080 * 4
host = C.c_void_p(); dev = C.c_void_p(); st = C.c_void_p()
assert hip.hipHostMalloc(C.byref(host), C.c_size_t(N), 0) == 0
assert hip.hipMalloc(C.byref(dev), C.c_size_t(N)) == 0
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
C.memset(host, 1, N)
H2D = 1


evs = [C.c_void_p() for _ in range(16)]
for e in evs:
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0          # hipEventDisableTiming
st2 = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st2), 1) == 0


def run(copies, pieces, record=False, waiter=False):
    hip.hipStreamSynchronize(st)
    t0 = time.perf_counter()
    for k in range(copies):
        for p in range(pieces):
            off = p * (N // pieces)
            assert hip.hipMemcpyAsync(C.c_void_p(dev.value + off), C.c_void_p(host.value + off), C.c_size_t(N // pieces), H2D, st) == 0
        if record:                                                   # what siftmi_stream_submit_host does after a step's upload
            assert hip.hipEventRecord(evs[k % 16], st) == 0
            if waiter:                                               # ... and another stream waits for it (the step's launch sequence)
                assert hip.hipStreamWaitEvent(st2, evs[k % 16], 0) == 0
    hip.hipStreamSynchronize(st)
    hip.hipStreamSynchronize(st2)
    return (time.perf_counter() - t0) / copies * 1e3


run(2, 1)
for pieces in (1, 2, 4, 8):
    print("%d piece(s) per step: " % pieces + ", ".join("%d queued %.3f ms" % (q, min(run(q, pieces) for _ in range(3))) for q in (1, 2, 4, 8)), flush=True)
for rec, wt in ((True, False), (True, True)):
    print("one copy per step, an event recorded behind every copy%s: " % (", another stream waiting on it" if wt else "") +
          ", ".join("%d queued %.3f ms" % (q, min(run(q, 1, rec, wt) for _ in range(3))) for q in (1, 2, 4, 8)), flush=True)
