"""ms per 1920x1080 frame against frames per step (lock-step batch = frames per step, device-resident, graph replay), one step
at a time and two in flight: where the small-launch paths (forked octave chains, chain kernel, tile-kernel flags) hand over to
the batch paths.  Run on the GPU box:  python tools/batch_size_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import siftmetal_amd as sm
from siftmetal_amd import _capi, stream as smstream
from tests.synth import blob_frame
L = _capi.load()
frames = np.stack([blob_frame(1920, 1080, i) for i in range(8)])
for F in [int(x) for x in os.environ.get("SWEEP_F", "1,2,3,4,6,8,16,32,64").split(",")]:
    eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=F, **({"blur_march_min_blocks": int(os.environ["MARCH_MIN"])} if os.environ.get("MARCH_MIN") else {}),
                    **({"blur_chain_max_tiles": int(os.environ["CHAIN_MAX"])} if os.environ.get("CHAIN_MAX") else {}))
    d = smstream.DeviceFrames(np.concatenate([frames] * ((F + 7) // 8))[:F])
    out = []
    for pipe in (1, 2):
        fs = smstream.FrameStream(eng, F, pipeline=pipe, result_sets=2 * pipe)
        for _ in range(8):
            fs.run(d)
        fs.synchronize()
        n = max(20, 400 // F)
        res = []
        for rep in range(5):
            t = time.perf_counter()
            for _ in range(n):
                fs.run(d)
                if pipe == 1:
                    fs.synchronize()
            fs.synchronize()
            res.append((time.perf_counter() - t) / n * 1e3)
        res.sort()
        out.append(res[2])
        fs.close()
    print("%2d frames per step: %.3f ms per step = %.3f ms per frame; two in flight %.3f ms per frame" % (F, out[0], out[0] / F, out[1] / F), flush=True)
    d.close(); eng.close()
