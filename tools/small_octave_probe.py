"""Times the Gaussian-layer launches of the small octaves of a 64-frame 1080p step (octaves 2 and 3: 960x540 and 480x270 per frame) under
the chunk-height / marching-threshold knobs of an experiment build (tools/build_variant.sh exp -DSIFTMI_EXPERIMENT).
usage: SIFTMI_LIB=tools/tmp_variants/libsiftmi_exp.so python tools/small_octave_probe.py [iters]
One child process per setting (the knobs are read from the environment by the library)."""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import numpy as np
    import siftmetal_amd as sm
    from tests.synth import blob_frame
    kw = json.loads(os.environ.get("SIFTMI_ENGINE_KW", "{}"))
    iters = int(sys.argv[2])
    eng = sm.Engine(1920, 1080, n_octaves=4, max_batch=64, **kw)
    eng.detect_describe_batch(np.stack([blob_frame(1920, 1080, 0)] * 64))
    out = {}
    for o in (2, 3):
        for l in range(1, 6):
            eng.time_blur(o, l, 3)
            out["%d.%d" % (o, l)] = round(min(eng.time_blur(o, l, iters) for _ in range(3)) * 1e3, 1)
    print(json.dumps(out))
    sys.exit(0)

iters = sys.argv[1] if len(sys.argv) > 1 else "50"
settings = [({}, {})]
for chunk in (64, 96, 128, 192, 288, 544):
    settings.append(({"SIFTMI_EXP_CHUNK_SMALL": str(chunk)}, {}))
    settings.append(({"SIFTMI_EXP_CHUNK_SMALL": str(chunk)}, {"blur_march_min_blocks": 200}))
settings.append(({}, {"blur_march_min_blocks": 200}))
settings.append(({}, {"blur_march_min_blocks": 100000}))
for env, kw in settings:
    e = dict(os.environ, SIFTMI_ENGINE_KW=json.dumps(kw), **env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", iters], env=e, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        print(env, kw, "FAILED", r.stderr[-300:]); continue
    t = json.loads(r.stdout.strip().splitlines()[-1])
    o2 = sum(v for k, v in t.items() if k[0] == "2"); o3 = sum(v for k, v in t.items() if k[0] == "3")
    print("%-36s %-32s octave 2 %6.1f us  octave 3 %6.1f us  sum %6.1f   %s" % (env, kw, o2, o3, o2 + o3, t), flush=True)
