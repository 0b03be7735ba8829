"""How much of a benchmark frame's DoG stack the flagged extrema scan has to read, and how much a per-scale choice of layers would
save (a design estimate on the CPU oracle's DoG, not a measurement of the kernel).  A (wavefront, row) is read when one of the
two 64-column cells under the wavefront's 62 columns has |DoG_s| > 0.8 * dog_threshold at some scale s in that row or a row next to
it; scale s needs Gaussian layers s-1 .. s+2.  usage: python tools/extrema_flag_stats.py [frame index] [octave]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from tests.synth import blob_frame

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
o = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1920, 1080
orc = pyoracle.Oracle(W, H, n_octaves=4)
orc.build_pyramid(blob_frame(W, H, idx))
w, h = orc.octave_size(o)
pre = 0.8 * 0.0133
ncell = (w + 63) // 64
act = np.zeros((3, h, ncell), bool)                     # [scale - 1][row][cell]
for s in (1, 2, 3):
    a = np.abs(orc.dog(o, s)) > pre
    pad = np.zeros((h, ncell * 64), bool); pad[:, :w] = a
    act[s - 1] = pad.reshape(h, ncell, 64).any(axis=2)
n_wave = (w - 2 + 61) // 62
rows_any = rows_layers = 0
hist = {}
for k in range(n_wave):
    xw = k * 62
    c0, c1 = min((xw + 1) >> 6, ncell - 1), min((xw + 62) >> 6, ncell - 1)
    centre = act[:, :, c0] | act[:, :, c1]              # [3][h]
    need = centre.copy(); need[:, 1:] |= centre[:, :-1]; need[:, :-1] |= centre[:, 1:]
    anyneed = need.any(axis=0)
    rows_any += int(anyneed.sum())
    layers = np.zeros((6, h), bool)
    for s in range(3):
        layers[s:s + 4] |= need[s][None, :]
    rows_layers += int(layers.sum())
    key = need[0].astype(int) + 2 * need[1] + 4 * need[2]
    for v, c in zip(*np.unique(key[anyneed], return_counts=True)):
        hist[int(v)] = hist.get(int(v), 0) + int(c)
tot = n_wave * h
print("octave %d (%d x %d), frame %d: %.1f %% of (wavefront, row) segments read; layers per read segment if chosen per scale: %.2f of 6 (-%.1f %% bytes)"
      % (o, w, h, idx, 100.0 * rows_any / tot, rows_layers / max(rows_any, 1), 100.0 * (1 - rows_layers / (6.0 * max(rows_any, 1)))))
print("needed-scale sets among read segments (bit s-1 = scale s):", {k: round(100.0 * v / rows_any, 1) for k, v in sorted(hist.items())})
