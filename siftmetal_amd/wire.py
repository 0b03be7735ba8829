"""Text wire format of keypoints / descriptors: one record per line, space separated,

    y x sigma theta f[0] ... f[127] [h[0] ... h[35]]

which is what the reference's test loaders read (Tests/SIFTMetalTests/DescriptorTests.swift:176-216
`loadDescriptors`, KeypointTests.swift:90-116 `loadKeypoints`) and what IPOL's `sift_cli` prints (floats as
"%f", features as integers, an optional 36-bin orientation histogram).  Parsed records are built exactly as
those loaders build them: octave 0, scale 0, subScale 0, scaledCoordinate zero, absoluteCoordinate (x, y),
sigma, value 0.
"""
from typing import Iterable, List, Optional, Sequence

from . import SIFTDescriptor, SIFTKeypoint


def _keypoint(y: float, x: float, sigma: float) -> SIFTKeypoint:
    return SIFTKeypoint(octave=0, scale=0, subScale=0.0, scaledCoordinate=(0, 0), absoluteCoordinate=(x, y),
                        normalizedCoordinate=(0.0, 0.0), sigma=sigma, value=0.0)


def parseKeypoints(text: str) -> List[SIFTKeypoint]:
    """KeypointTests.loadKeypoints: the first three columns are y, x, sigma; the rest of the line is ignored."""
    out = []
    for line in text.split("\n"):
        c = line.split()
        if not c:
            continue
        out.append(_keypoint(float(c[0]), float(c[1]), float(c[2])))
    return out


def parseDescriptors(text: str) -> List[SIFTDescriptor]:
    """DescriptorTests.loadDescriptors: y x sigma theta + 128 integer features (columns after 132 are ignored)."""
    out = []
    for line in text.split("\n"):
        c = line.split()
        if not c:
            continue
        if len(c) < 4 + 128:
            raise ValueError("descriptor line has %d columns, need at least 132" % len(c))
        features = [int(v) for v in c[4:4 + 128]]
        out.append(SIFTDescriptor(keypoint=_keypoint(float(c[0]), float(c[1]), float(c[2])), theta=float(c[3]), features=features))
    return out


def parseOrientationHistograms(text: str) -> List[Optional[List[float]]]:
    """The optional trailing 36 floats of each descriptor line (None where a line has none)."""
    out = []
    for line in text.split("\n"):
        c = line.split()
        if not c:
            continue
        out.append([float(v) for v in c[132:168]] if len(c) >= 168 else None)
    return out


def formatKeypoints(keypoints: Iterable[SIFTKeypoint]) -> str:
    return "".join("%f %f %f \n" % (k.absoluteCoordinate[1], k.absoluteCoordinate[0], k.sigma) for k in keypoints)


def formatDescriptors(descriptors: Iterable[SIFTDescriptor], orientationHistograms: Optional[Sequence[Sequence[float]]] = None) -> str:
    lines = []
    for i, d in enumerate(descriptors):
        k = d.keypoint
        parts = ["%f %f %f %f" % (k.absoluteCoordinate[1], k.absoluteCoordinate[0], k.sigma, d.theta)]
        parts.append(" ".join(str(int(f)) for f in d.features))
        if orientationHistograms is not None:
            parts.append(" ".join("%f" % h for h in orientationHistograms[i]))
        lines.append(" ".join(parts) + " \n")
    return "".join(lines)
