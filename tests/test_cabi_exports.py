"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/siftmi.h declares; record layouts match the reference's C structs; no compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from siftmetal_amd import _capi
    return _capi.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "siftmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(siftmi_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    from siftmetal_amd import _capi
    assert declared == set(_capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_record_layouts_match_reference_structs():
    from siftmetal_amd import _capi
    # Sources/MetalShaders/include: SIFTExtremaResult 12 B, SIFTOrientationResult 152 B,
    # SIFTDescriptorResult 524 B; SIFTKeypoint flattened = 44 B
    assert _capi.extremum_dtype.itemsize == 12
    assert _capi.keypoint_dtype.itemsize == 44
    assert _capi.orientation_dtype.itemsize == 152
    assert _capi.descriptor_reference_dtype.itemsize == 524
    assert _capi.descriptor_dtype.itemsize == 136
    assert C.sizeof(_capi.Config) == 4 * 29
    # stream / exchange records: the sizes stream_api.hip.h static_asserts for the C structs
    assert (C.sizeof(_capi.StreamConfig), C.sizeof(_capi.StepDevice), C.sizeof(_capi.StepHost), C.sizeof(_capi.Gathered),
            C.sizeof(_capi.GatherPlan)) == (64, 56, 48, 96, 72)
    assert _capi.NO_STREAM == 2 ** 64 - 1


def test_defaults_are_the_reference_literals(lib):
    from siftmetal_amd import _capi
    cfg = _capi.default_config(512, 340)
    assert (cfg.n_octaves, cfg.nspo, cfg.max_iterations, cfg.image_border) == (7, 3, 5, 5)
    assert abs(cfg.dog_threshold - 0.0133) < 1e-9 and cfg.edge_threshold == 10.0 and abs(cfg.max_offset - 0.6) < 1e-7
    assert cfg.lambda_orientation == 1.5 and abs(cfg.orientation_threshold - 0.8) < 1e-7 and cfg.orientation_smoothing == 6
    assert abs(cfg.sigma_min - 0.8) < 1e-7 and cfg.delta_min == 0.5 and cfg.sigma_in == 0.5
    assert cfg.descriptor_scales_per_octave == 3 and cfg.full_neighbourhood == 0 and cfg.max_batch == 1 and cfg.use_hip_graph == 1


def test_descriptor_record_conversion(lib):
    from siftmetal_amd import _capi
    rng = np.random.default_rng(0)
    d = np.zeros(5, _capi.descriptor_dtype)
    d["keypoint"] = np.arange(5); d["theta"] = rng.uniform(0, 6, 5)
    d["features"] = rng.integers(0, 256, (5, 128))
    out = np.zeros(5, _capi.descriptor_reference_dtype)
    lib.siftmi_descriptor_to_reference(d.ctypes.data, 5, out.ctypes.data)
    assert (out["valid"] == 1).all() and np.array_equal(out["keypoint"], d["keypoint"])
    assert np.array_equal(out["features"], d["features"].astype(np.int32)) and np.array_equal(out["theta"], d["theta"])


def test_match_plan_covers_every_target_once(lib):
    """siftmi_match_plan (host arithmetic only): chunks are whole staging quanta, cover the target list, and start from a bound only
    when they are long (the size class tests/test_gpu_parity.py::test_match_large_bounded_chunks_* exercises)."""
    sl, ns, b = C.c_int64(), C.c_int64(), C.c_int()
    seen_bounded = set()
    for n_src, n_tgt in [(1, 1), (5, 3), (300, 1000), (5000, 12000), (20000, 20000), (50000, 50000), (32845, 200013), (100000, 100000), (700, 3000000),
                         (100000, 3000)]:
        assert lib.siftmi_match_plan(n_src, n_tgt, C.byref(sl), C.byref(ns), C.byref(b)) == 0
        assert sl.value > 0 and sl.value % 64 == 0
        assert ns.value * sl.value >= n_tgt > (ns.value - 1) * sl.value
        groups = (n_src + 511) // 512
        if b.value:
            assert ns.value >= 2 and sl.value >= 2048              # the pre-pass (512 targets) lies inside the first chunk
            assert ns.value * groups <= 2304 + groups
        else:
            assert ns.value * groups <= max(512, groups * 2)       # one round of workgroups
        seen_bounded.add(bool(b.value))
    assert seen_bounded == {False, True}
    assert lib.siftmi_match_plan(0, 5, None, None, None) != 0


def test_no_gpu_means_loud_failure(lib):
    """The product path has no CPU fallback: without a HIP device create() fails with E_NODEVICE."""
    import siftmetal_amd
    from siftmetal_amd import _capi
    if lib.siftmi_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(siftmetal_amd.SiftmiError) as e:
        siftmetal_amd.Engine(64, 64)
    assert e.value.code == _capi.E_NODEVICE
    with pytest.raises(siftmetal_amd.SiftmiError):
        siftmetal_amd.SIFT(0, siftmetal_amd.SIFT.Configuration(siftmetal_amd.IntegralSize(64, 64)))


def test_product_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "siftmetal_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "sift_oracle" not in txt, f


def test_swift_stub_files_match_integration_md():
    """The reference-side binding (swift/) is committed as files; INTEGRATION.md quotes them verbatim."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```(\w*)\n(.*?)```", txt, re.S)
    modmap = [b for lang, b in blocks if b.startswith("module CSiftmi")]
    swift = [b for lang, b in blocks if lang == "swift"]
    assert len(modmap) == 1 and len(swift) == 4
    sw = os.path.join(ROOT, "swift", "Sources")
    assert open(os.path.join(sw, "CSiftmi", "module.modulemap")).read() == modmap[0]
    assert open(os.path.join(sw, "SIFTMetal", "SIFT", "SIFT+MI355X.swift")).read() == swift[0]
    assert open(os.path.join(sw, "SIFTMetal", "SIFT", "SIFT+MI355X+Match.swift")).read() == "import CSiftmi\nimport Foundation\n\n" + swift[1]
    # the reference's matchers are statics of the value type and its call sites (DescriptorTests.swift:141-169) name no device: the stub
    # must offer exactly those signatures (an instance method on SIFT did not compile against `SIFTDescriptor.match(source:...)`)
    ext = swift[1][swift[1].index("extension SIFTDescriptor {"):swift[1].index("extension SIFT {")]
    for name, dflt in (("match", "1.176"), ("approximateMatch", "300"), ("matchGeometry", "1.176")):
        assert re.search(r"public static func %s\(source: \[SIFTDescriptor\], target: \[SIFTDescriptor\],\s*absoluteThreshold: Float = %s, relativeThreshold: Float = 0.6\)" % (name, re.escape(dflt)), ext), name
    assert "-> [SIFTCorrespondence]" in ext and "-> Float" in ext
    assert open(os.path.join(sw, "SIFTMetal", "SIFT", "SIFT+MI355X+Stream.swift")).read() == swift[2]
    # the reference's other public compute types: same names, the reference's Configuration defaults
    assert open(os.path.join(sw, "SIFTMetal", "SIFT", "SIFT+MI355X+Kernels.swift")).read() == swift[3]
    assert "public final class DifferenceOfGaussians {" in swift[3] and "public final class SIFTDescriptorKernel {" in swift[3]
    for field, dflt in (("sigmaMinimum", "0.8"), ("deltaMinimum", "0.5"), ("sigmaInput", "0.5"), ("numberOfOctaves", "7"), ("numberOfScalesPerOctave", "3")):
        assert re.search(r"var %s: \w+ = %s\b" % (field, re.escape(dflt)), swift[3]), field
    assert "public init(inputDimensions: IntegralSize)" in swift[3] and "public func encode(" in swift[3]
    # every siftmi_* symbol the Swift uses is declared in the header
    hdr = open(os.path.join(ROOT, "include", "siftmi.h")).read()
    used = set(re.findall(r"\b(siftmi_[a-z_0-9]+)\s*\(", swift[0] + swift[1] + swift[2] + swift[3]))
    structs = {"siftmi_config", "siftmi_keypoint", "siftmi_descriptor", "siftmi_stream_config", "siftmi_step_host", "siftmi_gathered"}
    assert used and all(re.search(r"\b%s\s*\(" % u, hdr) for u in used - structs), used
    assert "siftmi_stream_submit_host" in used and "siftmi_exchange_gather" in used and "siftmi_copy_dog" in used and "siftmi_copy_gaussian" in used
    man = open(os.path.join(ROOT, "swift", "Package.swift")).read()
    assert '.systemLibrary(name: "CSiftmi"' in man and '.linkedLibrary("siftmi")' in man


def test_bench_refuses_more_gpus_than_visible():
    """bench.py --gpus N spawns its own ranks; with fewer than N devices it must fail loudly before any GPU call
    (`--gpus 8` used to run ONE rank silently and print n_gpus 1)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("more devices than the test asks for")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"--gpus 64" in p.stderr and not p.stdout.strip()
    # a torchrun environment whose WORLD_SIZE disagrees with --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env2, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE is 2" in p.stderr


def test_asm_prefetch_registers_are_not_touched_in_flight():
    """extrema_kernel issues its row loads as asm statements the compiler does not count (three rows ahead) and waits with a
    hand-counted vmcnt; hipcc may legally copy or reuse such a destination VGPR before the data lands.  The generated ISA is
    audited: no compiler instruction names a destination between its load and the wait that retires it."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "siftmetal_amd", "csrc")
    asm = os.path.join(csrc, "siftmi_api.s")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    if not os.path.exists(asm) or any(os.path.getmtime(f) > os.path.getmtime(asm) for f in srcs):
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("no hipcc to regenerate the ISA listing")
        subprocess.check_call(["make", "-C", csrc, "-s", "asm"])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py"), asm], stdout=subprocess.PIPE)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert int(re.search(r"audited (\d+) asm loads", out).group(1)) >= 100, out


def test_makefile_lists_every_part_of_the_translation_unit():
    """csrc/siftmi_api.hip is one translation unit in parts (`#include "x.hip.h"`): every part, and every header a part includes, is a
    prerequisite of libsiftmi.so in csrc/Makefile (a stale library after an edit to a part would pass every CPU-side check)."""
    csrc = os.path.join(ROOT, "siftmetal_amd", "csrc")
    hdr_line = [l for l in open(os.path.join(csrc, "Makefile")).read().splitlines() if l.startswith("HDR")][0]
    listed = {os.path.basename(t) for t in hdr_line.split("=", 1)[1].split()}
    seen, todo = set(), ["siftmi_api.hip"]
    while todo:
        f = todo.pop()
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(os.path.join(csrc, f)).read(), re.M):
            base = os.path.basename(inc)
            if base not in seen:
                seen.add(base)
                assert os.path.exists(os.path.join(csrc, inc)), inc
                if os.path.dirname(inc) == "":
                    todo.append(inc)
    assert seen <= listed, "not prerequisites of ../libsiftmi.so in csrc/Makefile: %s" % sorted(seen - listed)
    assert {"launch_sequence.hip.h", "batch_api.hip.h", "describe_match_api.hip.h", "inspect_api.hip.h"} <= seen
