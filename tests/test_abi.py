"""The C-ABI library loads and exports every symbol include/aprilslam.h declares; without a GPU the
product path fails loudly instead of falling back to anything."""
import os
import re

import pytest

from aprilslam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "aprilslam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asl_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported():
    L = _lib.load()
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), "libaprilslam.so does not export %s" % n
    assert set(_lib.EXPORTS) == set(names)
    assert b"gfx950" in L.asl_version()


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.AslDetection) == 4 + 4 + 4 + 4 + 16 + 64
    assert C.sizeof(_lib.AslPose) == 24 + 24 + 128 + 8
    assert _lib.DET_DTYPE.itemsize == C.sizeof(_lib.AslDetection)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.AslError):
        _lib.Detector("tagStandard41h12")
    from aprilslam_amd.apriltag import apriltag
    with pytest.raises(RuntimeError):
        apriltag("tagStandard41h12")


def test_product_does_not_touch_the_oracle():
    """Nothing under aprilslam_amd/ (or the drop-in shim) may import, link or open oracle/."""
    bad = []
    for base in ("aprilslam_amd", "lib"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".inc", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle_lib|liboracle|oracle/|aso_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_reference_shim_location():
    """`from apriltag import apriltag` resolves from the directory the reference puts on sys.path
    (reference src/detection/tag_detector.py:7-9)."""
    import importlib.util
    p = os.path.join(ROOT, "lib", "apriltag", "build", "apriltag.py")
    spec = importlib.util.spec_from_file_location("apriltag_shim_test", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from aprilslam_amd.apriltag import apriltag as impl
    assert m.apriltag is impl
