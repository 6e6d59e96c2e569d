"""`apriltag` module at the path the reference adds to sys.path
(reference src/detection/tag_detector.py:7-9: <root>/lib/apriltag/build).
Re-exports the HIP-backed shim so that `from apriltag import apriltag` resolves to it."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)

from aprilslam_amd.apriltag import apriltag  # noqa: E402,F401
