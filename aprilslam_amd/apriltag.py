"""Drop-in for the `apriltag` Python module the reference imports.

The reference does `from apriltag import apriltag`, constructs `apriltag(tag_type)` and
calls `.detect(gray)` (reference src/detection/tag_detector.py:11,18,26).  Upstream's module
is a CPython extension around the AprilRobotics C detector; this shim has the same
constructor keywords and result shape, and forwards to the HIP detector through the C ABI
(include/aprilslam.h: asl_detector_create / asl_detect_gray_u8).  There is no CPU path.

`lib/apriltag/build/apriltag.py` re-exports this module from the location the reference
puts on sys.path (tag_detector.py:7-9).
"""
import numpy as np

from . import _lib


class apriltag(object):
    def __init__(self, family, threads=1, maxhamming=1, decimate=2.0, blur=0.0, refine_edges=True, debug=False,
                 device=0, id_limit=None):
        if not isinstance(family, str):
            raise TypeError("family must be a string")
        try:
            # id_limit=None: only the ids the reference pins (0..4) are decoded; 0 opens the build-defined rest
            self._det = _lib.Detector(family, threads, maxhamming, decimate, blur, refine_edges, device, id_limit)
        except _lib.AslError as e:
            # upstream raises RuntimeError for an unrecognised family / bad options
            raise RuntimeError(str(e))
        self.family = family
        self.debug = bool(debug)

    def detect(self, image):
        """image: 2-D uint8 array.  Returns a tuple of dicts with the keys upstream's wrapper
        emits: 'hamming', 'margin', 'id', 'center', 'lb-rb-rt-lt'."""
        a = np.asarray(image)
        if a.ndim != 2 or a.dtype != np.uint8:
            raise RuntimeError("Expected a 2-D uint8 array (got ndim=%d dtype=%s)" % (a.ndim, a.dtype))
        dets, _ = self._det.detect_host(np.ascontiguousarray(a))
        return tuple(
            {"hamming": int(d["hamming"]), "margin": float(d["margin"]), "id": int(d["id"]),
             "center": np.array(d["center"], dtype=np.float64), "lb-rb-rt-lt": np.array(d["corners"], dtype=np.float64)}
            for d in dets)
