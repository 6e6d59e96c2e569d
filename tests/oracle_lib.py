"""ctypes binding of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("ASO_SO") or os.path.join(ROOT, "oracle", "liboracle.so")  # ASO_SO: variant builds for diagnosis


class AsoDetection(C.Structure):
    _fields_ = [("id", C.c_int32), ("hamming", C.c_int32), ("margin", C.c_float), ("reserved", C.c_int32),
                ("center", C.c_double * 2), ("corners", (C.c_double * 2) * 4)]


class AsoFamily(C.Structure):
    _fields_ = [("nbits", C.c_int), ("width_at_border", C.c_int), ("total_width", C.c_int),
                ("reversed_border", C.c_int), ("ncodes", C.c_int), ("codes", C.POINTER(C.c_uint64)),
                ("bit_x", C.POINTER(C.c_int)), ("bit_y", C.POINTER(C.c_int))]


class AsoParams(C.Structure):
    _fields_ = [("decimate", C.c_int), ("maxhamming", C.c_int), ("refine_edges", C.c_int)]


class AsoPoint(C.Structure):
    _fields_ = [("cluster", C.c_uint64), ("x", C.c_uint16), ("y", C.c_uint16), ("gx", C.c_int16), ("gy", C.c_int16)]


class AsoQuad(C.Structure):
    _fields_ = [("p", (C.c_double * 2) * 4), ("reversed_border", C.c_int), ("cluster", C.c_uint64)]


POINT_DTYPE = np.dtype([("cluster", "<u8"), ("x", "<u2"), ("y", "<u2"), ("gx", "<i2"), ("gy", "<i2")])
assert POINT_DTYPE.itemsize == C.sizeof(AsoPoint)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        L.aso_gradient_clusters.restype = C.c_long
        L.aso_detect_gray.restype = C.c_int
        L.aso_detect_bgr.restype = C.c_int
        L.aso_fit_quads.restype = C.c_int
        L.aso_decode_quad.restype = C.c_int
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Family:
    """Keeps the numpy arrays alive behind an AsoFamily struct."""

    def __init__(self, fam):
        self.codes = np.ascontiguousarray(fam.codes, dtype=np.uint64)
        self.bx = np.ascontiguousarray(fam.bit_x, dtype=np.int32)
        self.by = np.ascontiguousarray(fam.bit_y, dtype=np.int32)
        self.c = AsoFamily(fam.nbits, fam.width_at_border, fam.total_width, int(fam.reversed_border), len(self.codes),
                           self.codes.ctypes.data_as(C.POINTER(C.c_uint64)),
                           self.bx.ctypes.data_as(C.POINTER(C.c_int)), self.by.ctypes.data_as(C.POINTER(C.c_int)))


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().aso_bgr2gray(_u8(bgr), w, h, w * 3, _u8(out))
    return out


def decimate(gray, f):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    sw, sh = 1 + (w - 1) // f, 1 + (h - 1) // f
    out = np.empty((sh, sw), np.uint8)
    a, b = C.c_int(), C.c_int()
    lib().aso_decimate(_u8(gray), w, h, w, f, _u8(out), C.byref(a), C.byref(b))
    assert (a.value, b.value) == (sw, sh)
    return out


def threshold(im):
    im = np.ascontiguousarray(im, dtype=np.uint8)
    h, w = im.shape
    out = np.empty((h, w), np.uint8)
    lib().aso_threshold(_u8(im), w, h, _u8(out))
    return out


def connected_components(th):
    th = np.ascontiguousarray(th, dtype=np.uint8)
    h, w = th.shape
    labels = np.empty((h, w), np.uint32)
    sizes = np.empty((h, w), np.uint32)
    lib().aso_connected_components(_u8(th), w, h, labels.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p))
    return labels, sizes


def gradient_clusters(th, labels, sizes):
    h, w = th.shape
    cap = th.size * 4
    pts = np.zeros(cap, dtype=POINT_DTYPE)
    n = lib().aso_gradient_clusters(_u8(np.ascontiguousarray(th)), w, h, labels.ctypes.data_as(C.c_void_p),
                                    sizes.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), C.c_long(cap))
    assert n >= 0
    return pts[:n].copy()


def fit_quads(dec, pts, fam, decimate_f, cap=4096):
    h, w = dec.shape
    F = Family(fam)
    out = (AsoQuad * cap)()
    pts = np.ascontiguousarray(pts)
    n = lib().aso_fit_quads(_u8(np.ascontiguousarray(dec)), w, h, pts.ctypes.data_as(C.c_void_p), C.c_long(len(pts)),
                            C.byref(F.c), decimate_f, out, cap)
    return [dict(p=np.array([[q.p[i][0], q.p[i][1]] for i in range(4)]), reversed_border=q.reversed_border,
                 cluster=q.cluster) for q in out[:n]]


def _dets(arr, n):
    return [dict(id=d.id, hamming=d.hamming, margin=d.margin, center=np.array(list(d.center)),
                 corners=np.array([[d.corners[i][0], d.corners[i][1]] for i in range(4)])) for d in arr[:n]]


def detect_gray(gray, fam, decimate_f=2, maxhamming=1, refine_edges=1, cap=1024):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    F = Family(fam)
    prm = AsoParams(decimate_f, maxhamming, refine_edges)
    out = (AsoDetection * cap)()
    n = lib().aso_detect_gray(_u8(gray), w, h, w, C.byref(F.c), C.byref(prm), out, cap)
    return _dets(out, n)


def detect_bgr(bgr, fam, decimate_f=2, maxhamming=1, refine_edges=1, cap=1024):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    F = Family(fam)
    prm = AsoParams(decimate_f, maxhamming, refine_edges)
    out = (AsoDetection * cap)()
    n = lib().aso_detect_bgr(_u8(bgr), w, h, w * 3, C.byref(F.c), C.byref(prm), out, cap)
    return _dets(out, n)


def solve_pnp(corners, K, dist, tag_size):
    """corners: (n, 4, 2); returns rvec (n,3), tvec (n,3), T (n,4,4), ok (n,)"""
    corners = np.ascontiguousarray(np.asarray(corners, dtype=np.float32).astype(np.float64)).reshape(-1, 4, 2)
    n = corners.shape[0]
    K = np.ascontiguousarray(K, dtype=np.float64)
    dist = np.ascontiguousarray(np.asarray(dist, dtype=np.float64).ravel())
    rvec = np.zeros((n, 3)); tvec = np.zeros((n, 3)); T = np.zeros((n, 4, 4)); ok = np.zeros(n, np.uint8)
    dp = C.POINTER(C.c_double)
    lib().aso_solve_pnp(corners.ctypes.data_as(dp), n, K.ctypes.data_as(dp), dist.ctypes.data_as(dp), len(dist),
                        C.c_double(tag_size), rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), T.ctypes.data_as(dp),
                        _u8(ok))
    return rvec, tvec, T, ok.astype(bool)
