"""ctypes binding of libaprilslam.so (include/aprilslam.h).  No fallback: if the HIP library
is missing or no gfx950 device is usable, every call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASL_LIB") or os.path.join(_HERE, "libaprilslam.so")  # ASL_LIB: diagnostic builds only


class AslError(RuntimeError):
    pass


class AslDetection(C.Structure):
    _fields_ = [("id", C.c_int32), ("hamming", C.c_int32), ("margin", C.c_float), ("frame", C.c_int32),
                ("center", C.c_double * 2), ("corners", (C.c_double * 2) * 4)]


class AslPose(C.Structure):
    _fields_ = [("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("T", C.c_double * 16), ("ok", C.c_int32),
                ("reserved", C.c_int32)]


class AslDebugQuad(C.Structure):
    _fields_ = [("p", (C.c_double * 2) * 4), ("cluster", C.c_uint64), ("frame", C.c_int32), ("reversed_border", C.c_int32)]


PLANE_DTYPE = np.dtype([("Hi", "<f8", (9,)), ("bbox", "<i4", (4,)), ("tex", "<i4"), ("pad", "<i4")])  # asl_render_plane
assert PLANE_DTYPE.itemsize == 96
OBS_DTYPE = np.dtype([("id", "<i4"), ("flags", "<i4"), ("corners", "<f4", (8,)), ("T", "<f8", (12,))])  # asl_obs
assert OBS_DTYPE.itemsize == 136
DET_DTYPE = np.dtype([("id", "<i4"), ("hamming", "<i4"), ("margin", "<f4"), ("frame", "<i4"),
                      ("center", "<f8", (2,)), ("corners", "<f8", (4, 2))])
POSE_DTYPE = np.dtype([("rvec", "<f8", (3,)), ("tvec", "<f8", (3,)), ("T", "<f8", (4, 4)), ("ok", "<i4"), ("reserved", "<i4")])
QUAD_DTYPE = np.dtype([("p", "<f8", (4, 2)), ("cluster", "<u8"), ("frame", "<i4"), ("reversed_border", "<i4")])
assert DET_DTYPE.itemsize == C.sizeof(AslDetection)
assert POSE_DTYPE.itemsize == C.sizeof(AslPose)
assert QUAD_DTYPE.itemsize == C.sizeof(AslDebugQuad)

EXPORTS = [
    "asl_detector_create", "asl_detector_destroy", "asl_detector_set_id_limit", "asl_detector_set_pnp_both_minima", "asl_last_error", "asl_version", "asl_detect_gray_u8",
    "asl_detect_bgr_u8", "asl_detect_batch_u8", "asl_detect_batch_pose_u8", "asl_detect_batch_device", "asl_submit_batch_device", "asl_collect_batch", "asl_collect_batch_view", "asl_solve_pnp_batch", "asl_gn_solve", "asl_pack_observations_device", "asl_graph_frames_device", "asl_graph_picks_device", "asl_render_frames_device",
    "asl_debug_fetch", "asl_stage_times", "asl_set_profiling", "asl_debug_phase_cycles",
]

_lib = None


def load():
    """Load libaprilslam.so and declare the prototypes.  Raises AslError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AslError("libaprilslam.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'`"
                       % LIB_PATH)
    # ONE HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1, and a
    # second runtime initialised in the same process cannot see the GPU.  Importing torch first makes
    # libaprilslam.so's NEEDED "libamdhip64.so.7" resolve to the copy torch already loaded (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, dp, u8p = C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    L.asl_last_error.restype = C.c_char_p
    L.asl_version.restype = C.c_char_p
    L.asl_detector_create.argtypes = [C.c_char_p, i32, i32, C.c_float, C.c_float, i32, i32, C.POINTER(vp)]
    L.asl_detector_destroy.argtypes = [vp]
    L.asl_detector_destroy.restype = None
    L.asl_detector_set_id_limit.argtypes = [vp, i32]
    L.asl_detector_set_pnp_both_minima.argtypes = [vp, i32]
    L.asl_detect_gray_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    L.asl_detect_bgr_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    L.asl_detect_batch_u8.argtypes = [vp, C.POINTER(vp), i32, i32, i32, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.asl_detect_batch_pose_u8.argtypes = [vp, C.POINTER(vp), i32, i32, i32, i32, i32, dp, dp, i32, C.c_double, vp, vp, i32,
                                           C.POINTER(i32), C.POINTER(i32)]
    L.asl_detect_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_size_t, vp, dp, dp, i32, C.c_double,
                                          vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.asl_submit_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_size_t, vp, dp, dp, i32, C.c_double]
    L.asl_collect_batch.argtypes = [vp, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.asl_collect_batch_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32)]
    L.asl_solve_pnp_batch.argtypes = [vp, C.POINTER(C.c_float), dp, dp, i32, C.c_double, dp, dp, dp, u8p, i32]
    L.asl_gn_solve.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.c_double, i32,
                               dp, dp, i32, dp]
    L.asl_render_frames_device.argtypes = [vp, vp, i32, i32, i32, i32, C.c_size_t, vp, i32, vp, i32, i32, C.c_double, dp, dp, i32, vp]
    L.asl_pack_observations_device.argtypes = [vp, vp, i32, vp]
    L.asl_graph_frames_device.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]
    L.asl_graph_picks_device.argtypes = [vp, vp, i32, i32, i32, vp, C.c_uint32, C.c_uint32, vp, i32, vp, vp]
    L.asl_debug_fetch.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.asl_stage_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), i32, C.POINTER(i32)]
    L.asl_set_profiling.argtypes = [vp, i32]
    L.asl_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise AslError("libaprilslam error %d: %s" % (rc, load().asl_last_error().decode("utf-8", "replace")))


class Detector:
    """Owns one asl_detector (one GPU workspace).  Not re-entrant."""

    def __init__(self, family="tagStandard41h12", threads=1, maxhamming=1, decimate=2.0, blur=0.0, refine_edges=True,
                 device=0, id_limit=None):
        """id_limit: None = the ids the reference pins (0..4); 0 = the whole (build-defined) table; n = ids 0..n-1."""
        L = load()
        self._L = L
        self._h = C.c_void_p()
        check(L.asl_detector_create(family.encode(), int(threads), int(maxhamming), float(decimate), float(blur),
                                    1 if refine_edges else 0, int(device), C.byref(self._h)))
        self.device = int(device)
        if id_limit is not None:
            check(L.asl_detector_set_id_limit(self._h, int(id_limit)))

    def set_pnp_both_minima(self, enabled):
        """asl_detector_set_pnp_both_minima: off = the reference's (cv2's) single minimum, on = the better of the two planar poses"""
        check(self._L.asl_detector_set_pnp_both_minima(self._h, 1 if enabled else 0))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.asl_detector_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host images ------------------------------------------------------------------
    def detect_host(self, images, max_per_frame=256, channels=None, K=None, dist=None, tag_size=0.0):
        """images: (H,W) / (H,W,3) uint8 array, or (B,H,W[,3]); pass channels=1 for a gray batch whose W is 3.
        Returns (dets, n_per_frame); with a camera matrix K the per-tag PnP runs in the same submission
        (asl_detect_batch_pose_u8) and the result is (dets, poses, n_per_frame)."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim == 2:
            a = a[None]
        elif a.ndim == 3 and a.shape[2] == 3 and channels != 1:
            a = a[None]
        if a.ndim == 3:
            ch = 1
            B, H, W = a.shape
        elif a.ndim == 4 and a.shape[3] == 3:
            ch = 3
            B, H, W = a.shape[:3]
        else:
            raise ValueError("expected (H,W), (H,W,3), (B,H,W) or (B,H,W,3) uint8")
        stride = W * ch
        ptrs = (C.c_void_p * B)(*[a[i].ctypes.data for i in range(B)])
        cap = B * max_per_frame
        out = np.empty(cap, dtype=DET_DTYPE)
        npf = (C.c_int * B)()
        n = C.c_int()
        if K is not None:
            dp = C.POINTER(C.c_double)
            Kc = np.ascontiguousarray(K, dtype=np.float64)
            dc = np.ascontiguousarray(np.zeros(0) if dist is None else dist, dtype=np.float64).ravel()
            if len(dc) not in (0, 4, 5):
                raise ValueError("dist must have 0, 4 or 5 coefficients")
            poses = np.empty(cap, dtype=POSE_DTYPE)
            check(self._L.asl_detect_batch_pose_u8(self._h, ptrs, B, ch, W, H, stride, Kc.ctypes.data_as(dp),
                                                   dc.ctypes.data_as(dp) if len(dc) else None, len(dc), float(tag_size),
                                                   out.ctypes.data, poses.ctypes.data, cap, npf, C.byref(n)))
            if n.value > cap:
                return self.detect_host(images, (n.value + B - 1) // B + 1, channels, K, dist, tag_size)
            return out[:n.value], poses[:n.value], np.array(list(npf), dtype=np.int64)
        check(self._L.asl_detect_batch_u8(self._h, ptrs, B, ch, W, H, stride, out.ctypes.data, cap, npf, C.byref(n)))
        if n.value > cap:
            return self.detect_host(images, max_per_frame=(n.value + B - 1) // B + 1, channels=channels)
        return out[:n.value], np.array(list(npf), dtype=np.int64)

    # -- frames resident in HBM ---------------------------------------------------------
    def detect_device(self, data_ptr, n_frames, channels, width, height, stride=None, frame_pitch=None, stream=0,
                      K=None, dist=None, tag_size=0.0, max_per_frame=64, want_poses=None, reuse_buffers=False):
        stride = stride or width * channels
        frame_pitch = frame_pitch or stride * height
        cap = n_frames * max_per_frame
        want_poses = (K is not None) if want_poses is None else want_poses
        # result buffers are kept and reused (fresh multi-MB arrays cost ~1 ms of page faults per call);
        # the returned arrays are views into them, valid until the next call on this detector
        if reuse_buffers and getattr(self, "_outbuf", None) is not None and len(self._outbuf) >= cap:
            out = self._outbuf
        else:
            out = np.empty(cap, dtype=DET_DTYPE)
            self._outbuf = out if reuse_buffers else None
        if not want_poses:
            poses = np.empty(0, dtype=POSE_DTYPE)
        elif reuse_buffers and getattr(self, "_posebuf", None) is not None and len(self._posebuf) >= cap:
            poses = self._posebuf
        else:
            poses = np.empty(cap, dtype=POSE_DTYPE)
            self._posebuf = poses if reuse_buffers else None
        npf = (C.c_int * n_frames)()
        n = C.c_int()
        dp = C.POINTER(C.c_double)
        if K is not None:
            Kc = np.ascontiguousarray(K, dtype=np.float64)
            dc = np.ascontiguousarray(np.zeros(0) if dist is None else dist, dtype=np.float64).ravel()
            nd = len(dc)
            if nd not in (0, 4, 5):
                raise ValueError("dist must have 0, 4 or 5 coefficients")
            Kp = Kc.ctypes.data_as(dp)
            dpp = dc.ctypes.data_as(dp) if nd else None
        else:
            Kp, dpp, nd = None, None, 0
        check(self._L.asl_detect_batch_device(self._h, C.c_void_p(int(data_ptr)), n_frames, channels, width, height, stride,
                                              frame_pitch, C.c_void_p(int(stream)), Kp, dpp, nd, float(tag_size),
                                              out.ctypes.data, poses.ctypes.data if want_poses else None, cap, npf,
                                              C.byref(n)))
        if n.value > cap:
            return self.detect_device(data_ptr, n_frames, channels, width, height, stride, frame_pitch, stream, K, dist,
                                      tag_size, max_per_frame=(n.value + n_frames - 1) // n_frames + 1, want_poses=want_poses,
                                      reuse_buffers=reuse_buffers)
        return out[:n.value], (poses[:n.value] if want_poses else None), np.array(list(npf), dtype=np.int64)

    def submit_device(self, data_ptr, n_frames, channels, width, height, stride=None, frame_pitch=None, stream=0, K=None,
                      dist=None, tag_size=0.0):
        """Enqueue a batch resident in HBM and return immediately (asl_submit_batch_device); pair with collect()."""
        stride = stride or width * channels
        frame_pitch = frame_pitch or stride * height
        dp = C.POINTER(C.c_double)
        if K is not None:
            Kc = np.ascontiguousarray(K, dtype=np.float64)
            dc = np.ascontiguousarray(np.zeros(0) if dist is None else dist, dtype=np.float64).ravel()
            if len(dc) not in (0, 4, 5):
                raise ValueError("dist must have 0, 4 or 5 coefficients")
            Kp, dpp, nd = Kc.ctypes.data_as(dp), (dc.ctypes.data_as(dp) if len(dc) else None), len(dc)
        else:
            Kp, dpp, nd = None, None, 0
        check(self._L.asl_submit_batch_device(self._h, C.c_void_p(int(data_ptr)), n_frames, channels, width, height, stride,
                                              frame_pitch, C.c_void_p(int(stream)), Kp, dpp, nd, float(tag_size)))
        self._inflight = (n_frames, K is not None)

    def render_frames_device(self, frames_ptr, n_frames, width, height, planes_ptr, max_planes, textures_ptr, tw, th, half, K=None, dist=None,
                             stride=None, frame_pitch=None, stream=0):
        """asl_render_frames_device: BGR frames straight into device memory (frames_ptr, planes_ptr, textures_ptr are device
        addresses; K / dist are host arrays, only for a camera with lens distortion)."""
        stride = stride or 3 * width
        frame_pitch = frame_pitch or stride * height
        dp = C.POINTER(C.c_double)
        Kp = dpp = None
        nd = 0
        if dist is not None:
            Kc = np.ascontiguousarray(K, dtype=np.float64)
            dc = np.ascontiguousarray(dist, dtype=np.float64).ravel()
            Kp, dpp, nd = Kc.ctypes.data_as(dp), dc.ctypes.data_as(dp), len(dc)
        check(self._L.asl_render_frames_device(self._h, C.c_void_p(int(frames_ptr)), int(n_frames), int(width), int(height), int(stride),
                                               int(frame_pitch), C.c_void_p(int(planes_ptr)), int(max_planes), C.c_void_p(int(textures_ptr)),
                                               int(tw), int(th), float(half), Kp, dpp, nd, C.c_void_p(int(stream))))

    def pack_observations_device(self, out_ptr, max_tags, stream=0):
        """asl_pack_observations_device: the submitted batch's results as n_frames x max_tags asl_obs records at the
        device address `out_ptr`, enqueued on `stream` (use the stream the batch was submitted on)."""
        check(self._L.asl_pack_observations_device(self._h, C.c_void_p(int(out_ptr)), int(max_tags), C.c_void_p(int(stream))))

    def graph_frames_device(self, obs_ptr, world, n_frames, max_tags, coordinate_id, pose_ptr, status_ptr, last_ptr, n_ids, picks_ptr=0,
                            stream=0):
        """asl_graph_frames_device (all pointers are device addresses; picks_ptr = 0 skips the picks)."""
        check(self._L.asl_graph_frames_device(self._h, C.c_void_p(int(obs_ptr)), int(world), int(n_frames), int(max_tags),
                                              int(coordinate_id), C.c_void_p(int(pose_ptr)), C.c_void_p(int(status_ptr)),
                                              C.c_void_p(int(last_ptr)), int(n_ids), C.c_void_p(int(picks_ptr)) if picks_ptr else None,
                                              C.c_void_p(int(stream))))

    def graph_picks_device(self, obs_ptr, world, n_frames, max_tags, status_ptr, order_lo, order_hi, last_ptr, n_ids, picks_ptr, stream=0):
        """asl_graph_picks_device: last sightings + picks of the status-0 frames at positions [order_lo, order_hi)."""
        check(self._L.asl_graph_picks_device(self._h, C.c_void_p(int(obs_ptr)), int(world), int(n_frames), int(max_tags), C.c_void_p(int(status_ptr)),
                                             int(order_lo), int(order_hi), C.c_void_p(int(last_ptr)), int(n_ids), C.c_void_p(int(picks_ptr)),
                                             C.c_void_p(int(stream))))

    def collect_view(self):
        """Wait for the submitted batch; (dets, poses or None, n_per_frame) as numpy VIEWS of the detector's page-locked result
        buffers (asl_collect_batch_view: no copy) -- valid until the next submit on this detector."""
        n_frames, want_poses = self._inflight
        pd, pp, pn, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        check(self._L.asl_collect_batch_view(self._h, C.byref(pd), C.byref(pp), C.byref(pn), C.byref(n)))
        nd = n.value

        def view(ptr, dtype, count):
            if not ptr or count == 0:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_char * (count * dtype.itemsize)).from_address(ptr), dtype=dtype, count=count)
        dets = view(pd.value, DET_DTYPE, nd)
        poses = view(pp.value, POSE_DTYPE, nd) if (want_poses and pp.value) else None
        npf = view(pn.value, np.dtype(np.uint32), n_frames)
        return dets, poses, npf

    def collect(self, max_per_frame=64):
        """Wait for the submitted batch; returns (dets, poses or None, n_per_frame).  The arrays are views into
        buffers owned by the detector, valid until its next collect()."""
        n_frames, want_poses = self._inflight
        cap = n_frames * max_per_frame
        if getattr(self, "_outbuf", None) is None or len(self._outbuf) < cap:
            self._outbuf = np.empty(cap, dtype=DET_DTYPE)
        if want_poses and (getattr(self, "_posebuf", None) is None or len(self._posebuf) < cap):
            self._posebuf = np.empty(cap, dtype=POSE_DTYPE)
        npf = (C.c_int * n_frames)()
        n = C.c_int()
        check(self._L.asl_collect_batch(self._h, self._outbuf.ctypes.data, self._posebuf.ctypes.data if want_poses else None,
                                        cap, npf, C.byref(n)))
        if n.value > cap:
            raise AslError("more than %d detections per frame on average; raise max_per_frame" % max_per_frame)
        return self._outbuf[:n.value], (self._posebuf[:n.value] if want_poses else None), np.frombuffer(npf, dtype=np.int32)

    def solve_pnp(self, corners, K, dist, tag_size):
        c = np.ascontiguousarray(np.asarray(corners, dtype=np.float32).reshape(-1, 4, 2))
        N = c.shape[0]
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        dc = np.ascontiguousarray(np.zeros(0) if dist is None else dist, dtype=np.float64).ravel()
        if len(dc) not in (0, 4, 5):
            raise ValueError("dist must have 0, 4 or 5 coefficients")
        rvec = np.zeros((N, 3)); tvec = np.zeros((N, 3)); T = np.zeros((N, 4, 4)); ok = np.zeros(N, np.uint8)
        dp = C.POINTER(C.c_double)
        check(self._L.asl_solve_pnp_batch(self._h, c.ctypes.data_as(C.POINTER(C.c_float)), Kc.ctypes.data_as(dp),
                                          dc.ctypes.data_as(dp) if len(dc) else None, len(dc), float(tag_size),
                                          rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), T.ctypes.data_as(dp),
                                          ok.ctypes.data_as(C.POINTER(C.c_uint8)), N))
        return rvec, tvec, T, ok.astype(bool)

    def gn_solve(self, cam_T, tag_T, obs_cam, obs_tag, obs_corners, K, tag_size, fixed_tag=0, iters=10):
        """Pose-graph Levenberg-Marquardt on the device (asl_gn_solve).  cam_T (P,4,4) world<-camera,
        tag_T (L,4,4) world<-tag, observations (cam index, tag index, 4x2 pixel corners).
        Returns refined (cam_T, tag_T, stats=[cost0, cost, accepted])."""
        cam = np.ascontiguousarray(cam_T, dtype=np.float64).reshape(-1, 16).copy()
        tag = np.ascontiguousarray(tag_T, dtype=np.float64).reshape(-1, 16).copy()
        oc = np.ascontiguousarray(obs_cam, dtype=np.int32)
        ot = np.ascontiguousarray(obs_tag, dtype=np.int32)
        corners = np.ascontiguousarray(obs_corners, dtype=np.float64).reshape(-1, 8)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        stats = np.zeros(3)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        check(self._L.asl_gn_solve(self._h, cam.shape[0], tag.shape[0], len(oc), oc.ctypes.data_as(ip), ot.ctypes.data_as(ip),
                                   corners.ctypes.data_as(dp), Kc.ctypes.data_as(dp), float(tag_size), int(fixed_tag),
                                   cam.ctypes.data_as(dp), tag.ctypes.data_as(dp), int(iters), stats.ctypes.data_as(dp)))
        return cam.reshape(-1, 4, 4), tag.reshape(-1, 4, 4), stats

    # -- introspection for the parity tests ------------------------------------------------
    def debug_counters(self):
        buf = np.zeros(18, dtype=np.int64)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, 5, buf.ctypes.data, buf.nbytes, C.byref(n)))
        return buf

    def debug_image(self, what):
        c = self.debug_counters()
        B, sw, sh = int(c[0]), int(c[1]), int(c[2])
        dt = np.uint8 if what in (0, 1) else np.uint32
        buf = np.zeros((B, sh, sw), dtype=dt)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, what, buf.ctypes.data, buf.nbytes, C.byref(n)))
        return buf

    def debug_quads(self, cap=65536):
        buf = np.zeros(cap, dtype=QUAD_DTYPE)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, 4, buf.ctypes.data, buf.nbytes, C.byref(n)))
        return buf[:n.value]

    def debug_clusters(self):
        """(n, 3) uint64: key, point count, hash of the sorted point records of every cluster of the last batch, by key."""
        ncl = int(self.debug_counters()[3])
        buf = np.zeros((max(ncl, 1), 3), dtype=np.uint64)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, 6, buf.ctypes.data, buf.nbytes, C.byref(n)))
        return buf[:n.value]

    def debug_refit(self, reps):
        """Re-run the quad fit of the last batch `reps` times; int64[7]: reps, quads differing from the first run, by size class."""
        buf = np.zeros(7, dtype=np.int64)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, 7, buf.ctypes.data, int(reps), C.byref(n)))
        return buf

    def debug_division_check(self, exponent_limit=100):
        """(pairs, mismatches) of div_by(a, recip_of(d)) against a / d on the device, exponents within +-exponent_limit."""
        buf = np.zeros(2, dtype=np.int64)
        n = C.c_size_t()
        check(self._L.asl_debug_fetch(self._h, 8, buf.ctypes.data, int(exponent_limit), C.byref(n)))
        return int(buf[0]), int(buf[1])

    def phase_cycles(self, reset=True):
        buf = (C.c_uint64 * 64)()
        check(self._L.asl_debug_phase_cycles(self._h, buf, 1 if reset else 0))
        return np.array(list(buf), dtype=np.uint64)

    def set_profiling(self, on=True):
        check(self._L.asl_set_profiling(self._h, 1 if on else 0))

    def stage_times(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = C.c_int()
        check(self._L.asl_stage_times(self._h, names, ms, 32, C.byref(n)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}
