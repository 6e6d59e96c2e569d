"""`TagDetector`: host-side mirror of reference src/detection/tag_detector.py:14-88.

Same constructor, methods, argument meaning and return shapes.  The three native calls of
the reference are replaced by the C ABI of libaprilslam.so:
    cv2.cvtColor(BGR2GRAY) + apriltag.detect   -> asl_detect_bgr_u8   (gray conversion fused)
    cv2.solvePnP + cv2.Rodrigues               -> asl_solve_pnp_batch
`detect_batch` / `get_poses` are the batched forms the GPU wants; `detect` / `get_pose` keep
the reference's one-frame / one-tag call surface.
"""
import numpy as np

from . import _lib
from .apriltag import apriltag


def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0]: rotation vector -> 3x3 matrix (host, float64)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r @ r))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


class TagDetector:
    """Handles AprilTag detection and pose estimation (GPU-backed)."""

    def __init__(self, camera_params, tag_type="tagStandard41h12", tag_size=0.06, device=0, id_limit=None):
        self.detector = apriltag(tag_type, device=device, id_limit=id_limit)
        self.tag_size = tag_size
        self.camera_matrix = camera_params['camera_matrix']
        self.dist_coeffs = camera_params['dist_coeffs']

    # -- reference call surface ------------------------------------------------------
    def detect(self, image):
        """BGR (H,W,3) uint8 image -> list of detection dicts sorted by id (tag_detector.py:23-28)."""
        a = np.asarray(image)
        if a.ndim == 2:
            return sorted(self.detector.detect(a), key=lambda d: d['id'])
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError("expected an (H, W, 3) uint8 BGR image")
        # the poses are solved in the same device submission (asl_detect_batch_pose_u8) and ride along under a
        # private key, so that get_pose() of these detections needs no second trip to the GPU; the arithmetic is
        # the one get_pose() does on its own (float32-rounded corners -> asl_solve_pnp_batch), bit for bit
        dets, poses, _ = self.detector._det.detect_host(np.ascontiguousarray(a), K=self.camera_matrix, dist=self._dist(),
                                                        tag_size=self.tag_size)
        out = [{"hamming": int(d["hamming"]), "margin": float(d["margin"]), "id": int(d["id"]),
                "center": np.array(d["center"]), "lb-rb-rt-lt": np.array(d["corners"]),
                "_pose": (np.array(d["corners"]), self._pose_key(), bool(p["ok"]), np.array(p["rvec"]), np.array(p["tvec"]))}
               for d, p in zip(dets, poses)]
        return sorted(out, key=lambda d: d['id'])

    def get_pose(self, detection):
        """(retval, rvec(3,1), tvec(3,1), T 4x4) of one detection (tag_detector.py:30-43)."""
        cached = detection.get('_pose') if isinstance(detection, dict) else None
        if cached is not None and cached[1] == self._pose_key() and np.array_equal(cached[0], detection['lb-rb-rt-lt']):
            rv, tv = cached[3].reshape(3, 1).copy(), cached[4].reshape(3, 1).copy()
            return cached[2], rv, tv, self.transformation(rv, tv)
        corners = np.array(detection['lb-rb-rt-lt'], dtype=np.float32)
        rvec, tvec, _, ok = self.detector._det.solve_pnp(corners[None], self.camera_matrix, self._dist(), self.tag_size)
        rv, tv = rvec[0].reshape(3, 1), tvec[0].reshape(3, 1)
        return bool(ok[0]), rv, tv, self.transformation(rv, tv)

    def transformation(self, rvec, tvec):
        T = np.eye(4)
        T[:3, :3] = rodrigues(rvec)
        T[:3, 3] = np.asarray(tvec, dtype=np.float64).flatten()
        return T

    def euler_angles(self, rvec):
        """[yaw, pitch, roll] in degrees, the reference's convention (tag_detector.py:54-69)."""
        R = rodrigues(rvec)
        sy = np.sqrt(R[0, 0] ** 2 + R[1, 0] ** 2)
        if sy >= 1e-6:
            yaw = np.arctan2(R[0, 2], R[2, 2])
            pitch = np.arctan2(-R[1, 2], sy)
            roll = np.arctan2(R[1, 0], R[1, 1])
        else:
            yaw = np.arctan2(-R[2, 0], R[0, 0])
            pitch = np.arctan2(-R[1, 2], sy)
            roll = 0
        return np.degrees([yaw, pitch, roll])

    def distance(self, tvec):
        return np.linalg.norm(tvec)

    def draw(self, rvec, tvec, corners, image, tag_id):
        """Overlay drawing needs OpenCV's GUI primitives; without cv2 the image is returned as is."""
        try:
            import cv2  # noqa: F401
        except ImportError:
            return image
        for i in range(4):
            p1 = tuple(map(int, corners[i]))
            p2 = tuple(map(int, corners[(i + 1) % 4]))
            cv2.line(image, p1, p2, (0, 255, 0), 2)
        yaw, pitch, roll = self.euler_angles(rvec)
        cv2.putText(image, f'ID: {tag_id}, Dist: {self.distance(tvec):.1f} units', (p1[0], p1[1] - 20),
                    cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 165, 255), 2)
        cv2.putText(image, f'Yaw: {yaw:.1f}, Pitch: {pitch:.1f}, Roll: {roll:.1f}', (p1[0], p1[1] - 40),
                    cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 165, 255), 2)
        return image

    # -- batched forms ------------------------------------------------------------------
    def _pose_key(self):
        """what a cached pose depends on besides the corners: tag size and intrinsics, by value"""
        return (float(self.tag_size), np.asarray(self.camera_matrix, dtype=np.float64).tobytes(), self._dist().tobytes())

    def _dist(self):
        d = np.asarray(self.dist_coeffs, dtype=np.float64).ravel()
        if len(d) not in (0, 4, 5):
            raise ValueError("dist_coeffs must hold 0, 4 or 5 values")
        return d

    def get_poses(self, detections):
        """PnP for many detections in one launch: returns (ok[N], rvec[N,3], tvec[N,3], T[N,4,4])."""
        if not detections:
            return np.zeros(0, bool), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 4, 4))
        c = np.stack([np.asarray(d['lb-rb-rt-lt'], dtype=np.float32) for d in detections])
        rvec, tvec, T, ok = self.detector._det.solve_pnp(c, self.camera_matrix, self._dist(), self.tag_size)
        return ok, rvec, tvec, T

    def detect_batch_device(self, data_ptr, n_frames, channels, width, height, with_pose=True, stream=0, **kw):
        """Frames resident in HBM -> (dets, poses, n_per_frame) structured arrays (see _lib)."""
        K = self.camera_matrix if with_pose else None
        return self.detector._det.detect_device(data_ptr, n_frames, channels, width, height, stream=stream, K=K,
                                                dist=self._dist(), tag_size=self.tag_size, **kw)
