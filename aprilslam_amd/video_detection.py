"""Headless counterpart of the reference's webcam caller (src/detection/video_detection.py:33-296): the same
sequence per frame -- `TagDetector.detect`, then per detection `get_pose`, `distance`, `euler_angles` -- over any
iterable of BGR frames instead of `cv2.VideoCapture`, with the report lines the reference prints.  This is the
caller with a *calibrated* camera: `dist_coeffs` are the five Brown-Conrady coefficients of
`data/calibration/camera_calibration_parameters.npz` (calibrate.py:71-75), so the PnP stage runs with non-zero
k1, k2, p1, p2, k3.  Window handling, drawing and keyboard input (cv2 GUI) are out of scope.
"""
import os
import time

import numpy as np

from .tag_detector import TagDetector


def load_camera_calibration(calibration_path=None):
    """{'camera_matrix', 'dist_coeffs'} from an .npz written by the reference's calibration tool
    (video_detection.py:33-66: same keys, same exceptions: FileNotFoundError, KeyError)."""
    if calibration_path is None:
        calibration_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data', 'calibration',
                                        'camera_calibration_parameters.npz')
    try:
        with np.load(calibration_path) as data:
            camera_params = {'camera_matrix': data['camera_matrix'], 'dist_coeffs': data['dist_coeffs']}
        print(f"Loaded camera calibration from: {calibration_path}")
        return camera_params
    except FileNotFoundError:
        print(f"Camera calibration file not found: {calibration_path}")
        raise
    except KeyError as e:
        print(f"Missing calibration parameter: {e}")
        raise


def process_detections(detector, detections, frame=None, out=print):
    """What the reference does with one frame's detections (video_detection.py:105-160), minus the drawing: returns
    (frame, records) with one record per detection: id, ok, rvec, tvec, distance_mm, yaw / pitch / roll in degrees,
    integer corners."""
    records = []
    if not detections:
        return frame, records
    for detection in detections:
        tag_id = detection['id']
        retval, rvec, tvec, _T = detector.get_pose(detection)
        corners = np.array(detection['lb-rb-rt-lt'], dtype=np.float32)
        if retval:
            if frame is not None:
                frame = detector.draw(rvec, tvec, corners, frame, tag_id)
            distance_mm = detector.distance(tvec) * 1000
            yaw, pitch, roll = detector.euler_angles(rvec)
            out(f"Tag ID {tag_id}:")
            out(f"   Position (x,y,z): ({tvec[0][0]:.3f}, {tvec[1][0]:.3f}, {tvec[2][0]:.3f}) m")
            out(f"   Distance: {distance_mm:.1f} mm")
            out(f"   Orientation - Yaw: {yaw:.1f}\N{DEGREE SIGN}, Pitch: {pitch:.1f}\N{DEGREE SIGN}, Roll: {roll:.1f}\N{DEGREE SIGN}")
            out(f"   Corners: {corners.astype(int).tolist()}")
            out("   " + "-" * 50)
            records.append({'id': tag_id, 'ok': True, 'rvec': rvec, 'tvec': tvec, 'distance_mm': float(distance_mm),
                            'yaw': float(yaw), 'pitch': float(pitch), 'roll': float(roll), 'corners': corners.astype(int).tolist()})
        else:
            out(f"Tag ID {tag_id}: Detection OK, but pose estimation failed")
            records.append({'id': tag_id, 'ok': False, 'corners': corners.astype(int).tolist()})
    return frame, records


def run(frames, camera_params, tag_type="tagStandard41h12", tag_size=0.06, out=print, detector=None, **detector_kw):
    """The detection loop of the reference's main() (video_detection.py:209-296) over an iterable of BGR frames.
    Returns (per-frame record lists, frames per second over the run)."""
    if detector is None:
        detector = TagDetector(camera_params=camera_params, tag_type=tag_type, tag_size=tag_size, **detector_kw)
    per_frame = []
    t0 = time.perf_counter()
    for frame in frames:
        detections = detector.detect(frame)
        _, records = process_detections(detector, detections, None, out)
        per_frame.append(records)
    dt = time.perf_counter() - t0
    return per_frame, (len(per_frame) / dt if dt > 0 else 0.0)
