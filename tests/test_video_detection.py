"""Headless webcam caller (aprilslam_amd/video_detection.py; reference src/detection/video_detection.py)."""
import numpy as np
import pytest

from aprilslam_amd import synth, video_detection as V


class _FakeDetector:
    def get_pose(self, d):
        ok = d['id'] != 7
        return ok, np.array([[0.1], [0.2], [0.3]]), np.array([[0.01], [-0.02], [0.5]]), np.eye(4)

    def distance(self, tvec):
        return float(np.linalg.norm(tvec))

    def euler_angles(self, rvec):
        return np.array([10.0, -5.0, 1.5])

    def draw(self, *a):
        return a[3]


def test_calibration_loader_contract(tmp_path):
    p = tmp_path / "camera_calibration_parameters.npz"
    K = synth.camera_matrix(640, 480, 60.0)
    np.savez(p, camera_matrix=K, dist_coeffs=np.array([[0.1, -0.2, 0.001, 0.002, 0.05]]))
    cp = V.load_camera_calibration(str(p))
    assert np.array_equal(cp['camera_matrix'], K) and cp['dist_coeffs'].size == 5
    with pytest.raises(FileNotFoundError):
        V.load_camera_calibration(str(tmp_path / "missing.npz"))
    np.savez(tmp_path / "bad.npz", camera_matrix=K)
    with pytest.raises(KeyError):
        V.load_camera_calibration(str(tmp_path / "bad.npz"))


def test_process_detections_report_lines():
    lines = []
    dets = [{'id': 3, 'lb-rb-rt-lt': [[10.2, 20.7], [30, 20], [30, 40], [10, 40]]}, {'id': 7, 'lb-rb-rt-lt': [[0, 0], [1, 0], [1, 1], [0, 1]]}]
    _, rec = V.process_detections(_FakeDetector(), dets, None, lines.append)
    assert [r['id'] for r in rec] == [3, 7] and rec[0]['ok'] and not rec[1]['ok']
    assert abs(rec[0]['distance_mm'] - 1000 * np.linalg.norm([0.01, -0.02, 0.5])) < 1e-9
    assert lines[0] == "Tag ID 3:" and lines[1] == "   Position (x,y,z): (0.010, -0.020, 0.500) m"
    assert lines[3].startswith("   Orientation - Yaw: 10.0") and lines[4] == "   Corners: [[10, 20], [30, 20], [30, 40], [10, 40]]"
    assert lines[-1] == "Tag ID 7: Detection OK, but pose estimation failed"
    assert V.process_detections(_FakeDetector(), [], None, lines.append) == (None, [])


@pytest.mark.gpu
def test_webcam_loop_on_a_distorting_camera(tmp_path):
    """640x480 frames of a camera with the five calibration coefficients, through the reference's loop: every tag is found
    and its distance matches the renderer's ground truth."""
    w, h = 640, 480
    K = synth.camera_matrix(w, h, 60.0)
    dist = np.array([-0.12, 0.05, 0.001, -0.0015, 0.01])
    np.savez(tmp_path / "cal.npz", camera_matrix=K, dist_coeffs=dist.reshape(1, 5))
    cp = V.load_camera_calibration(str(tmp_path / "cal.npz"))
    rng = np.random.default_rng(5)
    tags = synth.random_scene(w, h, 4, rng, fov_y_deg=60.0)
    frames, gts = [], []
    for _ in range(3):
        pos, rot = tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(-3, 3, 3))
        f, gt = synth.render_frame(w, h, tags, 18.0, cam_position=pos, cam_rotation_deg=rot, fov_y_deg=60.0, dist=dist)
        frames.append(f); gts.append(gt)
    per_frame, fps = V.run(frames, cp, tag_size=10.0, out=lambda s: None, id_limit=0)
    assert fps > 0 and len(per_frame) == 3
    for rec, gt in zip(per_frame, gts):
        assert sorted(r['id'] for r in rec) == sorted(gt.keys())
        for r in rec:
            assert r['ok'] and abs(r['distance_mm'] / 1000 - np.linalg.norm(gt[r['id']][:3, 3])) < 0.02 * np.linalg.norm(gt[r['id']][:3, 3])
