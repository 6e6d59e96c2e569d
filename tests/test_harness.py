import numpy as np
import pytest

from aprilslam_amd import harness, synth


def test_ground_truth_matches_renderer_model():
    """The reference's ground-truth formula (ground_truth.py:48-90) and the renderer model agree for a
    camera that only translates."""
    sc = synth.default_scene()
    gt = harness.GroundTruth(sc["tags"])
    cam = np.array([1.5, -2.0, 3.0])
    for t in sc["tags"]:
        # reference rotation list is [roll(x), pitch(y), yaw(z)] -> Rz Ry Rx == renderer's Rz(rot[2]) Ry(rot[1]) Rx(rot[0])
        assert np.allclose(gt.camera_to_tag(t["id"], cam), synth.camera_from_tag(t["position"], t["rotation"], cam), atol=1e-12)
    T = gt.camera_to_tag(0, cam)
    assert np.allclose(gt.inverse_transform(0, cam) @ T, np.eye(4), atol=1e-12)


def test_euler_and_error_metrics():
    R = harness.euler_to_rotation_matrix([10.0, -20.0, 30.0])
    assert np.allclose(np.degrees(harness.rotation_matrix_to_euler(R)), [10.0, -20.0, 30.0])
    A, B = np.eye(4), np.eye(4)
    B[:3, 3] = [3, 4, 0]
    assert harness.calculate_pose_error(A, B) == (5.0, 0.0)


def test_csv_header_is_the_references():
    # column names of the reference's main CSV (data_logger.py:110-117): a data format, kept verbatim
    assert len(harness.MAIN_CSV_HEADER) == 17
    assert harness.MAIN_CSV_HEADER[:3] == ['Time', 'Number_of_Nodes', 'Average_Distance']
    assert harness.MAIN_CSV_HEADER[-2:] == ['Translation_Difference', 'Rotation_Difference']


@pytest.mark.gpu
def test_headless_run_default_scene(tmp_path):
    import csv
    import logging
    sim = harness.HeadlessSimulation(synth.default_scene(), logging, output_dir=str(tmp_path))
    rng = np.random.default_rng(5)
    for _ in range(6):
        out = sim.step(rng.uniform([-6, -6, -8], [6, 6, 10]))
        assert out is not None and 0 in out["ids"]
    sim.close()
    st = sim.statistics()
    # the reference's own logged run has translation RMSE 1.8 units / rotation RMSE 0.0071 on this scene
    assert st["frames"] == 6 and st["translation_rmse_units"] < 1.8 and st["rotation_fro_rmse"] < 0.05
    rows = list(csv.reader(open(tmp_path / "slam_simulation_data.csv")))
    assert rows[0] == harness.MAIN_CSV_HEADER and len(rows) == 7 and len(rows[1]) == 17
