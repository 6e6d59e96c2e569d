"""Starting values for the pose-graph back-end (aprilslam_amd.map_init): a chain that starts from corrupted
single-view poses is repaired by letting every camera / tag pick the most consistent pose its observations imply."""
import numpy as np
from scipy.spatial.transform import Rotation as R

from aprilslam_amd import map_init, synth


def _scene(rng, L=8, P=6, corrupt=0.3):
    def rnd():
        T = np.eye(4)
        T[:3, :3] = R.from_rotvec(rng.normal(size=3) * 0.3).as_matrix()
        return T
    tag = [np.eye(4)] + [rnd() for _ in range(L - 1)]
    for j in range(1, L):
        tag[j][:3, 3] = rng.uniform(-30, 30, 3) * [1, 1, 0.1]
    cam = []
    for _ in range(P):
        T = rnd()
        T[:3, 3] = rng.uniform(-5, 5, 3) + [0, 0, -80]
        cam.append(T)
    K = synth.camera_matrix(1280, 720)
    X = map_init._corners_obj(10.0)
    frames, oc, ot, oT, oC = [], [], [], [], []
    for f in range(P):
        fr = []
        for j in range(L):
            if f == 0 and j == 0:
                continue  # the world tag is not in the first frame
            Tct = np.linalg.inv(cam[f]) @ tag[j]
            p = (Tct[:3] @ X.T).T
            c = np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], 1)
            Tn = Tct.copy()
            if rng.random() < corrupt:
                Tn[:3, :3] = Tn[:3, :3] @ R.from_rotvec([0.5, 0, 0]).as_matrix()
            fr.append((j, Tn, c)); oc.append(f); ot.append(j); oT.append(Tn); oC.append(c)
        frames.append(fr)
    return np.array(tag), np.array(cam), K, frames, oc, ot, oT, oC


def test_chain_places_everything_and_is_exact_on_clean_poses():
    tag, cam, K, frames, *_ = _scene(np.random.default_rng(3), corrupt=0.0)
    world, tags, cams = map_init.chain_initial_map(frames)
    assert world == 0 and len(tags) == len(tag) and all(c is not None for c in cams)
    assert np.allclose(np.array([tags[j] for j in range(len(tag))]), tag, atol=1e-9) and np.allclose(np.array(cams), cam, atol=1e-9)
    assert map_init.chain_initial_map([[], []]) == (-1, {}, [None, None])


def test_reseed_repairs_a_chain_through_bad_single_view_poses():
    tag, cam, K, frames, oc, ot, oT, oC = _scene(np.random.default_rng(1))
    world, tags, cams = map_init.chain_initial_map(frames)
    t0, c0 = np.array([tags[j] for j in range(len(tag))]), np.array(cams)
    assert np.abs(t0 - tag).max() > 1.0  # the chain went through corrupted observations
    c1, t1 = map_init.reseed_poses(c0, t0, oc, ot, oT, oC, K, 10.0, fixed_tag=0, sweeps=3)
    assert np.allclose(t1, tag, atol=1e-9) and np.allclose(c1, cam, atol=1e-9)
    assert np.array_equal(t1[0], np.eye(4))


def test_flip_test_recovers_mirrored_tags():
    tag, cam, K, frames, oc, ot, oT, oC = _scene(np.random.default_rng(1), corrupt=0.0)
    bad = tag.copy()
    for j in (2, 5):
        bad[j] = cam[1] @ map_init.mirrored_pose(np.linalg.inv(cam[1]) @ tag[j])
    assert np.abs(bad - tag).max() > 0.1
    # the mirrored pose reprojects almost as well in the view it was built in
    c_bad = map_init.reprojection_cost(np.linalg.inv(cam[1]) @ bad[2], np.asarray(oC)[[i for i in range(len(oc)) if oc[i] == 1 and ot[i] == 2][0]], K, 10.0)
    assert c_bad < 25.0
    fixed, flipped = map_init.flip_test_tags(cam, bad, oc, ot, oC, K, 10.0, fixed_tag=0)
    assert flipped == [2, 5] and np.allclose(fixed, tag, atol=1e-6)
