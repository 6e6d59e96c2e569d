"""Synthetic frame source: the step BEFORE the hot path.

The reference renders its frames with fixed-function OpenGL through pygame
(reference src/simulation/renderer.py:197-274); neither is available here or on the GPU
box, so this module reproduces the same image-formation model in NumPy:

  * background clear colour (0.5, 0, 0.5) -> RGB (128, 0, 128)      renderer.py:206
  * each tag is a planar quad of side tag_size_outer centred on its pose, texture
    coordinates (0,0) at the (-h,-h) vertex, image drawn upright    renderer.py:243-249
  * tag pose = T(position) Rz(roll) Ry(yaw) Rx(pitch)               renderer.py:232-237
  * view = Rz(-roll) Rx(-pitch) Ry(-yaw) T(-position)               renderer.py:190-195
  * perspective: fov_y, square pixels, principal point at the image centre
    (fx = fy = 0.5 H / tan(fov_y/2), cx = W/2, cy = H/2)            simulation_engine.py:124-126
  * read-back flipped to top-left origin and converted to BGR       renderer.py:263-272
  * GL_LINEAR texture filtering, one sample per pixel centre, painter's order by z.

It also returns the analytic ground truth of ground_truth.py:48-90 (camera<-tag 4x4 in
the OpenCV camera frame: x right, y down, z forward).
"""
import numpy as np

from .families import get_family


def camera_matrix(width, height, fov_y_deg=45.0):
    f = 0.5 * height / np.tan(0.5 * np.radians(fov_y_deg))
    return np.array([[f, 0, 0.5 * width], [0, f, 0.5 * height], [0, 0, 1.0]])


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


_FLIP = np.diag([1.0, -1.0, -1.0])  # OpenGL camera (y up, -z forward) -> OpenCV camera


def tag_model_matrix(position, rotation_deg):
    """GL model matrix of a tag: rotation list is [pitch(x), yaw(y), roll(z)] in degrees."""
    p, y, r = np.radians(rotation_deg)
    M = np.eye(4)
    M[:3, :3] = _rz(r) @ _ry(y) @ _rx(p)
    M[:3, 3] = position
    return M


def view_matrix(cam_position, cam_rotation_deg):
    """GL view matrix: camera rotation list is [pitch, yaw, roll] in degrees."""
    p, y, r = np.radians(cam_rotation_deg)
    V = np.eye(4)
    V[:3, :3] = _rz(-r) @ _rx(-p) @ _ry(-y)
    V[:3, 3] = V[:3, :3] @ (-np.asarray(cam_position, dtype=np.float64))
    return V


def camera_from_tag(tag_position, tag_rotation_deg, cam_position=(0, 0, 0), cam_rotation_deg=(0, 0, 0)):
    """4x4 camera<-tag transform in the OpenCV camera frame (what PnP should return)."""
    M = view_matrix(cam_position, cam_rotation_deg) @ tag_model_matrix(tag_position, tag_rotation_deg)
    T = np.eye(4)
    T[:3, :3] = _FLIP @ M[:3, :3]
    T[:3, 3] = _FLIP @ M[:3, 3]
    return T


def _bilinear(tex, u, v):
    """GL_LINEAR, clamp-to-edge: tex (h, w, 3) uint8; u, v in texel units (texel centres at +0.5)."""
    h, w = tex.shape[:2]
    x = u - 0.5
    y = v - 0.5
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx = (x - x0)[..., None]
    fy = (y - y0)[..., None]
    x0c = np.clip(x0, 0, w - 1)
    x1c = np.clip(x0 + 1, 0, w - 1)
    y0c = np.clip(y0, 0, h - 1)
    y1c = np.clip(y0 + 1, 0, h - 1)
    t = tex.astype(np.float64)
    out = (t[y0c, x0c] * (1 - fx) * (1 - fy) + t[y0c, x1c] * fx * (1 - fy) +
           t[y1c, x0c] * (1 - fx) * fy + t[y1c, x1c] * fx * fy)
    return out


UNDISTORT_ITERS = 8  # fixed-point iterations of the per-pixel inverse lens distortion (renderer side)


def undistort_normalized(xd, yd, dist):
    """Inverse of the Brown-Conrady model (k1, k2, p1, p2, k3) on normalised image coordinates by fixed-point iteration
    (the scheme cv2.undistortPoints uses), UNDISTORT_ITERS steps.  Works on arrays."""
    k1, k2, p1, p2, k3 = (list(np.asarray(dist, dtype=np.float64).ravel()) + [0.0] * 5)[:5]
    x, y = xd, yd
    for _ in range(UNDISTORT_ITERS):
        r2 = x * x + y * y
        icdist = 1 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) * icdist
        y = (yd - dy) * icdist
    return x, y


def frame_geometry(width, height, tags, tag_size_outer, cam_position=(0, 0, 0), cam_rotation_deg=(0, 0, 0), fov_y_deg=45.0,
                   dist=None):
    """What a renderer needs for one frame: per visible tag (painter's order, far to near) the inverse of the plane
    homography -- pixel -> tag plane without `dist`, normalised image coordinates -> tag plane with it -- and the pixel
    bounding box of the tag.  Returns (planes, gt): planes = list of dicts {"id", "Hi" 3x3, "bbox" (x0, x1, y0, y1)},
    gt[id] = camera<-tag 4x4."""
    K = camera_matrix(width, height, fov_y_deg)
    half = 0.5 * tag_size_outer
    gt, order = {}, []
    for tag in tags:
        T = camera_from_tag(tag["position"], tag["rotation"], cam_position, cam_rotation_deg)
        gt[int(tag["id"])] = T
        order.append((T[2, 3], tag, T))
    order.sort(key=lambda e: -e[0])  # far to near
    planes = []
    for _, tag, T in order:
        Hn = np.column_stack([T[:3, 0], T[:3, 1], T[:3, 3]])
        Hm = K @ Hn  # homography tag plane (X, Y, 1) -> pixel
        corners = np.array([[-half, -half, 1], [half, -half, 1], [half, half, 1], [-half, half, 1]]).T
        pc = Hm @ corners
        if np.any(pc[2] <= 1e-9):
            continue  # crosses the camera plane: skip (never happens in the generated scenes)
        px = pc[0] / pc[2]
        py = pc[1] / pc[2]
        pad = 0
        if dist is not None:
            # the bounding box is that of the undistorted projection: leave room for the lens to move the outline
            pad = 2 + int(0.08 * max(width, height))
        x0 = max(int(np.floor(px.min())) - pad, 0)
        x1 = min(int(np.ceil(px.max())) + 1 + pad, width)
        y0 = max(int(np.floor(py.min())) - pad, 0)
        y1 = min(int(np.ceil(py.max())) + 1 + pad, height)
        if x0 >= x1 or y0 >= y1:
            continue
        planes.append({"id": int(tag["id"]), "Hi": np.linalg.inv(Hm if dist is None else Hn), "bbox": (x0, x1, y0, y1)})
    return planes, gt


def render_frame(width, height, tags, tag_size_outer, cam_position=(0, 0, 0), cam_rotation_deg=(0, 0, 0),
                 fov_y_deg=45.0, family="tagStandard41h12", cell_px=40, noise_sigma=0.0, rng=None, textures=None, dist=None):
    """Render one H x W x 3 BGR uint8 frame.

    tags: iterable of dicts {"id", "position" [x,y,z], "rotation" [pitch,yaw,roll] deg} (the
    reference's sim_settings.json schema).  Returns (frame, gt) with gt[id] = camera<-tag 4x4.
    dist = (k1, k2, p1, p2[, k3]): the frame a camera with that lens distortion would deliver (the reference's second
    caller, a calibrated webcam: video_detection.py:209-296, calibrate.py:71-75); None = the simulator's ideal pinhole.
    """
    fam = get_family(family)
    K = camera_matrix(width, height, fov_y_deg)
    frame = np.empty((height, width, 3), dtype=np.uint8)
    frame[:] = (128, 0, 128)  # BGR of RGB (128, 0, 128)
    half = 0.5 * tag_size_outer
    planes, gt = frame_geometry(width, height, tags, tag_size_outer, cam_position, cam_rotation_deg, fov_y_deg, dist)
    for pl in planes:
        x0, x1, y0, y1 = pl["bbox"]
        Hi = pl["Hi"]
        xs, ys = np.meshgrid(np.arange(x0, x1) + 0.5, np.arange(y0, y1) + 0.5)
        if dist is not None:
            xs, ys = undistort_normalized((xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], dist)
        q0 = Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]
        q1 = Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]
        q2 = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
        X = q0 / q2
        Y = q1 / q2
        inside = (X >= -half) & (X <= half) & (Y >= -half) & (Y <= half)
        if dist is not None:
            inside &= q2 > 0
        if not inside.any():
            continue
        tex = textures[pl["id"]] if textures is not None else fam.texture(pl["id"], cell_px)
        th, tw = tex.shape[:2]
        u = (X + half) / (2 * half) * tw
        v = (1.0 - (Y + half) / (2 * half)) * th  # image row 0 is the top (t = 1)
        rgb = _bilinear(tex, u[inside], v[inside])
        sub = frame[y0:y1, x0:x1]
        sub[inside] = np.clip(np.floor(rgb[:, ::-1] + 0.5), 0, 255).astype(np.uint8)
    if noise_sigma > 0:
        rng = rng or np.random.default_rng(0)
        frame = np.clip(frame.astype(np.float64) + rng.normal(0, noise_sigma, frame.shape), 0, 255).astype(np.uint8)
    return frame, gt


def render_planes(width, height, tags, tag_size_outer, cameras, fov_y_deg=45.0, dist=None, tex_index=None):
    """The per-frame plane lists asl_render_frames_device consumes, for a list of cameras [(position, rotation_deg), ...]:
    (planes (n_frames, max_planes) array of _lib.PLANE_DTYPE, gts list of {id: camera<-tag}).  tex_index maps a tag id to
    its texture (default: the id itself)."""
    from ._lib import PLANE_DTYPE
    max_planes = max(1, len(tags))
    planes = np.zeros((len(cameras), max_planes), dtype=PLANE_DTYPE)
    planes["tex"] = -1
    gts = []
    for f, (pos, rot) in enumerate(cameras):
        pls, gt = frame_geometry(width, height, tags, tag_size_outer, pos, rot, fov_y_deg, dist)
        gts.append(gt)
        for k, pl in enumerate(pls):
            planes["Hi"][f, k] = pl["Hi"].ravel()
            planes["bbox"][f, k] = pl["bbox"]
            planes["tex"][f, k] = pl["id"] if tex_index is None else tex_index[pl["id"]]
    return planes, gts


def gray_textures(ids, family="tagStandard41h12", cell_px=40, textures=None):
    """(n, th, tw) uint8 stack of the tag textures for ids 0..max(ids) (rows top to bottom), as the device renderer wants them."""
    fam = get_family(family)
    n = max(ids) + 1
    first = textures[ids[0]] if textures is not None else fam.texture(ids[0], cell_px)
    out = np.zeros((n,) + first.shape[:2], dtype=np.uint8)
    for i in ids:
        t = textures[i] if textures is not None else fam.texture(i, cell_px)
        out[i] = t[:, :, 0]
    return out


def default_scene():
    """The reference's default scene (reference config/sim_settings.json:1-43), as data."""
    return {
        "display_width": 1000, "display_height": 1000, "fov_y": 45, "size_scale": 2,
        "tag_size_inner": 5, "tag_size_outer": 9, "actual_size_in_mm": 55.6,
        "tags": [
            {"id": 0, "position": [0, 0, -50], "rotation": [0, 0, 0]},
            {"id": 1, "position": [-30, 0, -120], "rotation": [0, 0, 0]},
            {"id": 2, "position": [25, 15, -85], "rotation": [0, 0, 0]},
            {"id": 3, "position": [55, -10, -75], "rotation": [0, 20, 10]},
            {"id": 4, "position": [80, 5, -65], "rotation": [0, 20, 0]},
        ],
    }


def random_scene(width, height, ntags, rng, fov_y_deg=45.0, tag_size_outer=18.0, min_edge_px=40.0,
                 max_angle_deg=25.0, max_tries=20000):
    """Seeded scene of `ntags` non-overlapping tags on a jittered grid.

    Tags get ids 0..ntags-1, depth chosen so the projected edge is >= min_edge_px, and
    yaw/pitch/roll uniform in +-max_angle_deg.  Positions are GL world coordinates for a camera
    at the origin looking down -z.
    """
    K = camera_matrix(width, height, fov_y_deg)
    f = K[0, 0]
    cols = int(np.ceil(np.sqrt(ntags * width / height)))
    rows = int(np.ceil(ntags / cols))
    cell_w, cell_h = width / cols, height / rows
    # projected outer edge must fit inside a grid cell with margin, and be >= min_edge_px
    max_edge = 0.62 * min(cell_w, cell_h)
    lo_edge = max(min_edge_px * 9.0 / 5.0, 0.45 * max_edge)  # min_edge applies to the 5-cell border
    if lo_edge > max_edge:
        lo_edge = max_edge
    tags = []
    cells = [(r, c) for r in range(rows) for c in range(cols)]
    rng.shuffle(cells)
    for i in range(ntags):
        r, c = cells[i]
        edge = rng.uniform(lo_edge, max_edge)
        z = f * tag_size_outer / edge
        jitter = 0.5 * (min(cell_w, cell_h) - edge / 0.62 * 0.62) * 0.4
        u = (c + 0.5) * cell_w + rng.uniform(-jitter, jitter)
        v = (r + 0.5) * cell_h + rng.uniform(-jitter, jitter)
        x = (u - K[0, 2]) / f * z
        y = -(v - K[1, 2]) / f * z
        rot = rng.uniform(-max_angle_deg, max_angle_deg, size=3)
        tags.append({"id": i, "position": [float(x), float(y), float(-z)], "rotation": [float(a) for a in rot]})
    return tags
