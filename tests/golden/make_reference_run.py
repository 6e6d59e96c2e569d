#!/usr/bin/env python3
"""Golden data of the reference's committed run (runs only in the build container).

Writes two fixtures (data only, no reference source text):

  tag_textures.npz            the five tag images the reference's renderer uploads as textures
                              (reference assets/tags/tag{0..4}.png, 354x354, anti-aliased cell edges;
                              renderer.py:160-171 uploads RGBA and draws without blending, so only RGB counts;
                              R == G == B in all five) as one uint8 array [5, 354, 354], rows top to bottom.
  reference_trajectory.json   every camera pose of reference data/csv/slam_clustered_data.csv (570 rows, written by
                              the legacy loop sim.py:372-470) with the pose the reference estimated there:
                              consecutive duplicates folded, order kept (the graph is stateful).
"""
import json
import os

import numpy as np
import pandas as pd
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

tex = []
for i in range(5):
    a = np.asarray(Image.open(os.path.join(REF, "assets/tags/tag%d.png" % i)).convert("RGBA"))
    assert np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2])
    tex.append(a[..., 0].copy())
np.savez_compressed(os.path.join(HERE, "tag_textures.npz"), textures=np.stack(tex).astype(np.uint8))

d = pd.read_csv(os.path.join(REF, "data/csv/slam_clustered_data.csv"))
key = d[["GT_X", "GT_Y", "GT_Z", "GT_Roll", "GT_Pitch", "GT_Yaw"]].apply(tuple, axis=1)
u = d[key != key.shift()]
rows = []
for idx, r in u.iterrows():
    rows.append({
        "csv_line": int(idx) + 2,
        "num_nodes": int(r["Number of Nodes"]),
        "gt_xyz": [float(r.GT_X), float(r.GT_Y), float(r.GT_Z)],
        "gt_rpy": [float(r.GT_Roll), float(r.GT_Pitch), float(r.GT_Yaw)],
        "est_xyz": [float(r.Est_X), float(r.Est_Y), float(r.Est_Z)],
        "est_rpy": [float(r.Est_Roll), float(r.Est_Pitch), float(r.Est_Yaw)],
        "translation_difference": float(r["Translation Difference"]),
        "rotation_difference": float(r["Rotation Difference"]),
    })
out = {
    "source": "reference data/csv/slam_clustered_data.csv, consecutive duplicate poses folded",
    "scene": "reference config/sim_settings.json (default scene); camera rotation is zero in every row",
    "note": "gt_xyz is the camera position in tag 0's frame: camera_position = (x, y, z - 50)",
    "rows": rows,
}
with open(os.path.join(HERE, "reference_trajectory.json"), "w") as f:
    json.dump(out, f, indent=0)
print(len(rows), "poses;", os.path.getsize(os.path.join(HERE, "tag_textures.npz")), "bytes of textures")
