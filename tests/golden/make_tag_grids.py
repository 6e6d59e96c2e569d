#!/usr/bin/env python3
"""Extract the 9x9 cell grids of the reference's tag images into tests/golden/tag_grids.json.

Runs only in the build container (reads /root/reference/assets/tags/tag{0..4}.png).  The
fixture is data: for each id the 81 cells (row-major, '.' = white, '#' = black) as rendered
upright.  These five grids are the only code-book known-answers the reference holds.
"""
import json
import os

import numpy as np
from PIL import Image

out = {}
for i in range(5):
    im = Image.open("/root/reference/assets/tags/tag%d.png" % i).convert("RGBA")
    a = np.asarray(im).astype(float)
    g = a[..., 0] * a[..., 3] / 255 + 255 * (1 - a[..., 3] / 255)  # composite over white
    n = a.shape[0] / 9.0
    rows = []
    for r in range(9):
        row = ""
        for c in range(9):
            cell = g[int((r + 0.3) * n):int((r + 0.7) * n), int((c + 0.3) * n):int((c + 0.7) * n)]
            row += "." if cell.mean() > 127 else "#"
        rows.append(row)
    out[str(i)] = rows
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tag_grids.json")
json.dump({"source": "reference assets/tags/tag{0..4}.png (354x354, 9x9 cells)", "grids": out}, open(path, "w"), indent=1)
print(open(path).read()[:300])
