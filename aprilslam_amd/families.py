"""Tag family tables for the AprilTag detector (host side).

The reference selects the family by name, `apriltag(tag_type)` with
tag_type="tagStandard41h12" (reference src/detection/tag_detector.py:17-18).  The code
table ships as data (aprilslam_amd/data/tagStandard41h12.json, written by
tools/gen_family.py): ids 0..4 are pinned by the reference's assets/tags/tag{0..4}.png,
ids >= 5 are build-defined.
"""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class TagFamily:
    def __init__(self, d):
        self.name = d["name"]
        self.nbits = int(d["nbits"])
        self.h = int(d["h"])
        self.width_at_border = int(d["width_at_border"])
        self.total_width = int(d["total_width"])
        self.reversed_border = bool(d["reversed_border"])
        self.bit_x = np.asarray(d["bit_x"], dtype=np.int32)
        self.bit_y = np.asarray(d["bit_y"], dtype=np.int32)
        self.codes = np.asarray([int(c, 16) for c in d["codes"]], dtype=np.uint64)
        self.pinned_ids = int(d.get("pinned_ids", 0))

    @property
    def ncodes(self):
        return int(self.codes.shape[0])

    def grid(self, tag_id):
        """total_width x total_width uint8 cell grid (1 = white) of tag `tag_id`, as drawn upright."""
        tw, wb = self.total_width, self.width_at_border
        off = (tw - wb) // 2
        g = np.zeros((tw, tw), dtype=np.uint8)
        # border: for a reversed-border family the ring just outside the border is black
        # and the ring just inside is white; everything else defaults to data/0.
        inner0, inner1 = off, off + wb - 1
        if self.reversed_border:
            g[inner0:inner1 + 1, inner0:inner1 + 1] = 1            # white ring (and interior, data overwrites)
            g[inner0 + 1:inner1, inner0 + 1:inner1] = 0
            g[inner0 + 1:inner1, inner0 + 1:inner1] = 0
        else:
            g[:, :] = 1
            g[inner0:inner1 + 1, inner0:inner1 + 1] = 0
        code = int(self.codes[tag_id])
        for i in range(self.nbits):
            bit = (code >> (self.nbits - 1 - i)) & 1
            g[self.bit_y[i] + off, self.bit_x[i] + off] = bit
        return g

    def texture(self, tag_id, cell_px=40):
        """RGB uint8 texture of the upright tag, `cell_px` texels per cell."""
        g = self.grid(tag_id)
        img = np.kron(g, np.ones((cell_px, cell_px), dtype=np.uint8)) * 255
        return np.repeat(img[:, :, None], 3, axis=2)


_CACHE = {}


def get_family(name="tagStandard41h12"):
    if name not in _CACHE:
        path = os.path.join(_DATA, name + ".json")
        if not os.path.exists(path):
            raise ValueError("unknown tag family: %r" % (name,))
        with open(path) as f:
            _CACHE[name] = TagFamily(json.load(f))
    return _CACHE[name]
