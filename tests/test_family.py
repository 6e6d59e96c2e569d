import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rotate90(w, nbits=41):
    p, l = (nbits - 1, 1) if nbits % 4 == 1 else (nbits, 0)
    w = ((w >> l) << (p // 4 + l)) | (w >> (3 * p // 4 + l) << l) | (w & l)
    return w & ((1 << nbits) - 1)


def test_family_table_consistency(family):
    assert (family.nbits, family.width_at_border, family.total_width, family.reversed_border) == (41, 5, 9, True)
    assert family.ncodes >= 256 and len(set(int(c) for c in family.codes)) == family.ncodes
    inc = open(os.path.join(ROOT, "aprilslam_amd", "csrc", "tag_standard41h12.inc")).read()
    codes = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{11})ULL", inc)]
    assert codes == [int(c) for c in family.codes], "C table and JSON table differ: rerun tools/gen_family.py"
    # layout is a 4-fold rotationally symmetric arrangement plus the centre bit
    cells = set(zip(family.bit_x.tolist(), family.bit_y.tolist()))
    assert len(cells) == 41
    assert {(4 - y, x) for (x, y) in cells} == cells  # 90 degree rotation about (2,2) maps the layout onto itself


def test_min_hamming_distance_12(family):
    codes = [int(c) for c in family.codes]
    rots = []
    for c in codes:
        r = [c]
        for _ in range(3):
            r.append(rotate90(r[-1]))
        rots.append(r)
    allr = np.array([x for r in rots for x in r], dtype=np.uint64)
    owner = np.repeat(np.arange(len(codes)), 4)
    pop = np.array([bin(i).count("1") for i in range(256)])
    for i, c in enumerate(codes):
        d = pop[(allr ^ np.uint64(c)).view(np.uint8).reshape(-1, 8)].sum(axis=1)
        d[(owner == i) & (allr == np.uint64(c))] = 99
        assert d.min() >= 12, (i, int(d.min()))


def test_rotate90_matches_grid_rotation(family):
    """rotate90 on the code word is a quarter turn of the drawn tag."""
    for tid in (0, 3, 77):
        g = family.grid(tid)
        code = int(family.codes[tid])
        gr = np.rot90(g, -1)  # clockwise quarter turn of the image
        bits = 0
        for i in range(41):
            bits = (bits << 1) | int(gr[family.bit_y[i] + 2, family.bit_x[i] + 2])
        assert bits in (rotate90(code), rotate90(rotate90(rotate90(code))))
