"""Convert the golden images the reference ships (samples/images/*.ppm, P3 text) into
compact fixtures: tests/golden/ref_<name>.npz holding `rgb` uint8 [H, W, 3].

These are data files of the reference (its only integration fixtures), kept
as fixtures because /root/reference does not exist on the GPU box.  Run here, once:
    python tests/golden/make_reference_goldens.py
"""
import os

import numpy as np

SRC = "/root/reference/samples/images"
DST = os.path.dirname(os.path.abspath(__file__))
NAMES = ["skewed_triangle", "cone", "cylinder_world", "glass_spheres", "teapot"]


def load_p3(path):
    toks = open(path).read().split()
    assert toks[0] == "P3" and toks[3] == "255"
    w, h = int(toks[1]), int(toks[2])
    return np.array(toks[4:], dtype=np.uint8).reshape(h, w, 3)


if __name__ == "__main__":
    for n in NAMES:
        rgb = load_p3(os.path.join(SRC, n + ".ppm"))
        np.savez_compressed(os.path.join(DST, "ref_%s.npz" % n), rgb=rgb)
        print(n, rgb.shape)
