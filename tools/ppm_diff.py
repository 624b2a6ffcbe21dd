#!/usr/bin/env python3
"""Compare two PPM images (P3 as the reference writes them, or P6) under the +-1 LSB rule.

    python tools/ppm_diff.py a.ppm b.ppm [--frac 0.999] [--mask x0,y0,x1,y1]

--mask excludes the box [x0, x1) x [y0, y1) from the comparison (SURVEY.md §8c masks the
glass cylinder's footprint of cylinder_world.ppm that way).  Exit status 0 when the images
agree within 1 LSB on at least --frac of the compared pixels, 1 otherwise.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ppm = importlib.import_module("ray-tracer_amd.ppm")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--frac", type=float, default=0.999)
    ap.add_argument("--mask", default=None)
    args = ap.parse_args()
    a, b = ppm.read(args.a), ppm.read(args.b)
    mask = None
    if args.mask:
        x0, y0, x1, y1 = (int(v) for v in args.mask.split(","))
        mask = np.ones(a.shape[:2], dtype=bool)
        mask[y0:y1, x0:x1] = False
    d = ppm.diff(a, b, mask)
    print("%d px compared: exact %.4f %%, within 1 LSB %.4f %%, max |diff| %d" % (
        d["pixels"], 100 * d["frac_exact"], 100 * d["frac_within1"], d["max_abs"]))
    return 0 if d["frac_within1"] >= args.frac else 1


if __name__ == "__main__":
    sys.exit(main())
