#!/usr/bin/env python3
"""Deterministic generator of a Utah-style teapot as an OBJ file restricted to the subset the
reference's parser accepts (`v`, `f`, `g`; src/obj_file.zig:99-138).

The reference's samples/teapot.zig opens "teapot.obj" from the working directory, but no .obj
ships with it, so the mesh is synthesised: y-up, base at y = 0, x in about
[-3.0, 3.4], y in [0, 3.15] — the framing of samples/images/teapot.ppm.  Body and lid are
surfaces of revolution of the classic teapot profile curves (cubic Beziers), handle and spout
are tubes swept along cubic Beziers.  Faces are quads (the parser fan-triangulates them,
src/obj_file.zig:156-164) except at the poles.

    python samples/make_teapot_obj.py [level] > teapot.obj      # level 1: ~6k triangles after fanning
"""
import math
import sys


def bezier(p, t):
    a = [(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t * t * (1 - t), t ** 3]
    return tuple(sum(a[i] * p[i][k] for i in range(4)) for k in range(len(p[0])))


def bezier_tangent(p, t):
    a = [-3 * (1 - t) ** 2, 3 * (1 - t) ** 2 - 6 * t * (1 - t), 6 * t * (1 - t) - 3 * t * t, 3 * t * t]
    return tuple(sum(a[i] * p[i][k] for i in range(4)) for k in range(len(p[0])))


class Mesh:
    def __init__(self):
        self.lines = []
        self.nv = 0
        self.nf = 0

    def v(self, x, y, z):
        self.lines.append("v %.6f %.6f %.6f" % (x + 0.0, y + 0.0, z + 0.0))
        self.nv += 1
        return self.nv  # OBJ indices start at 1

    def f(self, *idx):
        idx = [i for k, i in enumerate(idx) if i != idx[k - 1]]  # collapse pole duplicates
        if len(idx) >= 3:
            self.lines.append("f " + " ".join(str(i) for i in idx))
            self.nf += 1

    def g(self, name):
        self.lines.append("g " + name)


def lathe(mesh, profile, segments):
    """Surface of revolution of a list of (radius, height) points around the y axis."""
    rings = []
    for r, y in profile:
        if r <= 1e-9:
            rings.append([mesh.v(0.0, y, 0.0)] * segments)
        else:
            rings.append([mesh.v(r * math.cos(2 * math.pi * k / segments), y, r * math.sin(2 * math.pi * k / segments))
                          for k in range(segments)])
    for a, b in zip(rings[:-1], rings[1:]):
        for k in range(segments):
            k2 = (k + 1) % segments
            mesh.f(a[k], a[k2], b[k2], b[k])


def tube(mesh, path, radius0, radius1, steps, sides, squash=1.0):
    """Tube swept along a cubic Bezier in the z = 0 plane; radius tapers linearly."""
    rings = []
    for i in range(steps + 1):
        t = i / steps
        cx, cy = bezier(path, t)
        tx, ty = bezier_tangent(path, t)
        n = math.hypot(tx, ty)
        nx, ny = -ty / n, tx / n  # in-plane normal
        r = radius0 + (radius1 - radius0) * t
        ring = []
        for k in range(sides):
            a = 2 * math.pi * k / sides
            ring.append(mesh.v(cx + nx * r * math.cos(a), cy + ny * r * math.cos(a), r * squash * math.sin(a)))
        rings.append(ring)
    for a, b in zip(rings[:-1], rings[1:]):
        for k in range(sides):
            k2 = (k + 1) % sides
            mesh.f(a[k], a[k2], b[k2], b[k])


def sample_curves(curves, n):
    pts = []
    for c in curves:
        for i in range(n + 1):
            p = bezier(c, i / n)
            if not pts or abs(p[0] - pts[-1][0]) + abs(p[1] - pts[-1][1]) > 1e-9:
                pts.append(p)
    return pts


def teapot(level=1):
    seg = 16 * level       # around the axis
    prof = 4 * level       # per profile curve
    mesh = Mesh()
    # (radius, height) control points of the classic profile
    rim = [(1.4, 2.25), (1.3375, 2.38125), (1.4375, 2.38125), (1.5, 2.25)]
    body = [(1.5, 2.25), (1.75, 1.725), (2.0, 1.2), (2.0, 0.75)]
    lower = [(2.0, 0.75), (2.0, 0.3), (1.5, 0.075), (1.5, 0.0)]
    bottom = [(1.5, 0.0), (1.0, 0.0), (0.5, 0.0), (0.0, 0.0)]
    knob = [(0.0, 3.15), (0.8, 3.15), (0.0, 2.7), (0.2, 2.55)]
    lid = [(0.2, 2.55), (0.4, 2.4), (1.3, 2.4), (1.3, 2.25)]
    mesh.g("body")
    lathe(mesh, sample_curves([rim, body, lower, bottom], prof), seg)
    mesh.g("lid")
    lathe(mesh, sample_curves([knob, lid], prof), seg)
    mesh.g("handle")
    tube(mesh, [(-1.55, 2.0), (-3.15, 2.15), (-3.05, 0.75), (-1.9, 0.55)], 0.16, 0.16, 10 * level, 8 * level, squash=1.3)
    mesh.g("spout")
    tube(mesh, [(1.7, 0.75), (2.9, 0.85), (2.35, 2.0), (3.3, 2.3)], 0.36, 0.13, 10 * level, 8 * level, squash=1.15)
    return mesh


def obj_text(level=1):
    """The OBJ text of the level-`level` tessellation (what this script prints)."""
    return "\n".join(teapot(level).lines) + "\n"


if __name__ == "__main__":
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    m = teapot(level)
    sys.stdout.write("\n".join(m.lines) + "\n")
    sys.stderr.write("teapot level %d: %d vertices, %d faces\n" % (level, m.nv, m.nf))
