"""ctypes mirror of include/rt_scene.h (the by-value data model of the reference's scene).

Data only: no algorithm lives here.  Field order and types must match the header
exactly; tests/test_abi.py checks the sizes against the C library.
"""
import ctypes as C

import numpy as np

f32 = C.c_float
u32 = C.c_uint32
i32 = C.c_int32

PATTERN_STRIPE, PATTERN_GRADIENT, PATTERN_RING, PATTERN_CHECKER = 0, 1, 2, 3


class Mat4(C.Structure):
    _fields_ = [("m", (f32 * 4) * 4)]

    @staticmethod
    def from_rows(rows):
        out = Mat4()
        a = np.asarray(rows, dtype=np.float32).reshape(4, 4)
        for r in range(4):
            for c in range(4):
                out.m[r][c] = float(a[r, c])
        return out

    def to_numpy(self):
        return np.array([[self.m[r][c] for c in range(4)] for r in range(4)], dtype=np.float32)


def identity():
    return Mat4.from_rows(np.eye(4, dtype=np.float32))


class Pattern(C.Structure):
    _fields_ = [("a", f32 * 3), ("b", f32 * 3), ("transform", Mat4), ("color_map", i32)]


class Material(C.Structure):
    _fields_ = [
        ("color", f32 * 3),
        ("has_pattern", i32),
        ("pattern", Pattern),
        ("ambient", f32),
        ("diffuse", f32),
        ("specular", f32),
        ("shininess", f32),
        ("reflective", f32),
        ("refractive_index", f32),
        ("transparency", f32),
    ]


def default_material():
    """Material defaults, /root/reference src/material.zig:10-37."""
    m = Material()
    m.color[:] = [1.0, 1.0, 1.0]
    m.has_pattern = 0
    m.pattern.transform = identity()
    m.ambient, m.diffuse, m.specular, m.shininess = 0.1, 0.9, 0.9, 200.0
    m.reflective, m.refractive_index, m.transparency = 0.0, 1.0, 0.0
    return m


class Attrs(C.Structure):
    _fields_ = [("transform", Mat4), ("material", Material)]


def default_attrs():
    a = Attrs()
    a.transform = identity()
    a.material = default_material()
    return a


class Sphere(C.Structure):
    _fields_ = [("common_attrs", Attrs)]


class Plane(C.Structure):
    _fields_ = [("common_attrs", Attrs)]


class Cube(C.Structure):
    _fields_ = [("common_attrs", Attrs)]


class Cylinder(C.Structure):
    _fields_ = [("common_attrs", Attrs), ("minimum", f32), ("maximum", f32), ("closed", i32)]


class Cone(C.Structure):
    _fields_ = [("common_attrs", Attrs), ("minimum", f32), ("maximum", f32), ("closed", i32)]


class Triangle(C.Structure):
    _fields_ = [
        ("common_attrs", Attrs),
        ("p0", f32 * 4), ("p1", f32 * 4), ("p2", f32 * 4),
        ("e0", f32 * 4), ("e1", f32 * 4),
        ("normal", f32 * 4),
    ]


class Bounds(C.Structure):
    _fields_ = [("minimum", f32 * 4), ("maximum", f32 * 4)]


class Group(C.Structure):
    pass


Group._fields_ = [
    ("transform", Mat4),
    ("has_bounding_box", i32),
    ("bounding_box", Bounds),
    ("n_spheres", u32), ("spheres", C.POINTER(Sphere)),
    ("n_planes", u32), ("planes", C.POINTER(Plane)),
    ("n_cubes", u32), ("cubes", C.POINTER(Cube)),
    ("n_cylinders", u32), ("cylinders", C.POINTER(Cylinder)),
    ("n_cones", u32), ("cones", C.POINTER(Cone)),
    ("n_triangles", u32), ("triangles", C.POINTER(Triangle)),
    ("n_subgroups", u32), ("subgroups", C.POINTER(Group)),
]


class World(C.Structure):
    _fields_ = [
        ("light_position", f32 * 4),
        ("light_intensity", f32 * 3),
        ("n_spheres", u32), ("spheres", C.POINTER(Sphere)),
        ("n_planes", u32), ("planes", C.POINTER(Plane)),
        ("n_cubes", u32), ("cubes", C.POINTER(Cube)),
        ("n_cylinders", u32), ("cylinders", C.POINTER(Cylinder)),
        ("n_cones", u32), ("cones", C.POINTER(Cone)),
        ("n_triangles", u32), ("triangles", C.POINTER(Triangle)),
        ("n_groups", u32), ("groups", C.POINTER(Group)),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("horizontal_size", u32),
        ("vertical_size", u32),
        ("field_of_view", f32),
        ("half_width", f32),
        ("half_height", f32),
        ("pixel_size", f32),
        ("transform", Mat4),
    ]


class Ray(C.Structure):
    _fields_ = [("origin", f32 * 4), ("direction", f32 * 4)]


class RayCounts(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("secondary", C.c_uint64)]

    @property
    def total(self):
        return self.primary + self.shadow + self.secondary


_LISTS = [
    ("spheres", Sphere), ("planes", Plane), ("cubes", Cube),
    ("cylinders", Cylinder), ("cones", Cone), ("triangles", Triangle),
]


def _set_list(owner, keep, name, ctype, items):
    n = len(items)
    arr = (ctype * max(n, 1))(*items)
    keep.append(arr)
    setattr(owner, "n_" + name, n)
    setattr(owner, name, C.cast(arr, C.POINTER(ctype)))


def make_group(transform=None, bounding_box=None, subgroups=(), keep=None, **shapes):
    """Build an rt_group; `keep` collects the backing arrays (keep it alive)."""
    g = Group()
    g.transform = transform if transform is not None else identity()
    g.has_bounding_box = 0
    if bounding_box is not None:
        g.has_bounding_box = 1
        g.bounding_box = bounding_box
    for name, ctype in _LISTS:
        _set_list(g, keep, name, ctype, list(shapes.get(name, ())))
    _set_list(g, keep, "subgroups", Group, list(subgroups))
    return g


def make_world(light_position, light_intensity, groups=(), keep=None, **shapes):
    """Build an rt_world; returns (world, keepalive list)."""
    if keep is None:
        keep = []
    w = World()
    w.light_position[:] = list(light_position)
    w.light_intensity[:] = list(light_intensity)
    for name, ctype in _LISTS:
        _set_list(w, keep, name, ctype, list(shapes.get(name, ())))
    _set_list(w, keep, "groups", Group, list(groups))
    return w, keep
