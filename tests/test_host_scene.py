"""Host side (no GPU): scene construction through the C++ mirror is bit-identical to the
oracle's own builders; the flattener keeps the reference's insertion order, gates and
composed transforms; group bounds agree with the oracle."""
import ctypes as C
import importlib

import numpy as np
import pytest

import oracle_lib as o
import scene_util as su
from backends import Backend

rta = importlib.import_module("ray-tracer_amd")
st = rta.st


@pytest.mark.parametrize("name", rta.SAMPLES)
def test_sample_scene_bytes_match_oracle_builder(name):
    a, b = rta.Sample(name), o.Sample(name)
    assert su.world_bytes(a.world) == su.world_bytes(b.world)
    assert bytes(a.camera) == bytes(b.camera)


@pytest.mark.parametrize("size", [(160, 90), (33, 77), (1920, 1080)])
def test_camera_matches_oracle(size):
    for name in ("glass_spheres", "cylinder_world"):
        assert bytes(rta.Sample(name, *size).camera) == bytes(o.Sample(name, *size).camera)


def test_ray_for_pixel_matches_oracle():
    s = rta.Sample("sphere_world", 50, 40)
    for px, py in [(0, 0), (49, 39), (25, 20), (3, 37)]:
        a, b = st.Ray(), st.Ray()
        rta.lib().rt_ray_for_pixel(C.byref(s.camera), px, py, C.byref(a))
        o.lib().oracle_ray_for_pixel(C.byref(s.camera), px, py, C.byref(b))
        assert bytes(a) == bytes(b)


def test_flatten_insertion_order_and_nodes():
    w, keep = su.random_world(3, n_shapes=9, groups=2)
    flat = rta.FlatScene(w)
    d = flat.desc
    # world node first, then per group: node, subgroup node
    assert d.n_nodes == 1 + 2 * 2
    assert d.nodes[0].has_bounds == 0 and d.nodes[0].first_shape == 0 and d.nodes[0].skip == 1
    types = [d.shapes[i].type for i in range(d.nodes[0].n_shapes)]
    assert types == sorted(types)  # spheres, planes, cubes, cylinders, cones, triangles (world.zig:39-44)
    nxt = 0
    for i in range(d.n_nodes):
        n = d.nodes[i]
        assert n.first_shape == nxt and n.skip > i
        own = [d.shapes[j].type for j in range(n.first_shape, n.first_shape + n.n_shapes)]
        assert own == sorted(own)
        nxt += n.n_shapes
    assert nxt == d.n_shapes
    # group 0 node is at index 1, its subgroup at 2, skip of group = 3
    assert d.nodes[1].skip == 3 and d.nodes[2].skip == 3 and d.nodes[3].skip == 5


def test_flatten_composes_group_transforms_like_group_zig():
    B = Backend("oracle")
    rng = np.random.default_rng(5)
    keep = []
    inner_shape = su.random_shape(rng, "spheres")
    sub = st.make_group(transform=su.random_transform(rng), keep=keep, spheres=[inner_shape])
    g = st.make_group(transform=su.random_transform(rng), subgroups=[sub], keep=keep, cubes=[su.random_shape(rng, "cubes")])
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], groups=[g], keep=keep)
    d = rta.FlatScene(w).desc
    # member of the subgroup: inverse(mul(mul(G, Gsub), S))  (group.zig:75,103-104)
    composed = B.mul(B.mul(g.transform.to_numpy(), sub.transform.to_numpy()), inner_shape.common_attrs.transform.to_numpy())
    expect = B.inverse(composed)
    got = np.array(d.shapes[1].inverse[:], dtype=np.float32).reshape(4, 4)
    assert got.tobytes() == expect.tobytes()


def test_flatten_marks_identical_shapes_invisible_to_n1():
    a = st.Sphere(st.default_attrs())
    b = st.Sphere(st.default_attrs())
    c = st.Sphere(st.default_attrs())
    c.common_attrs.material.refractive_index = 1.5
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], spheres=[a, b, c])
    d = rta.FlatScene(w).desc
    assert [d.shapes[i].flags & 2 for i in range(3)] == [2, 2, 0]  # a twin pair cancels (intersection.zig:126-137)
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], spheres=[a, b, a])
    d = rta.FlatScene(w).desc
    assert [d.shapes[i].flags & 2 for i in range(3)] == [2, 2, 0]  # three equal: only the last survives


def test_flatten_sends_twins_behind_different_boxes_to_the_per_ray_rule():
    """Equal shapes cancel only for rays that reach both (intersection.zig:126-137 after group.zig:58): twins behind the
    same chain of boxes get the static flag, twins behind different chains carry a class number instead."""
    a = st.Sphere(st.default_attrs())
    keep = []
    box = st.Bounds()
    box.minimum[:] = [-1, -1, -1, 1]
    box.maximum[:] = [1, 1, 1, 1]
    small = st.Bounds()
    small.minimum[:] = [-0.5, -0.5, -0.5, 1]
    small.maximum[:] = [0.5, 0.5, 0.5, 1]
    g_same = st.make_group(bounding_box=box, keep=keep, spheres=[a, a])           # same chain: static rule
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], groups=[g_same], keep=keep)
    d = rta.FlatScene(w).desc
    assert [d.shapes[i].flags & 2 for i in range(2)] == [2, 2] and [d.shapes[i].reserved for i in range(2)] == [0, 0]
    g1 = st.make_group(bounding_box=box, keep=keep, spheres=[a])
    g2 = st.make_group(bounding_box=small, keep=keep, spheres=[a])
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], groups=[g1, g2], keep=keep, spheres=[a])
    d = rta.FlatScene(w).desc
    assert [d.shapes[i].flags & 2 for i in range(3)] == [0, 0, 0] and [d.shapes[i].reserved for i in range(3)] == [1, 1, 1]
    # an unbounded group adds no gate: its member shares the world-level twin's (empty) chain
    g3 = st.make_group(keep=keep, spheres=[a])
    w, keep = st.make_world([0, 5, -5, 1], [1, 1, 1], groups=[g3], keep=keep, spheres=[a])
    d = rta.FlatScene(w).desc
    assert [d.shapes[i].flags & 2 for i in range(2)] == [2, 2] and [d.shapes[i].reserved for i in range(2)] == [0, 0]


def test_materials_and_patterns_are_deduplicated():
    s = rta.Sample("sphere_world")
    d = rta.FlatScene(s.world).desc
    assert d.n_shapes == 6 and d.n_materials == 4 and d.n_patterns == 0
    s = rta.Sample("glass_spheres")
    d = rta.FlatScene(s.world).desc
    assert d.n_shapes == 3 and d.n_materials == 3 and d.n_patterns == 1


@pytest.mark.parametrize("seed", range(6))
def test_bound_group_matches_oracle(seed):
    w, keep = su.random_world(100 + seed, n_shapes=6, groups=2)
    P, O = Backend("product"), Backend("oracle")
    for i in range(w.n_groups):
        a, b = P.bound_group(w.groups[i]), O.bound_group(w.groups[i])
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_libm_restatements_agree():
    L, O = rta.lib(), o.lib()
    rng = np.random.default_rng(1)
    for x in rng.uniform(-7, 7, 300):
        x = float(np.float32(x))
        assert L.rt_sin(x) == O.oracle_sin(x) and L.rt_cos(x) == O.oracle_cos(x) and L.rt_tan(x) == O.oracle_tan(x)
    for x, y in zip(rng.uniform(0, 1.2, 300), rng.choice([5.0, 10.0, 200.0, 300.0, 2.5, 0.5, 33.3], 300)):
        assert L.rt_pow(float(np.float32(x)), float(y)) == O.oracle_pow(float(np.float32(x)), float(y))
    # against libm: the Cephes/Go ports are within a few ulp
    for x in rng.uniform(-3, 3, 50):
        assert abs(L.rt_sin(float(x)) - np.sin(x)) < 3e-7 and abs(L.rt_cos(float(x)) - np.cos(x)) < 3e-7
    for x in rng.uniform(0.01, 1, 50):
        assert abs(L.rt_pow(float(x), 200.0) - x ** 200) <= 2e-5 * x ** 200 + 1e-38
