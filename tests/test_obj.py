"""OBJ ingest (src/obj_file.zig) — host only.  The reference has no tests for its parser, so
these pin the behaviour read off its source: v/f/g only, fan triangulation with the degenerate
first triangle, named groups -> identity subgroups, error cases."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

rta = importlib.import_module("ray-tracer_amd")
st = rta.st
import scene_util as su  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRI = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\nf 1 2 3\nf 1 3 4\n"


def tri_points(t):
    return [tuple(t.p0[:3]), tuple(t.p1[:3]), tuple(t.p2[:3])]


def test_triangle_faces():
    o = rta.ObjFile(TRI, bound=False)
    assert (o.n_vertices, o.n_faces, o.n_groups) == (4, 2, 0)
    g = o.group
    assert g.n_triangles == 2 and g.n_subgroups == 0 and g.has_bounding_box == 0
    assert tri_points(g.triangles[0]) == [(-1, 1, 0), (-1, 0, 0), (1, 0, 0)]
    assert tri_points(g.triangles[1]) == [(-1, 1, 0), (1, 0, 0), (1, 1, 0)]
    # Triangle.init precomputes e0, e1 and normal = normalize(cross(e1, e0)) (triangle.zig:19-30)
    t = g.triangles[0]
    assert tuple(t.e0[:]) == (0, -1, 0, 0) and tuple(t.e1[:]) == (2, -1, 0, 0)
    assert tuple(t.normal[:3]) == (0, 0, -1)


def test_polygon_fan_includes_the_reference_degenerate_triangle():
    text = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nf 1 2 3 4 5\n"
    g = rta.ObjFile(text, bound=False).group
    # obj_file.zig:158-163 walks indices[1..] with an index starting at 0: (v1,v1,v2) comes first
    assert g.n_triangles == 4
    assert tri_points(g.triangles[0]) == [(-1, 1, 0), (-1, 1, 0), (-1, 0, 0)]
    assert tri_points(g.triangles[1]) == [(-1, 1, 0), (-1, 0, 0), (1, 0, 0)]
    assert tri_points(g.triangles[3]) == [(-1, 1, 0), (1, 1, 0), (0, 2, 0)]


def test_named_groups_become_identity_subgroups():
    text = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\ng FirstGroup\nf 1 3 4\n"
    o = rta.ObjFile(text, bound=True)
    g = o.group
    assert g.n_triangles == 1 and g.n_subgroups == 2 and o.n_groups == 2
    assert g.subgroups[0].n_triangles == 2 and g.subgroups[1].n_triangles == 1
    assert np.array_equal(g.subgroups[0].transform.to_numpy(), np.eye(4, dtype=np.float32))
    assert g.subgroups[0].has_bounding_box == 0 and g.has_bounding_box == 1  # only the top group is bounded (teapot.zig:19)
    assert tuple(g.bounding_box.minimum[:3]) == (-1, 0, 0) and tuple(g.bounding_box.maximum[:3]) == (1, 1, 0)


@pytest.mark.parametrize("text,code", [
    ("vn 0 0 1\n", -10),            # command longer than one character
    ("x 1 2 3\n", -10),             # unknown command
    ("v 1 2\n", -10),               # missing literal
    ("v 1 2 3 4\n", -10),           # unused literal
    ("g a b\n", -10),
    ("v 1 2 abc\n", -11),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2/2 3/3\n", -11),  # only plain indices are understood
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", -12),
    ("v 0 0 0\nv 1 0 0\nf 1 2\n", -12),
])
def test_parse_errors(text, code):
    h = C.c_void_p()
    assert rta.lib().rt_obj_parse(text.encode(), len(text), C.byref(h)) == code
    assert not h


def test_blank_lines_and_extra_spaces_are_skipped():
    o = rta.ObjFile("\n\nv  1   2 3\n   \nv 4 5 6\nv 7 8 9\n\nf 1  2 3\n", bound=False)
    assert (o.n_vertices, o.n_faces) == (3, 1)


def test_committed_teapot_is_what_the_generator_produces():
    text = open(os.path.join(ROOT, "samples", "teapot.obj")).read()
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "samples", "make_teapot_obj.py"), "1"],
                         capture_output=True, text=True, check=True).stdout
    assert gen == text
    o = rta.ObjFile(text)
    assert o.n_vertices == 562 and o.n_groups == 4 and o.n_faces == 1568
    b = o.group.bounding_box
    assert -3.2 < b.minimum[0] < -2.8 and 3.2 < b.maximum[0] < 3.5 and b.minimum[1] == 0 and abs(b.maximum[1] - 3.15) < 1e-6


def test_teapot_scene_flattens_in_insertion_order():
    text = open(os.path.join(ROOT, "samples", "teapot.obj")).read()
    world, cam, keep = rta.teapot_scene(text, 64, 64)
    d = rta.FlatScene(world).desc
    assert d.n_nodes == 1 + 1 + 4 and d.n_shapes == 1568
    assert d.nodes[1].has_bounds == 1 and d.nodes[1].n_shapes == 0 and d.nodes[1].skip == 6
    assert all(d.nodes[i].has_bounds == 0 for i in range(2, 6))


def test_cpp_teapot_sample_equals_the_assembled_scene(tmp_path, monkeypatch):
    # samples/teapot.zig through the C++ mirror (rt::obj::parseObjFile -> toShapeGroup -> bound)
    text = open(os.path.join(ROOT, "samples", "teapot.obj")).read()
    world, cam, keep = rta.teapot_scene(text, 64, 48)
    s = rta.Sample("teapot", 64, 48, obj_text=text)
    assert su.world_bytes(s.world) == su.world_bytes(world)
    assert bytes(s.camera) == bytes(cam)
    # like the reference's program, the plain name reads ./teapot.obj
    with pytest.raises(KeyError):
        monkeypatch.chdir(tmp_path)
        rta.Sample("teapot", 64, 48)
    (tmp_path / "teapot.obj").write_text(text)
    s2 = rta.Sample("teapot", 64, 48)
    assert su.world_bytes(s2.world) == su.world_bytes(world)
    shipped = rta.Sample("teapot", obj_text=text)
    assert (shipped.camera.horizontal_size, shipped.camera.vertical_size) == (256, 256)


def test_cpp_sample_rejects_bad_obj_and_unknown_names():
    with pytest.raises(KeyError):
        rta.Sample("teapot", obj_text="v 1 2\n")
    with pytest.raises(KeyError):
        rta.Sample("no_such_scene", obj_text="v 1 2 3\n")


def test_composite_scene_is_frozen():
    # BASELINE.json configs[4]: teapot group + glass pair + red sphere + mirror floor
    text = open(os.path.join(ROOT, "samples", "teapot.obj")).read()
    world, cam, keep = rta.composite_scene(text, 160, 90)
    assert (world.n_spheres, world.n_planes, world.n_groups) == (3, 1, 1)
    assert (cam.horizontal_size, cam.vertical_size) == (160, 90)
    assert list(world.light_position) == [-5, 10, -10, 1]
    glass = rta.Sample("glass_spheres")
    t = world.spheres[1].common_attrs.transform.to_numpy()
    assert np.array_equal(t, np.array([[0.5, 0, 0, 4.2], [0, 0.5, 0, 1.0], [0, 0, 0.5, -1.5], [0, 0, 0, 1]], dtype=np.float32))
    assert bytes(world.spheres[0].common_attrs.material) == bytes(glass.world.spheres[0].common_attrs.material)
    assert world.planes[0].common_attrs.material.reflective == np.float32(0.25)
    d = rta.FlatScene(world).desc
    assert d.n_shapes == 4 + 1568
    default = rta.composite_scene(text, 0, 0)[1]
    assert (default.horizontal_size, default.vertical_size) == (7680, 4320)
