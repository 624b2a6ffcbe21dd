"""Triangle BVH (new design) on the CPU: rtx_bvh_selftest builds the trees a scene would get and
runs the traversal code the kernels compile — here compiled for the host — against brute force.
No GPU involved; the GPU tests (test_gpu_parity.py) repeat the brute-force comparison on the device."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

import scene_util as su

rta = importlib.import_module("ray-tracer_amd")
st = rta.st
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def teapot_text(level=1):
    return rta.teapot_obj_text(level)


def selftest(world, origins, dirs):
    flat = rta.FlatScene(world)
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 4)
    nn, nt, bad = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = rta.lib().rtx_bvh_selftest(C.byref(flat.desc), rta.fptr(o), rta.fptr(d), o.shape[0], C.byref(nn), C.byref(nt), C.byref(bad))
    assert rc == 0, rta.lib().rtx_last_error()
    return nn.value, nt.value, bad.value, flat


def rays_for(flat, rng, n):
    """Random rays, rays through triangle vertices and edge midpoints (grazing), axis-parallel rays,
    rays that start inside the mesh, rays pointing away from it."""
    d = flat.desc
    tri = [i for i in range(d.n_shapes) if d.shapes[i].type == 5]
    pick = rng.choice(tri, size=n)
    targets = []
    for k, i in enumerate(pick):
        sh = d.shapes[i]
        p0, e0, e1 = np.array(sh.p0[:3]), np.array(sh.e0[:3]), np.array(sh.e1[:3])
        kind = k % 4
        if kind == 0:
            targets.append(p0)                      # a vertex shared by several triangles
        elif kind == 1:
            targets.append(p0 + 0.5 * e0)           # an edge midpoint
        elif kind == 2:
            u, v = rng.uniform(0, 0.5, 2)
            targets.append(p0 + u * e0 + v * e1)    # interior
        else:
            targets.append(rng.uniform(-4, 4, 3))   # anywhere
    targets = np.array(targets, dtype=np.float32)
    origins = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    origins[::5] = rng.uniform(-1, 1, (len(origins[::5]), 3)) + np.array([0, 1.5, 0])  # inside the pot
    dirs = targets - origins
    dirs /= np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-6)
    dirs[::7, 0] = 0.0       # exactly axis-parallel components: 1/d = inf in the slab test
    dirs[::11, 1] = 0.0
    dirs[3::13] *= -1.0      # pointing away: only negative roots (predecessor walk)
    o4 = np.concatenate([origins, np.ones((n, 1), np.float32)], axis=1)
    d4 = np.concatenate([dirs.astype(np.float32), np.zeros((n, 1), np.float32)], axis=1)
    return o4, d4


@pytest.mark.parametrize("level,tris", [(1, 1568), (2, 6400)])
def test_teapot_bvh_equals_brute_force_on_the_host(level, tris):
    world, cam, keep = rta.teapot_scene(teapot_text(level), 64, 64)
    flat = rta.FlatScene(world)
    assert flat.desc.n_shapes == tris
    o4, d4 = rays_for(flat, np.random.default_rng(level), 6000)
    n_nodes, n_tris, bad, _ = selftest(world, o4, d4)
    degenerate = sum(1 for i in range(tris) if not any(flat.desc.shapes[i].e0[:3]))
    assert n_tris == tris - degenerate and n_nodes >= n_tris // 4
    assert bad == 0


def test_transformed_mesh_and_camera_rays():
    # a rotated, scaled and translated group: every triangle carries a non-identity inverse
    obj = rta.ObjFile(teapot_text(), bound=False)
    obj.group.transform = su.compose(su.translation(1.5, -0.5, 2.0), su.rotation("y", 0.7), su.scaling(0.8, 1.3, 0.6))
    world, keep = st.make_world([0, 10, -10, 1], [1, 1, 1], groups=[obj.group])
    keep.append(obj)
    flat = rta.FlatScene(world)
    assert not any(flat.desc.shapes[i].flags & (1 << 8) for i in range(10))
    cam = su.default_camera(96, 64)
    rays = []
    for y in range(0, 64, 3):
        for x in range(0, 96, 3):
            r = st.Ray()
            rta.lib().rt_ray_for_pixel(C.byref(cam), x, y, C.byref(r))
            rays.append((list(r.origin), list(r.direction)))
    o4 = np.array([r[0] for r in rays], np.float32)
    d4 = np.array([r[1] for r in rays], np.float32)
    n_nodes, n_tris, bad, _ = selftest(world, o4, d4)
    assert n_tris == 1568 - 512 and bad == 0  # the OBJ fan emits one degenerate triangle per quad
    o5, d5 = rays_for(flat, np.random.default_rng(9), 2000)
    # rays_for aims at object-space vertices; still valid rays, mostly misses: the walk must agree anyway
    assert selftest(world, o5, d5)[2] == 0


def test_short_triangle_runs_and_plain_scenes_get_no_bvh():
    s = rta.Sample("skewed_triangle")
    assert selftest(s.world, np.zeros((0, 4)), np.zeros((0, 4)))[:2] == (0, 0)
    s = rta.Sample("glass_spheres")
    assert selftest(s.world, np.zeros((0, 4)), np.zeros((0, 4)))[:2] == (0, 0)


def test_triangle_run_is_taken_from_the_tail_whatever_the_type_order():
    """The C ABI does not require a node's shapes in type order (rt_flatten happens to emit them so).
    Only the maximal trailing run of triangles may go into a BVH: with [triangles..., sphere] nothing
    does (the sphere must not be swallowed as a triangle), with [triangle, sphere, triangles...] the tail does."""
    world, cam, keep = rta.teapot_scene(teapot_text(), 64, 64)
    flat = rta.FlatScene(world)
    d = flat.desc
    nodes = [d.nodes[i] for i in range(d.n_nodes)]
    big = max(nodes, key=lambda n: n.n_shapes)
    assert big.n_shapes >= 64
    last = d.shapes[big.first_shape + big.n_shapes - 1]
    saved = (last.type, list(last.inverse))
    nn, nt, bad = C.c_uint32(), C.c_uint32(), C.c_uint32()

    def run():
        rc = rta.lib().rtx_bvh_selftest(C.byref(d), None, None, 0, C.byref(nn), C.byref(nt), C.byref(bad))
        assert rc == 0, rta.lib().rtx_last_error()
        return nn.value, nt.value

    whole = run()
    last.type = 0  # the node now ends in a sphere
    others = sum(n.n_shapes for n in nodes if n is not big and n.n_shapes >= 16)
    n_nodes, n_tris = run()
    assert n_tris < whole[1] and n_tris <= others  # nothing of the big node went into a tree
    last.type = saved[0]
    mid = d.shapes[big.first_shape + 1]
    mid.type = 0  # [triangle, sphere, triangles...]: the tail after the sphere is still a run
    n_nodes, n_tris = run()
    assert 0 < whole[1] - n_tris <= 2
    mid.type = 5
    assert run() == whole


@pytest.mark.parametrize("label,tf", [
    ("scale 1000", lambda: su.scaling(1000.0, 1000.0, 1000.0)),
    ("scale 0.01", lambda: su.scaling(0.01, 0.01, 0.01)),
    ("far translation", lambda: su.translation(20000.0, -5000.0, 30000.0)),
    ("rotated, anisotropic, far", lambda: su.compose(su.translation(-3000.0, 100.0, 800.0), su.rotation("z", 0.4), su.rotation("y", 1.1),
                                                     su.scaling(30.0, 2.0, 9.0))),
])
def test_bvh_walk_is_conservative_away_from_the_unit_scale(label, tf):
    """The slab slack is relative to the slab parameters and the box padding to the coordinates:
    scaled, far-away and sheared-looking meshes must cull exactly like the unit-scale teapot — nothing the
    f32 triangle test accepts may be lost.  Rays aim at world-space vertices, edges and interiors from
    origins near and far (up to 100 mesh diameters away)."""
    obj = rta.ObjFile(teapot_text(), bound=False)
    obj.group.transform = tf()
    world, keep = st.make_world([0, 10, -10, 1], [1, 1, 1], groups=[obj.group])
    keep.append(obj)
    flat = rta.FlatScene(world)
    d = flat.desc
    m = np.array(obj.group.transform.to_numpy(), dtype=np.float64)
    rng = np.random.default_rng(5)
    n = 4000
    tri = [i for i in range(d.n_shapes) if d.shapes[i].type == 5 and any(d.shapes[i].e0[:3])]
    pick = rng.choice(tri, size=n)
    targets = np.zeros((n, 3))
    for k, i in enumerate(pick):
        sh = d.shapes[i]
        p0, e0, e1 = np.array(sh.p0[:3]), np.array(sh.e0[:3]), np.array(sh.e1[:3])
        u, v = [(0, 0), (0.5, 0), (0.3, 0.3), (1, 0)][k % 4]
        targets[k] = (m @ np.append(p0 + u * e0 + v * e1, 1.0))[:3]   # object -> world
    centre, size = targets.mean(axis=0), np.ptp(targets, axis=0).max()
    dist = size * np.where(np.arange(n) % 3 == 0, 100.0, np.where(np.arange(n) % 3 == 1, 3.0, 0.2))[:, None]
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    origins = targets - dirs * dist
    dirs[::7, 0] = 0.0
    o4 = np.concatenate([origins, np.ones((n, 1))], axis=1).astype(np.float32)
    d4 = np.concatenate([dirs, np.zeros((n, 1))], axis=1).astype(np.float32)
    n_nodes, n_tris, bad, _ = selftest(world, o4, d4)
    assert n_tris == 1568 - 512 and bad == 0, label
