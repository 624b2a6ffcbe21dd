"""How far would f64 arithmetic sit from the f32 reference?  (SURVEY.md §8c sensitivity table.)

BASELINE.json's north_star asks for "f64 math kept to match reference precision"; the reference is
f32 end to end (src/tuple.zig:6,10), so f32 in the reference's operation order is what reproduces
it bit for bit.  This test keeps the evidence: the oracle rebuilt with its ray and shading
arithmetic in double (oracle/liboracle_f64.so, -DORACLE_F64; same f32 scene records) differs from
the f32 oracle on a small fraction of pixels — inside north_star's ±1 LSB / 99.9 % band on the
opaque scenes, outside it where refraction amplifies the rounding (glass_spheres) — so an f64
kernel could never be the bit-exact drop-in the f32 kernels are.

Test infrastructure only: nothing in the product loads either oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as o

F64_SO = os.path.join(o.ORACLE_DIR, "liboracle_f64.so")


def _lib64():
    if not os.path.exists(F64_SO):
        subprocess.check_call(["make", "-s", "-C", o.ORACLE_DIR, "liboracle_f64.so"])
    L = C.CDLL(F64_SO)
    P = C.POINTER
    L.oracle_sample.restype = C.c_void_p
    L.oracle_sample.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    L.oracle_scene_world.restype = P(o.st.World)
    L.oracle_scene_world.argtypes = [C.c_void_p]
    L.oracle_scene_camera.restype = P(o.st.Camera)
    L.oracle_scene_camera.argtypes = [C.c_void_p]
    L.oracle_sample_free.argtypes = [C.c_void_p]
    L.oracle_render.restype = C.c_int
    L.oracle_render.argtypes = [P(o.st.Camera), P(o.st.World), P(C.c_float), C.c_int, C.c_uint32, P(o.st.RayCounts)]
    return L


def _render64(L, name, w, h):
    hnd = L.oracle_sample(name.encode(), w, h)
    assert hnd
    cam = L.oracle_scene_camera(hnd).contents
    world = L.oracle_scene_world(hnd).contents
    canvas = np.zeros((h, w, 3), dtype=np.float32)
    counts = o.st.RayCounts()
    assert L.oracle_render(C.byref(cam), C.byref(world), canvas.ctypes.data_as(C.POINTER(C.c_float)), 1, 1,
                           C.byref(counts)) == 0
    L.oracle_sample_free(hnd)
    return canvas


# (scene, lower bound on the fraction of pixels within ±1 LSB, must the f64 image differ somewhere?)
CASES = [
    ("skewed_triangle", 0.999, False),
    ("cone", 0.999, False),
    ("cylinder_world", 0.995, True),
    ("glass_spheres", 0.99, True),
    ("fresnel", 0.99, True),
]


@pytest.mark.parametrize("name,floor,must_differ", CASES)
def test_f64_arithmetic_is_close_but_not_identical(name, floor, must_differ):
    L = _lib64()
    w = h = 240
    s = o.Sample(name, w, h)
    ref, _ = o.render(s.camera, s.world)
    f64 = _render64(L, name, w, h)
    a = o.to_ppm_bytes(ref).astype(np.int16)
    b = o.to_ppm_bytes(f64).astype(np.int16)
    d = np.abs(a - b).max(axis=-1)
    within1 = float((d <= 1).mean())
    exact_f32 = float((ref == f64).all(axis=-1).mean())
    print(f"{name}: f64 vs f32 oracle: {100 * within1:.3f} % of pixels within ±1 LSB, "
          f"{100 * exact_f32:.2f} % with identical f32 colour")
    assert within1 >= floor
    if must_differ:
        # the point of the test: double arithmetic does not reproduce the f32 canvas
        assert exact_f32 < 1.0
