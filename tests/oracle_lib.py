"""ctypes loader for the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")

import sys

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
st = importlib.import_module("ray-tracer_amd.scene_types")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build()
    L = C.CDLL(ORACLE_SO)
    P = C.POINTER
    L.oracle_set_libm.argtypes = [C.c_int]
    L.oracle_set_containers.argtypes = [C.c_int]
    L.oracle_set_depth_limit.argtypes = [C.c_int]
    L.oracle_sample.restype = C.c_void_p
    L.oracle_sample.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    L.oracle_scene_world.restype = P(st.World)
    L.oracle_scene_world.argtypes = [C.c_void_p]
    L.oracle_scene_camera.restype = P(st.Camera)
    L.oracle_scene_camera.argtypes = [C.c_void_p]
    L.oracle_sample_free.argtypes = [C.c_void_p]
    L.oracle_camera.argtypes = [C.c_uint32, C.c_uint32, C.c_float, P(C.c_float), P(C.c_float), P(C.c_float), P(st.Camera)]
    L.oracle_render.restype = C.c_int
    L.oracle_render.argtypes = [P(st.Camera), P(st.World), P(C.c_float), C.c_int, C.c_uint32, P(st.RayCounts)]
    L.oracle_color_at.argtypes = [P(st.World), P(st.Ray), P(C.c_float)]
    L.oracle_ray_for_pixel.argtypes = [P(st.Camera), C.c_uint32, C.c_uint32, P(st.Ray)]
    L.oracle_ppm_byte.restype = C.c_int
    L.oracle_ppm_byte.argtypes = [C.c_float]
    for name in ("oracle_pow",):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float, C.c_float]
    for name in ("oracle_sin", "oracle_cos", "oracle_tan"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.oracle_mat_determinant.restype = C.c_float
    L.oracle_mat_determinant_n.restype = C.c_float
    L.oracle_mat_cofactor_n.restype = C.c_float
    L.oracle_mat_minor_n.restype = C.c_float
    L.oracle_tup_dot.restype = C.c_float
    L.oracle_tup_magnitude.restype = C.c_float
    _lib = L
    return L


class Sample:
    """One of the reference's sample scenes, built by the oracle's own builders."""

    def __init__(self, name, width=0, height=0):
        L = lib()
        self.name = name
        self.handle = L.oracle_sample(name.encode(), width, height)
        if not self.handle:
            raise KeyError(name)
        self.world = L.oracle_scene_world(self.handle).contents
        self.camera = L.oracle_scene_camera(self.handle).contents

    def __del__(self):
        if getattr(self, "handle", None):
            lib().oracle_sample_free(self.handle)
            self.handle = None


def render(camera, world, rows_quirk=True, row_step=1, threads=8):
    """oracle_render -> (canvas float32 [H, W, 3], RayCounts).  threads = the reference's
    number_of_threads (src/camera.zig:87): static row bands of H // threads rows."""
    L = lib()
    h, w = camera.vertical_size, camera.horizontal_size
    canvas = np.zeros((h, w, 3), dtype=np.float32)
    counts = st.RayCounts()
    rc = L.oracle_render_threads(C.byref(camera), C.byref(world), canvas.ctypes.data_as(C.POINTER(C.c_float)),
                                 1 if rows_quirk else 0, row_step, threads, C.byref(counts))
    assert rc == 0
    return canvas, counts


def color_at(world, origin, direction):
    L = lib()
    r = st.Ray()
    r.origin[:] = list(origin)
    r.direction[:] = list(direction)
    out = (C.c_float * 3)()
    L.oracle_color_at(C.byref(world), C.byref(r), out)
    return np.array(out[:], dtype=np.float32)


def to_ppm_bytes(canvas):
    """canvas.zig:42-50 quantisation: round-half-away(clamp(c*255, 0, 255)) as uint8 [H, W, 3]."""
    v = canvas.astype(np.float32) * np.float32(255)
    v = np.maximum(np.float32(0), np.minimum(v, np.float32(255)))
    # v >= 0 after the clamp, so round-half-away is floor(v + 0.5), done in f64 to stay exact.
    return np.floor(v.astype(np.float64) + 0.5).astype(np.uint8)
