"""ray-tracer_amd — MI355X-native renderer for the hot path `Camera.render -> World.colorAt`.

This package is a thin ctypes binding over the C ABI of libraytracer_amd.so
(include/rt_api.h, include/rtx.h): the product is the shared library (HIP kernels + C++
host side); Python only moves pointers.  Import with
    rta = importlib.import_module("ray-tracer_amd")
(the directory name carries the reference's hyphen).

There is no CPU fallback: rendering entry points raise RtxError when the HIP library or a
GPU is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import scene_types as st

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(HERE, "libraytracer_amd.so")  # RTX_LIB: A/B builds in experiments

f32 = C.c_float
u32 = C.c_uint32
P = C.POINTER


class RtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtx error %d: %s" % (code, msg))
        self.code = code


class RtxShape(C.Structure):
    _fields_ = [("inverse", f32 * 16), ("type", u32), ("material", u32), ("flags", u32), ("reserved", u32),
                ("minimum", f32), ("maximum", f32), ("pad", f32 * 2),
                ("p0", f32 * 4), ("e0", f32 * 4), ("e1", f32 * 4), ("normal", f32 * 4)]


class RtxMaterial(C.Structure):
    _fields_ = [("color", f32 * 3), ("ambient", f32), ("diffuse", f32), ("specular", f32), ("shininess", f32),
                ("reflective", f32), ("refractive_index", f32), ("transparency", f32),
                ("pattern", C.c_int32), ("reserved", C.c_int32)]


class RtxPattern(C.Structure):
    _fields_ = [("inverse", f32 * 16), ("a", f32 * 3), ("b", f32 * 3), ("color_map", u32), ("reserved", u32)]


class RtxNode(C.Structure):
    _fields_ = [("bounds_min", f32 * 3), ("has_bounds", u32), ("bounds_max", f32 * 3), ("skip", u32),
                ("first_shape", u32), ("n_shapes", u32), ("reserved", u32 * 2)]


class RtxSceneDesc(C.Structure):
    _fields_ = [("n_shapes", u32), ("shapes", P(RtxShape)),
                ("n_materials", u32), ("materials", P(RtxMaterial)),
                ("n_patterns", u32), ("patterns", P(RtxPattern)),
                ("n_nodes", u32), ("nodes", P(RtxNode)),
                ("light_position", f32 * 4), ("light_intensity", f32 * 3), ("flags", u32)]


class RtxCamera(C.Structure):
    _fields_ = [("horizontal_size", u32), ("vertical_size", u32), ("half_width", f32), ("half_height", f32),
                ("pixel_size", f32), ("inverse_transform", f32 * 16)]


class RtxOpts(C.Structure):
    _fields_ = [("depth", u32), ("rows_quirk", u32), ("batch_pixels", u32), ("strip_rows", u32),
                ("strip_count", u32), ("strip_index", u32), ("compact_rows", u32), ("profile", u32),
                ("no_bvh", u32), ("full_w", u32), ("lanes", u32), ("containers", u32), ("real", u32), ("all_shadows", u32), ("work_set", u32)]


class RtxStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_secondary", C.c_uint64),
                ("level_rays", C.c_uint64 * 6), ("nodes", C.c_uint64),
                ("level_hits", C.c_uint64 * 6), ("hits_patterned", C.c_uint64), ("hits_transparent", C.c_uint64),
                ("ms_total", f32), ("ms_trace_shade", f32), ("ms_combine", f32),
                ("launches_trace_shade", u32), ("launches_combine", u32), ("rows_rendered", u32), ("kernel_flags", u32),
                ("bvh_box_tests", C.c_uint64), ("bvh_triangle_tests", C.c_uint64), ("gate_tests", C.c_uint64),
                ("shape_tests", C.c_uint64 * 6), ("shadow_rays_skipped", C.c_uint64)]

    KERNEL_CANONICAL_W, KERNEL_BVH, KERNEL_LDS_TABLES, KERNEL_TRACK_N1, KERNEL_LDS_HIT_RECORDS = 1, 2, 4, 8, 16  # rtx.h RTX_KERNEL_*
    KERNEL_F64 = 32

    @property
    def rays_total(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary


# Every symbol include/rt_api.h and include/rtx.h declare: (name, restype, argtypes).
_SIGNATURES = [
    ("rtx_device_count", C.c_int, []),
    ("rtx_scene_create", C.c_int, [P(RtxSceneDesc), C.c_int, P(C.c_void_p)]),
    ("rtx_scene_destroy", None, [C.c_void_p]),
    ("rtx_scene_refit", C.c_int, [C.c_void_p, P(RtxSceneDesc)]),
    ("rtx_rows_for", C.c_int, [P(RtxCamera), P(RtxOpts), P(u32)]),
    ("rtx_render", C.c_int, [C.c_void_p, P(RtxCamera), P(RtxOpts), P(f32), P(RtxStats)]),
    ("rtx_render_device", C.c_int, [C.c_void_p, P(RtxCamera), P(RtxOpts), C.c_void_p, C.c_void_p, C.c_void_p, P(RtxStats)]),
    ("rtx_render_device_static", C.c_int, [C.c_void_p, P(RtxCamera), P(RtxOpts), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rtx_static_max_width", u32, []),
    ("rtx_multi_create", C.c_int, [P(RtxSceneDesc), P(C.c_int), u32, P(C.c_void_p)]),
    ("rtx_multi_destroy", None, [C.c_void_p]),
    ("rtx_multi_device_count", u32, [C.c_void_p]),
    ("rtx_multi_peer_links", u32, [C.c_void_p]),
    ("rtx_multi_render", C.c_int, [C.c_void_p, P(RtxCamera), P(RtxOpts), u32, P(f32), P(C.c_uint8), P(RtxStats)]),
    ("rtx_multi_render_device", C.c_int, [C.c_void_p, P(RtxCamera), P(RtxOpts), u32, C.c_void_p, C.c_void_p, P(RtxStats)]),
    ("rtx_strip_row_map", C.c_int, [P(RtxCamera), P(RtxOpts), P(u32), u32, P(u32)]),
    ("rtx_profile_collect", C.c_int, [C.c_void_p, P(RtxStats)]),
    ("rtx_color_at", C.c_int, [C.c_void_p, P(f32), P(f32), u32, u32, P(f32)]),
    ("rtx_bvh_selftest", C.c_int, [P(RtxSceneDesc), P(f32), P(f32), u32, P(u32), P(u32), P(u32)]),
    ("rtx_bvh_selftest_work", None, [P(C.c_uint64), P(C.c_uint64)]),
    ("rtx_last_error", C.c_char_p, []),
    ("rtx_struct_size", u32, [u32]),
    ("rtx_build_id", C.c_char_p, []),
    ("rt_camera_init", None, [u32, u32, f32, P(f32), P(f32), P(f32), P(st.Camera)]),
    ("rt_ray_for_pixel", None, [P(st.Camera), u32, u32, P(st.Ray)]),
    ("rt_render", C.c_int, [P(st.Camera), P(st.World), P(f32), P(st.RayCounts)]),
    ("rt_color_at", C.c_int, [P(st.World), P(st.Ray), u32, P(f32)]),
    ("rt_canvas_to_ppm", C.c_uint64, [P(f32), u32, u32, C.c_char_p, C.c_uint64]),
    ("rt_ppm_byte", C.c_int, [f32]),
    ("rt_flatten", C.c_int, [P(st.World), P(C.c_void_p)]),
    ("rt_flat_desc", P(RtxSceneDesc), [C.c_void_p]),
    ("rt_flat_free", None, [C.c_void_p]),
    ("rt_flatten_camera", None, [P(st.Camera), P(RtxCamera)]),
    ("rt_triangle_init", None, [P(f32), P(f32), P(f32), P(st.Triangle)]),
    ("rt_bound_group", None, [P(st.Group), P(st.Bounds)]),
    ("rt_bounds_intersect", C.c_int, [P(st.Bounds), P(st.Ray)]),
    ("rt_bounds_merge", None, [P(st.Bounds), P(st.Bounds)]),
    ("rt_bounds_transform", None, [P(st.Bounds), P(st.Mat4), P(st.Bounds)]),
    ("rt_default_material", None, [P(st.Material)]),
    ("rt_default_attrs", None, [P(st.Attrs)]),
    ("rt_mat_identity", None, [P(st.Mat4)]),
    ("rt_mat_mul", None, [P(st.Mat4), P(st.Mat4), P(st.Mat4)]),
    ("rt_mat_mul_tuple", None, [P(st.Mat4), P(f32), P(f32)]),
    ("rt_mat_transpose", None, [P(st.Mat4), P(st.Mat4)]),
    ("rt_mat_determinant", f32, [P(st.Mat4)]),
    ("rt_mat_determinant_n", f32, [P(f32), C.c_int]),
    ("rt_mat_minor_n", f32, [P(f32), C.c_int, C.c_int, C.c_int]),
    ("rt_mat_cofactor_n", f32, [P(f32), C.c_int, C.c_int, C.c_int]),
    ("rt_mat_inverse", None, [P(st.Mat4), P(st.Mat4)]),
    ("rt_mat_translation", None, [f32, f32, f32, P(st.Mat4)]),
    ("rt_mat_scaling", None, [f32, f32, f32, P(st.Mat4)]),
    ("rt_mat_rotation_x", None, [f32, P(st.Mat4)]),
    ("rt_mat_rotation_y", None, [f32, P(st.Mat4)]),
    ("rt_mat_rotation_z", None, [f32, P(st.Mat4)]),
    ("rt_mat_shearing", None, [f32, f32, f32, f32, f32, f32, P(st.Mat4)]),
    ("rt_tup_dot", f32, [P(f32), P(f32)]),
    ("rt_tup_cross", None, [P(f32), P(f32), P(f32)]),
    ("rt_tup_magnitude", f32, [P(f32)]),
    ("rt_tup_normalize", None, [P(f32), P(f32)]),
    ("rt_tup_reflect", None, [P(f32), P(f32), P(f32)]),
    ("rt_tup_equal", C.c_int, [P(f32), P(f32)]),
    ("rt_pow", f32, [f32, f32]),
    ("rt_sin", f32, [f32]),
    ("rt_cos", f32, [f32]),
    ("rt_tan", f32, [f32]),
    ("rt_obj_parse", C.c_int, [C.c_char_p, C.c_uint64, P(C.c_void_p)]),
    ("rt_obj_vertex_count", u32, [C.c_void_p]),
    ("rt_obj_face_count", u32, [C.c_void_p]),
    ("rt_obj_group_count", u32, [C.c_void_p]),
    ("rt_obj_to_group", C.c_int, [C.c_void_p, C.c_int, P(P(st.Group))]),
    ("rt_obj_free", None, [C.c_void_p]),
    ("rt_sample_create", C.c_void_p, [C.c_char_p, u32, u32]),
    ("rt_sample_create_obj", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_uint64, u32, u32]),
    ("rt_sample_world", P(st.World), [C.c_void_p]),
    ("rt_sample_camera", P(st.Camera), [C.c_void_p]),
    ("rt_sample_destroy", None, [C.c_void_p]),
]

SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def build():
    """Compile libraytracer_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "-j8"])


def lib():
    """Load the shared library; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtxError(-2, "%s is missing: run `make -C ray-tracer_amd` (or __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def build_id():
    """rtx_build_id(): hash of the sources the loaded library was built from."""
    return lib().rtx_build_id().decode()


def _check(rc):
    if rc != 0:
        raise RtxError(rc, lib().rtx_last_error().decode(errors="replace"))


def fptr(a):
    return a.ctypes.data_as(P(f32))


class Sample:
    """One of the reference's sample scenes (samples/<name>.zig) built by the C++ mirror."""

    def __init__(self, name, width=0, height=0, obj_text=None):
        L = lib()
        self.name = name
        if obj_text is not None:  # "teapot" / "composite": the scenes built around an OBJ mesh
            data = obj_text.encode() if isinstance(obj_text, str) else obj_text
            self.handle = L.rt_sample_create_obj(name.encode(), data, len(data), width, height)
        else:
            self.handle = L.rt_sample_create(name.encode(), width, height)
        if not self.handle:
            raise KeyError(name)
        self.world = L.rt_sample_world(self.handle).contents
        self.camera = L.rt_sample_camera(self.handle).contents

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:  # `lib` is gone when the interpreter shuts down
            lib().rt_sample_destroy(self.handle)
            self.handle = None


class ObjFile:
    """obj.parseObjFile + toShapeGroup (src/obj_file.zig): text -> rt_group owned by this object."""

    def __init__(self, text, bound=True):
        data = text.encode() if isinstance(text, str) else text
        h = C.c_void_p()
        rc = lib().rt_obj_parse(data, len(data), C.byref(h))
        if rc != 0:
            raise RtxError(rc, "OBJ parse error")
        self.handle = h
        g = P(st.Group)()
        _check(lib().rt_obj_to_group(h, 1 if bound else 0, C.byref(g)))
        self.group = g.contents
        self.n_vertices = lib().rt_obj_vertex_count(h)
        self.n_faces = lib().rt_obj_face_count(h)
        self.n_groups = lib().rt_obj_group_count(h)

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:
            lib().rt_obj_free(self.handle)
            self.handle = None


def teapot_obj_text(level=1):
    """OBJ text of the generated teapot (samples/make_teapot_obj.py), tessellation `level` (1 = the
    committed samples/teapot.obj).  Generated in this process: nothing is spawned, so it is safe
    under rocprofv3, whose preloaded library has initialised the GPU before Python starts."""
    root = os.path.dirname(HERE)
    if level == 1:
        with open(os.path.join(root, "samples", "teapot.obj")) as f:
            return f.read()
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_teapot_obj", os.path.join(root, "samples", "make_teapot_obj.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.obj_text(level)


def teapot_scene(obj_text, width=256, height=256, extra=None):
    """samples/teapot.zig: the OBJ group (bounded) under its light and camera.

    `extra` may add world-level shapes (dict of lists as for scene_types.make_world).
    Returns (world, camera, keepalive)."""
    import math

    obj = ObjFile(obj_text, bound=True)
    world, keep = st.make_world([0, 0, -10, 1], [1, 1, 1], groups=[obj.group], **(extra or {}))
    cam = camera(width, height, float(np.float32(math.pi / 3.0)), (0, 3, -10, 1), (0, 0, 0, 1), (0, 1, 0, 0))
    keep.append(obj)
    return world, cam, keep


def composite_scene(obj_text, width=7680, height=4320):
    """BASELINE.json configs[4]'s scaled stress scene (frozen in host/rt_samples.cpp).
    Returns (world, camera, keepalive) like teapot_scene."""
    s = Sample("composite", width, height, obj_text=obj_text)
    return s.world, s.camera, [s]


SAMPLES = ["sphere_world", "glass_spheres", "fresnel", "cylinder_world", "cone", "cube", "reflective_floor",
           "skewed_triangle", "striped_scene", "sphere_on_plane"]


def camera(hsize, vsize, fov, frm, to, up):
    """camera() — src/camera.zig:25-51."""
    c = st.Camera()
    a = [(f32 * 4)(*v) for v in (frm, to, up)]
    lib().rt_camera_init(hsize, vsize, fov, a[0], a[1], a[2], C.byref(c))
    return c


def flatten_camera(cam):
    out = RtxCamera()
    lib().rt_flatten_camera(C.byref(cam), C.byref(out))
    return out


class FlatScene:
    """Host-side flattened scene (rt_flatten): owns the arrays of an rtx_scene_desc."""

    def __init__(self, world):
        h = C.c_void_p()
        _check(lib().rt_flatten(C.byref(world), C.byref(h)))
        self.handle = h
        self.desc = lib().rt_flat_desc(h).contents

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:
            lib().rt_flat_free(self.handle)
            self.handle = None


class Scene:
    """A scene resident on one GPU (rtx_scene)."""

    def __init__(self, world, device=0, device_bvh=False):
        flat = FlatScene(world)
        if device_bvh:
            flat.desc.flags |= 1  # RTX_SCENE_BVH_DEVICE
        h = C.c_void_p()
        _check(lib().rtx_scene_create(C.byref(flat.desc), device, C.byref(h)))
        self.handle = h
        self.n_shapes = flat.desc.n_shapes

    def refit(self, world):
        """rtx_scene_refit: the same scene structure with moved shapes."""
        flat = FlatScene(world)
        _check(lib().rtx_scene_refit(self.handle, C.byref(flat.desc)))

    def close(self):
        if getattr(self, "handle", None) and lib is not None:
            lib().rtx_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    def render(self, cam, rows_quirk=True, depth=0, batch_pixels=0, strips=None, compact_rows=False, profile=False,
               no_bvh=False, full_w=False, lanes=0, book=False, f64=False, all_shadows=False):
        """render() — src/camera.zig:84-105 -> (canvas float32 [rows, W, 3], RtxStats)."""
        L = lib()
        rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
        opts = RtxOpts(depth=depth, rows_quirk=1 if rows_quirk else 0, batch_pixels=batch_pixels,
                       compact_rows=1 if compact_rows else 0, profile=1 if profile else 0, no_bvh=1 if no_bvh else 0,
                       full_w=1 if full_w else 0, lanes=lanes, containers=1 if book else 0, real=1 if f64 else 0,
                       all_shadows=1 if all_shadows else 0)
        if strips is not None:
            opts.strip_rows, opts.strip_count, opts.strip_index = strips
        rows = u32()
        _check(L.rtx_rows_for(C.byref(rc), C.byref(opts), C.byref(rows)))
        h = rows.value if compact_rows else rc.vertical_size
        canvas = np.zeros((h, rc.horizontal_size, 3), dtype=np.float32)
        stats = RtxStats()
        _check(L.rtx_render(self.handle, C.byref(rc), C.byref(opts), fptr(canvas), C.byref(stats)))
        return canvas, stats

    def render_device(self, cam, d_rgb, d_rgb8=None, stream=None, want_stats=False, **kw):
        """rtx_render_device: canvas stays on the GPU (d_rgb / d_rgb8 are raw device pointers)."""
        L = lib()
        rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
        opts = RtxOpts(depth=kw.get("depth", 0), rows_quirk=1 if kw.get("rows_quirk", True) else 0,
                       batch_pixels=kw.get("batch_pixels", 0), compact_rows=1 if kw.get("compact_rows", False) else 0,
                       profile=1 if kw.get("profile", False) else 0, no_bvh=1 if kw.get("no_bvh", False) else 0,
                       full_w=1 if kw.get("full_w", False) else 0, lanes=kw.get("lanes", 0), real=1 if kw.get("f64", False) else 0,
                       all_shadows=1 if kw.get("all_shadows", False) else 0, work_set=kw.get("work_set", 0))
        if kw.get("strips") is not None:
            opts.strip_rows, opts.strip_count, opts.strip_index = kw["strips"]
        stats = RtxStats() if want_stats else None
        _check(L.rtx_render_device(self.handle, C.byref(rc), C.byref(opts), C.c_void_p(d_rgb),
                                   C.c_void_p(d_rgb8) if d_rgb8 else None, C.c_void_p(stream) if stream else None,
                                   C.byref(stats) if stats is not None else None))
        return stats

    def render_device_static(self, cam, d_rgb, d_rgb8=None, stream=None, **kw):
        """rtx_render_device_static: the same canvas through the never-allocated work set (width <= 1024)."""
        rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
        opts = RtxOpts(depth=kw.get("depth", 0), rows_quirk=1 if kw.get("rows_quirk", True) else 0,
                       batch_pixels=kw.get("batch_pixels", 0), compact_rows=1 if kw.get("compact_rows", False) else 0,
                       no_bvh=1 if kw.get("no_bvh", False) else 0, full_w=1 if kw.get("full_w", False) else 0)
        if kw.get("strips") is not None:
            opts.strip_rows, opts.strip_count, opts.strip_index = kw["strips"]
        _check(lib().rtx_render_device_static(self.handle, C.byref(rc), C.byref(opts), C.c_void_p(d_rgb),
                                              C.c_void_p(d_rgb8) if d_rgb8 else None, C.c_void_p(stream) if stream else None))

    def profile_collect(self):
        """Sum the pending per-launch HIP-event timings (profile=True frames) -> RtxStats."""
        stats = RtxStats()
        _check(lib().rtx_profile_collect(self.handle, C.byref(stats)))
        return stats

    def color_at(self, origins, directions, depth=0):
        """colorAt() — src/world.zig:81-83 for n rays -> float32 [n, 3]."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 4)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 4)
        out = np.zeros((o.shape[0], 3), dtype=np.float32)
        _check(lib().rtx_color_at(self.handle, fptr(o), fptr(d), o.shape[0], depth, fptr(out)))
        return out


class MultiScene:
    """A scene replicated on several GPUs of this process (rtx_multi)."""

    def __init__(self, world, devices=None):
        flat = FlatScene(world)
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices else None
        _check(lib().rtx_multi_create(C.byref(flat.desc), arr, len(devices) if devices else 0, C.byref(h)))
        self.handle = h
        self.n_devices = lib().rtx_multi_device_count(h)
        self.peer_links = lib().rtx_multi_peer_links(h)

    def close(self):
        if getattr(self, "handle", None) and lib is not None:
            lib().rtx_multi_destroy(self.handle)
            self.handle = None

    __del__ = close

    def render(self, cam, chunks_per_gpu=1, rows_quirk=True, depth=0, want_u8=True, want_stats=True, **kw):
        """rtx_multi_render -> (f32 canvas [H, W, 3], u8 canvas or None, RtxStats or None), host arrays."""
        rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
        opts = RtxOpts(depth=depth, rows_quirk=1 if rows_quirk else 0, batch_pixels=kw.get("batch_pixels", 0),
                       no_bvh=1 if kw.get("no_bvh", False) else 0, full_w=1 if kw.get("full_w", False) else 0)
        h, w = rc.vertical_size, rc.horizontal_size
        canvas = np.full((h, w, 3), np.nan, dtype=np.float32)
        canvas8 = np.full((h, w, 3), 77, dtype=np.uint8) if want_u8 else None
        stats = RtxStats() if want_stats else None
        _check(lib().rtx_multi_render(self.handle, C.byref(rc), C.byref(opts), chunks_per_gpu, fptr(canvas),
                                      canvas8.ctypes.data_as(P(C.c_uint8)) if want_u8 else None,
                                      C.byref(stats) if want_stats else None))
        return canvas, canvas8, stats

    def render_device(self, cam, d_rgb, d_rgb8=None, chunks_per_gpu=1, rows_quirk=True, want_stats=False):
        """rtx_multi_render_device: canvases on the first device of the list (raw device pointers)."""
        rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
        opts = RtxOpts(rows_quirk=1 if rows_quirk else 0)
        stats = RtxStats() if want_stats else None
        _check(lib().rtx_multi_render_device(self.handle, C.byref(rc), C.byref(opts), chunks_per_gpu, C.c_void_p(d_rgb),
                                             C.c_void_p(d_rgb8) if d_rgb8 else None, C.byref(stats) if want_stats else None))
        return stats


def strip_row_map(cam, strips=None, rows_quirk=True):
    """rtx_strip_row_map: the global row of every local row of a strip set (host only)."""
    rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
    opts = RtxOpts(rows_quirk=1 if rows_quirk else 0)
    if strips is not None:
        opts.strip_rows, opts.strip_count, opts.strip_index = strips
    n = u32()
    _check(lib().rtx_strip_row_map(C.byref(rc), C.byref(opts), None, 0, C.byref(n)))
    rows = (u32 * max(1, n.value))()
    _check(lib().rtx_strip_row_map(C.byref(rc), C.byref(opts), rows, n.value, C.byref(n)))
    return list(rows[: n.value])


def rows_for(cam, strips=None, rows_quirk=True):
    rc = flatten_camera(cam) if isinstance(cam, st.Camera) else cam
    opts = RtxOpts(rows_quirk=1 if rows_quirk else 0)
    if strips is not None:
        opts.strip_rows, opts.strip_count, opts.strip_index = strips
    rows = u32()
    _check(lib().rtx_rows_for(C.byref(rc), C.byref(opts), C.byref(rows)))
    return rows.value


def render(cam, world):
    """The by-value drop-in rt_render (flatten + upload + render + copy back)."""
    canvas = np.zeros((cam.vertical_size, cam.horizontal_size, 3), dtype=np.float32)
    counts = st.RayCounts()
    _check(lib().rt_render(C.byref(cam), C.byref(world), fptr(canvas), C.byref(counts)))
    return canvas, counts


def to_ppm_bytes(canvas):
    """canvas.zig:45-47 per channel through the library's rt_ppm_byte rule (vectorised)."""
    v = canvas.astype(np.float32) * np.float32(255)
    v = np.maximum(np.float32(0), np.minimum(v, np.float32(255)))
    return np.floor(v.astype(np.float64) + 0.5).astype(np.uint8)
