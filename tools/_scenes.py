"""Scene lookup shared by the tools: a sample name, or teapot[:level] / composite[:level] (generated mesh)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
rta = importlib.import_module("ray-tracer_amd")


def load(name, w, h):
    """-> (world, camera, keepalive)"""
    base = name.split(":")[0]
    if base == "composite_nomesh":  # the composite without its mesh group: what the analytic part of the frame costs alone
        world, cam, keep = load("composite" + name[len(base):], w, h)
        world.n_groups = 0
        return world, cam, keep
    if base == "composite_tinymesh":  # the composite with its mesh swapped for 32 far-away triangles: the cost of carrying a tree at all
        import ctypes as C
        st = rta.st
        world, cam, keep = load("composite" + name[len(base):], w, h)
        tris = []
        for k in range(32):
            t = st.Triangle()
            p = [(C.c_float * 4)(500.0 + k, 500.0, 500.0, 1.0), (C.c_float * 4)(501.0 + k, 500.0, 500.0, 1.0), (C.c_float * 4)(500.0 + k, 501.0, 500.0, 1.0)]
            rta.lib().rt_triangle_init(p[0], p[1], p[2], C.byref(t))
            t.common_attrs = st.default_attrs()
            tris.append(t)
        g = st.make_group(keep=keep, triangles=tris)
        b = st.Bounds()
        rta.lib().rt_bound_group(C.byref(g), C.byref(b))
        g.has_bounding_box = 1
        g.bounding_box = b
        arr = (st.Group * 1)(g)
        keep.append(arr)
        world.groups = arr
        world.n_groups = 1
        return world, cam, keep
    if base in ("teapot", "composite"):
        level = int(name.split(":")[1]) if ":" in name else 1
        text = rta.teapot_obj_text(level)  # in-process: nothing may be spawned under rocprofv3
        return (rta.teapot_scene if base == "teapot" else rta.composite_scene)(text, w, h)
    s = rta.Sample(name, w, h)
    return s.world, s.camera, [s]
