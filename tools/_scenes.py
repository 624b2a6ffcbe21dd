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
    if base in ("teapot", "composite"):
        level = int(name.split(":")[1]) if ":" in name else 1
        text = rta.teapot_obj_text(level)  # in-process: nothing may be spawned under rocprofv3
        return (rta.teapot_scene if base == "teapot" else rta.composite_scene)(text, w, h)
    s = rta.Sample(name, w, h)
    return s.world, s.camera, [s]
