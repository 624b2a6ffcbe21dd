"""rtx_render_device_static — the entry point that allocates nothing (its work buffers are a file-scope
device array) — against rtx_render_device, through the C ABI, bit for bit: two calls in a row, the
second with a smaller input, then another scene on the same array, and the size limit; and the property the
entry point exists for: the first call on a fresh scene changes the device's free memory by nothing and can be
captured into a HIP graph (where an allocation or a synchronisation is an error) and replayed."""
import importlib

import numpy as np
import pytest

rta = importlib.import_module("ray-tracer_amd")

pytestmark = pytest.mark.gpu


def _both(scene, cam, w, h, **kw):
    import torch

    ref32 = torch.full((h, w, 3), 3.0, dtype=torch.float32, device="cuda")
    ref8 = torch.full((h, w, 3), 9, dtype=torch.uint8, device="cuda")
    scene.render_device(cam, ref32.data_ptr(), ref8.data_ptr(), **kw)
    got32 = torch.full((h, w, 3), 5.0, dtype=torch.float32, device="cuda")
    got8 = torch.full((h, w, 3), 77, dtype=torch.uint8, device="cuda")
    scene.render_device_static(cam, got32.data_ptr(), got8.data_ptr(), **kw)
    torch.cuda.synchronize()
    return ref32.cpu().numpy(), ref8.cpu().numpy(), got32.cpu().numpy(), got8.cpu().numpy()


def test_static_work_set_equals_the_allocating_entry_point():
    import torch

    assert rta.lib().rtx_static_max_width() == 1024
    big = rta.Sample("glass_spheres", 640, 360)    # reflective and transparent: levels can double, 29 batches
    small = rta.Sample("glass_spheres", 200, 120)  # the second call: a smaller input over the first one's leftovers
    scene = rta.Scene(big.world)
    for s, (w, h) in ((big, (640, 360)), (small, (200, 120))):
        ref32, ref8, got32, got8 = _both(scene, s.camera, w, h)
        assert ref32.any() and got32.tobytes() == ref32.tobytes(), "f32 canvas differs at %dx%d" % (w, h)
        assert np.array_equal(got8, ref8)
    # another scene (mesh + BVH, general-w kernels forced, strips) shares the GPU's array
    world, cam, keep = rta.teapot_scene(rta.teapot_obj_text(), 333, 100)
    mesh = rta.Scene(world)
    for kw in ({}, {"full_w": True}, {"strips": (16, 3, 1)}, {"depth": 2, "rows_quirk": False}):
        ref32, ref8, got32, got8 = _both(mesh, cam, 333, 100, **kw)
        assert ref32.any() and got32.tobytes() == ref32.tobytes() and np.array_equal(got8, ref8), kw
    # and the first scene again, after the other one used the array
    ref32, ref8, got32, got8 = _both(scene, small.camera, 200, 120)
    assert got32.tobytes() == ref32.tobytes() and np.array_equal(got8, ref8)
    # the widest frame it takes, and one pixel more
    wide = rta.Sample("fresnel", 1024, 16)
    fres = rta.Scene(wide.world)
    ref32, ref8, got32, got8 = _both(fres, wide.camera, 1024, 16)
    assert got32.tobytes() == ref32.tobytes() and np.array_equal(got8, ref8)
    too_wide = rta.Sample("fresnel", 1025, 16)
    buf = torch.zeros((16, 1025, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(rta.RtxError, match="exceeds the static work set"):
        fres.render_device_static(too_wide.camera, buf.data_ptr())
    for sc in (scene, mesh, fres):
        sc.close()


def test_static_first_call_allocates_nothing_and_can_be_captured():
    import torch

    torch.cuda.init()
    s = rta.Sample("glass_spheres", 320, 200)
    scene = rta.Scene(s.world)  # fresh: no render call has grown a work set on it
    cam = rta.flatten_camera(s.camera)
    got32 = torch.full((200, 320, 3), 5.0, dtype=torch.float32, device="cuda")
    got8 = torch.full((200, 320, 3), 77, dtype=torch.uint8, device="cuda")
    ref32 = torch.zeros_like(got32)
    ref8 = torch.zeros_like(got8)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    scene.render_device_static(cam, got32.data_ptr(), got8.data_ptr(), stream=stream.cuda_stream)
    free1, _ = torch.cuda.mem_get_info()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free0 == free1 == free2, (free0, free1, free2)
    scene.render_device(cam, ref32.data_ptr(), ref8.data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(got32, ref32) and torch.equal(got8, ref8)
    # under stream capture an allocation or a synchronisation is an error: capture the call, replay it twice
    got32.fill_(1.0)
    got8.fill_(3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        scene.render_device_static(cam, got32.data_ptr(), got8.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for _ in range(2):
        got32.fill_(1.0)
        got8.fill_(3)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got32, ref32) and torch.equal(got8, ref8)
    del g
    scene.close()
