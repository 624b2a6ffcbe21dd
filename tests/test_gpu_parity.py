"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): PPM bytes within +-1 LSB on >= 99.9 % of pixels.  What the
implementation actually delivers is stronger — the f32 canvas is bit-identical to the oracle
(same IEEE operations in the same order; pow restated identically on both sides) — and the
tests assert that stronger property, with the north-star bar as the fallback message.
"""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import oracle_lib as o
import scene_util as su

rta = importlib.import_module("ray-tracer_amd")
st = rta.st

pytestmark = pytest.mark.gpu

SMALL = {
    "sphere_world": (200, 100), "glass_spheres": (120, 120), "fresnel": (160, 80), "cylinder_world": (200, 100),
    "cone": (128, 128), "cube": (128, 128), "reflective_floor": (128, 128), "skewed_triangle": (128, 128),
    "striped_scene": (128, 128), "sphere_on_plane": (128, 128),
}


def compare(gpu, ref, what):
    assert gpu.shape == ref.shape
    same = gpu.view(np.uint32) == ref.view(np.uint32)
    both_zero = (gpu == 0) & (ref == 0)  # +0 vs -0 print the same
    exact = (same | both_zero).all(axis=-1)
    pb = np.abs(rta.to_ppm_bytes(gpu).astype(np.int32) - o.to_ppm_bytes(ref).astype(np.int32)).max(axis=-1)
    msg = "%s: bit-exact pixels %.4f%%, PPM exact %.4f%%, within 1 LSB %.4f%%, max |diff| %g" % (
        what, 100 * exact.mean(), 100 * (pb == 0).mean(), 100 * (pb <= 1).mean(), np.abs(gpu - ref).max())
    assert (pb <= 1).mean() >= 0.999, "north-star bar missed: " + msg
    assert exact.all(), msg


@pytest.mark.parametrize("name", rta.SAMPLES)
def test_sample_matches_oracle(name):
    w, h = SMALL[name]
    s = rta.Sample(name, w, h)
    ref, counts = o.render(s.camera, s.world)
    scene = rta.Scene(s.world)
    gpu, stats = scene.render(s.camera)
    compare(gpu, ref, name)
    assert (stats.rays_primary, stats.rays_shadow, stats.rays_secondary) == (counts.primary, counts.shadow, counts.secondary)
    # the canonical-w kernels (chosen per scene) and the four-lane kernels give the same canvas
    full, stats_full = scene.render(s.camera, full_w=True)
    compare(full, ref, name + " (full_w)")
    assert stats_full.rays_total == stats.rays_total
    scene.close()


def test_by_value_drop_in_rt_render():
    s = rta.Sample("cylinder_world", 96, 48)
    ref, counts = o.render(s.camera, s.world)
    gpu, c = rta.render(s.camera, s.world)
    compare(gpu, ref, "rt_render")
    assert c.total == counts.total


@pytest.mark.parametrize("seed", range(20))
def test_random_world_matches_oracle(seed):
    w, keep = su.random_world(seed, n_shapes=6 + seed % 12, groups=seed % 4)
    cam = su.default_camera(96, 64)
    ref, counts = o.render(cam, w)
    scene = rta.Scene(w)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "random world %d" % seed)
    assert stats.rays_total == counts.total
    full, _ = scene.render(cam, full_w=True)
    compare(full, ref, "random world %d (full_w)" % seed)
    scene.close()


@pytest.mark.parametrize("seed", range(8))
def test_axis_aligned_world_matches_oracle(seed):
    """Scaled / translated / untransformed shapes of every type: the diagonal-inverse ray transform, the
    identity shortcuts and the per-shape world normals of planes and triangles, against the oracle's
    full 4x4 arithmetic; the four-lane kernels (no diagonal form) must agree too."""
    w, keep = su.axis_aligned_world(100 + seed, n_shapes=8 + seed)
    cam = su.default_camera(96, 64)
    ref, counts = o.render(cam, w)
    scene = rta.Scene(w)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "axis-aligned world %d" % seed)
    assert stats.rays_total == counts.total
    full, _ = scene.render(cam, full_w=True)
    compare(full, ref, "axis-aligned world %d (full_w)" % seed)
    scene.close()


def test_color_at_matches_oracle_including_rays_from_inside():
    s = rta.Sample("glass_spheres")
    scene = rta.Scene(s.world)
    rng = np.random.default_rng(2)
    origins = np.concatenate([rng.uniform(-0.4, 0.4, (40, 3)), rng.uniform(-3, 3, (40, 3))]).astype(np.float32)
    origins = np.concatenate([origins, np.ones((80, 1), np.float32)], axis=1)
    d = rng.normal(size=(80, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dirs = np.concatenate([d, np.zeros((80, 1), np.float32)], axis=1)
    gpu = scene.color_at(origins, dirs)
    ref = np.stack([o.color_at(s.world, origins[i], dirs[i]) for i in range(80)])
    compare(gpu[None], ref[None], "color_at")
    # rays that are not canonical (origin w != 1, direction w != 0) take the four-lane kernels
    origins[::7, 3] = 0.75
    dirs[::5, 3] = 0.125
    gpu = scene.color_at(origins, dirs)
    ref = np.stack([o.color_at(s.world, origins[i], dirs[i]) for i in range(80)])
    compare(gpu[None], ref[None], "color_at, general w")
    scene.close()


def test_duplicate_intersections_cancel_in_the_n1_walk():
    """Tangent sphere behind the ray origin: its two equal roots cancel (std.meta.eql on whole
    Intersections, intersection.zig:126-137), so n1 comes from the entry before them."""
    glass = st.default_attrs()
    glass.material.transparency = 0.9
    glass.material.reflective = 0.5
    glass.material.refractive_index = 1.5
    front = st.Sphere(glass)
    front.common_attrs.transform = su.translation(0, 0, 5)
    dense = st.default_attrs()
    dense.material.refractive_index = 2.4
    dense.material.transparency = 0.5
    tangent = st.Sphere(dense)
    tangent.common_attrs.transform = su.translation(1, 0, -5)  # ray x=0 is tangent at t=-5 (disc == 0)
    behind = st.default_attrs()
    behind.material.refractive_index = 1.2
    behind.material.transparency = 0.5
    wall = st.Plane(behind)
    wall.common_attrs.transform = su.compose(su.translation(0, 0, -8), su.rotation("x", float(np.float32(np.pi / 2))))
    floor = st.Plane(st.default_attrs())
    floor.common_attrs.transform = su.translation(0, -3, 0)
    floor.common_attrs.material.has_pattern = 1
    floor.common_attrs.material.pattern.a[:] = [1, 0, 0]
    floor.common_attrs.material.pattern.b[:] = [0, 1, 0]
    floor.common_attrs.material.pattern.color_map = st.PATTERN_CHECKER
    w, keep = st.make_world([2, 6, -3, 1], [1, 1, 1], spheres=[front, tangent], planes=[wall, floor])
    origins = np.array([[0, 0, 0, 1], [0, 0.1, 0, 1], [0.05, 0, 0, 1]], np.float32)
    dirs = np.array([[0, 0, 1, 0]] * 3, np.float32)
    scene = rta.Scene(w)
    gpu = scene.color_at(origins, dirs)
    ref = np.stack([o.color_at(w, origins[i], dirs[i]) for i in range(3)])
    compare(gpu[None], ref[None], "tangent duplicate")
    # twin shapes (byte-identical) cancel each other as well
    w2, keep2 = st.make_world([2, 6, -3, 1], [1, 1, 1], spheres=[front, tangent, tangent], planes=[wall, wall, floor])
    scene2 = rta.Scene(w2)
    gpu = scene2.color_at(origins, dirs)
    ref = np.stack([o.color_at(w2, origins[i], dirs[i]) for i in range(3)])
    compare(gpu[None], ref[None], "twin shapes")
    scene.close()
    scene2.close()


def _twin_world(n_twins=2):
    """Byte-identical dense glass spheres, each in a group of its own whose box the test sets by hand: the first box is
    a thin column inside the sphere, the others hold it whole — so some rays reach every twin and others all but the
    first.  The reference cancels equal Intersections per ray (src/intersection.zig:126-137, src/group.zig:58)."""
    glass = st.default_attrs()
    glass.material.transparency = 0.9
    glass.material.reflective = 0.5
    glass.material.refractive_index = 1.5
    front = st.Sphere(glass)
    front.common_attrs.transform = su.translation(0, 0, 5)
    dense = st.default_attrs()
    dense.material.refractive_index = 2.4
    dense.material.transparency = 0.5
    dense.material.color[:] = [0.3, 0.5, 0.9]
    twin = st.Sphere(dense)
    twin.common_attrs.transform = su.translation(0, 0, -5)
    behind = st.default_attrs()
    behind.material.refractive_index = 1.2
    behind.material.transparency = 0.5
    wall = st.Plane(behind)
    wall.common_attrs.transform = su.compose(su.translation(0, 0, -8), su.rotation("x", float(np.float32(np.pi / 2))))
    floor = st.Plane(st.default_attrs())
    floor.common_attrs.transform = su.translation(0, -3, 0)
    floor.common_attrs.material.has_pattern = 1
    floor.common_attrs.material.pattern.a[:] = [1, 0, 0]
    floor.common_attrs.material.pattern.b[:] = [0, 1, 0]
    floor.common_attrs.material.pattern.color_map = st.PATTERN_CHECKER
    keep = []
    groups = []
    for k in range(n_twins):
        b = st.Bounds()
        half = 0.3 if k == 0 else 1.0 + 0.25 * k
        b.minimum[:] = [-half, -half, -6.5, 1]
        b.maximum[:] = [half, half, -3.5, 1]
        groups.append(st.make_group(bounding_box=b, keep=keep, spheres=[twin]))
    w, keep = st.make_world([2, 6, -3, 1], [1, 1, 1], groups=groups, keep=keep, spheres=[front], planes=[wall, floor])
    return w, keep


@pytest.mark.parametrize("n_twins", [2, 3])
def test_twins_behind_different_boxes_cancel_per_ray(n_twins):
    w, keep = _twin_world(n_twins)
    flat = rta.FlatScene(w)
    cls = [flat.desc.shapes[i].reserved for i in range(flat.desc.n_shapes)]
    assert sorted(c for c in cls if c) == [1] * n_twins and not any(flat.desc.shapes[i].flags & 2 for i in range(flat.desc.n_shapes))
    # rays along +z from z = 0: x = 0 passes every box, x = 0.5 misses the thin first one, x = 2.5 misses all
    origins = np.array([[0, 0, 0, 1], [0.1, 0.05, 0, 1], [0.5, 0, 0, 1], [0.45, 0.2, 0, 1], [2.5, 0, 0, 1], [0.29, 0, 0, 1], [0.31, 0, 0, 1]], np.float32)
    dirs = np.array([[0, 0, 1, 0]] * len(origins), np.float32)
    scene = rta.Scene(w)
    gpu = scene.color_at(origins, dirs)
    ref = np.stack([o.color_at(w, origins[i], dirs[i]) for i in range(len(origins))])
    compare(gpu[None], ref[None], "twins behind different boxes (%d)" % n_twins)
    assert not np.array_equal(ref[0], ref[2])  # the two kinds of ray do see different predecessors
    # a camera behind the twins looks through them: rays refracted inside pass one box or several
    cam = su.default_camera(96, 96, 0.6, frm=(0.2, 0.1, -9, 1), to=(0, 0, -5, 1))
    g, sg = scene.render(cam)
    r, counts = o.render(cam, w)
    compare(g, r, "twins behind different boxes (%d), rendered" % n_twins)
    assert sg.rays_total == counts.total
    g2, _ = scene.render(cam, no_bvh=True, batch_pixels=2048)
    assert g2.tobytes() == g.tobytes()
    scene.close()


def test_rows_quirk_leaves_remainder_rows_black():
    s = rta.Sample("sphere_world", 100, 70)  # 8*floor(70/8) = 64 rows rendered (camera.zig:87-88)
    scene = rta.Scene(s.world)
    gpu, stats = scene.render(s.camera)
    ref, _ = o.render(s.camera, s.world)
    compare(gpu, ref, "rows quirk")
    assert stats.rows_rendered == 64 and not gpu[64:].any() and gpu[:64].any()
    full, stats = scene.render(s.camera, rows_quirk=False)
    ref, _ = o.render(s.camera, s.world, rows_quirk=False)
    compare(full, ref, "no quirk")
    scene.close()


def test_depth_option_limits_recursion():
    s = rta.Sample("glass_spheres", 64, 64)
    scene = rta.Scene(s.world)
    _, s5 = scene.render(s.camera)
    _, s2 = scene.render(s.camera, depth=2)
    assert s2.level_rays[3] == 0 and s5.level_rays[5] > 0 and s2.level_rays[1] == s5.level_rays[1]
    scene.close()


# ---- size-independent properties at BASELINE.json's full sizes -----------------------------

def test_full_size_is_independent_of_batching_and_strips():
    s = rta.Sample("glass_spheres", 1920, 1080)
    scene = rta.Scene(s.world)
    a, sa = scene.render(s.camera)
    b, sb = scene.render(s.camera, batch_pixels=1 << 16)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    # four interleaved strip sets (what four ranks would render) reassemble the same canvas bit for bit
    acc = np.zeros_like(a)
    rays = 0
    rows = 0
    for r in range(4):
        part, sp = scene.render(s.camera, strips=(16, 4, r))
        assert not (acc.astype(bool) & part.astype(bool)).any()
        acc += part
        rays += sp.rays_total
        rows += sp.rows_rendered
        comp, _ = scene.render(s.camera, strips=(16, 4, r), compact_rows=True)
        ys = [y for y in range(1080) if (y // 16) % 4 == r]
        assert comp.shape[0] == len(ys) == rta.rows_for(s.camera, strips=(16, 4, r))
        assert comp.tobytes() == a[ys].tobytes()
    assert acc.tobytes() == a.tobytes() and rays == sa.rays_total and rows == 1080
    # sampled rows agree with the oracle (every 135th row keeps the CPU side to seconds)
    ref, _ = o.render(s.camera, s.world, row_step=135)
    ys = [y for y in range(1080) if (y % 135) == 0]
    compare(a[ys], ref[ys], "1920x1080 sampled rows")
    scene.close()


def test_fresnel_full_size_sampled_rows_match_oracle():
    """The other half of BASELINE configs[1]: fresnel at 1920x1080, every 135th row against the oracle."""
    s = rta.Sample("fresnel", 1920, 1080)
    scene = rta.Scene(s.world)
    a, sa = scene.render(s.camera)
    b, sb = scene.render(s.camera, batch_pixels=1 << 17)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    ref, _ = o.render(s.camera, s.world, row_step=135)
    ys = [y for y in range(1080) if (y % 135) == 0]
    compare(a[ys], ref[ys], "fresnel 1920x1080 sampled rows")
    scene.close()


def test_frames_in_flight_on_independent_work_sets():
    """rtx_opts.work_set: frames queued on different streams through different work sets overlap on the GPU and give the
    canvas a frame gives alone; sets are checked, and a frame cut into lanes or carrying per-launch events must use set 0."""
    import torch

    world, cam, keep = rta.composite_scene(_teapot_text(), 640, 360)
    scene = rta.Scene(world)
    fcam = rta.flatten_camera(cam)
    want, _ = scene.render(fcam)
    streams = [torch.cuda.Stream() for _ in range(3)]
    rgb = [torch.full((360, 640, 3), float(k + 2), dtype=torch.float32, device="cuda") for k in range(3)]
    for rounds in range(4):  # twelve frames, three in flight
        for k in range(3):
            scene.render_device(fcam, rgb[k].data_ptr(), None, stream=streams[k].cuda_stream, work_set=k)
    torch.cuda.synchronize()
    for k in range(3):
        assert rgb[k].cpu().numpy().tobytes() == want.tobytes(), "work set %d" % k
    with pytest.raises(rta.RtxError, match="work_set"):
        scene.render_device(fcam, rgb[0].data_ptr(), None, stream=streams[0].cuda_stream, work_set=3)
    with pytest.raises(rta.RtxError, match="work_set"):
        scene.render_device(fcam, rgb[0].data_ptr(), None, stream=streams[0].cuda_stream, work_set=1, lanes=2)
    with pytest.raises(rta.RtxError, match="work_set"):
        scene.render_device(fcam, rgb[0].data_ptr(), None, stream=streams[0].cuda_stream, work_set=2, profile=True)
    scene.close()


def test_device_canvas_and_u8_quantisation():
    import torch

    s = rta.Sample("cylinder_world", 256, 128)
    scene = rta.Scene(s.world)
    host, _ = scene.render(s.camera)
    rgb = torch.zeros((128, 256, 3), dtype=torch.float32, device="cuda")
    rgb8 = torch.zeros((128, 256, 3), dtype=torch.uint8, device="cuda")
    stats = scene.render_device(s.camera, rgb.data_ptr(), rgb8.data_ptr(), want_stats=True)
    torch.cuda.synchronize()
    assert stats.rays_primary == 256 * 128
    assert rgb.cpu().numpy().tobytes() == host.tobytes()
    assert np.array_equal(rgb8.cpu().numpy(), rta.to_ppm_bytes(host))  # canvas.zig:45-47
    scene.close()


@pytest.mark.parametrize("name,w,h", [("glass_spheres", 250, 127), ("striped_scene", 131, 70)])
def test_u8_canvas_covers_every_pixel(name, w, h):
    """f32 and u8 device canvases of one call: every pixel gets its bytes — ragged width (tile
    padding), rows the 8-row quirk leaves black (src/camera.zig:87-88), multi-level pixels —
    whatever the buffers held before, and whatever the batch size."""
    import torch

    s = rta.Sample(name, w, h)
    scene = rta.Scene(s.world)
    host, _ = scene.render(s.camera)
    want = rta.to_ppm_bytes(host)
    assert (want[8 * (h // 8):] == 0).all() and want.any()
    for batch in (0, 16 * w):
        rgb = torch.full((h, w, 3), 7.0, dtype=torch.float32, device="cuda")
        rgb8 = torch.full((h, w, 3), 255, dtype=torch.uint8, device="cuda")
        scene.render_device(s.camera, rgb.data_ptr(), rgb8.data_ptr(), batch_pixels=batch)
        torch.cuda.synchronize()
        assert rgb.cpu().numpy().tobytes() == host.tobytes()
        assert np.array_equal(rgb8.cpu().numpy(), want)
    scene.close()


def test_frames_without_statistics_leave_the_counters_right_for_the_next_frame():
    """Frames that do not ask for statistics skip the totals kernel and rely on the alternating
    counter sets; a frame with statistics after any number of them must still count right, for
    multi-batch frames and for a different scene size in between."""
    import torch

    s = rta.Sample("glass_spheres", 200, 120)
    small = rta.Sample("glass_spheres", 64, 40)
    scene = rta.Scene(s.world)
    host, ref_stats = scene.render(s.camera)
    rgb = torch.zeros((120, 200, 3), dtype=torch.float32, device="cuda")
    rgb_small = torch.zeros((40, 64, 3), dtype=torch.float32, device="cuda")
    for quiet in (1, 2, 3):
        for _ in range(quiet):
            rgb.zero_()
            assert scene.render_device(s.camera, rgb.data_ptr(), None) is None
            torch.cuda.synchronize()
            assert rgb.cpu().numpy().tobytes() == host.tobytes()
        scene.render_device(small.camera, rgb_small.data_ptr(), None, batch_pixels=64 * 8)
        scene.color_at([[0.0, 0.0, -5.0, 1.0]] * quiet, [[0.0, 0.0, 1.0, 0.0]] * quiet)
        st =scene.render_device(s.camera, rgb.data_ptr(), None, want_stats=True, batch_pixels=200 * 8 * (quiet + 1))
        torch.cuda.synchronize()
        assert st.rays_total == ref_stats.rays_total and list(st.level_rays) == list(ref_stats.level_rays)
        assert rgb.cpu().numpy().tobytes() == host.tobytes()
    scene.close()


def test_invalid_arguments_are_rejected():
    s = rta.Sample("cone", 32, 32)
    flat = rta.FlatScene(s.world)
    flat.desc.shapes[0].material = 99
    h = C.c_void_p()
    assert rta.lib().rtx_scene_create(C.byref(flat.desc), 0, C.byref(h)) == -1
    assert b"material" in rta.lib().rtx_last_error()
    scene = rta.Scene(s.world)
    with pytest.raises(rta.RtxError):
        scene.render(s.camera, depth=6)
    with pytest.raises(rta.RtxError):
        scene.render(s.camera, strips=(8, 2, 2))
    scene.close()


# ---- OBJ meshes: group gate + triangle BVH (new design) must equal brute force and the oracle ----

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def _teapot_text(level=1):
    return rta.teapot_obj_text(level)


def test_teapot_matches_oracle_with_and_without_bvh():
    world, cam, keep = rta.teapot_scene(_teapot_text(), 72, 72)
    ref, counts = o.render(cam, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "teapot (BVH)")
    lin, stats_lin = scene.render(cam, no_bvh=True)
    compare(lin, ref, "teapot (linear)")
    full, stats_full = scene.render(cam, full_w=True)
    compare(full, ref, "teapot (BVH, full_w)")
    assert stats.rays_total == counts.total == stats_lin.rays_total
    # The OBJ fan's degenerate triangles (NaN normals, never hit) must not push the frame off the
    # canonical-w kernels: the three renders above are three different kernel variants.
    K = rta.RtxStats
    assert stats.kernel_flags & (K.KERNEL_CANONICAL_W | K.KERNEL_BVH) == K.KERNEL_CANONICAL_W | K.KERNEL_BVH
    assert stats_lin.kernel_flags & (K.KERNEL_CANONICAL_W | K.KERNEL_BVH) == K.KERNEL_CANONICAL_W
    assert stats_full.kernel_flags & (K.KERNEL_CANONICAL_W | K.KERNEL_BVH) == K.KERNEL_BVH
    assert not stats.kernel_flags & K.KERNEL_LDS_TABLES  # 1568 shapes x 160 B do not fit the LDS budget ...
    assert stats.kernel_flags & K.KERNEL_LDS_HIT_RECORDS  # ... their 16-byte hit records do
    scene.close()


def _torus_obj(nu, nv, R=2.0, r=0.8):
    """A torus of nu x nv quads around the y axis, centred at y = 1 (each quad becomes three fan
    triangles, the first degenerate: src/obj_file.zig:156-164)."""
    import math

    lines = []
    for i in range(nu):
        for j in range(nv):
            a, b = 2 * math.pi * i / nu, 2 * math.pi * j / nv
            lines.append("v %.6f %.6f %.6f" % ((R + r * math.cos(b)) * math.cos(a), 1.0 + r * math.sin(b), (R + r * math.cos(b)) * math.sin(a)))
    for i in range(nu):
        for j in range(nv):
            q = [i * nv + j, ((i + 1) % nu) * nv + j, ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv]
            lines.append("f " + " ".join(str(k + 1) for k in q))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("glass", [False, True])
def test_small_mesh_runs_the_canonical_bvh_kernels_with_lds_tables(glass):
    """A mesh small enough for the LDS-staged tables (<= 32 KB of records) and large enough for a BVH:
    the <LDS, C, BVH> kernels, with and without n1 tracking, against the oracle and the other variants."""
    extra = None
    if glass:
        g = rta.Sample("glass_spheres")
        ball = g.world.spheres[0]
        ball.common_attrs.transform = su.compose(su.translation(0.0, 1.2, -2.5), su.scaling(0.8, 0.8, 0.8))
        extra = {"spheres": [ball]}
    world, cam, keep = rta.teapot_scene(_torus_obj(8, 6), 96, 80, extra=extra)
    ref, counts = o.render(cam, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "torus mesh")
    K = rta.RtxStats
    want = K.KERNEL_CANONICAL_W | K.KERNEL_BVH | K.KERNEL_LDS_TABLES | (K.KERNEL_TRACK_N1 if glass else 0)
    assert stats.kernel_flags == want, stats.kernel_flags
    assert stats.rays_total == counts.total and stats.rays_shadow > 500
    for kw in ({"full_w": True}, {"no_bvh": True}, {"full_w": True, "no_bvh": True}):
        other, so = scene.render(cam, **kw)
        compare(other, ref, "torus mesh %r" % kw)
        assert so.rays_total == counts.total and so.kernel_flags != stats.kernel_flags
    scene.close()


def test_composite_at_3840x2160_north_star_target():
    """The frame bench.py's headline is measured on (teapot + glass spheres + reflective floor, depth 5):
    canonical-w BVH kernels, sampled rows against the oracle, the four-lane kernels and strips agree."""
    world, cam, keep = rta.composite_scene(_teapot_text(), 3840, 2160)
    scene = rta.Scene(world)
    a, sa = scene.render(cam)
    K = rta.RtxStats
    assert sa.kernel_flags & (K.KERNEL_CANONICAL_W | K.KERNEL_BVH | K.KERNEL_TRACK_N1) == K.KERNEL_CANONICAL_W | K.KERNEL_BVH | K.KERNEL_TRACK_N1
    assert sa.rays_primary == 3840 * 2160 and sa.rays_secondary > sa.rays_primary // 2
    b, sb = scene.render(cam, full_w=True)
    assert not sb.kernel_flags & K.KERNEL_CANONICAL_W
    compare(b, a, "composite 3840x2160, canonical vs four-lane kernels")
    assert sb.rays_total == sa.rays_total
    part, sp = scene.render(cam, strips=(16, 8, 3), compact_rows=True)
    ys = [y for y in range(2160) if (y // 16) % 8 == 3]
    assert part.tobytes() == a[ys].tobytes()
    scene.close()
    ys = list(range(0, 2160, 270))  # the first row of each of the oracle's 8 bands: sky, teapot, spheres, floor reflections
    ref, _ = o.render(cam, world, row_step=270)
    compare(a[ys], ref[ys], "composite 3840x2160 sampled rows")


def test_teapot_with_glass_and_mirror_matches_oracle():
    """The composite of BASELINE.json's stress config at test size: teapot group + the two glass
    spheres + a reflective checkered floor; n1 tracking forces whole-line BVH traversal."""
    g = rta.Sample("glass_spheres")
    f = rta.Sample("reflective_floor")
    spheres = [g.world.spheres[0], g.world.spheres[1]]
    for s_ in spheres:
        s_.common_attrs.transform = su.compose(su.translation(2.2, 1.0, -3.0), st.Mat4.from_rows(s_.common_attrs.transform.to_numpy()))
    floor = f.world.planes[0]
    world, cam, keep = rta.teapot_scene(_teapot_text(), 80, 64, extra={"spheres": spheres, "planes": [floor]})
    ref, counts = o.render(cam, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "teapot + glass + floor (BVH)")
    lin, _ = scene.render(cam, no_bvh=True)
    assert lin.tobytes() == gpu.tobytes()
    assert stats.rays_total == counts.total
    scene.close()


def test_large_mesh_with_global_memory_tables_matches_oracle():
    """6400 shapes: neither the shape records nor the hit records fit the LDS budget."""
    world, cam, keep = rta.teapot_scene(_teapot_text(2), 64, 56)
    ref, counts = o.render(cam, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "teapot level 2")
    K = rta.RtxStats
    assert not stats.kernel_flags & (K.KERNEL_LDS_TABLES | K.KERNEL_LDS_HIT_RECORDS) and stats.kernel_flags & K.KERNEL_BVH
    assert stats.rays_total == counts.total
    scene.close()


def test_bvh_equals_brute_force_at_full_size():
    world, cam, keep = rta.teapot_scene(_teapot_text(2), 960, 540)
    scene = rta.Scene(world)
    a, sa = scene.render(cam)
    b, sb = scene.render(cam, no_bvh=True)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    assert sa.rays_shadow > 10000
    scene.close()


# ---- the reference's own renders (samples/images/*.ppm, as tests/golden/ref_*.npz) -----------

ppm = importlib.import_module("ray-tracer_amd.ppm")


def _golden(name):
    return np.load(ROOT + "/tests/golden/ref_%s.npz" % name)["rgb"]


@pytest.mark.parametrize("name", ["skewed_triangle", "cone"])
def test_gpu_render_meets_the_north_star_bar_against_shipped_images(name):
    # Files the real Zig program wrote (an older revision, see test_oracle_golden.py): the v1
    # algorithm on the GPU must be within +-1 LSB on >= 99.9 % of their pixels.
    s = rta.Sample(name)
    scene = rta.Scene(s.world)
    gpu, _ = scene.render(s.camera)
    scene.close()
    d = ppm.diff(rta.to_ppm_bytes(gpu), _golden(name))
    assert d["frac_within1"] >= 0.999 and d["frac_exact"] >= 0.998, d


def test_gpu_cylinder_world_against_shipped_image_outside_the_glass_cylinder():
    s = rta.Sample("cylinder_world")
    scene = rta.Scene(s.world)
    gpu, _ = scene.render(s.camera)
    scene.close()
    mask = np.ones((400, 800), dtype=bool)
    mask[199:347, 112:260] = False  # the glass cylinder's footprint is unpinned at v1
    assert ppm.diff(rta.to_ppm_bytes(gpu), _golden("cylinder_world"), mask)["frac_within1"] >= 0.995


def test_render_sample_program_writes_the_reference_ppm(tmp_path):
    # samples/main.cpp: the reference's sample program flow (scene -> render -> <name>.ppm), C++ over the C ABI
    import os
    import subprocess

    exe = ROOT + "/samples/render_sample"
    if not os.path.exists(exe):
        pytest.fail("samples/render_sample is not built: run __graft_entry__.build()")
    out = str(tmp_path / "skewed_triangle.ppm")
    r = subprocess.run([exe, "skewed_triangle", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "262144 primary, 11502 shadow, 0 secondary" in r.stdout  # SURVEY §8c reference point
    img = ppm.read(out)
    assert open(out, "rb").read(15) == b"P3\n512 512\n255\n"
    assert ppm.within_tolerance(img, _golden("skewed_triangle"))
    # the switches of SURVEY section 5: book containers reproduce the Scene API's book render, the rows quirk can be
    # switched off (sphere_world 100x70: 64 of 70 rows with it), f64 runs, a bad value is a usage error
    book = str(tmp_path / "glass_book.ppm")
    r = subprocess.run([exe, "glass_spheres", "64", "64", book, "--containers", "book", "--gpus", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "containers book" in r.stdout, r.stderr
    s = rta.Sample("glass_spheres", 64, 64)
    scene = rta.Scene(s.world)
    assert np.array_equal(ppm.read(book), rta.to_ppm_bytes(scene.render(s.camera, book=True)[0]))
    assert not np.array_equal(ppm.read(book), rta.to_ppm_bytes(scene.render(s.camera)[0]))
    scene.close()
    full = str(tmp_path / "all_rows.ppm")
    r = subprocess.run([exe, "sphere_world", "100", "70", full, "--rows-quirk", "off", "--real", "f64"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "7000 primary" in r.stdout and "f64" in r.stdout, r.stderr
    assert ppm.read(full)[64:].any()
    r = subprocess.run([exe, "sphere_world", "--real", "f16"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr


# ---- BASELINE.json configs[2..4] at their full sizes ---------------------------------------

@pytest.mark.parametrize("name", ["cube", "cylinder_world", "cone", "striped_scene"])
def test_mixed_primitives_at_3840x2160(name):
    s = rta.Sample(name, 3840, 2160)
    scene = rta.Scene(s.world)
    a, sa = scene.render(s.camera)
    b, sb = scene.render(s.camera, batch_pixels=1 << 18)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    assert sa.rays_primary == 3840 * 2160 and np.isfinite(a).all()
    scene.close()
    ref, _ = o.render(s.camera, s.world, row_step=270)
    ys = list(range(0, 2160, 270))
    compare(a[ys], ref[ys], name + " 3840x2160 sampled rows")


def test_teapot_at_3840x2160():
    world, cam, keep = rta.teapot_scene(_teapot_text(), 3840, 2160)
    scene = rta.Scene(world)
    a, sa = scene.render(cam)
    b, sb = scene.render(cam, no_bvh=True)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    parts = np.zeros_like(a)
    for r in range(2):
        part, _ = scene.render(cam, strips=(16, 2, r))
        parts += part
    assert parts.tobytes() == a.tobytes()
    scene.close()
    ref, _ = o.render(cam, world, row_step=270)
    ys = list(range(0, 2160, 270))
    compare(a[ys], ref[ys], "teapot 3840x2160 sampled rows")


def test_composite_at_7680x4320():
    world, cam, keep = rta.composite_scene(_teapot_text(), 7680, 4320)
    scene = rta.Scene(world)
    a, sa = scene.render(cam)
    assert sa.rays_primary == 7680 * 4320 and sa.rays_secondary > sa.rays_primary // 2
    # one of eight ranks' strips, in place and compacted
    part, sp = scene.render(cam, strips=(16, 8, 5))
    ys = [y for y in range(4320) if (y // 16) % 8 == 5]
    assert part[ys].tobytes() == a[ys].tobytes() and sp.rows_rendered == len(ys)
    others = np.ones(4320, dtype=bool)
    others[ys] = False
    assert not part[others].any()
    scene.close()
    ref, _ = o.render(cam, world, row_step=2160)  # the top row and the row through teapot and spheres
    ys = [0, 2160]
    compare(a[ys], ref[ys], "composite 7680x4320 sampled rows")


def test_untranslated_cube_runs_canonical_and_translated_cube_does_not_change_it():
    """A cube normal keeps its w lane through normalize (src/cube.zig:28-44): only an untranslated
    cube leaves w == 0, so only then do the canonical-w kernels apply.  Both cases match the oracle."""
    for shift in (0.0, 0.6):
        mirror = st.default_attrs()
        mirror.material.reflective = 0.6
        mirror.material.transparency = 0.4
        mirror.material.refractive_index = 1.3
        cube = st.Cube(mirror)
        cube.common_attrs.transform = su.compose(su.translation(shift, 0, 0), su.compose(su.rotation("y", 0.5), su.scaling(1.2, 0.7, 0.9)))
        ball = st.Sphere(st.default_attrs())
        ball.common_attrs.transform = su.translation(-2.0, 0.5, 1.0)
        floor = st.Plane(st.default_attrs())
        floor.common_attrs.transform = su.translation(0, -1.5, 0)
        floor.common_attrs.material.reflective = 0.3
        w, keep = st.make_world([-4, 6, -6, 1], [1, 1, 1], cubes=[cube], spheres=[ball], planes=[floor])
        cam = su.default_camera(96, 64)
        ref, counts = o.render(cam, w)
        scene = rta.Scene(w)
        gpu, stats = scene.render(cam)
        compare(gpu, ref, "cube shift %g" % shift)
        full, _ = scene.render(cam, full_w=True)
        compare(full, ref, "cube shift %g (full_w)" % shift)
        assert stats.rays_total == counts.total and stats.rays_secondary > 0
        scene.close()


# ---- edge cases: empty and ragged inputs, many shapes, nested gates ---------------------------

def test_empty_world_and_ragged_canvas_sizes():
    w, keep = st.make_world([0, 10, -10, 1], [1, 1, 1])
    for width, height in [(1, 8), (7, 9), (13, 24), (65, 16)]:
        cam = su.default_camera(width, height)
        scene = rta.Scene(w)
        gpu, stats = scene.render(cam)
        assert gpu.shape == (height, width, 3) and not gpu.any()
        assert stats.rays_primary == width * 8 * (height // 8) and stats.rays_shadow == 0 and stats.rays_secondary == 0
        scene.close()
    # a real scene on the same ragged sizes (tiles are 8x8: edges are padded lanes)
    s = rta.Sample("sphere_world", 37, 19)
    ref, counts = o.render(s.camera, s.world)
    scene = rta.Scene(s.world)
    gpu, stats = scene.render(s.camera)
    compare(gpu, ref, "37x19")
    gpu_all, stats_all = scene.render(s.camera, rows_quirk=False)
    ref_all, _ = o.render(s.camera, s.world, rows_quirk=False)
    compare(gpu_all, ref_all, "37x19, all rows")
    assert stats.rays_total == counts.total and stats_all.rays_primary == 37 * 19
    assert rta.Scene(w).color_at(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)).shape == (0, 3)
    scene.close()


def test_many_shapes_take_the_global_memory_tables():
    # 300 small spheres: the scene tables no longer fit the 32 KB LDS budget
    rng = np.random.default_rng(11)
    spheres = []
    for k in range(300):
        a = st.default_attrs()
        a.material.color[:] = rng.uniform(0.1, 1.0, 3).astype(np.float32)
        a.material.reflective = 0.3 if k % 7 == 0 else 0.0
        a.material.transparency = 0.5 if k % 11 == 0 else 0.0
        a.material.refractive_index = 1.3
        sp = st.Sphere(a)
        x, y, z = rng.uniform(-4, 4), rng.uniform(-2, 2), rng.uniform(0, 6)
        sp.common_attrs.transform = su.compose(su.translation(float(x), float(y), float(z)), su.scaling(0.25, 0.25, 0.25))
        spheres.append(sp)
    w, keep = st.make_world([-5, 8, -8, 1], [1, 1, 1], spheres=spheres)
    cam = su.default_camera(80, 48)
    ref, counts = o.render(cam, w)
    scene = rta.Scene(w)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "300 spheres")
    assert stats.rays_total == counts.total and stats.rays_secondary > 0
    scene.close()


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5])
def test_every_depth_limit_matches_the_oracle(depth):
    s = rta.Sample("glass_spheres", 64, 64)
    o.lib().oracle_set_depth_limit(depth)
    try:
        ref, counts = o.render(s.camera, s.world)
    finally:
        o.lib().oracle_set_depth_limit(5)
    scene = rta.Scene(s.world)
    gpu, stats = scene.render(s.camera, depth=depth)
    compare(gpu, ref, "depth %d" % depth)
    assert stats.rays_total == counts.total
    full, stats5 = scene.render(s.camera)
    assert sum(stats.level_rays[depth + 1:]) == 0
    assert list(stats.level_rays[: depth + 1]) == list(stats5.level_rays[: depth + 1])  # the same tree, cut
    scene.close()


def test_two_lanes_equal_one_lane():
    # rtx_opts.lanes: batches alternate between two work-buffer sets on two streams
    s = rta.Sample("glass_spheres", 640, 360)
    scene = rta.Scene(s.world)
    one, s1 = scene.render(s.camera, lanes=1)
    two, s2 = scene.render(s.camera, lanes=2)
    many, s3 = scene.render(s.camera, lanes=2, batch_pixels=640 * 40)  # nine batches over two lanes
    assert one.tobytes() == two.tobytes() == many.tobytes()
    for a in (s2, s3):
        assert (a.rays_primary, a.rays_shadow, a.rays_secondary, a.nodes) == (s1.rays_primary, s1.rays_shadow, s1.rays_secondary, s1.nodes)
        assert list(a.level_rays) == list(s1.level_rays) and list(a.level_hits) == list(s1.level_hits)
    part, _ = scene.render(s.camera, lanes=2, strips=(16, 3, 1), compact_rows=True)
    ys = [y for y in range(360) if (y // 16) % 3 == 1]
    assert part.tobytes() == one[ys].tobytes()
    with pytest.raises(rta.RtxError):
        scene.render(s.camera, lanes=3)
    scene.close()


@pytest.mark.parametrize("name,w,h", [("glass_spheres", 320, 200), ("cylinder_world", 320, 200), ("reflective_floor", 320, 200), ("striped_scene", 256, 160)])
def test_skipped_shadow_rays_cannot_be_seen(name, w, h):
    # A hit with the light behind its surface returns the ambient term shadowed or not (src/material.zig:72-79):
    # its isShadowed call is counted but not traversed.  The canvas must be the same BITS (+0 / -0 included) as
    # with every shadow ray traversed (rtx_opts.all_shadows), and both must be the oracle's.
    s = rta.Sample(name, w, h)
    scene = rta.Scene(s.world)
    fast, st_fast = scene.render(s.camera)
    full, st_full = scene.render(s.camera, all_shadows=True)
    assert fast.tobytes() == full.tobytes()
    assert (st_fast.rays_primary, st_fast.rays_shadow, st_fast.rays_secondary) == (st_full.rays_primary, st_full.rays_shadow, st_full.rays_secondary)
    assert list(st_fast.level_hits) == list(st_full.level_hits)
    assert 0 < st_fast.shadow_rays_skipped < st_fast.rays_shadow and st_full.shadow_rays_skipped == 0
    ref, counts = o.render(s.camera, s.world)
    compare(fast, ref, name + " (shadow skip)")
    assert st_fast.rays_shadow == counts.shadow
    scene.close()


def test_shadow_skip_is_off_where_ambient_could_be_negative_zero():
    # ambient + 0 + 0 is not ambient when ambient is -0: a scene with a negative (or -0, or non-finite) colour,
    # ambient or light intensity traverses every shadow ray.
    def world(tweak):
        ball = st.Sphere(st.default_attrs())
        floor = st.Plane(st.default_attrs())
        floor.common_attrs.transform = su.translation(0, -1, 0)
        tweak(ball.common_attrs.material)
        return st.make_world([-10, 10, -10, 1], [1, 1, 1], spheres=[ball], planes=[floor])

    def neg_colour(m):
        m.color[:] = [-0.5, 0.2, 0.0]

    def neg_zero_ambient(m):
        m.ambient = -0.0

    def plain(m):
        m.color[:] = [0.5, 0.2, 0.0]

    cam = su.default_camera(160, 96)
    for tweak, skips in ((neg_colour, False), (neg_zero_ambient, False), (plain, True)):
        w, keep = world(tweak)
        scene = rta.Scene(w)
        gpu, stats = scene.render(cam)
        ref, counts = o.render(cam, w)
        assert gpu.view(np.uint32).tobytes() == ref.view(np.uint32).tobytes(), tweak.__name__  # -0 must stay -0
        assert stats.rays_shadow == counts.shadow
        assert (stats.shadow_rays_skipped > 0) == skips, tweak.__name__
        scene.close()


def test_scene_lifecycle_returns_device_memory():
    import torch

    def cycle():
        s = rta.Sample("glass_spheres", 512, 256)
        scene = rta.Scene(s.world)
        scene.render(s.camera, lanes=2, profile=False)
        scene.render(s.camera, profile=True)
        scene.color_at(np.array([[0, 0, -5, 1]], np.float32), np.array([[0, 0, 1, 0]], np.float32))
        scene.close()

    cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory shrank by %d MiB over 20 scene lifecycles" % ((free0 - free1) >> 20)


def test_by_value_rt_color_at_and_camera_init():
    # colorAt (src/world.zig:81-83) through the by-value drop-in: world struct in, colours out
    s = rta.Sample("cylinder_world", 64, 32)
    cam = rta.camera(64, 32, float(np.float32(np.pi / 3.0)), (0, 1.5, -5, 1), (0, 1, 0, 1), (0, 1, 0, 0))  # rt_camera_init
    rays = (st.Ray * 48)()
    for k in range(48):
        rta.lib().rt_ray_for_pixel(C.byref(cam), (k * 5) % 64, (k * 3) % 32, C.byref(rays[k]))
    out = np.zeros((48, 3), np.float32)
    assert rta.lib().rt_color_at(C.byref(s.world), rays, 48, rta.fptr(out)) == 0
    ref = np.stack([o.color_at(s.world, list(rays[k].origin), list(rays[k].direction)) for k in range(48)])
    compare(out[None], ref[None], "rt_color_at")


# ---- several GPUs behind the C ABI (rtx_multi_*): one-GPU boxes rehearse with device 0 listed more than once ----

@pytest.mark.parametrize("devices,chunks", [([0], 1), ([0, 0], 1), ([0, 0, 0], 2), ([0], 5)])
def test_multi_gpu_render_equals_the_single_gpu_canvas(devices, chunks):
    import torch

    s = rta.Sample("glass_spheres", 250, 190)  # ragged width, 184 rows rendered: the last strip is short
    single = rta.Scene(s.world)
    want, want_stats = single.render(s.camera)
    single.close()
    multi = rta.MultiScene(s.world, devices=devices)
    assert multi.n_devices == len(devices)
    got, got8, st = multi.render(s.camera, chunks_per_gpu=chunks)
    assert got.tobytes() == want.tobytes()
    assert np.array_equal(got8, rta.to_ppm_bytes(want))
    assert (st.rays_primary, st.rays_shadow, st.rays_secondary) == (want_stats.rays_primary, want_stats.rays_shadow, want_stats.rays_secondary)
    assert st.rows_rendered == 184 and list(st.level_rays) == list(want_stats.level_rays)
    # the same into canvases on the first device (peer copies)
    rgb = torch.full((190, 250, 3), 7.0, dtype=torch.float32, device="cuda:0")
    rgb8 = torch.full((190, 250, 3), 9, dtype=torch.uint8, device="cuda:0")
    multi.render_device(s.camera, rgb.data_ptr(), rgb8.data_ptr(), chunks_per_gpu=chunks)
    torch.cuda.synchronize()
    assert rgb.cpu().numpy().tobytes() == want.tobytes() and np.array_equal(rgb8.cpu().numpy(), rta.to_ppm_bytes(want))
    full, _, _ = multi.render(s.camera, chunks_per_gpu=chunks, rows_quirk=False, want_u8=False, want_stats=False)
    ref_full, _ = o.render(s.camera, s.world, rows_quirk=False)
    compare(full, ref_full, "multi, all rows")
    multi.close()


def test_multi_gpu_render_of_the_mesh_composite_and_errors():
    world, cam, keep = rta.composite_scene(_teapot_text(), 640, 360)
    single = rta.Scene(world)
    want, _ = single.render(cam)
    single.close()
    multi = rta.MultiScene(world, devices=[0, 0])
    got, _, _ = multi.render(cam, chunks_per_gpu=3, want_u8=False, want_stats=False)
    assert got.tobytes() == want.tobytes()
    multi.close()
    with pytest.raises(rta.RtxError):
        rta.MultiScene(world, devices=[0, 99])


# ---- rtx_opts.containers = BOOK: the container walk that compares shapes only --------------

def _oracle_book(cam, world, **kw):
    o.lib().oracle_set_containers(1)
    try:
        return o.render(cam, world, **kw)
    finally:
        o.lib().oracle_set_containers(0)


@pytest.mark.parametrize("name", ["glass_spheres", "fresnel", "cylinder_world", "reflective_floor"])
def test_book_containers_match_the_oracles_book_walk(name):
    w, h = SMALL[name]
    s = rta.Sample(name, w, h)
    ref, counts = _oracle_book(s.camera, s.world)
    v1, _ = o.render(s.camera, s.world)
    scene = rta.Scene(s.world)
    gpu, stats = scene.render(s.camera, book=True)
    compare(gpu, ref, name + " (book)")
    assert stats.rays_total == counts.total
    if name in ("glass_spheres", "cylinder_world"):
        assert not np.array_equal(ref, v1)  # the two walks really differ on nested / overlapping glass
    again, _ = scene.render(s.camera)       # and the default is still the reference as written
    compare(again, v1, name + " (v1 after book)")
    scene.close()


@pytest.mark.parametrize("seed", range(12))
def test_book_containers_on_random_worlds_and_meshes(seed):
    w, keep = su.random_world(seed, n_shapes=6 + seed % 10, groups=seed % 3)
    cam = su.default_camera(80, 56)
    ref, counts = _oracle_book(cam, w)
    scene = rta.Scene(w)
    gpu, stats = scene.render(cam, book=True)
    compare(gpu, ref, "random world %d (book)" % seed)
    assert stats.rays_total == counts.total
    scene.close()


def test_book_containers_with_a_mesh_inside_glass():
    """Glass around the camera-side of a mesh and glass triangles: the predecessor walk over the BVH keeps the best two."""
    g = rta.Sample("glass_spheres")
    shell = g.world.spheres[0]
    shell.common_attrs.transform = su.compose(su.translation(0.0, 1.0, -4.0), su.scaling(2.5, 2.5, 2.5))
    world, cam, keep = rta.teapot_scene(_torus_obj(10, 8), 72, 60, extra={"spheres": [shell]})
    grp = world.groups[0]
    for i in range(grp.n_triangles):
        m = grp.triangles[i].common_attrs.material
        m.transparency, m.refractive_index, m.reflective = 0.6, 1.3 + 0.1 * (i % 3), 0.2
    ref, counts = _oracle_book(cam, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(cam, book=True)
    compare(gpu, ref, "glass torus in a glass shell (book)")
    lin, _ = scene.render(cam, book=True, no_bvh=True)
    compare(lin, ref, "glass torus in a glass shell (book, linear)")
    assert stats.rays_total == counts.total and stats.kernel_flags & rta.RtxStats.KERNEL_BVH
    v1ref, _ = o.render(cam, world)
    v1, _ = scene.render(cam)
    compare(v1, v1ref, "glass torus in a glass shell (v1)")
    scene.close()


def test_book_containers_against_the_shipped_refractive_images(capsys):
    """Report, not a gate on the numbers: the reference's shipped refractive images were rendered by an older
    revision (glass_spheres.ppm has another light; the glass cylinder of cylinder_world.ppm follows
    the book walk).  v1 refraction stays "parity unpinned"; this is the secondary number SURVEY 8c asks for."""
    out = {}
    for name, mask_box in (("glass_spheres", None), ("cylinder_world", (199, 347, 112, 260))):
        s = rta.Sample(name)
        scene = rta.Scene(s.world)
        gold = _golden(name)
        for mode in ("v1", "book"):
            img = rta.to_ppm_bytes(scene.render(s.camera, book=(mode == "book"))[0])
            mask = None
            if mask_box:  # only the glass cylinder's footprint: the part v1 cannot be pinned on
                mask = np.zeros(gold.shape[:2], dtype=bool)
                mask[mask_box[0]:mask_box[1], mask_box[2]:mask_box[3]] = True
            out[(name, mode)] = ppm.diff(img, gold, mask)["frac_within1"]
        scene.close()
    with capsys.disabled():
        print("\nshipped refractive images, pixels within 1 LSB: " + ", ".join("%s %s %.2f%%" % (k[0], k[1], 100 * v) for k, v in out.items()))
    assert out[("cylinder_world", "book")] > out[("cylinder_world", "v1")]   # the glass cylinder follows the book walk
    assert out[("glass_spheres", "book")] > out[("glass_spheres", "v1")]     # so does the (otherwise stale) glass_spheres image


# ---- BVH conservativeness away from the unit scale (the slab slack is relative) ------------

@pytest.mark.parametrize("label,group_tf,frm,to", [
    ("scale 1000", lambda: su.scaling(1000.0, 1000.0, 1000.0), (0, 3000, -10000, 1), (0, 1000, 0, 1)),
    ("scale 0.01", lambda: su.scaling(0.01, 0.01, 0.01), (0, 0.03, -0.1, 1), (0, 0.01, 0, 1)),
    ("far translation", lambda: su.translation(20000.0, -5000.0, 30000.0), (20000, -4997, 29990, 1), (20000, -4999, 30000, 1)),
    ("rotated, anisotropic, far", lambda: su.compose(su.translation(-3000.0, 100.0, 800.0), su.rotation("z", 0.4), su.rotation("y", 1.1),
                                                     su.scaling(30.0, 20.0, 9.0)), (-3000, 160, 500, 1), (-3000, 130, 800, 1)),
])
def test_bvh_equals_brute_force_under_large_group_transforms(label, group_tf, frm, to):
    obj = rta.ObjFile(_teapot_text(), bound=False)
    obj.group.transform = group_tf()
    light = [frm[0] + 0.3 * abs(frm[2] - to[2]), frm[1] + abs(frm[2] - to[2]), frm[2], 1]
    world, keep = st.make_world(light, [1, 1, 1], groups=[obj.group])
    keep.append(obj)
    cam = rta.camera(160, 120, 1.0, frm, to, (0, 1, 0, 0))
    scene = rta.Scene(world)
    a, sa = scene.render(cam)
    b, sb = scene.render(cam, no_bvh=True)
    assert sa.kernel_flags & rta.RtxStats.KERNEL_BVH and not sb.kernel_flags & rta.RtxStats.KERNEL_BVH
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total, label
    assert sa.rays_shadow > 300, "%s: the mesh is not in view (%d hits)" % (label, sa.rays_shadow)
    scene.close()
    small = rta.camera(48, 36, 1.0, frm, to, (0, 1, 0, 0))
    ref, counts = o.render(small, world)
    scene = rta.Scene(world)
    gpu, stats = scene.render(small)
    compare(gpu, ref, "teapot under " + label)
    assert stats.rays_total == counts.total
    scene.close()


# ---- triangle BVH built and refitted on the device (RTX_SCENE_BVH_DEVICE: Morton sort + Karras hierarchy) ----

@pytest.mark.parametrize("level,size", [(1, (160, 120)), (2, (160, 120)), (4, (320, 240))])
def test_device_built_bvh_gives_the_same_canvas(level, size):
    world, cam, keep = rta.teapot_scene(_teapot_text(level), *size)
    host = rta.Scene(world)
    want, sw = host.render(cam)
    host.close()
    dev = rta.Scene(world, device_bvh=True)
    got, sg = dev.render(cam)
    lin, sl = dev.render(cam, no_bvh=True)
    assert sg.kernel_flags & rta.RtxStats.KERNEL_BVH and sg.bvh_box_tests > 0 and not sl.kernel_flags & rta.RtxStats.KERNEL_BVH
    assert got.tobytes() == want.tobytes() == lin.tobytes()
    assert sg.rays_total == sw.rays_total == sl.rays_total and sg.rays_shadow > 1000
    dev.close()


def test_device_built_bvh_on_the_composite_matches_the_oracle():
    world, cam, keep = rta.composite_scene(_teapot_text(), 96, 72)
    ref, counts = o.render(cam, world)
    scene = rta.Scene(world, device_bvh=True)
    gpu, stats = scene.render(cam)
    compare(gpu, ref, "composite, device-built BVH")
    assert stats.rays_total == counts.total
    book, _ = scene.render(cam, book=True)  # the predecessor walk over a device tree
    o.lib().oracle_set_containers(1)
    try:
        ref_book, _ = o.render(cam, world)
    finally:
        o.lib().oracle_set_containers(0)
    compare(book, ref_book, "composite, device-built BVH (book)")
    scene.close()


def test_refit_follows_a_moving_mesh():
    def world_at(angle, lift):
        obj = rta.ObjFile(_teapot_text(), bound=False)
        obj.group.transform = su.compose(su.translation(0.0, lift, 0.0), su.rotation("y", angle))
        w, keep = st.make_world([-6, 9, -8, 1], [1, 1, 1], groups=[obj.group])
        keep.append(obj)
        return w, keep

    cam = rta.camera(200, 150, 1.0, (0, 4, -11, 1), (0, 1.5, 0, 1), (0, 1, 0, 0))
    w0, k0 = world_at(0.0, 0.0)
    scene = rta.Scene(w0, device_bvh=True)
    first, _ = scene.render(cam)
    for angle, lift in ((0.4, 0.0), (1.3, 0.8), (3.0, -0.5)):
        w1, k1 = world_at(angle, lift)
        scene.refit(w1)
        moved, sm = scene.render(cam)
        fresh_scene = rta.Scene(w1)  # host-built tree over the moved mesh
        fresh, sf = fresh_scene.render(cam)
        fresh_scene.close()
        assert moved.tobytes() == fresh.tobytes() and sm.rays_total == sf.rays_total
        assert moved.tobytes() != first.tobytes()
        assert moved.tobytes() == scene.render(cam, no_bvh=True)[0].tobytes()
    scene.refit(w0)
    assert scene.render(cam)[0].tobytes() == first.tobytes()
    scene.close()
    host_built = rta.Scene(w0)
    with pytest.raises(rta.RtxError, match="built on the device"):
        host_built.refit(w0)
    host_built.close()
    other, _cam, ko = rta.teapot_scene(_teapot_text(2), 64, 64)
    dev = rta.Scene(w0, device_bvh=True)
    with pytest.raises(rta.RtxError):
        dev.refit(other)  # another structure
    dev.close()


def test_refit_rejects_what_it_cannot_honour_and_leaves_the_scene_whole():
    """A refit may move shapes, not re-assign materials (the work buffers were sized for the materials in use at creation)
    or make a triangle unboundable; either is refused before any device record changes."""
    def world(glassy, singular):
        obj = rta.ObjFile(_teapot_text(), bound=False)
        if singular:
            obj.group.transform = su.scaling(1.0, 0.0, 1.0)  # not invertible: its triangles cannot be bounded
        ball = st.Sphere(st.default_attrs())
        ball.common_attrs.transform = su.translation(4, 1, 0)
        if glassy:
            ball.common_attrs.material.transparency = 0.9
            ball.common_attrs.material.reflective = 0.9
            ball.common_attrs.material.refractive_index = 1.5
        w, keep = st.make_world([-6, 9, -8, 1], [1, 1, 1], groups=[obj.group], spheres=[ball])
        keep.append(obj)
        return w, keep

    cam = rta.camera(160, 120, 1.0, (0, 4, -11, 1), (0, 1.5, 0, 1), (0, 1, 0, 0))
    w0, k0 = world(False, False)
    scene = rta.Scene(w0, device_bvh=True)
    first, s0 = scene.render(cam)
    assert s0.rays_secondary == 0
    w1, k1 = world(True, False)  # same structure, but the sphere now points at a glass material: two children per hit
    with pytest.raises(rta.RtxError, match="material"):
        scene.refit(w1)
    assert scene.render(cam)[0].tobytes() == first.tobytes()
    w2, k2 = world(False, True)
    with pytest.raises(rta.RtxError, match="bounded"):
        scene.refit(w2)
    assert scene.render(cam)[0].tobytes() == first.tobytes()  # refused before anything was overwritten
    scene.refit(w0)
    assert scene.render(cam)[0].tobytes() == first.tobytes()
    scene.close()


def test_device_build_falls_back_to_the_linear_path_like_the_host_build():
    """A triangle run that cannot be bounded (singular group transform) stays on the linear path with either builder:
    same scene, same canvas, no error."""
    obj = rta.ObjFile(_teapot_text(), bound=False)
    obj.group.transform = su.scaling(1.0, 0.0, 1.0)
    ball = st.Sphere(st.default_attrs())
    w, keep = st.make_world([-6, 9, -8, 1], [1, 1, 1], groups=[obj.group], spheres=[ball])
    cam = rta.camera(64, 48, 1.0, (0, 4, -11, 1), (0, 1.5, 0, 1), (0, 1, 0, 0))
    host = rta.Scene(w)
    a, sa = host.render(cam)
    dev = rta.Scene(w, device_bvh=True)
    b, sb = dev.render(cam)
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total
    assert not sa.kernel_flags & rta.RtxStats.KERNEL_BVH and not sb.kernel_flags & rta.RtxStats.KERNEL_BVH
    host.close()
    dev.close()


# ---- rtx_opts.real = RTX_REAL_F64: the double-precision reporting instantiation ----------

def _oracle_f64():
    import test_oracle_f64_sensitivity as f64t

    return f64t._lib64()


@pytest.mark.parametrize("name", ["sphere_world", "glass_spheres", "fresnel", "cylinder_world", "cone", "cube", "striped_scene", "skewed_triangle"])
def test_f64_instantiation_matches_the_oracles_f64_build(name, capsys):
    """Parity of the f64 mode is against the oracle's own f64 build (double arithmetic over the same f32 scene
    records), to north_star's bar: PPM bytes within 1 LSB on >= 99.9 % of the pixels.  The f32 canvas stays the
    reference's; how far f64 sits from it is reported (it is the divergence SURVEY fact 1 measured)."""
    import test_oracle_f64_sensitivity as f64t

    w, h = SMALL[name]
    ref64 = f64t._render64(_oracle_f64(), name, w, h)
    s = rta.Sample(name, w, h)
    scene = rta.Scene(s.world)
    gpu64, st64 = scene.render(s.camera, f64=True)
    gpu32, st32 = scene.render(s.camera)
    scene.close()
    assert st64.kernel_flags & rta.RtxStats.KERNEL_F64 and not st32.kernel_flags & rta.RtxStats.KERNEL_F64
    d = np.abs(rta.to_ppm_bytes(gpu64).astype(np.int32) - o.to_ppm_bytes(ref64).astype(np.int32)).max(axis=-1)
    d32 = np.abs(rta.to_ppm_bytes(gpu64).astype(np.int32) - rta.to_ppm_bytes(gpu32).astype(np.int32)).max(axis=-1)
    with capsys.disabled():
        print("\n%s: f64 GPU vs f64 oracle: %.4f%% PPM-exact, %.4f%% within 1 LSB, f32-rounded canvas identical on %.4f%%; f64 vs f32 GPU within 1 LSB %.4f%%"
              % (name, 100 * (d == 0).mean(), 100 * (d <= 1).mean(), 100 * (gpu64 == ref64).all(axis=-1).mean(), 100 * (d32 <= 1).mean()))
    # striped_scene: a stripe edge is a step in colour; the oracle's f64 build inverts the pattern and shape
    # transforms in double at every lookup, the product precomputes them in f32 (its design), so
    # pixels whose pattern coordinate sits within ~1e-7 of an integer land on different stripes (58 of 16384)
    assert (d <= 1).mean() >= (0.995 if name == "striped_scene" else 0.999)
    if name in ("sphere_world", "cone", "cube", "striped_scene", "skewed_triangle"):
        assert st64.rays_total == st32.rays_total  # no refraction: rounding does not change the ray tree


def test_f64_instantiation_on_a_mesh_with_bvh():
    world, cam, keep = rta.teapot_scene(_teapot_text(), 96, 72)
    scene = rta.Scene(world)
    a, sa = scene.render(cam, f64=True)
    b, sb = scene.render(cam, f64=True, no_bvh=True)
    c, sc = scene.render(cam)
    assert sa.kernel_flags & rta.RtxStats.KERNEL_BVH and sa.kernel_flags & rta.RtxStats.KERNEL_F64
    assert a.tobytes() == b.tobytes() and sa.rays_total == sb.rays_total == sc.rays_total
    d = np.abs(rta.to_ppm_bytes(a).astype(np.int32) - rta.to_ppm_bytes(c).astype(np.int32)).max(axis=-1)
    assert (d <= 1).mean() >= 0.999
    scene.close()


def test_f64_with_the_book_walk_is_rejected():
    """The f64 reporting mode renders with the reference's own container walk only: on a scene that tracks n1
    (glass) the combination is an error, not a silent fallback."""
    s = rta.Sample("glass_spheres", 48, 48)
    scene = rta.Scene(s.world)
    _, st32 = scene.render(s.camera)
    assert st32.kernel_flags & rta.RtxStats.KERNEL_TRACK_N1
    with pytest.raises(rta.RtxError, match="container walk"):
        scene.render(s.camera, f64=True, book=True)
    a, _ = scene.render(s.camera, f64=True)  # each of them alone is fine
    b, _ = scene.render(s.camera, book=True)
    assert a.any() and b.any()
    scene.close()


# ---------------------------------------------------------------------------------------- camera-ray tile culling
def _render_with_and_without_tile_cull(scene, cam, **kw):
    """(canvas with the culling on, its stats, canvas with RTX_NO_TILE_CULL=1, its stats); the switch is read per frame."""
    os.environ.pop("RTX_NO_TILE_CULL", None)
    on, st_on = scene.render(cam, **kw)
    os.environ["RTX_NO_TILE_CULL"] = "1"
    try:
        off, st_off = scene.render(cam, **kw)
    finally:
        os.environ.pop("RTX_NO_TILE_CULL", None)
    return on, st_on, off, st_off


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,culls", [("cylinder_world", 203, 117, True), ("glass_spheres", 160, 90, False), ("fresnel", 100, 70, False),
                                           ("cube", 96, 64, True)])
def test_tile_cull_leaves_the_canvas_alone_and_skips_tests(name, w, h, culls):
    # Params.cull: a wave whose 8x8 tile of camera rays lies outside a shape's (or a gate's) projected box skips the test.  Same
    # bits with the switch off, the same ray counts, fewer tests counted; sizes with partial tiles on purpose.
    s = rta.Sample(name, w, h)
    scene = rta.Scene(s.world)
    on, st_on, off, st_off = _render_with_and_without_tile_cull(scene, s.camera)
    assert on.tobytes() == off.tobytes()
    assert (st_on.rays_primary, st_on.rays_shadow, st_on.rays_secondary) == (st_off.rays_primary, st_off.rays_shadow, st_off.rays_secondary)
    if culls:  # a handful of spheres get no entries (their rootless test is cheap: measured, no gain); cubes, cylinders and crowds do
        assert sum(st_on.shape_tests) < sum(st_off.shape_tests), "nothing was culled"
    else:
        assert sum(st_on.shape_tests) == sum(st_off.shape_tests)
    # strips dealt to three ranks, compact rows (a tile's rows are image rows of one 16-row strip)
    for k in range(3):
        a, _, b, _ = _render_with_and_without_tile_cull(scene, s.camera, strips=(16, 3, k), compact_rows=True)
        assert a.tobytes() == b.tobytes()
    scene.close()


@pytest.mark.gpu
def test_tile_cull_on_the_mesh_composite_and_random_worlds():
    obj = rta.teapot_obj_text(1)
    world, cam, keep = rta.composite_scene(obj, 640, 360)
    scene = rta.Scene(world)
    on, st_on, off, st_off = _render_with_and_without_tile_cull(scene, cam)
    assert on.tobytes() == off.tobytes()
    # one gate and three spheres: too few cheap tests to pay for entries, the frame runs as before
    assert st_on.gate_tests == st_off.gate_tests and list(st_on.shape_tests) == list(st_off.shape_tests)
    assert st_on.bvh_box_tests == st_off.bvh_box_tests
    scene.close()
    for seed in range(12):  # every shape type, nested groups with and without boxes; cameras outside and inside the shapes
        w, keep = su.random_world(seed, n_shapes=6 + seed % 12, groups=seed % 4)
        scene = rta.Scene(w)
        for cam in (su.default_camera(96, 64), su.default_camera(72, 72, 1.3, frm=(0.3, 0.2, -1.5, 1), to=(0, 0.5, 2, 1))):
            on, _, off, _ = _render_with_and_without_tile_cull(scene, cam)
            assert on.tobytes() == off.tobytes(), "random world %d" % seed
        scene.close()


@pytest.mark.gpu
def test_tile_cull_keeps_what_lies_behind_the_eye():
    # The reference intersects LINES: a glass triangle behind the camera is crossed at negative t and stays in the container
    # list, so it is n1 for what the ray hits in front (src/intersection.zig:92-159).  Its culling rectangle is the mirror
    # image of its projection; tiles inside must still test it, and the canvas must be the oracle's.
    glass = st.default_attrs()
    glass.material.transparency = 0.9
    glass.material.reflective = 0.3
    glass.material.refractive_index = 1.5
    ball = st.Sphere(glass)
    dense = st.default_attrs()
    dense.material.transparency = 0.8
    dense.material.refractive_index = 2.2
    tri = st.Triangle()
    pts = [(C.c_float * 4)(*p) for p in ((-0.6, 1.2, -9.0, 1.0), (0.9, 1.4, -9.5, 1.0), (0.1, 2.4, -8.5, 1.0))]
    rta.lib().rt_triangle_init(pts[0], pts[1], pts[2], C.byref(tri))
    tri.common_attrs = dense
    behind_ball = st.Sphere(dense)
    behind_ball.common_attrs.transform = su.compose(su.translation(1.5, 2.5, -11), su.scaling(0.7, 0.7, 0.7))
    floor = st.Plane(st.default_attrs())
    floor.common_attrs.transform = su.translation(0, -1, 0)
    crowd = []  # enough cheap shapes for spheres and triangles to get entries at all
    for k in range(7):
        pebble = st.Sphere(st.default_attrs())
        pebble.common_attrs.transform = su.compose(su.translation(-3.0 + k, -0.7, 2.0 + 0.5 * k), su.scaling(0.3, 0.3, 0.3))
        crowd.append(pebble)
    w, keep = st.make_world([-4, 6, -5, 1], [1, 1, 1], spheres=[ball, behind_ball] + crowd, planes=[floor], triangles=[tri])
    cam = su.default_camera(128, 96, 1.0)  # from (0, 1.5, -7) towards the origin: the triangle and the second sphere are behind it
    scene = rta.Scene(w)
    on, st_on, off, st_off = _render_with_and_without_tile_cull(scene, cam)
    assert on.tobytes() == off.tobytes()
    assert sum(st_on.shape_tests) < sum(st_off.shape_tests)
    ref, counts = o.render(cam, w)
    compare(on, ref, "glass behind the eye")
    # the triangle does change what the front sphere looks like: without it the canvas differs
    w2, keep2 = st.make_world([-4, 6, -5, 1], [1, 1, 1], spheres=[ball, behind_ball] + crowd, planes=[floor])
    scene2 = rta.Scene(w2)
    other, _ = scene2.render(cam)
    assert other.tobytes() != on.tobytes()
    scene.close()
    scene2.close()


@pytest.mark.gpu
def test_tile_cull_margin_follows_far_small_spheres():
    # A sphere thousands of radii away: the discriminant's cancellation lets the reference's f32 test report roots for rays
    # that pass it by many radii (pixels, at that size).  The rectangle's margin grows with distance / radius; the canvas must
    # stay the one with every test made, and the oracle's.
    rng = np.random.default_rng(11)
    balls = []
    for k in range(14):
        a = st.default_attrs()
        a.material.color[:] = [float(v) for v in rng.uniform(0.2, 1.0, 3)]
        r = float(10.0 ** rng.uniform(-3.2, -1.0))
        pos = [float(rng.uniform(-6, 6)), float(rng.uniform(-3, 5)), float(rng.uniform(5, 120))]
        b = st.Sphere(a)
        b.common_attrs.transform = su.compose(su.translation(*pos), su.scaling(r, r, r))
        balls.append(b)
    floor = st.Plane(st.default_attrs())
    floor.common_attrs.transform = su.translation(0, -4, 0)
    w, keep = st.make_world([-10, 10, -10, 1], [1, 1, 1], spheres=balls, planes=[floor])
    scene = rta.Scene(w)
    for cam in (su.default_camera(160, 120, 0.5), su.default_camera(96, 64, 1.2)):
        on, st_on, off, st_off = _render_with_and_without_tile_cull(scene, cam)
        assert on.tobytes() == off.tobytes()
        ref, _ = o.render(cam, w)
        compare(on, ref, "far small spheres")
    scene.close()
