"""Pin the CPU oracle against the golden images the reference ships.

What the goldens turned out to pin (measured here, see DESIGN.md "Oracle"):
* skewed_triangle.ppm and cone.ppm were rendered by an older revision that evaluated Phong
  at comps.point rather than at over_point (v1: src/world.zig:57-63).  With that one
  documented switch the oracle reproduces them to 1 px / 8 px out of 262 144 — which pins
  camera, ray generation, triangle and cone intersection/normals, Phong incl. pow, and the
  PPM quantisation bit-for-bit.  In v1 mode the only differences are the <=1 LSB shifts that
  the 0.001 offset of the light vector causes.
* cylinder_world.ppm (lit at over_point) pins planes, cylinders + caps, checker + pattern
  transform, shadows and reflection outside the glass cylinder's footprint.
* glass_spheres.ppm is stale (different light, older refraction code): reported only.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, "ref_%s.npz" % name))["rgb"].astype(np.int32)


@pytest.fixture(autouse=True)
def _v1_defaults():
    L = o.lib()
    L.oracle_set_over_eps_factor.argtypes = [C.c_float]
    yield
    L.oracle_set_libm(1)
    L.oracle_set_containers(0)
    L.oracle_set_legacy_light_at_point(0)
    L.oracle_set_over_eps_factor(100.0)


def diff(name, **kw):
    s = o.Sample(name)
    img, counts = o.render(s.camera, s.world)
    d = np.abs(o.to_ppm_bytes(img).astype(np.int32) - gold(name)).max(axis=2)
    return d, counts


@pytest.mark.parametrize("name,shadow", [("skewed_triangle", 11502), ("cone", 40212)])
def test_v1_mode_matches_shipped_render(name, shadow):
    d, counts = diff(name)
    assert (counts.primary, counts.shadow, counts.secondary) == (262144, shadow, 0)  # SURVEY §8c reference points
    assert (d == 0).mean() >= 0.998
    assert (d <= 1).mean() >= 0.99999


@pytest.mark.parametrize("name,max_bad", [("skewed_triangle", 1), ("cone", 8)])
def test_legacy_light_point_reproduces_shipped_render(name, max_bad):
    o.lib().oracle_set_legacy_light_at_point(1)
    d, _ = diff(name)
    assert (d > 0).sum() <= max_bad


def test_glibc_and_zig_libm_agree_on_goldens():
    # Both readings of the un-vendored libm calls give the same 8-bit images on the pinned goldens.
    for name in ("skewed_triangle", "cone"):
        o.lib().oracle_set_libm(0)
        d0, _ = diff(name)
        o.lib().oracle_set_libm(1)
        d1, _ = diff(name)
        assert (d0 > 0).sum() == (d1 > 0).sum()


def test_cylinder_world_outside_glass_cylinder():
    d, counts = diff("cylinder_world")
    assert (counts.primary, counts.shadow, counts.secondary) == (320000, 491499, 223595)
    mask = np.ones(d.shape, bool)
    mask[199:348, 112:261] = False  # the glass cylinder's footprint (stale refraction code in the golden)
    assert (d[mask] <= 1).mean() >= 0.995


def test_glass_spheres_ray_counts():
    s = o.Sample("glass_spheres", 150, 150)
    img, counts = o.render(s.camera, s.world)
    assert counts.primary == 144 * 150  # Q6: rows = 8*floor(150/8); rows 144..149 stay black
    assert not img[144:].any()
    assert counts.shadow > counts.primary and counts.secondary > counts.primary
    assert np.isfinite(img).all()


def test_unrolled_inverse_is_bit_identical_to_literal():
    L = o.lib()
    rng = np.random.default_rng(7)
    for _ in range(200):
        a = o.st.Mat4.from_rows(rng.uniform(-3, 3, (4, 4)))
        x, y = o.st.Mat4(), o.st.Mat4()
        L.oracle_mat_inverse(C.byref(a), C.byref(x))
        L.oracle_mat_inverse_literal(C.byref(a), C.byref(y))
        assert x.to_numpy().tobytes() == y.to_numpy().tobytes()
