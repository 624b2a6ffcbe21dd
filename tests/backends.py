"""The two host-math implementations the known-answer tests run against: the product's
(libraytracer_amd.so, rt_* symbols) and the oracle's (liboracle.so, oracle_* symbols)."""
import ctypes as C
import importlib

import numpy as np

import oracle_lib as o

st = o.st
f32 = C.c_float


def vec(a):
    return (f32 * 4)(*[float(x) for x in a])


class Backend:
    def __init__(self, name):
        self.name = name
        if name == "product":
            self.L = importlib.import_module("ray-tracer_amd").lib()
            self.p = "rt_"
        else:
            self.L = o.lib()
            self.p = "oracle_"
            L = self.L
            P = C.POINTER
            for n, res, args in [
                ("mat_mul", None, [P(st.Mat4)] * 3), ("mat_mul_tuple", None, [P(st.Mat4), P(f32), P(f32)]),
                ("mat_transpose", None, [P(st.Mat4)] * 2), ("mat_determinant", f32, [P(st.Mat4)]),
                ("mat_determinant_n", f32, [P(f32), C.c_int]), ("mat_minor_n", f32, [P(f32), C.c_int, C.c_int, C.c_int]),
                ("mat_cofactor_n", f32, [P(f32), C.c_int, C.c_int, C.c_int]), ("mat_inverse", None, [P(st.Mat4)] * 2),
                ("mat_identity", None, [P(st.Mat4)]),
                ("mat_translation", None, [f32] * 3 + [P(st.Mat4)]), ("mat_scaling", None, [f32] * 3 + [P(st.Mat4)]),
                ("mat_rotation_x", None, [f32, P(st.Mat4)]), ("mat_rotation_y", None, [f32, P(st.Mat4)]),
                ("mat_rotation_z", None, [f32, P(st.Mat4)]), ("mat_shearing", None, [f32] * 6 + [P(st.Mat4)]),
                ("tup_dot", f32, [P(f32)] * 2), ("tup_cross", None, [P(f32)] * 3), ("tup_magnitude", f32, [P(f32)]),
                ("tup_normalize", None, [P(f32)] * 2), ("tup_reflect", None, [P(f32)] * 3),
                ("tup_equal", C.c_int, [P(f32)] * 2), ("bounds_intersect", C.c_int, [P(st.Bounds), P(st.Ray)]),
                ("bounds_merge", None, [P(st.Bounds)] * 2), ("bounds_transform", None, [P(st.Bounds), P(st.Mat4), P(st.Bounds)]),
                ("bound_group", None, [P(st.Group), P(st.Bounds)]),
                ("triangle_init", None, [P(f32)] * 3 + [P(st.Triangle)]),
            ]:
                fn = getattr(L, "oracle_" + n)
                fn.restype = res
                fn.argtypes = args

    def f(self, n):
        return getattr(self.L, self.p + n)

    # matrices are numpy 4x4 float32 in / out
    def _m(self, fn, *args):
        out = st.Mat4()
        self.f(fn)(*args, C.byref(out))
        return out.to_numpy()

    def mul(self, a, b):
        return self._m("mat_mul", C.byref(st.Mat4.from_rows(a)), C.byref(st.Mat4.from_rows(b)))

    def mul_tuple(self, a, t):
        out = (f32 * 4)()
        self.f("mat_mul_tuple")(C.byref(st.Mat4.from_rows(a)), vec(t), out)
        return np.array(out[:], dtype=np.float32)

    def transpose(self, a):
        return self._m("mat_transpose", C.byref(st.Mat4.from_rows(a)))

    def inverse(self, a):
        return self._m("mat_inverse", C.byref(st.Mat4.from_rows(a)))

    def identity(self):
        return self._m("mat_identity")

    def determinant(self, a):
        a = np.asarray(a, dtype=np.float32)
        n = a.shape[0]
        flat = (f32 * (n * n))(*a.ravel().tolist())
        return self.f("mat_determinant_n")(flat, n)

    def minor(self, a, r, c):
        a = np.asarray(a, dtype=np.float32)
        n = a.shape[0]
        return self.f("mat_minor_n")((f32 * (n * n))(*a.ravel().tolist()), n, r, c)

    def cofactor(self, a, r, c):
        a = np.asarray(a, dtype=np.float32)
        n = a.shape[0]
        return self.f("mat_cofactor_n")((f32 * (n * n))(*a.ravel().tolist()), n, r, c)

    def translation(self, x, y, z):
        return self._m("mat_translation", x, y, z)

    def scaling(self, x, y, z):
        return self._m("mat_scaling", x, y, z)

    def rotation_x(self, r):
        return self._m("mat_rotation_x", r)

    def rotation_y(self, r):
        return self._m("mat_rotation_y", r)

    def rotation_z(self, r):
        return self._m("mat_rotation_z", r)

    def shearing(self, *a):
        return self._m("mat_shearing", *a)

    def dot(self, a, b):
        return self.f("tup_dot")(vec(a), vec(b))

    def _t(self, fn, *args):
        out = (f32 * 4)()
        self.f(fn)(*[vec(a) for a in args], out)
        return np.array(out[:], dtype=np.float32)

    def cross(self, a, b):
        return self._t("tup_cross", a, b)

    def normalize(self, a):
        return self._t("tup_normalize", a)

    def reflect(self, a, n):
        return self._t("tup_reflect", a, n)

    def magnitude(self, a):
        return self.f("tup_magnitude")(vec(a))

    def equal(self, a, b):
        return bool(self.f("tup_equal")(vec(a), vec(b)))

    def mat_equal(self, a, b):
        return all(self.equal(a[r], b[r]) for r in range(4))

    def bounds_intersect(self, mn, mx, origin, direction):
        b = st.Bounds()
        b.minimum[:] = list(mn)
        b.maximum[:] = list(mx)
        r = st.Ray()
        r.origin[:] = list(origin)
        r.direction[:] = list(direction)
        return bool(self.f("bounds_intersect")(C.byref(b), C.byref(r)))

    def bounds_merge(self, a, b):
        x, y = st.Bounds(), st.Bounds()
        x.minimum[:], x.maximum[:] = list(a[0]), list(a[1])
        y.minimum[:], y.maximum[:] = list(b[0]), list(b[1])
        self.f("bounds_merge")(C.byref(x), C.byref(y))
        return np.array(x.minimum[:], dtype=np.float32), np.array(x.maximum[:], dtype=np.float32)

    def bounds_transform(self, mn, mx, m):
        b, out = st.Bounds(), st.Bounds()
        b.minimum[:], b.maximum[:] = list(mn), list(mx)
        self.f("bounds_transform")(C.byref(b), C.byref(st.Mat4.from_rows(m)), C.byref(out))
        return np.array(out.minimum[:], dtype=np.float32), np.array(out.maximum[:], dtype=np.float32)

    def bound_group(self, group):
        out = st.Bounds()
        self.f("bound_group")(C.byref(group), C.byref(out))
        return np.array(out.minimum[:], dtype=np.float32), np.array(out.maximum[:], dtype=np.float32)

    def triangle_init(self, p0, p1, p2):
        t = st.Triangle()
        self.f("triangle_init")(vec(p0), vec(p1), vec(p2), C.byref(t))
        return t
