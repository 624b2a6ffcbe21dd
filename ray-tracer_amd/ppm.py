"""PPM files as the reference writes them (Canvas.toPPM, src/canvas.zig:42-50) and the image
comparison BASELINE.json's tolerance is stated in ("within +-1 LSB on >= 99.9 % of pixels").

The P3 text layout is the reference's: "P3\\n{W} {H}\\n255\\n", then one pixel per line as
"{r} {g} {b}\\n".  Writing goes through the library's rt_canvas_to_ppm so the bytes are the
product's; reading accepts any whitespace-separated P3 (and binary P6).
"""
import ctypes as C

import numpy as np


def write_p3(canvas, path=None):
    """f32 canvas [H, W, 3] -> P3 bytes (rt_canvas_to_ppm); also written to `path` if given."""
    from . import f32, lib, P

    c = np.ascontiguousarray(canvas, dtype=np.float32)
    h, w, _ = c.shape
    L = lib()
    need = L.rt_canvas_to_ppm(c.ctypes.data_as(P(f32)), w, h, None, 0)
    buf = C.create_string_buffer(need)
    L.rt_canvas_to_ppm(c.ctypes.data_as(P(f32)), w, h, buf, need)
    data = buf.raw[:need]
    if path:
        with open(path, "wb") as f:
            f.write(data)
    return data


def write_p6(rgb8, path):
    """u8 image [H, W, 3] -> binary PPM (not a reference format; 4x smaller for big frames)."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def _header_tokens(data, n):
    """First n whitespace-separated tokens (comments skipped) and the offset after them."""
    toks, i = [], 0
    while len(toks) < n:
        while i < len(data) and data[i:i + 1].isspace():
            i += 1
        if data[i:i + 1] == b"#":
            while i < len(data) and data[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while j < len(data) and not data[j:j + 1].isspace():
            j += 1
        if j == i:
            raise ValueError("truncated PPM header")
        toks.append(data[i:j])
        i = j
    return toks, i


def read(path_or_bytes):
    """P3 or P6 -> uint8 [H, W, 3].  Raises ValueError on anything else (maxval must be 255)."""
    data = path_or_bytes
    if not isinstance(data, (bytes, bytearray)):
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    toks, off = _header_tokens(data, 4)
    magic, w, h, maxval = toks[0], int(toks[1]), int(toks[2]), int(toks[3])
    if maxval != 255:
        raise ValueError("maxval %d is not 255" % maxval)
    if magic == b"P3":
        vals = np.array(data[off:].split(), dtype=np.int64)
        if vals.size != w * h * 3:
            raise ValueError("P3 holds %d samples, header says %d" % (vals.size, w * h * 3))
        if vals.size and (vals.min() < 0 or vals.max() > 255):
            raise ValueError("P3 sample out of range")
        return vals.astype(np.uint8).reshape(h, w, 3)
    if magic == b"P6":
        body = data[off + 1:]  # exactly one whitespace byte after maxval
        if len(body) < w * h * 3:
            raise ValueError("P6 body is short")
        return np.frombuffer(body, dtype=np.uint8, count=w * h * 3).reshape(h, w, 3).copy()
    raise ValueError("not a P3/P6 file")


def diff(a, b, mask=None):
    """Pixel statistics of two u8 images: a pixel is `exact` when all three channels are
    equal and `within1` when every channel differs by at most 1 LSB.  `mask` (bool [H, W],
    True = compare) restricts the comparison.  Returns a dict."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        raise ValueError("shapes differ: %s vs %s" % (a.shape, b.shape))
    d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=2)
    if mask is not None:
        d = d[mask]
    n = int(d.size)
    return {
        "pixels": n,
        "exact": int((d == 0).sum()),
        "within1": int((d <= 1).sum()),
        "max_abs": int(d.max()) if n else 0,
        "frac_exact": float((d == 0).mean()) if n else 1.0,
        "frac_within1": float((d <= 1).mean()) if n else 1.0,
    }


def within_tolerance(a, b, frac=0.999, mask=None):
    """BASELINE.json's gate: +-1 LSB on >= 99.9 % of pixels."""
    return diff(a, b, mask)["frac_within1"] >= frac
