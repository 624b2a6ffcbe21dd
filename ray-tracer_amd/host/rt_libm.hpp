// rt_libm.hpp — the libm routines the reference calls but does not vendor, restated so the
// host (scene construction) and the device (Phong specular, Schlick) evaluate them with
// the same f32 operations.
//
// Zig 0.9's std.math.pow / sin / cos / tan are ports of Go's math.Pow / Sin / Cos / Tan
// (the latter three Cephes kernels); call sites: src/material.zig:96,
// src/intersection.zig:179 (pow), src/camera.zig:39 (tan), src/matrix.zig:188-213
// (@sin/@cos lower to the same std.math routines in a libc-free Zig build).  The Zig
// version is unpinned (no build.zig.zon); the dialect is 0.9.x.
//
// pow with an integer-valued exponent (every exponent the samples use: shininess 200/300,
// Schlick's 5) is repeated squaring on the frexp'd significand — pure f32 multiplies,
// hence bit-identical on CPU and GPU.  A fractional exponent additionally needs
// exp(yf*ln(x)); device expf/logf differ from the host's by a few ulp there.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt {
namespace zmath {

RT_HD inline bool is_odd_integer(float y) {
    float ip;
    const float fp = modff(y, &ip);
    return fp == 0 && ((static_cast<int64_t>(ip)) & 1) == 1;
}

// std.math.pow(f32, x, y)
RT_HD inline float pow_f32(float x, float y) {
    if (y == 0 || x == 1) return 1;
    if (x != x || y != y) return NAN;
    if (y == 1) return x;
    if (x == 0) {
        if (y < 0) return is_odd_integer(y) ? copysignf(INFINITY, x) : INFINITY;
        return is_odd_integer(y) ? x : 0.0f;
    }
    if (y == INFINITY || y == -INFINITY) {
        if (x == -1) return 1.0f;
        if ((fabsf(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (x == INFINITY || x == -INFINITY) {
        if (x < 0) {  // pow(-inf, y) = pow(-0, -y)
            if (y > 0) return is_odd_integer(-y) ? -INFINITY : INFINITY;
            return is_odd_integer(-y) ? -0.0f : 0.0f;
        }
        if (y < 0) return 0;
        if (y > 0) return INFINITY;
    }
    if (y == 0.5f) return sqrtf(x);
    if (y == -0.5f) return 1 / sqrtf(x);

    float yi;
    float yf = modff(fabsf(y), &yi);
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 2147483648.0f) return expf(y * logf(x));

    float a1 = 1.0f;
    int32_t ae = 0;
    if (yf != 0) {
        if (yf > 0.5f) {
            yf -= 1;
            yi += 1;
        }
        a1 = expf(yf * logf(x));
    }
    int xe;
    float x1 = frexpf(x, &xe);
    for (int32_t i = static_cast<int32_t>(yi); i != 0; i >>= 1) {
        const int overflow_shift = 8 + 1;  // floatExponentBits(f32) + 1
        if (xe < -(1 << overflow_shift) || (1 << overflow_shift) < xe) {
            ae += xe;
            break;
        }
        if ((i & 1) == 1) {
            a1 *= x1;
            ae += xe;
        }
        x1 *= x1;
        xe <<= 1;
        if (x1 < 0.5f) {
            x1 += x1;
            xe -= 1;
        }
    }
    if (y < 0) {
        a1 = 1 / a1;
        ae = -ae;
    }
    return ldexpf(a1, ae);
}

}  // namespace zmath
}  // namespace rt
