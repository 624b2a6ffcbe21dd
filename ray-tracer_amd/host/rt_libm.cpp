// rt_libm.cpp — host-only half of rt_libm.hpp: Zig 0.9 std.math.{sin,cos,tan} for f32
// (ports of Go's math.Sin/Cos/Tan, themselves Cephes sinf/cosf/tanf range reduction and
// polynomials evaluated entirely in f32).  Used by the transform constructors
// (src/matrix.zig:188-213) and camera() (src/camera.zig:39) only — never on the device.
#include "rt_libm.hpp"
#include "rt_math.hpp"

namespace rt {
namespace zmath {

namespace {
// sin / cos polynomial coefficients and the three-part pi/4.
constexpr float S[6] = {1.58962301576546568060E-10f, -2.50507477628578072866E-8f, 2.75573136213857245213E-6f,
                        -1.98412698295895385996E-4f, 8.33333333332211858878E-3f,  -1.66666666666666307295E-1f};
constexpr float Cc[6] = {-1.13585365213876817300E-11f, 2.08757008419747316778E-9f, -2.75573141792967388112E-7f,
                         2.48015872888517045348E-5f,   -1.38888888888730564116E-3f, 4.16666666666665929218E-2f};
constexpr float pi4a = 7.85398125648498535156e-1f;
constexpr float pi4b = 3.77489470793079817668E-8f;
constexpr float pi4c = 2.69515142907905952645E-15f;
constexpr float m4pi = 1.273239544735162542821171882678754627704620361328125f;

constexpr float Tp[3] = {-1.30936939181383777646E4f, 1.15351664838587416140E6f, -1.79565251976484877988E7f};
constexpr float Tq[5] = {0.0f, 1.36812963470692954678E4f, -1.32089234440210967447E6f, 2.50083801823357915839E7f,
                         -5.38695755929454629881E7f};

inline float sin_poly(float z, float w) {
    return z + z * w * (S[5] + w * (S[4] + w * (S[3] + w * (S[2] + w * (S[1] + w * S[0])))));
}
inline float cos_poly(float w) {
    return 1.0f - 0.5f * w + w * w * (Cc[5] + w * (Cc[4] + w * (Cc[3] + w * (Cc[2] + w * (Cc[1] + w * Cc[0])))));
}

struct Reduced {
    float z;
    int32_t j;
};

// Octant j (made even) and the reduced argument z = x - y*pi/4 in three steps.
inline Reduced reduce(float x) {
    float y = floorf(x * m4pi);
    int32_t j = static_cast<int32_t>(y);
    if ((j & 1) == 1) {
        j += 1;
        y += 1;
    }
    return Reduced{((x - y * pi4a) - y * pi4b) - y * pi4c, j};
}
}  // namespace

float sin(float x) {
    if (x == 0 || x != x) return x;
    if (std::isinf(x)) return NAN;
    bool sign = x < 0;
    const Reduced r = reduce(fabsf(x));
    int32_t j = r.j & 7;
    if (j > 3) {
        j -= 4;
        sign = !sign;
    }
    const float w = r.z * r.z;
    const float v = (j == 1 || j == 2) ? cos_poly(w) : sin_poly(r.z, w);
    return sign ? -v : v;
}

float cos(float x) {
    if (x != x || std::isinf(x)) return NAN;
    bool sign = false;
    const Reduced r = reduce(fabsf(x));
    int32_t j = r.j & 7;
    if (j > 3) {
        j -= 4;
        sign = !sign;
    }
    if (j > 1) sign = !sign;
    const float w = r.z * r.z;
    const float v = (j == 1 || j == 2) ? sin_poly(r.z, w) : cos_poly(w);
    return sign ? -v : v;
}

float tan(float x) {
    if (x == 0 || x != x) return x;
    if (std::isinf(x)) return NAN;
    const bool sign = x < 0;
    const Reduced r = reduce(fabsf(x));
    const float z = r.z;
    const float w = z * z;
    float v = (w > 1e-14f)
                  ? z + z * (w * ((Tp[0] * w + Tp[1]) * w + Tp[2]) / ((((w + Tq[1]) * w + Tq[2]) * w + Tq[3]) * w + Tq[4]))
                  : z;
    if ((r.j & 2) == 2) v = -1 / v;
    return sign ? -v : v;
}

float pow(float x, float y) { return pow_f32(x, y); }

}  // namespace zmath
}  // namespace rt
