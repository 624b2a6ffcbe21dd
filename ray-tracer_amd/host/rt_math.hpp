// rt_math.hpp — host-side f32 math of the renderer: rt::tup and rt::mat.
//
// Mirrors the reference's `tup` and `mat` namespaces (src/tuple.zig, src/matrix.zig,
// re-exported by src/main.zig:3-5) name for name.  Everything the device consumes that
// is derived from a transform (inverses, composed group transforms, camera inverse) is
// computed HERE, once per scene, with the reference's exact f32 operation order, so the
// pre-computation is bit-identical to what the reference recomputes per ray-shape test.
//
// Must be compiled with -ffp-contract=off and without -ffast-math.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/rt_scene.h"

namespace rt {

// Zig 0.9 std.math.{pow,sin,cos,tan} (ports of Go's math package / Cephes) — the
// un-vendored libm the reference calls.  Defined in rt_libm.cpp.
namespace zmath {
float pow(float x, float y);
float sin(float x);
float cos(float x);
float tan(float x);
}  // namespace zmath

namespace tup {

struct Tuple {  // src/tuple.zig:6 — @Vector(4, f32)
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
using Vector = Tuple;
using Point = Tuple;

constexpr float epsilon = 0.00001f;  // src/tuple.zig:10

inline Tuple tuple(float x, float y, float z, float w) { return Tuple{{x, y, z, w}}; }  // :12
inline Point point(float x, float y, float z) { return Tuple{{x, y, z, 1.0f}}; }        // :16
inline Vector vector(float x, float y, float z) { return Tuple{{x, y, z, 0.0f}}; }      // :20
inline Tuple from(const float* p) { return Tuple{{p[0], p[1], p[2], p[3]}}; }

inline bool isPoint(Tuple t) { return t[3] == 1; }   // :24
inline bool isVector(Tuple t) { return t[3] == 0; }  // :28

inline Tuple operator+(Tuple a, Tuple b) { return Tuple{{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
inline Tuple operator-(Tuple a, Tuple b) { return Tuple{{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}}; }
inline Tuple operator-(Tuple a) { return Tuple{{-a[0], -a[1], -a[2], -a[3]}}; }
inline Tuple operator*(Tuple a, float s) { return Tuple{{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
inline Tuple operator/(Tuple a, float s) { return Tuple{{a[0] / s, a[1] / s, a[2] / s, a[3] / s}}; }

inline bool equal(Tuple a, Tuple b) {  // :32-36
    for (int i = 0; i < 4; ++i)
        if (!(std::fabs(a[i] - b[i]) <= epsilon)) return false;
    return true;
}

// :38-40 — all four lanes, summed left to right.
inline float dot(Vector u, Vector v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2] + u[3] * v[3]; }

// :42-48 — the reference initialises three lanes only; lane w is 0.
inline Vector cross(Vector u, Vector v) {
    return Tuple{{u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0], 0.0f}};
}

inline float magnitude(Vector v) { return std::sqrt(dot(v, v)); }  // :50-52
inline Vector normalize(Vector v) { return v / magnitude(v); }     // :54-56

inline Vector reflect(Vector incident, Vector normal) {  // :58-77
    const Vector orthogonal_component = normal * dot(incident, normal);
    return incident - orthogonal_component * 2.0f;
}

}  // namespace tup

namespace mat {

using Matrix = rt_mat4;  // src/matrix.zig:11-12, row-major

inline Matrix make(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13,
                   float a20, float a21, float a22, float a23, float a30, float a31, float a32, float a33) {
    Matrix m = {{{a00, a01, a02, a03}, {a10, a11, a12, a13}, {a20, a21, a22, a23}, {a30, a31, a32, a33}}};
    return m;
}

inline Matrix identity() { return make(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); }  // :16-21

inline tup::Tuple row(const Matrix& a, int r) { return tup::tuple(a.m[r][0], a.m[r][1], a.m[r][2], a.m[r][3]); }

inline bool equal(const Matrix& a, const Matrix& b) {  // :23-29
    for (int r = 0; r < 4; ++r)
        if (!tup::equal(row(a, r), row(b, r))) return false;
    return true;
}

inline Matrix mulMatrix(const Matrix& a, const Matrix& b) {  // :43-60
    Matrix c;
    for (int r = 0; r < 4; ++r)
        for (int col = 0; col < 4; ++col)
            c.m[r][col] = a.m[r][0] * b.m[0][col] + a.m[r][1] * b.m[1][col] + a.m[r][2] * b.m[2][col] +
                          a.m[r][3] * b.m[3][col];
    return c;
}

inline tup::Tuple mulTuple(const Matrix& a, tup::Tuple b) {  // :62-64
    return tup::tuple(tup::dot(row(a, 0), b), tup::dot(row(a, 1), b), tup::dot(row(a, 2), b), tup::dot(row(a, 3), b));
}

// `mat.mul` is overloaded on the second argument in the reference (:31-41).
inline Matrix mul(const Matrix& a, const Matrix& b) { return mulMatrix(a, b); }
inline tup::Tuple mul(const Matrix& a, tup::Tuple b) { return mulTuple(a, b); }

inline Matrix transpose(const Matrix& m) {  // :74-87
    Matrix n;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) n.m[r][c] = m.m[c][r];
    return n;
}

// determinant / minor / cofactor: first-row cofactor expansion accumulating from 0
// (:89-147).  N is the matrix dimension (2, 3 or 4); storage is row-major N*N.
template <int N>
struct Square {
    float m[N][N];
};

template <int N>
float determinant(const Square<N>& m);

template <int N>
Square<N - 1> submatrix(const Square<N>& m, int row, int col) {  // :109-127
    Square<N - 1> n;
    int y = 0;
    for (int j = 0; j < N; ++j) {
        if (j == row) continue;
        int x = 0;
        for (int i = 0; i < N; ++i) {
            if (i == col) continue;
            n.m[y][x] = m.m[j][i];
            x += 1;
        }
        y += 1;
    }
    return n;
}

template <int N>
float minor_(const Square<N>& m, int row, int col) {  // :137-139
    return determinant(submatrix(m, row, col));
}

template <int N>
float cofactor(const Square<N>& m, int row, int col) {  // :141-147
    if ((row + col) % 2 == 0) return minor_(m, row, col);
    return -minor_(m, row, col);
}

template <>
inline float determinant<2>(const Square<2>& m) {  // :99-100
    return m.m[0][0] * m.m[1][1] - m.m[0][1] * m.m[1][0];
}

template <int N>
float determinant(const Square<N>& m) {  // :101-106
    float det = 0;
    for (int col = 0; col < N; ++col) det += m.m[0][col] * cofactor(m, 0, col);
    return det;
}

inline Square<4> square(const Matrix& a) {
    Square<4> s;
    std::memcpy(s.m, a.m, sizeof s.m);
    return s;
}

inline float determinant(const Matrix& a) { return determinant(square(a)); }
inline float cofactor(const Matrix& a, int row, int col) { return cofactor(square(a), row, col); }
inline bool isInvertible(const Matrix& a) { return determinant(a) != 0; }  // :149-151

// :153-168 — n[col][row] = cofactor(m,row,col) / determinant(m).  determinant(m) is a pure
// function of m, so evaluating it once instead of once per element is bit-identical.
inline Matrix inverse(const Matrix& a) {
    const Square<4> m = square(a);
    const float det = determinant(m);
    Matrix n;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) n.m[c][r] = cofactor(m, r, c) / det;
    return n;
}

inline Matrix translation(float x, float y, float z) { return make(1, 0, 0, x, 0, 1, 0, y, 0, 0, 1, z, 0, 0, 0, 1); }  // :170
inline Matrix scaling(float x, float y, float z) { return make(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1); }      // :179
inline Matrix rotationX(float r) {  // :188
    return make(1, 0, 0, 0, 0, zmath::cos(r), -zmath::sin(r), 0, 0, zmath::sin(r), zmath::cos(r), 0, 0, 0, 0, 1);
}
inline Matrix rotationY(float r) {  // :197
    return make(zmath::cos(r), 0, zmath::sin(r), 0, 0, 1, 0, 0, -zmath::sin(r), 0, zmath::cos(r), 0, 0, 0, 0, 1);
}
inline Matrix rotationZ(float r) {  // :206
    return make(zmath::cos(r), -zmath::sin(r), 0, 0, zmath::sin(r), zmath::cos(r), 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
}
inline Matrix shearing(float xy, float xz, float yx, float yz, float zx, float zy) {  // :215
    return make(1, xy, xz, 0, yx, 1, yz, 0, zx, zy, 1, 0, 0, 0, 0, 1);
}

}  // namespace mat
}  // namespace rt
