// rt.hpp — the host-side C++ API of the renderer, mirroring the names the reference exports
// from src/main.zig:3-19 (mat, tup, obj; Camera/camera/render, Canvas, Color, Cone, Cube,
// Cylinder, Group, Material, Pattern + stripe/gradient/ring/checker, Plane, Ray, Sphere,
// Triangle, World/colorAt).  The reference's toolchain (Zig 0.9) is not available in this
// image, so the host side above the C ABI is C++; a scene written against the reference
// reads the same here:
//
//     rt::Sphere middle;
//     middle.common_attrs.transform = rt::mat::translation(-0.5f, 1, 0.5f);
//     middle.common_attrs.material.color = {0.1f, 1, 0.5f};
//     rt::World world;  world.spheres = {floor, middle};
//     auto camera = rt::camera(1000, 500, fov, from, to, up);
//     rt::Canvas canvas = rt::render(camera, world);      // runs on the MI355X
//     canvas.toPPM(file);
//
// Value types are thin wrappers over the plain structs of include/rt_scene.h (same layout),
// so a World hands itself to the C ABI without copying shapes.
#pragma once

#include <array>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_api.h"
#include "rt_math.hpp"

namespace rt {

using Color = std::array<float, 3>;  // src/canvas.zig:6

struct Pattern;
using ColorMap = int;                                  // src/pattern.zig:17 — tag of the function
constexpr ColorMap stripe = RT_PATTERN_STRIPE;         // src/pattern.zig:33
constexpr ColorMap gradient = RT_PATTERN_GRADIENT;     // src/pattern.zig:37
constexpr ColorMap ring = RT_PATTERN_RING;             // src/pattern.zig:43
constexpr ColorMap checker = RT_PATTERN_CHECKER;       // src/pattern.zig:48

inline void set(float (&dst)[3], const Color& c) {
    dst[0] = c[0];
    dst[1] = c[1];
    dst[2] = c[2];
}

// src/pattern.zig:14-17
struct Pattern : rt_pattern {
    Pattern(Color a_, Color b_, ColorMap map, mat::Matrix transform_ = mat::identity()) {
        set(a, a_);
        set(b, b_);
        transform = transform_;
        color_map = map;
    }
};

// src/material.zig:10-37
struct Material : rt_material {
    Material() { rt_default_material(this); }
    Material& setColor(Color c) {
        set(color, c);
        return *this;
    }
    Material& setPattern(const Pattern& p) {
        has_pattern = 1;
        pattern = p;
        return *this;
    }
};

inline Material& material(rt_attrs& a) { return static_cast<Material&>(a.material); }

// src/shape.zig:20-23 and the six shapes; each default-constructs like `rt.Sphere{}`.
struct Sphere : rt_sphere {
    Sphere() { rt_default_attrs(&common_attrs); }
};
struct Plane : rt_plane {
    Plane() { rt_default_attrs(&common_attrs); }
};
struct Cube : rt_cube {
    Cube() { rt_default_attrs(&common_attrs); }
};
struct Cylinder : rt_cylinder {  // src/cylinder.zig:8-12
    Cylinder() {
        rt_default_attrs(&common_attrs);
        minimum = -INFINITY;
        maximum = INFINITY;
        closed = 0;
    }
};
struct Cone : rt_cone {  // src/cone.zig:8-12
    Cone() {
        rt_default_attrs(&common_attrs);
        minimum = -INFINITY;
        maximum = INFINITY;
        closed = 0;
    }
};
struct Triangle : rt_triangle {  // src/triangle.zig:8-30
    static Triangle init(tup::Point p0, tup::Point p1, tup::Point p2) {
        Triangle t;
        rt_default_attrs(&t.common_attrs);
        rt_triangle_init(p0.v, p1.v, p2.v, &t);
        return t;
    }

  private:
    Triangle() = default;
};

static_assert(sizeof(Sphere) == sizeof(rt_sphere) && sizeof(Cylinder) == sizeof(rt_cylinder) &&
                  sizeof(Triangle) == sizeof(rt_triangle) && sizeof(Material) == sizeof(rt_material),
              "wrappers must not add state");

struct Ray {  // src/ray.zig:5-18
    tup::Point origin;
    tup::Vector direction;
    tup::Point position(float t) const { return origin + direction * t; }
    Ray transform(const mat::Matrix& m) const { return Ray{mat::mul(m, origin), mat::mul(m, direction)}; }
};

struct PointLight {  // src/light.zig:5-8
    tup::Point position = tup::point(-10, 10, -10);
    Color intensity = {1, 1, 1};
};

// src/group.zig:16-53
struct Group {
    mat::Matrix transform = mat::identity();
    std::optional<rt_bounds> bounding_box;
    std::vector<Sphere> spheres;
    std::vector<Plane> planes;
    std::vector<Cube> cubes;
    std::vector<Cylinder> cylinders;
    std::vector<Cone> cones;
    std::vector<Triangle> triangles;
    std::vector<Group> subgroups;

    static Group init(const mat::Matrix& transform) {
        Group g;
        g.transform = transform;
        return g;
    }
    void bound();  // bnd.boundGroup, src/bounds.zig:85-107
};

// A view of Groups as the C ABI's rt_group tree; owns the rt_group arrays.
struct GroupView {
    std::vector<std::vector<rt_group>> storage;
    std::vector<Group> keep;  // optional: groups this view owns
    const rt_group* build(const std::vector<Group>& groups);
};

// src/obj_file.zig — rt.obj of src/main.zig:5.
namespace obj {
struct ParserError : std::runtime_error {  // src/obj_file.zig:25-27 (+ the number / index failures)
    int code;                              // RT_OBJ_*
    ParserError(int c, const char* what) : std::runtime_error(what), code(c) {}
};
struct ObjFile {
    std::shared_ptr<rt_obj> handle;
    Group toShapeGroup() const;  // :45-62 — caller owns the group, bound() is the caller's call
    uint32_t vertexCount() const { return rt_obj_vertex_count(handle.get()); }
    uint32_t faceCount() const { return rt_obj_face_count(handle.get()); }
};
ObjFile parseObjFile(const std::string& text);  // :77-147; throws ParserError
}  // namespace obj

// src/world.zig:22-36
struct World {
    PointLight light;
    std::vector<Sphere> spheres;
    std::vector<Plane> planes;
    std::vector<Cube> cubes;
    std::vector<Cylinder> cylinders;
    std::vector<Cone> cones;
    std::vector<Triangle> triangles;
    std::vector<Group> groups;

    rt_world view(GroupView& gv) const;
};

using Camera = rt_camera;  // src/camera.zig:17-23

inline Camera camera(uint32_t horizontal_size, uint32_t vertical_size, float field_of_view, tup::Point from,
                     tup::Point to, tup::Vector up) {  // src/camera.zig:25-51
    Camera c;
    rt_camera_init(horizontal_size, vertical_size, field_of_view, from.v, to.v, up.v, &c);
    return c;
}

inline Ray rayForPixel(const Camera& c, uint32_t px, uint32_t py) {  // src/camera.zig:70-82
    rt_ray r;
    rt_ray_for_pixel(&c, px, py, &r);
    return Ray{tup::from(r.origin), tup::from(r.direction)};
}

// src/canvas.zig
struct Canvas {
    uint32_t width = 0, height = 0;
    std::vector<float> pixels;  // width*height*3, row-major

    Canvas(uint32_t w, uint32_t h) : width(w), height(h), pixels((size_t)w * h * 3, 0.0f) {}
    void writePixel(uint32_t x, uint32_t y, Color c) {
        float* p = &pixels[3 * ((size_t)y * width + x)];
        p[0] = c[0];
        p[1] = c[1];
        p[2] = c[2];
    }
    Color pixelAt(uint32_t x, uint32_t y) const {
        const float* p = &pixels[3 * ((size_t)y * width + x)];
        return Color{p[0], p[1], p[2]};
    }
    void toPPM(std::ostream& writer) const {  // src/canvas.zig:42-50
        std::string buf(rt_canvas_to_ppm(pixels.data(), width, height, nullptr, 0), '\0');
        rt_canvas_to_ppm(pixels.data(), width, height, buf.data(), buf.size());
        writer.write(buf.data(), (std::streamsize)buf.size());
    }
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const char* what) : std::runtime_error(what), code(c) {}
};

// src/camera.zig:84-105 — on the GPU; throws rt::Error where the reference returns an error.
inline Canvas render(const Camera& cam, const World& world, rt_ray_counts* counts = nullptr) {
    Canvas image(cam.horizontal_size, cam.vertical_size);
    GroupView gv;
    const rt_world w = world.view(gv);
    const int rc = rt_render(&cam, &w, image.pixels.data(), counts);
    if (rc != 0) throw Error(rc, rtx_last_error());
    return image;
}

// src/world.zig:81-83
inline Color colorAt(const World& world, const Ray& r) {
    GroupView gv;
    const rt_world w = world.view(gv);
    rt_ray rr;
    for (int i = 0; i < 4; ++i) {
        rr.origin[i] = r.origin[i];
        rr.direction[i] = r.direction[i];
    }
    Color out;
    const int rc = rt_color_at(&w, &rr, 1, out.data());
    if (rc != 0) throw Error(rc, rtx_last_error());
    return out;
}

}  // namespace rt
