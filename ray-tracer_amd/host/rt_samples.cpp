// rt_samples.cpp — the reference's sample scenes (samples/<name>.zig) written against the
// C++ mirror of its API (rt.hpp).  Scene definitions only: these are the inputs of the hot
// path that BASELINE.json's configs name.  samples/main.cpp renders them to PPM files like
// the reference's sample programs do; tests fetch them through rt_sample_create().
#include <cstring>
#include <fstream>
#include <sstream>
#include <map>
#include <memory>
#include <string>

#include "rt.hpp"

using namespace rt;
using mat::mul;
using mat::rotationX;
using mat::rotationY;
using mat::rotationZ;
using mat::scaling;
using mat::translation;
using tup::point;
using tup::vector;

namespace {

constexpr double pi = 3.14159265358979323846264338327950288;  // std.math.pi; coerced to f32 per use

struct Shot {  // the camera block every sample ends with
    uint32_t width, height;
    float field_of_view;
    tup::Point from, to;
    tup::Tuple up;
};

struct Built {
    World world;
    Shot shot;
};

Built sphere_world() {  // samples/sphere_world.zig
    Sphere floor;
    floor.common_attrs.transform = scaling(10, 0.01f, 10);
    material(floor.common_attrs).setColor({1, 0.9f, 0.9f}).specular = 0;

    Sphere left_wall;
    left_wall.common_attrs.transform = mul(mul(translation(0, 0, 5), rotationY((float)(-pi / 4.0))),
                                           mul(rotationX((float)(pi / 2.0)), scaling(10, 0.01f, 10)));
    left_wall.common_attrs.material = floor.common_attrs.material;

    Sphere right_wall;
    right_wall.common_attrs.transform = mul(mul(translation(0, 0, 5), rotationY((float)(pi / 4.0))),
                                            mul(rotationX((float)(pi / 2.0)), scaling(10, 0.01f, 10)));
    right_wall.common_attrs.material = floor.common_attrs.material;

    Sphere middle;
    middle.common_attrs.transform = translation(-0.5f, 1, 0.5f);
    material(middle.common_attrs).setColor({0.1f, 1, 0.5f});
    middle.common_attrs.material.diffuse = 0.7f;
    middle.common_attrs.material.specular = 0.3f;

    Sphere right;
    right.common_attrs.transform = mul(translation(1.5f, 0.5f, -0.5f), scaling(0.5f, 0.5f, 0.5f));
    material(right.common_attrs).setColor({0.5f, 1, 0.1f});
    right.common_attrs.material.diffuse = 0.7f;
    right.common_attrs.material.specular = 0.3f;

    Sphere left;
    left.common_attrs.transform = mul(translation(-1.5f, 0.33f, -0.75f), scaling(0.33f, 0.33f, 0.33f));
    material(left.common_attrs).setColor({1, 0.8f, 0.1f});
    left.common_attrs.material.diffuse = 0.7f;
    left.common_attrs.material.specular = 0.3f;

    Built b;
    b.world.light = {point(-10, 10, -10), {1, 1, 1}};
    b.world.spheres = {floor, left_wall, right_wall, middle, right, left};
    b.shot = {1000, 500, (float)(pi / 3.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0)};
    return b;
}

Built glass_spheres() {  // samples/glass_spheres.zig
    Plane background;
    background.common_attrs.transform = mul(translation(0, 0, 10), rotationX((float)(pi / 2.0)));
    Material& bm = material(background.common_attrs);
    bm.ambient = 0.8f;
    bm.diffuse = 0.2f;
    bm.specular = 0.0f;
    bm.setPattern(Pattern({0.15f, 0.15f, 0.15f}, {0.85f, 0.85f, 0.85f}, checker));

    Sphere outer_sphere;
    Material& om = material(outer_sphere.common_attrs);
    om.setColor({1, 1, 1});
    om.ambient = 0.0f;
    om.diffuse = 0.0f;
    om.specular = 0.9f;
    om.shininess = 300.0f;
    om.reflective = 0.9f;
    om.transparency = 0.9f;
    om.refractive_index = 1.5f;

    Sphere inner_sphere;
    inner_sphere.common_attrs.transform = scaling(0.5f, 0.5f, 0.5f);
    inner_sphere.common_attrs.material = outer_sphere.common_attrs.material;
    inner_sphere.common_attrs.material.refractive_index = 1.0003f;

    Built b;
    b.world.light = {point(-2, 5, -10), {0.9f, 0.9f, 0.9f}};
    b.world.planes = {background};
    b.world.spheres = {outer_sphere, inner_sphere};
    b.shot = {600, 600, 0.45f, point(0, 0, -5), point(0, 0, 0), point(0, 1, 0)};
    return b;
}

Built fresnel() {  // samples/fresnel.zig
    Plane bottom;
    bottom.common_attrs.transform = translation(0, -15, 0);
    material(bottom.common_attrs).ambient = 1.0f;
    material(bottom.common_attrs).setPattern(Pattern({0.85f, 0.1f, 0.1f}, {0.3f, 0.85f, 0.15f}, checker, scaling(3, 3, 3)));

    Plane water;
    Material& wm = material(water.common_attrs);
    wm.diffuse = 0.1f;
    wm.shininess = 300.0f;
    wm.refractive_index = 1.33f;
    wm.transparency = 1.0f;
    wm.reflective = 0.9f;

    Plane far;
    far.common_attrs.transform = mul(translation(0, 0, 30), rotationX((float)(pi / 2.0)));
    Material& fm = material(far.common_attrs);
    fm.transparency = 0.75f;
    fm.specular = 0.0f;
    fm.setPattern(Pattern({1, 1, 1}, {0.5f, 0.5f, 0.5f}, checker, scaling(0.25f, 0.25f, 0.25f)));

    Built b;
    b.world.light = {point(0, 30, -15), {1, 1, 1}};
    b.world.planes = {bottom, water, far};
    b.shot = {512, 256, (float)(pi / 2.0), point(0, 1.5f, -15), point(0, 1, 0), vector(0, 1, 0)};
    return b;
}

Cylinder cylinder(float minimum, float maximum, bool closed, const mat::Matrix& transform, Color color, float ambient,
                  float diffuse) {
    Cylinder c;
    c.minimum = minimum;
    c.maximum = maximum;
    c.closed = closed ? 1 : 0;
    c.common_attrs.transform = transform;
    Material& m = material(c.common_attrs);
    m.setColor(color);
    m.ambient = ambient;
    m.diffuse = diffuse;
    m.specular = 0.9f;
    m.shininess = 300.0f;
    return c;
}

Built cylinder_world() {  // samples/cylinder_world.zig
    Plane floor;
    Material& fm = material(floor.common_attrs);
    fm.ambient = 0.2f;
    fm.diffuse = 0.9f;
    fm.specular = 0.0f;
    fm.setPattern(Pattern({0.5f, 0.5f, 0.5f}, {0.75f, 0.75f, 0.75f}, checker,
                          mul(rotationY(0.3f), scaling(0.25f, 0.25f, 0.25f))));

    Cylinder big_cylinder = cylinder(0.0f, 0.75f, true, mul(translation(-1, 0, 1), scaling(0.5f, 1, 0.5f)),
                                     {0, 0, 0.6f}, 0.1f, 0.1f);
    big_cylinder.common_attrs.material.reflective = 0.9f;

    const Cylinder a = cylinder(0.0f, 0.2f, false, mul(translation(1, 0, 0), scaling(0.8f, 1, 0.8f)), {1, 1, 0.3f}, 0.1f, 0.8f);
    const Cylinder b_ = cylinder(0.0f, 0.3f, false, mul(translation(1, 0, 0), scaling(0.6f, 1, 0.6f)), {1, 0.9f, 0.4f}, 0.1f, 0.8f);
    const Cylinder c = cylinder(0.0f, 0.4f, false, mul(translation(1, 0, 0), scaling(0.4f, 1, 0.4f)), {1, 0.8f, 0.5f}, 0.1f, 0.8f);
    const Cylinder d = cylinder(0.0f, 0.5f, true, mul(translation(1, 0, 0), scaling(0.2f, 1, 0.2f)), {1, 0.7f, 0.6f}, 0.1f, 0.8f);

    const Cylinder w = cylinder(0.0f, 0.3f, true, mul(translation(0, 0, -0.75f), scaling(0.05f, 1, 0.05f)), {1, 0, 0}, 0.1f, 0.9f);
    auto fan = [](float angle) {
        return mul(mul(translation(0, 0, -2.25f), rotationY(angle)), mul(translation(0, 0, 1.5f), scaling(0.05f, 1, 0.05f)));
    };
    const Cylinder x = cylinder(0.0f, 0.3f, true, fan(-0.15f), {1, 1, 0}, 0.1f, 0.9f);
    const Cylinder y = cylinder(0.0f, 0.3f, true, fan(-0.3f), {0, 1, 0}, 0.1f, 0.9f);
    const Cylinder z = cylinder(0.0f, 0.3f, true, fan(-0.45f), {0, 1, 1}, 0.1f, 0.9f);

    Cylinder glass_cylinder = cylinder(0.0001f, 0.5f, true, mul(translation(0, 0, -1.5f), scaling(0.33f, 1, 0.33f)),
                                       {0.25f, 0, 0}, 0.1f, 0.1f);
    glass_cylinder.common_attrs.material.reflective = 0.9f;
    glass_cylinder.common_attrs.material.transparency = 0.9f;
    glass_cylinder.common_attrs.material.refractive_index = 1.5f;

    Built b;
    b.world.light = {point(1, 6.9f, -4.9f), {1, 1, 1}};
    b.world.planes = {floor};
    b.world.cylinders = {big_cylinder, a, b_, c, d, w, x, y, z, glass_cylinder};
    b.shot = {800, 400, (float)(pi / 10.0), point(8, 3.5f, -9), point(0, 0.3f, 0), vector(0, 1, 0)};
    return b;
}

Built cone() {  // samples/cone.zig
    Cone c;
    c.common_attrs.transform = rotationX((float)(-pi / 6.0));
    c.minimum = -2.0f;
    c.maximum = 2.0f;
    c.closed = 0;
    Built b;
    b.world.light = {point(0, 0, -10), {1, 1, 1}};
    b.world.cones = {c};
    b.shot = {512, 512, (float)(pi / 2.0), point(0, 0, -5), point(0, 0.3f, 0), vector(0, 1, 0)};
    return b;
}

Built cube() {  // samples/cube.zig
    Cube c;
    c.common_attrs.transform = rotationY((float)(pi / 4.0));
    Built b;
    b.world.light = {point(0, 0, -10), {0.9f, 0.9f, 0.9f}};
    b.world.cubes = {c};
    b.shot = {1024, 1024, (float)(pi / 3.0), point(0, 5, -5), point(0, 0, 0), vector(0, 1, 0)};
    return b;
}

Built reflective_floor() {  // samples/reflective_floor.zig
    Plane floor;
    material(floor.common_attrs).setPattern(Pattern({0, 0, 0}, {0.25f, 0.25f, 0.25f}, checker));
    floor.common_attrs.material.reflective = 0.25f;
    Sphere sphere;
    material(sphere.common_attrs).setColor({1, 0.3f, 0.25f});
    sphere.common_attrs.transform = translation(0, 1, 0.5f);
    Built b;
    b.world.light = {point(0, 30, -5), {1, 1, 1}};
    b.world.spheres = {sphere};
    b.world.planes = {floor};
    b.shot = {1024, 1024, (float)(pi / 3.0), point(0, 1.5f, -5), point(0, 1, 0), vector(0, 1, 0)};
    return b;
}

Built skewed_triangle() {  // samples/skewed_triangle.zig
    Triangle triangle = Triangle::init(point(-5, -5, 10), point(0, 2.5f, 0), point(5, -5, 3));
    Material& m = material(triangle.common_attrs);
    m.setColor({1, 1, 0.25f});
    m.ambient = 0.5f;
    m.specular = 0;
    Built b;
    b.world.light = {point(0, 0, -10), {0.9f, 0.9f, 0.9f}};
    b.world.triangles = {triangle};
    b.shot = {512, 512, (float)(pi / 2.0), point(0, 0, -10), point(0, 0, 0), vector(0, 1, 0)};
    return b;
}

Built striped_scene() {  // samples/striped_scene.zig
    const Color blue = {0.05f, 0.25f, 1}, orange = {1, 0.5f, 0.25f};
    Plane floor;
    material(floor.common_attrs).setPattern(Pattern(orange, blue, stripe));
    Sphere sphere;
    material(sphere.common_attrs).setPattern(Pattern(blue, orange, stripe, rotationZ((float)(-pi / 4.0))));
    sphere.common_attrs.transform = translation(0, 1, 0);
    Built b;
    b.world.light = {point(0, 10, -10), {1, 1, 1}};
    b.world.spheres = {sphere};
    b.world.planes = {floor};
    b.shot = {1024, 1024, (float)(pi / 2.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0)};
    return b;
}

Built sphere_on_plane() {  // samples/sphere_on_plane.zig
    Plane floor;
    Plane backdrop;
    backdrop.common_attrs.transform = mul(translation(0, 0, 3), rotationX((float)(pi / 2.0)));
    Sphere sphere;
    sphere.common_attrs.transform = translation(0, 1, 0.5f);
    material(sphere.common_attrs).setColor({0.1f, 0.75f, 1});
    sphere.common_attrs.material.diffuse = 0.7f;
    sphere.common_attrs.material.specular = 0.3f;
    Built b;
    b.world.light = {point(10, 10, -10), {1, 1, 1}};
    b.world.spheres = {sphere};
    b.world.planes = {floor, backdrop};
    b.shot = {1024, 1024, (float)(pi / 4.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0)};
    return b;
}

Built teapot(const std::string& obj_text) {  // samples/teapot.zig
    Group obj_group = obj::parseObjFile(obj_text).toShapeGroup();
    obj_group.bound();
    Built b;
    b.world.light = {point(0, 0, -10), {1, 1, 1}};
    b.world.groups = {obj_group};
    b.shot = {256, 256, (float)(pi / 3.0), point(0, 3, -10), point(0, 0, 0), vector(0, 1, 0)};
    return b;
}

// The scaled stress scene of BASELINE.json configs[4] — NOT a reference scene; defined once
// here and frozen: the teapot group standing on reflective_floor's checkered mirror plane,
// glass_spheres' outer/inner glass pair to its right, reflective_floor's red sphere to its
// left, teapot.zig's camera, light at (-5, 10, -10).
Built composite(const std::string& obj_text) {
    Built tea = teapot(obj_text), glass = glass_spheres(), floor = reflective_floor();
    auto moved = [](Sphere s, float x, float y, float z) {
        s.common_attrs.transform = mul(translation(x, y, z), mat::Matrix(s.common_attrs.transform));
        return s;
    };
    Built b;
    b.world.light = {point(-5, 10, -10), {1, 1, 1}};
    b.world.spheres = {moved(glass.world.spheres[0], 4.2f, 1.0f, -1.5f), moved(glass.world.spheres[1], 4.2f, 1.0f, -1.5f),
                       moved(floor.world.spheres[0], -4.0f, 0.0f, -2.0f)};
    b.world.planes = {floor.world.planes[0]};
    b.world.groups = tea.world.groups;
    b.shot = tea.shot;
    b.shot.width = 7680;
    b.shot.height = 4320;
    return b;
}

using Builder = Built (*)();
const std::map<std::string, Builder>& builders() {
    static const std::map<std::string, Builder> m = {
        {"sphere_world", sphere_world},     {"glass_spheres", glass_spheres},     {"fresnel", fresnel},
        {"cylinder_world", cylinder_world}, {"cone", cone},                       {"cube", cube},
        {"reflective_floor", reflective_floor}, {"skewed_triangle", skewed_triangle},
        {"striped_scene", striped_scene},   {"sphere_on_plane", sphere_on_plane},
    };
    return m;
}

}  // namespace

struct rt_sample {
    World world;
    GroupView groups;
    rt_world view;
    rt_camera camera;
};

extern "C" {

static rt_sample* finish(Built&& b, uint32_t width, uint32_t height);

rt_sample* rt_sample_create(const char* name, uint32_t width, uint32_t height) {
    if (!name) return nullptr;
    if (!std::strcmp(name, "teapot")) {  // like samples/teapot.zig:7: "teapot.obj" in the working directory
        std::ifstream f("teapot.obj", std::ios::binary);
        if (!f) return nullptr;
        std::stringstream text;
        text << f.rdbuf();
        const std::string t = text.str();
        return rt_sample_create_obj(name, t.data(), t.size(), width, height);
    }
    auto it = builders().find(name);
    if (it == builders().end()) return nullptr;
    return finish(it->second(), width, height);
}

rt_sample* rt_sample_create_obj(const char* name, const char* obj_text, uint64_t len, uint32_t width, uint32_t height) {
    if (!name || !obj_text) return nullptr;
    try {
        const std::string text(obj_text, (size_t)len);
        if (!std::strcmp(name, "teapot")) return finish(teapot(text), width, height);
        if (!std::strcmp(name, "composite")) return finish(composite(text), width, height);
    } catch (const obj::ParserError&) {
    }
    return nullptr;
}

static rt_sample* finish(Built&& b, uint32_t width, uint32_t height) {
    rt_sample* s = new rt_sample();
    s->world = std::move(b.world);
    s->view = s->world.view(s->groups);
    if (width == 0 || height == 0) {
        width = b.shot.width;
        height = b.shot.height;
    }
    s->camera = rt::camera(width, height, b.shot.field_of_view, b.shot.from, b.shot.to, b.shot.up);
    return s;
}

const rt_world* rt_sample_world(const rt_sample* s) { return &s->view; }
const rt_camera* rt_sample_camera(const rt_sample* s) { return &s->camera; }
void rt_sample_destroy(rt_sample* s) { delete s; }

}  // extern "C"
