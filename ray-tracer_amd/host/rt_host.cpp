// rt_host.cpp — host side of the renderer above the device C ABI (include/rtx.h):
// camera construction, group bounds, the World -> flat-scene flattener, the by-value
// drop-ins of include/rt_api.h, and the math exports the known-answer tests call.
//
// No pixel is shaded here.  Everything in this file runs once per scene (or per frame for
// the camera) and produces the precomputed, bit-exact inputs the kernels consume.
// Compile with -ffp-contract=off, no fast-math.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rt.hpp"

using namespace rt;
using tup::Tuple;

// ------------------------------------------------------------------------- bounds

namespace {

rt_bounds make_bounds(Tuple mn, Tuple mx) {
    rt_bounds b;
    std::memcpy(b.minimum, mn.v, sizeof b.minimum);
    std::memcpy(b.maximum, mx.v, sizeof b.maximum);
    return b;
}

// @minimum/@maximum on vectors: lane-wise minnum/maxnum, all four lanes.
void expand_to_fit(rt_bounds& b, Tuple p) {  // src/bounds.zig:114-117
    for (int i = 0; i < 4; ++i) {
        b.minimum[i] = std::fmin(b.minimum[i], p[i]);
        b.maximum[i] = std::fmax(b.maximum[i], p[i]);
    }
}

void merge(rt_bounds& a, const rt_bounds& b) {  // src/bounds.zig:109-112
    for (int i = 0; i < 4; ++i) {
        a.minimum[i] = std::fmin(a.minimum[i], b.minimum[i]);
        a.maximum[i] = std::fmax(a.maximum[i], b.maximum[i]);
    }
}

// src/bounds.zig:119-133 — starts from the untransformed box and adds six of the eight
// transformed corners (min,min,min and max,max,max are never visited).  Kept as is: the
// box gates intersection tests, so "fixing" it would change pixels.
rt_bounds transform_bounds(const rt_bounds& b, const mat::Matrix& transform) {
    rt_bounds out = b;
    const float *mn = b.minimum, *mx = b.maximum;
    const Tuple corners[6] = {
        tup::tuple(mn[0], mn[1], mx[2], 1), tup::tuple(mn[0], mx[1], mn[2], 1), tup::tuple(mn[0], mx[1], mx[2], 1),
        tup::tuple(mx[0], mn[1], mn[2], 1), tup::tuple(mx[0], mn[1], mx[2], 1), tup::tuple(mx[0], mx[1], mn[2], 1),
    };
    for (const Tuple& c : corners) expand_to_fit(out, mat::mul(transform, c));
    return out;
}

rt_bounds bound_group(const rt_group& g);

template <class S>
void merge_shape(rt_bounds& acc, const rt_bounds& object_box, const rt_group& g, const S& shape) {  // :135-138
    merge(acc, transform_bounds(object_box, mat::mul(g.transform, shape.common_attrs.transform)));
}

rt_bounds bound_group(const rt_group& g) {  // src/bounds.zig:85-107
    const float inf = INFINITY;
    rt_bounds acc = make_bounds(tup::point(inf, inf, inf), tup::point(-inf, -inf, -inf));
    const rt_bounds unit = make_bounds(tup::point(-1, -1, -1), tup::point(1, 1, 1));  // sphere :39, cube :55
    for (uint32_t i = 0; i < g.n_spheres; ++i) merge_shape(acc, unit, g, g.spheres[i]);
    for (uint32_t i = 0; i < g.n_planes; ++i)  // :47-53
        merge_shape(acc, make_bounds(tup::point(-inf, 0, -inf), tup::point(inf, 0, inf)), g, g.planes[i]);
    for (uint32_t i = 0; i < g.n_cubes; ++i) merge_shape(acc, unit, g, g.cubes[i]);
    for (uint32_t i = 0; i < g.n_cylinders; ++i) {  // :63-68
        const rt_cylinder& c = g.cylinders[i];
        merge_shape(acc, make_bounds(tup::point(-1, c.minimum, -1), tup::point(1, c.maximum, 1)), g, c);
    }
    for (uint32_t i = 0; i < g.n_cones; ++i) {  // :70-76
        const rt_cone& c = g.cones[i];
        const float xz = std::fmax(std::fabs(c.minimum), std::fabs(c.maximum));
        merge_shape(acc, make_bounds(tup::point(-xz, c.minimum, -xz), tup::point(xz, c.maximum, xz)), g, c);
    }
    for (uint32_t i = 0; i < g.n_triangles; ++i) {  // :78-83
        const rt_triangle& t = g.triangles[i];
        rt_bounds b;
        for (int k = 0; k < 4; ++k) {
            b.minimum[k] = std::fmin(t.p0[k], std::fmin(t.p1[k], t.p2[k]));
            b.maximum[k] = std::fmax(t.p0[k], std::fmax(t.p1[k], t.p2[k]));
        }
        merge_shape(acc, b, g, t);
    }
    for (uint32_t i = 0; i < g.n_subgroups; ++i)  // :103-104,140-142
        merge(acc, transform_bounds(bound_group(g.subgroups[i]), g.transform));
    return acc;
}

struct MinMax {
    float tmin, tmax;
};

MinMax check_axis(float mn, float mx, float origin, float direction) {  // src/bounds.zig:160-175
    const float tmin_numerator = mn - origin;
    const float tmax_numerator = mx - origin;
    float tmin, tmax;
    if (std::fabs(direction) >= tup::epsilon) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = tmin_numerator * INFINITY;
        tmax = tmax_numerator * INFINITY;
    }
    if (tmin > tmax) return MinMax{tmax, tmin};
    return MinMax{tmin, tmax};
}

}  // namespace

// --------------------------------------------------------------------- C++ facade glue

void Group::bound() {
    GroupView gv;
    std::vector<Group> self{*this};
    const rt_group* g = gv.build(self);
    bounding_box = bound_group(*g);
}

const rt_group* GroupView::build(const std::vector<Group>& groups) {
    std::vector<rt_group> out(groups.size());
    for (size_t i = 0; i < groups.size(); ++i) {
        const Group& g = groups[i];
        rt_group& r = out[i];
        std::memset(&r, 0, sizeof r);
        r.transform = g.transform;
        r.has_bounding_box = g.bounding_box.has_value() ? 1 : 0;
        if (g.bounding_box) r.bounding_box = *g.bounding_box;
        r.n_spheres = (uint32_t)g.spheres.size();
        r.spheres = g.spheres.data();
        r.n_planes = (uint32_t)g.planes.size();
        r.planes = g.planes.data();
        r.n_cubes = (uint32_t)g.cubes.size();
        r.cubes = g.cubes.data();
        r.n_cylinders = (uint32_t)g.cylinders.size();
        r.cylinders = g.cylinders.data();
        r.n_cones = (uint32_t)g.cones.size();
        r.cones = g.cones.data();
        r.n_triangles = (uint32_t)g.triangles.size();
        r.triangles = g.triangles.data();
        r.n_subgroups = (uint32_t)g.subgroups.size();
        r.subgroups = g.subgroups.empty() ? nullptr : build(g.subgroups);
    }
    storage.push_back(std::move(out));
    return storage.back().data();
}

rt_world World::view(GroupView& gv) const {
    rt_world w;
    std::memset(&w, 0, sizeof w);
    std::memcpy(w.light_position, light.position.v, sizeof w.light_position);
    std::memcpy(w.light_intensity, light.intensity.data(), sizeof w.light_intensity);
    w.n_spheres = (uint32_t)spheres.size();
    w.spheres = spheres.data();
    w.n_planes = (uint32_t)planes.size();
    w.planes = planes.data();
    w.n_cubes = (uint32_t)cubes.size();
    w.cubes = cubes.data();
    w.n_cylinders = (uint32_t)cylinders.size();
    w.cylinders = cylinders.data();
    w.n_cones = (uint32_t)cones.size();
    w.cones = cones.data();
    w.n_triangles = (uint32_t)triangles.size();
    w.triangles = triangles.data();
    w.n_groups = (uint32_t)groups.size();
    w.groups = groups.empty() ? nullptr : gv.build(groups);
    return w;
}

// ---------------------------------------------------------------------- flattener

struct rt_flat_scene {
    std::vector<rtx_shape> shapes;
    std::vector<rtx_material> materials;
    std::vector<rtx_pattern> patterns;
    std::vector<rtx_node> nodes;
    rtx_scene_desc desc;
    // for the duplicate-shape rule of the n1/n2 container walk
    std::vector<std::string> eq_keys;
};

namespace {

std::string bytes_of(const void* p, size_t n) { return std::string(static_cast<const char*>(p), n); }

// Floats compare with == in std.meta.eql: fold -0 onto +0; a NaN never equals anything, so
// give every NaN-carrying key a unique suffix.
void push_floats(std::string& key, const float* f, size_t n, bool& has_nan) {
    for (size_t i = 0; i < n; ++i) {
        float v = f[i];
        if (v != v) has_nan = true;
        if (v == 0.0f) v = 0.0f;
        key.append(reinterpret_cast<const char*>(&v), sizeof v);
    }
}

struct Flattener {
    rt_flat_scene& out;
    std::map<std::string, uint32_t> material_ids, pattern_ids;

    uint32_t pattern_id(const rt_pattern& p) {
        const std::string key = bytes_of(&p, sizeof p);
        auto it = pattern_ids.find(key);
        if (it != pattern_ids.end()) return it->second;
        rtx_pattern r;
        std::memset(&r, 0, sizeof r);
        const mat::Matrix inv = mat::inverse(p.transform);  // src/pattern.zig:29
        std::memcpy(r.inverse, inv.m, sizeof r.inverse);
        std::memcpy(r.a, p.a, sizeof r.a);
        std::memcpy(r.b, p.b, sizeof r.b);
        r.color_map = (uint32_t)p.color_map;
        out.patterns.push_back(r);
        return pattern_ids[key] = (uint32_t)out.patterns.size() - 1;
    }

    uint32_t material_id(const rt_material& m) {
        rt_material norm = m;
        if (!norm.has_pattern) std::memset(&norm.pattern, 0, sizeof norm.pattern);
        norm.has_pattern = norm.has_pattern ? 1 : 0;
        const std::string key = bytes_of(&norm, sizeof norm);
        auto it = material_ids.find(key);
        if (it != material_ids.end()) return it->second;
        rtx_material r;
        std::memset(&r, 0, sizeof r);
        std::memcpy(r.color, m.color, sizeof r.color);
        r.ambient = m.ambient;
        r.diffuse = m.diffuse;
        r.specular = m.specular;
        r.shininess = m.shininess;
        r.reflective = m.reflective;
        r.refractive_index = m.refractive_index;
        r.transparency = m.transparency;
        r.pattern = m.has_pattern ? (int32_t)pattern_id(m.pattern) : -1;
        out.materials.push_back(r);
        return material_ids[key] = (uint32_t)out.materials.size() - 1;
    }

    // One shape with its composed transform (src/group.zig:101-106 for members; the shape's
    // own transform at world level).
    rtx_shape& add(uint32_t type, const rt_attrs& attrs, const mat::Matrix& composed) {
        rtx_shape s;
        std::memset(&s, 0, sizeof s);
        const mat::Matrix inv = mat::inverse(composed);
        std::memcpy(s.inverse, inv.m, sizeof s.inverse);
        s.type = type;
        s.material = material_id(attrs.material);
        out.shapes.push_back(s);

        // Equality key of the (transformed) shape as std.meta.eql sees it.
        std::string key;
        bool has_nan = false;
        key.append(reinterpret_cast<const char*>(&type), sizeof type);
        push_floats(key, &composed.m[0][0], 16, has_nan);
        const rt_material& m = attrs.material;
        push_floats(key, m.color, 3, has_nan);
        const float scalars[7] = {m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.refractive_index,
                                  m.transparency};
        push_floats(key, scalars, 7, has_nan);
        key.push_back(m.has_pattern ? 1 : 0);
        if (m.has_pattern) {
            push_floats(key, m.pattern.a, 3, has_nan);
            push_floats(key, m.pattern.b, 3, has_nan);
            push_floats(key, &m.pattern.transform.m[0][0], 16, has_nan);
            key.append(reinterpret_cast<const char*>(&m.pattern.color_map), sizeof m.pattern.color_map);
        }
        if (has_nan) key += "#nan" + std::to_string(out.shapes.size());
        out.eq_keys.push_back(key);
        return out.shapes.back();
    }

    void extra_floats(const float* f, size_t n) {
        bool has_nan = false;
        push_floats(out.eq_keys.back(), f, n, has_nan);
        if (has_nan) out.eq_keys.back() += "#nan" + std::to_string(out.shapes.size());
    }

    template <class S>
    mat::Matrix compose(const mat::Matrix* group, const S& shape) {
        return group ? mat::mul(*group, shape.common_attrs.transform) : shape.common_attrs.transform;
    }

    // The six typed lists in the reference's order (src/world.zig:39-44, src/group.zig:59-70).
    template <class Owner>
    void add_members(const Owner& o, const mat::Matrix* group) {
        for (uint32_t i = 0; i < o.n_spheres; ++i) add(RTX_SPHERE, o.spheres[i].common_attrs, compose(group, o.spheres[i]));
        for (uint32_t i = 0; i < o.n_planes; ++i) add(RTX_PLANE, o.planes[i].common_attrs, compose(group, o.planes[i]));
        for (uint32_t i = 0; i < o.n_cubes; ++i) add(RTX_CUBE, o.cubes[i].common_attrs, compose(group, o.cubes[i]));
        for (uint32_t i = 0; i < o.n_cylinders; ++i) {
            const rt_cylinder& c = o.cylinders[i];
            rtx_shape& s = add(RTX_CYLINDER, c.common_attrs, compose(group, c));
            s.minimum = c.minimum;
            s.maximum = c.maximum;
            if (c.closed) s.flags |= RTX_SHAPE_CLOSED;
            const float e[3] = {c.minimum, c.maximum, c.closed ? 1.0f : 0.0f};
            extra_floats(e, 3);
        }
        for (uint32_t i = 0; i < o.n_cones; ++i) {
            const rt_cone& c = o.cones[i];
            rtx_shape& s = add(RTX_CONE, c.common_attrs, compose(group, c));
            s.minimum = c.minimum;
            s.maximum = c.maximum;
            if (c.closed) s.flags |= RTX_SHAPE_CLOSED;
            const float e[3] = {c.minimum, c.maximum, c.closed ? 1.0f : 0.0f};
            extra_floats(e, 3);
        }
        for (uint32_t i = 0; i < o.n_triangles; ++i) {
            const rt_triangle& t = o.triangles[i];
            rtx_shape& s = add(RTX_TRIANGLE, t.common_attrs, compose(group, t));
            std::memcpy(s.p0, t.p0, sizeof s.p0);
            std::memcpy(s.e0, t.e0, sizeof s.e0);
            std::memcpy(s.e1, t.e1, sizeof s.e1);
            std::memcpy(s.normal, t.normal, sizeof s.normal);
            extra_floats(t.p0, 4);
            extra_floats(t.p1, 4);
            extra_floats(t.p2, 4);
            extra_floats(t.e0, 4);
            extra_floats(t.e1, 4);
            extra_floats(t.normal, 4);
        }
    }

    // Group.intersect (src/group.zig:55-87): gate, own members, then subgroups with
    // transform = mul(parent, sub.transform) (:75).
    void add_group(const rt_group& g, const mat::Matrix& composed) {
        const uint32_t me = (uint32_t)out.nodes.size();
        rtx_node n;
        std::memset(&n, 0, sizeof n);
        n.has_bounds = g.has_bounding_box ? 1u : 0u;
        if (g.has_bounding_box) {
            std::memcpy(n.bounds_min, g.bounding_box.minimum, sizeof n.bounds_min);
            std::memcpy(n.bounds_max, g.bounding_box.maximum, sizeof n.bounds_max);
        }
        n.first_shape = (uint32_t)out.shapes.size();
        out.nodes.push_back(n);
        add_members(g, &composed);
        out.nodes[me].n_shapes = (uint32_t)out.shapes.size() - out.nodes[me].first_shape;
        for (uint32_t i = 0; i < g.n_subgroups; ++i)
            add_group(g.subgroups[i], mat::mul(composed, g.subgroups[i].transform));
        out.nodes[me].skip = (uint32_t)out.nodes.size();
    }

    void run(const rt_world& w) {
        rtx_node root;
        std::memset(&root, 0, sizeof root);
        out.nodes.push_back(root);
        add_members(w, nullptr);
        out.nodes[0].n_shapes = (uint32_t)out.shapes.size();
        for (uint32_t i = 0; i < w.n_groups; ++i) add_group(w.groups[i], w.groups[i].transform);
        // The world node has no box: a ray never skips it, and its own "subtree" is just itself.
        out.nodes[0].skip = 1;

        // Shapes equal under std.meta.eql produce equal Intersections that toggle each other out of the n1/n2
        // container walk (src/intersection.zig:126-137): of m equal shapes a ray intersects, only the last
        // survives, and only when m is odd.  Which of them a ray intersects depends on the group boxes it passes
        // (src/group.zig:58).  Twins behind the SAME chain of boxes are intersected together or not at all, so the
        // outcome is the same for every ray and is recorded here (RTX_SHAPE_N1_INVISIBLE).  Twins behind DIFFERENT
        // chains are resolved per ray on the device: they carry their class number in rtx_shape.reserved.
        std::vector<uint32_t> node_of(out.shapes.size(), 0u);
        for (uint32_t n = 0; n < out.nodes.size(); ++n)
            for (uint32_t k = 0; k < out.nodes[n].n_shapes; ++k) node_of[out.nodes[n].first_shape + k] = n;
        auto gate_chain = [&](uint32_t shape) {  // the boxed nodes a ray must pass to reach the shape, outermost first
            std::vector<uint32_t> chain;
            const uint32_t at = node_of[shape];
            for (uint32_t n = 1; n <= at; ++n)
                if (out.nodes[n].skip > at && out.nodes[n].has_bounds) chain.push_back(n);
            return chain;
        };
        std::map<std::string, std::vector<uint32_t>> classes;
        for (uint32_t i = 0; i < out.eq_keys.size(); ++i) classes[out.eq_keys[i]].push_back(i);
        uint32_t split_classes = 0;
        for (auto& kv : classes) {
            const std::vector<uint32_t>& ids = kv.second;
            if (ids.size() < 2) continue;
            bool same_chain = true;
            const std::vector<uint32_t> first_chain = gate_chain(ids[0]);
            for (size_t k = 1; k < ids.size() && same_chain; ++k) same_chain = gate_chain(ids[k]) == first_chain;
            if (!same_chain) {
                ++split_classes;
                for (uint32_t id : ids) out.shapes[id].reserved = split_classes;
                continue;
            }
            for (size_t k = 0; k < ids.size(); ++k) {
                const bool survivor = (ids.size() & 1) && k + 1 == ids.size();
                if (!survivor) out.shapes[ids[k]].flags |= RTX_SHAPE_N1_INVISIBLE;
            }
        }
        if (out.materials.empty()) {
            rt_material m;
            rt_default_material(&m);
            material_id(m);
        }
    }
};

}  // namespace

// ------------------------------------------------------------------------- C ABI

namespace {
template <int N>
mat::Square<N> load_square(const float* m) {
    mat::Square<N> s;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) s.m[r][c] = m[r * N + c];
    return s;
}
}  // namespace

extern "C" {

void rt_default_material(rt_material* m) {  // src/material.zig:10-37
    std::memset(m, 0, sizeof *m);
    m->color[0] = m->color[1] = m->color[2] = 1.0f;
    m->has_pattern = 0;
    m->pattern.transform = mat::identity();
    m->ambient = 0.1f;
    m->diffuse = 0.9f;
    m->specular = 0.9f;
    m->shininess = 200.0f;
    m->reflective = 0.0f;
    m->refractive_index = 1.0f;
    m->transparency = 0.0f;
}

void rt_default_attrs(rt_attrs* a) {  // src/shape.zig:20-23
    a->transform = mat::identity();
    rt_default_material(&a->material);
}

void rt_triangle_init(const float p0[4], const float p1[4], const float p2[4], rt_triangle* out) {  // src/triangle.zig:19-30
    const Tuple a = tup::from(p0), b = tup::from(p1), c = tup::from(p2);
    const Tuple e0 = b - a;
    const Tuple e1 = c - a;
    const Tuple n = tup::normalize(tup::cross(e1, e0));
    std::memcpy(out->p0, a.v, 16);
    std::memcpy(out->p1, b.v, 16);
    std::memcpy(out->p2, c.v, 16);
    std::memcpy(out->e0, e0.v, 16);
    std::memcpy(out->e1, e1.v, 16);
    std::memcpy(out->normal, n.v, 16);
}

void rt_bound_group(const rt_group* g, rt_bounds* out) { *out = bound_group(*g); }

int rt_bounds_intersect(const rt_bounds* b, const rt_ray* r) {  // src/bounds.zig:149-158
    const MinMax x = check_axis(b->minimum[0], b->maximum[0], r->origin[0], r->direction[0]);
    const MinMax y = check_axis(b->minimum[1], b->maximum[1], r->origin[1], r->direction[1]);
    const MinMax z = check_axis(b->minimum[2], b->maximum[2], r->origin[2], r->direction[2]);
    const float tmin = std::fmax(x.tmin, std::fmax(y.tmin, z.tmin));
    const float tmax = std::fmin(x.tmax, std::fmin(y.tmax, z.tmax));
    return tmin <= tmax;
}

void rt_bounds_merge(rt_bounds* a, const rt_bounds* b) { merge(*a, *b); }
void rt_bounds_transform(const rt_bounds* b, const rt_mat4* m, rt_bounds* out) { *out = transform_bounds(*b, *m); }

// src/camera.zig:25-67
void rt_camera_init(uint32_t horizontal_size, uint32_t vertical_size, float field_of_view, const float from_[4],
                    const float to_[4], const float up_[4], rt_camera* out) {
    const Tuple from = tup::from(from_), to = tup::from(to_), up = tup::from(up_);
    const float h = static_cast<float>(horizontal_size);
    const float v = static_cast<float>(vertical_size);
    rt_camera c;
    c.horizontal_size = horizontal_size;
    c.vertical_size = vertical_size;
    c.field_of_view = field_of_view;
    {  // transformCameraPerspective, :53-67.  normalize(up) keeps up's w lane (w = 1 when a sample
       // passes a point), which scales `left`/`true_up` — reproduced by doing the same math.
        const Tuple forward = tup::normalize(to - from);
        const Tuple left = tup::cross(forward, tup::normalize(up));
        const Tuple true_up = tup::cross(left, forward);
        const mat::Matrix orientation = mat::make(left[0], left[1], left[2], 0, true_up[0], true_up[1], true_up[2], 0,
                                                  -forward[0], -forward[1], -forward[2], 0, 0, 0, 0, 1);
        c.transform = mat::mul(orientation, mat::translation(-from[0], -from[1], -from[2]));
    }
    const float half_view = zmath::tan(c.field_of_view / 2.0f);
    const float aspect_ratio = h / v;
    if (aspect_ratio >= 1) {
        c.half_width = half_view;
        c.half_height = half_view / aspect_ratio;
    } else {
        c.half_width = half_view * aspect_ratio;
        c.half_height = half_view;
    }
    c.pixel_size = (c.half_width * 2) / h;
    *out = c;
}

void rt_ray_for_pixel(const rt_camera* c, uint32_t px, uint32_t py, rt_ray* out) {  // src/camera.zig:70-82
    const float x_offset = (static_cast<float>(px) + 0.5f) * c->pixel_size;
    const float y_offset = (static_cast<float>(py) + 0.5f) * c->pixel_size;
    const float world_x = c->half_width - x_offset;
    const float world_y = c->half_height - y_offset;
    const mat::Matrix inv = mat::inverse(c->transform);
    const Tuple pixel = mat::mul(inv, tup::point(world_x, world_y, -1));
    const Tuple origin = mat::mul(inv, tup::point(0, 0, 0));
    const Tuple direction = tup::normalize(pixel - origin);
    std::memcpy(out->origin, origin.v, 16);
    std::memcpy(out->direction, direction.v, 16);
}

void rt_flatten_camera(const rt_camera* cam, rtx_camera* out) {
    out->horizontal_size = cam->horizontal_size;
    out->vertical_size = cam->vertical_size;
    out->half_width = cam->half_width;
    out->half_height = cam->half_height;
    out->pixel_size = cam->pixel_size;
    const mat::Matrix inv = mat::inverse(cam->transform);  // hoisted from src/camera.zig:77-78
    std::memcpy(out->inverse_transform, inv.m, sizeof out->inverse_transform);
}

int rt_flatten(const rt_world* world, rt_flat_scene** out) {
    if (!world || !out) return RTX_E_INVALID;
    rt_flat_scene* f = new rt_flat_scene();
    Flattener fl{*f, {}, {}};
    fl.run(*world);
    std::memset(&f->desc, 0, sizeof f->desc);
    f->desc.n_shapes = (uint32_t)f->shapes.size();
    f->desc.shapes = f->shapes.data();
    f->desc.n_materials = (uint32_t)f->materials.size();
    f->desc.materials = f->materials.data();
    f->desc.n_patterns = (uint32_t)f->patterns.size();
    f->desc.patterns = f->patterns.data();
    f->desc.n_nodes = (uint32_t)f->nodes.size();
    f->desc.nodes = f->nodes.data();
    std::memcpy(f->desc.light_position, world->light_position, sizeof f->desc.light_position);
    std::memcpy(f->desc.light_intensity, world->light_intensity, sizeof f->desc.light_intensity);
    *out = f;
    return RTX_OK;
}

const rtx_scene_desc* rt_flat_desc(const rt_flat_scene* flat) { return &flat->desc; }
void rt_flat_free(rt_flat_scene* flat) { delete flat; }

int rt_render(const rt_camera* cam, const rt_world* world, float* canvas, rt_ray_counts* counts) {
    if (!cam || !world || !canvas) return RTX_E_INVALID;
    rt_flat_scene* flat = nullptr;
    int rc = rt_flatten(world, &flat);
    if (rc) return rc;
    rtx_camera rc_cam;
    rt_flatten_camera(cam, &rc_cam);
    rtx_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.rows_quirk = 1;  // src/camera.zig:87-88
    rtx_stats stats;
    // The callee fans out and joins, like the reference's eight threads (src/camera.zig:87-101): over
    // every visible GPU (RT_GPUS caps the number), one host thread each (rtx_multi_*).
    int gpus = rtx_device_count();
    if (const char* cap = std::getenv("RT_GPUS")) {
        const int want = std::atoi(cap);
        if (want > 0 && want < gpus) gpus = want;
    }
    if (gpus > 1) {
        rtx_multi* multi = nullptr;
        std::vector<int> devices;
        for (int d = 0; d < gpus; ++d) devices.push_back(d);
        rc = rtx_multi_create(rt_flat_desc(flat), devices.data(), (uint32_t)devices.size(), &multi);
        rt_flat_free(flat);
        if (rc) return rc;
        // two sets of strips per GPU: a GPU that finishes its first set early takes another (dynamic stealing,
        // where the reference's bands are static, src/camera.zig:88,99)
        rc = rtx_multi_render(multi, &rc_cam, &opts, 2, canvas, nullptr, &stats);
        rtx_multi_destroy(multi);
    } else {
        rtx_scene* scene = nullptr;
        rc = rtx_scene_create(rt_flat_desc(flat), 0, &scene);
        rt_flat_free(flat);
        if (rc) return rc;
        rc = rtx_render(scene, &rc_cam, &opts, canvas, &stats);
        rtx_scene_destroy(scene);
    }
    if (rc == 0 && counts) {
        counts->primary = stats.rays_primary;
        counts->shadow = stats.rays_shadow;
        counts->secondary = stats.rays_secondary;
    }
    return rc;
}

int rt_color_at(const rt_world* world, const rt_ray* rays, uint32_t n, float* rgb_out) {
    if (!world || (n && (!rays || !rgb_out))) return RTX_E_INVALID;
    rt_flat_scene* flat = nullptr;
    int rc = rt_flatten(world, &flat);
    if (rc) return rc;
    rtx_scene* scene = nullptr;
    rc = rtx_scene_create(rt_flat_desc(flat), 0, &scene);
    rt_flat_free(flat);
    if (rc) return rc;
    std::vector<float> o((size_t)n * 4), d((size_t)n * 4);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(&o[4 * (size_t)i], rays[i].origin, 16);
        std::memcpy(&d[4 * (size_t)i], rays[i].direction, 16);
    }
    rc = rtx_color_at(scene, o.data(), d.data(), n, 0, rgb_out);
    rtx_scene_destroy(scene);
    return rc;
}

int rt_ppm_byte(float c) {  // src/canvas.zig:45-47
    float v = c * 255;
    v = std::fmax(0.0f, std::fmin(v, 255.0f));
    return (int)std::round(v);
}

uint64_t rt_canvas_to_ppm(const float* canvas, uint32_t width, uint32_t height, char* buf, uint64_t cap) {  // canvas.zig:42-50
    char line[64];
    uint64_t need = 0;
    auto put = [&](const char* s, int n) {
        if (buf && need + (uint64_t)n <= cap) std::memcpy(buf + need, s, (size_t)n);
        need += (uint64_t)n;
    };
    int n = std::snprintf(line, sizeof line, "P3\n%u %u\n255\n", width, height);
    put(line, n);
    const size_t pixels = (size_t)width * height;
    for (size_t i = 0; i < pixels; ++i) {
        n = std::snprintf(line, sizeof line, "%d %d %d\n", rt_ppm_byte(canvas[3 * i]), rt_ppm_byte(canvas[3 * i + 1]),
                          rt_ppm_byte(canvas[3 * i + 2]));
        put(line, n);
    }
    return need;
}

// ---- math exports (known-answer tests, bindings) ----

void rt_mat_identity(rt_mat4* out) { *out = mat::identity(); }
void rt_mat_mul(const rt_mat4* a, const rt_mat4* b, rt_mat4* out) { *out = mat::mulMatrix(*a, *b); }
void rt_mat_mul_tuple(const rt_mat4* a, const float b[4], float out[4]) {
    const Tuple r = mat::mulTuple(*a, tup::from(b));
    std::memcpy(out, r.v, 16);
}
void rt_mat_transpose(const rt_mat4* a, rt_mat4* out) { *out = mat::transpose(*a); }
float rt_mat_determinant(const rt_mat4* a) { return mat::determinant(*a); }

float rt_mat_determinant_n(const float* m, int n) {
    if (n == 2) return mat::determinant(load_square<2>(m));
    if (n == 3) return mat::determinant(load_square<3>(m));
    return mat::determinant(load_square<4>(m));
}
float rt_mat_minor_n(const float* m, int n, int row, int col) {
    if (n == 3) return mat::minor_(load_square<3>(m), row, col);
    return mat::minor_(load_square<4>(m), row, col);
}
float rt_mat_cofactor_n(const float* m, int n, int row, int col) {
    if (n == 3) return mat::cofactor(load_square<3>(m), row, col);
    return mat::cofactor(load_square<4>(m), row, col);
}
void rt_mat_inverse(const rt_mat4* a, rt_mat4* out) { *out = mat::inverse(*a); }
void rt_mat_translation(float x, float y, float z, rt_mat4* out) { *out = mat::translation(x, y, z); }
void rt_mat_scaling(float x, float y, float z, rt_mat4* out) { *out = mat::scaling(x, y, z); }
void rt_mat_rotation_x(float r, rt_mat4* out) { *out = mat::rotationX(r); }
void rt_mat_rotation_y(float r, rt_mat4* out) { *out = mat::rotationY(r); }
void rt_mat_rotation_z(float r, rt_mat4* out) { *out = mat::rotationZ(r); }
void rt_mat_shearing(float xy, float xz, float yx, float yz, float zx, float zy, rt_mat4* out) {
    *out = mat::shearing(xy, xz, yx, yz, zx, zy);
}

float rt_tup_dot(const float a[4], const float b[4]) { return tup::dot(tup::from(a), tup::from(b)); }
void rt_tup_cross(const float a[4], const float b[4], float out[4]) {
    const Tuple r = tup::cross(tup::from(a), tup::from(b));
    std::memcpy(out, r.v, 16);
}
float rt_tup_magnitude(const float a[4]) { return tup::magnitude(tup::from(a)); }
void rt_tup_normalize(const float a[4], float out[4]) {
    const Tuple r = tup::normalize(tup::from(a));
    std::memcpy(out, r.v, 16);
}
void rt_tup_reflect(const float incident[4], const float normal[4], float out[4]) {
    const Tuple r = tup::reflect(tup::from(incident), tup::from(normal));
    std::memcpy(out, r.v, 16);
}
int rt_tup_equal(const float a[4], const float b[4]) { return tup::equal(tup::from(a), tup::from(b)); }

float rt_pow(float x, float y) { return zmath::pow(x, y); }
float rt_sin(float x) { return zmath::sin(x); }
float rt_cos(float x) { return zmath::cos(x); }
float rt_tan(float x) { return zmath::tan(x); }

}  // extern "C"
