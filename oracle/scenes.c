/*
 * oracle/scenes.c — TEST INFRASTRUCTURE.  The reference's sample programs
 * (samples/<name>.zig) restated as scene builders on top of the oracle's own math
 * (cpu_ref.c), so the oracle can be pinned against the shipped golden images
 * without touching product code.  This is the translation unit the oracle's
 * shared library is built from (it includes cpu_ref.c).
 */
#include "cpu_ref.c"

#include <stdio.h>

#define PI 3.14159265358979323846264338327950288 /* std.math.pi, coerced to f32 at each use */

typedef struct oracle_scene {
    rt_world  world;
    rt_camera camera;
    /* owned storage */
    rt_sphere*   spheres;
    rt_plane*    planes;
    rt_cube*     cubes;
    rt_cylinder* cylinders;
    rt_cone*     cones;
    rt_triangle* triangles;
} oracle_scene;

static rt_material default_material(void) { /* material.zig:10-37 */
    rt_material m;
    memset(&m, 0, sizeof m);
    m.color[0] = m.color[1] = m.color[2] = 1;
    m.has_pattern = 0;
    m.pattern.transform = identity; /* payload of a null optional: unused */
    m.ambient = 0.1f;
    m.diffuse = 0.9f;
    m.specular = 0.9f;
    m.shininess = 200.0f;
    m.reflective = 0.0f;
    m.refractive_index = 1.0f;
    m.transparency = 0.0f;
    return m;
}

static rt_attrs default_attrs(void) { /* shape.zig:20-23 */
    rt_attrs a;
    memset(&a, 0, sizeof a);
    a.transform = identity;
    a.material = default_material();
    return a;
}

static void set_pattern(rt_material* m, int kind, float a0, float a1, float a2, float b0, float b1, float b2) {
    m->has_pattern = 1;
    m->pattern.a[0] = a0; m->pattern.a[1] = a1; m->pattern.a[2] = a2;
    m->pattern.b[0] = b0; m->pattern.b[1] = b1; m->pattern.b[2] = b2;
    m->pattern.transform = identity; /* pattern.zig:16 */
    m->pattern.color_map = kind;
}

static void set_color(rt_material* m, float r, float g, float b) { m->color[0] = r; m->color[1] = g; m->color[2] = b; }

static void set_light(oracle_scene* s, float x, float y, float z, float r, float g, float b) {
    tuple p = point(x, y, z);
    tup_store(s->world.light_position, p);
    s->world.light_intensity[0] = r;
    s->world.light_intensity[1] = g;
    s->world.light_intensity[2] = b;
}

static void set_camera(oracle_scene* s, uint32_t w, uint32_t h, uint32_t dw, uint32_t dh, float fov,
                       tuple from, tuple to, tuple up) {
    if (w == 0 || h == 0) { w = dw; h = dh; }
    float f4[4], t4[4], u4[4];
    tup_store(f4, from); tup_store(t4, to); tup_store(u4, up);
    oracle_camera(w, h, fov, f4, t4, u4, &s->camera);
}

static void finish(oracle_scene* s, uint32_t ns, uint32_t np, uint32_t ncu, uint32_t ncy, uint32_t nco, uint32_t nt) {
    s->world.n_spheres = ns; s->world.spheres = s->spheres;
    s->world.n_planes = np; s->world.planes = s->planes;
    s->world.n_cubes = ncu; s->world.cubes = s->cubes;
    s->world.n_cylinders = ncy; s->world.cylinders = s->cylinders;
    s->world.n_cones = nco; s->world.cones = s->cones;
    s->world.n_triangles = nt; s->world.triangles = s->triangles;
    s->world.n_groups = 0; s->world.groups = NULL;
}

static rt_cylinder make_cyl(float mn, float mx, int closed, mat4 tr) {
    rt_cylinder c;
    memset(&c, 0, sizeof c);
    c.common_attrs = default_attrs();
    c.common_attrs.transform = tr;
    c.minimum = mn;
    c.maximum = mx;
    c.closed = closed;
    return c;
}

static void cyl_mat(rt_cylinder* c, float r, float g, float b, float ambient, float diffuse) {
    set_color(&c->common_attrs.material, r, g, b);
    c->common_attrs.material.ambient = ambient;
    c->common_attrs.material.diffuse = diffuse;
    c->common_attrs.material.specular = 0.9f;
    c->common_attrs.material.shininess = 300.0f;
}

EXPORT oracle_scene* oracle_sample(const char* name, uint32_t w, uint32_t h) {
    oracle_scene* s = (oracle_scene*)calloc(1, sizeof *s);
    s->spheres = (rt_sphere*)calloc(8, sizeof(rt_sphere));
    s->planes = (rt_plane*)calloc(4, sizeof(rt_plane));
    s->cubes = (rt_cube*)calloc(2, sizeof(rt_cube));
    s->cylinders = (rt_cylinder*)calloc(12, sizeof(rt_cylinder));
    s->cones = (rt_cone*)calloc(2, sizeof(rt_cone));
    s->triangles = (rt_triangle*)calloc(2, sizeof(rt_triangle));

    if (!strcmp(name, "sphere_world")) { /* samples/sphere_world.zig */
        rt_attrs fl = default_attrs();
        fl.transform = scaling(10, 0.01f, 10);
        set_color(&fl.material, 1, 0.9f, 0.9f);
        fl.material.specular = 0;
        s->spheres[0].common_attrs = fl;
        rt_attrs lw = fl;
        lw.transform = mul_matrix(mul_matrix(translation(0, 0, 5), rotation_y((float)(-PI / 4.0))),
                                  mul_matrix(rotation_x((float)(PI / 2.0)), scaling(10, 0.01f, 10)));
        s->spheres[1].common_attrs = lw;
        rt_attrs rw = fl;
        rw.transform = mul_matrix(mul_matrix(translation(0, 0, 5), rotation_y((float)(PI / 4.0))),
                                  mul_matrix(rotation_x((float)(PI / 2.0)), scaling(10, 0.01f, 10)));
        s->spheres[2].common_attrs = rw;
        rt_attrs mid = default_attrs();
        mid.transform = translation(-0.5f, 1, 0.5f);
        set_color(&mid.material, 0.1f, 1, 0.5f);
        mid.material.diffuse = 0.7f;
        mid.material.specular = 0.3f;
        s->spheres[3].common_attrs = mid;
        rt_attrs right = default_attrs();
        right.transform = mul_matrix(translation(1.5f, 0.5f, -0.5f), scaling(0.5f, 0.5f, 0.5f));
        set_color(&right.material, 0.5f, 1, 0.1f);
        right.material.diffuse = 0.7f;
        right.material.specular = 0.3f;
        s->spheres[4].common_attrs = right;
        rt_attrs left = default_attrs();
        left.transform = mul_matrix(translation(-1.5f, 0.33f, -0.75f), scaling(0.33f, 0.33f, 0.33f));
        set_color(&left.material, 1, 0.8f, 0.1f);
        left.material.diffuse = 0.7f;
        left.material.specular = 0.3f;
        s->spheres[5].common_attrs = left;
        set_light(s, -10, 10, -10, 1, 1, 1);
        finish(s, 6, 0, 0, 0, 0, 0);
        set_camera(s, w, h, 1000, 500, (float)(PI / 3.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0));
    } else if (!strcmp(name, "glass_spheres")) { /* samples/glass_spheres.zig */
        rt_attrs bg = default_attrs();
        bg.transform = mul_matrix(translation(0, 0, 10), rotation_x((float)(PI / 2.0)));
        bg.material.ambient = 0.8f;
        bg.material.diffuse = 0.2f;
        bg.material.specular = 0.0f;
        set_pattern(&bg.material, RT_PATTERN_CHECKER, 0.15f, 0.15f, 0.15f, 0.85f, 0.85f, 0.85f);
        s->planes[0].common_attrs = bg;
        rt_attrs outer = default_attrs();
        set_color(&outer.material, 1, 1, 1);
        outer.material.ambient = 0.0f;
        outer.material.diffuse = 0.0f;
        outer.material.specular = 0.9f;
        outer.material.shininess = 300.0f;
        outer.material.reflective = 0.9f;
        outer.material.transparency = 0.9f;
        outer.material.refractive_index = 1.5f;
        s->spheres[0].common_attrs = outer;
        rt_attrs inner = outer;
        inner.transform = scaling(0.5f, 0.5f, 0.5f);
        inner.material.refractive_index = 1.0003f;
        s->spheres[1].common_attrs = inner;
        set_light(s, -2, 5, -10, 0.9f, 0.9f, 0.9f);
        finish(s, 2, 1, 0, 0, 0, 0);
        set_camera(s, w, h, 600, 600, 0.45f, point(0, 0, -5), point(0, 0, 0), point(0, 1, 0));
    } else if (!strcmp(name, "fresnel")) { /* samples/fresnel.zig */
        rt_attrs bottom = default_attrs();
        bottom.transform = translation(0, -15, 0);
        bottom.material.ambient = 1.0f;
        set_pattern(&bottom.material, RT_PATTERN_CHECKER, 0.85f, 0.1f, 0.1f, 0.3f, 0.85f, 0.15f);
        bottom.material.pattern.transform = scaling(3, 3, 3);
        s->planes[0].common_attrs = bottom;
        rt_attrs water = default_attrs();
        water.material.diffuse = 0.1f;
        water.material.shininess = 300.0f;
        water.material.refractive_index = 1.33f;
        water.material.transparency = 1.0f;
        water.material.reflective = 0.9f;
        s->planes[1].common_attrs = water;
        rt_attrs far = default_attrs();
        far.transform = mul_matrix(translation(0, 0, 30), rotation_x((float)(PI / 2.0)));
        far.material.transparency = 0.75f;
        far.material.specular = 0.0f;
        set_pattern(&far.material, RT_PATTERN_CHECKER, 1, 1, 1, 0.5f, 0.5f, 0.5f);
        far.material.pattern.transform = scaling(0.25f, 0.25f, 0.25f);
        s->planes[2].common_attrs = far;
        set_light(s, 0, 30, -15, 1, 1, 1);
        finish(s, 0, 3, 0, 0, 0, 0);
        set_camera(s, w, h, 512, 256, (float)(PI / 2.0), point(0, 1.5f, -15), point(0, 1, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "cylinder_world")) { /* samples/cylinder_world.zig */
        rt_attrs fl = default_attrs();
        fl.material.ambient = 0.2f;
        fl.material.diffuse = 0.9f;
        fl.material.specular = 0.0f;
        set_pattern(&fl.material, RT_PATTERN_CHECKER, 0.5f, 0.5f, 0.5f, 0.75f, 0.75f, 0.75f);
        fl.material.pattern.transform = mul_matrix(rotation_y(0.3f), scaling(0.25f, 0.25f, 0.25f));
        s->planes[0].common_attrs = fl;

        rt_cylinder big = make_cyl(0.0f, 0.75f, 1, mul_matrix(translation(-1, 0, 1), scaling(0.5f, 1, 0.5f)));
        set_color(&big.common_attrs.material, 0, 0, 0.6f);
        big.common_attrs.material.diffuse = 0.1f;
        big.common_attrs.material.specular = 0.9f;
        big.common_attrs.material.shininess = 300.0f;
        big.common_attrs.material.reflective = 0.9f;
        s->cylinders[0] = big;

        rt_cylinder a = make_cyl(0.0f, 0.2f, 0, mul_matrix(translation(1, 0, 0), scaling(0.8f, 1, 0.8f)));
        cyl_mat(&a, 1, 1, 0.3f, 0.1f, 0.8f);
        s->cylinders[1] = a;
        rt_cylinder b = make_cyl(0.0f, 0.3f, 0, mul_matrix(translation(1, 0, 0), scaling(0.6f, 1, 0.6f)));
        cyl_mat(&b, 1, 0.9f, 0.4f, 0.1f, 0.8f);
        s->cylinders[2] = b;
        rt_cylinder c = make_cyl(0.0f, 0.4f, 0, mul_matrix(translation(1, 0, 0), scaling(0.4f, 1, 0.4f)));
        cyl_mat(&c, 1, 0.8f, 0.5f, 0.1f, 0.8f);
        s->cylinders[3] = c;
        rt_cylinder d = make_cyl(0.0f, 0.5f, 1, mul_matrix(translation(1, 0, 0), scaling(0.2f, 1, 0.2f)));
        cyl_mat(&d, 1, 0.7f, 0.6f, 0.1f, 0.8f);
        s->cylinders[4] = d;

        rt_cylinder wc = make_cyl(0.0f, 0.3f, 1, mul_matrix(translation(0, 0, -0.75f), scaling(0.05f, 1, 0.05f)));
        cyl_mat(&wc, 1, 0, 0, 0.1f, 0.9f);
        s->cylinders[5] = wc;
        rt_cylinder xc = make_cyl(0.0f, 0.3f, 1,
            mul_matrix(mul_matrix(translation(0, 0, -2.25f), rotation_y(-0.15f)),
                       mul_matrix(translation(0, 0, 1.5f), scaling(0.05f, 1, 0.05f))));
        cyl_mat(&xc, 1, 1, 0, 0.1f, 0.9f);
        s->cylinders[6] = xc;
        rt_cylinder yc = make_cyl(0.0f, 0.3f, 1,
            mul_matrix(mul_matrix(translation(0, 0, -2.25f), rotation_y(-0.3f)),
                       mul_matrix(translation(0, 0, 1.5f), scaling(0.05f, 1, 0.05f))));
        cyl_mat(&yc, 0, 1, 0, 0.1f, 0.9f);
        s->cylinders[7] = yc;
        rt_cylinder zc = make_cyl(0.0f, 0.3f, 1,
            mul_matrix(mul_matrix(translation(0, 0, -2.25f), rotation_y(-0.45f)),
                       mul_matrix(translation(0, 0, 1.5f), scaling(0.05f, 1, 0.05f))));
        cyl_mat(&zc, 0, 1, 1, 0.1f, 0.9f);
        s->cylinders[8] = zc;

        rt_cylinder glass = make_cyl(0.0001f, 0.5f, 1, mul_matrix(translation(0, 0, -1.5f), scaling(0.33f, 1, 0.33f)));
        set_color(&glass.common_attrs.material, 0.25f, 0, 0);
        glass.common_attrs.material.diffuse = 0.1f;
        glass.common_attrs.material.specular = 0.9f;
        glass.common_attrs.material.shininess = 300.0f;
        glass.common_attrs.material.reflective = 0.9f;
        glass.common_attrs.material.transparency = 0.9f;
        glass.common_attrs.material.refractive_index = 1.5f;
        s->cylinders[9] = glass;

        set_light(s, 1, 6.9f, -4.9f, 1, 1, 1);
        finish(s, 0, 1, 0, 10, 0, 0);
        set_camera(s, w, h, 800, 400, (float)(PI / 10.0), point(8, 3.5f, -9), point(0, 0.3f, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "cone")) { /* samples/cone.zig */
        s->cones[0].common_attrs = default_attrs();
        s->cones[0].common_attrs.transform = rotation_x((float)(-PI / 6.0));
        s->cones[0].minimum = -2.0f;
        s->cones[0].maximum = 2.0f;
        s->cones[0].closed = 0;
        set_light(s, 0, 0, -10, 1, 1, 1);
        finish(s, 0, 0, 0, 0, 1, 0);
        set_camera(s, w, h, 512, 512, (float)(PI / 2.0), point(0, 0, -5), point(0, 0.3f, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "cube")) { /* samples/cube.zig */
        s->cubes[0].common_attrs = default_attrs();
        s->cubes[0].common_attrs.transform = rotation_y((float)(PI / 4.0));
        set_light(s, 0, 0, -10, 0.9f, 0.9f, 0.9f);
        finish(s, 0, 0, 1, 0, 0, 0);
        set_camera(s, w, h, 1024, 1024, (float)(PI / 3.0), point(0, 5, -5), point(0, 0, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "reflective_floor")) { /* samples/reflective_floor.zig */
        rt_attrs fl = default_attrs();
        set_pattern(&fl.material, RT_PATTERN_CHECKER, 0, 0, 0, 0.25f, 0.25f, 0.25f);
        fl.material.reflective = 0.25f;
        s->planes[0].common_attrs = fl;
        rt_attrs sp = default_attrs();
        set_color(&sp.material, 1, 0.3f, 0.25f);
        sp.transform = translation(0, 1, 0.5f);
        s->spheres[0].common_attrs = sp;
        set_light(s, 0, 30, -5, 1, 1, 1);
        finish(s, 1, 1, 0, 0, 0, 0);
        set_camera(s, w, h, 1024, 1024, (float)(PI / 3.0), point(0, 1.5f, -5), point(0, 1, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "skewed_triangle")) { /* samples/skewed_triangle.zig */
        tuple p0 = point(-5, -5, 10), p1 = point(0, 2.5f, 0), p2 = point(5, -5, 3);
        float q0[4], q1[4], q2[4];
        tup_store(q0, p0); tup_store(q1, p1); tup_store(q2, p2);
        oracle_triangle_init(q0, q1, q2, &s->triangles[0]);
        s->triangles[0].common_attrs = default_attrs();
        set_color(&s->triangles[0].common_attrs.material, 1, 1, 0.25f);
        s->triangles[0].common_attrs.material.ambient = 0.5f;
        s->triangles[0].common_attrs.material.specular = 0;
        set_light(s, 0, 0, -10, 0.9f, 0.9f, 0.9f);
        finish(s, 0, 0, 0, 0, 0, 1);
        set_camera(s, w, h, 512, 512, (float)(PI / 2.0), point(0, 0, -10), point(0, 0, 0), vector(0, 1, 0));
    } else if (!strcmp(name, "striped_scene")) { /* samples/striped_scene.zig */
        const float blue[3] = {0.05f, 0.25f, 1}, orange[3] = {1, 0.5f, 0.25f};
        rt_attrs fl = default_attrs();
        set_pattern(&fl.material, RT_PATTERN_STRIPE, orange[0], orange[1], orange[2], blue[0], blue[1], blue[2]);
        s->planes[0].common_attrs = fl;
        rt_attrs sp = default_attrs();
        set_pattern(&sp.material, RT_PATTERN_STRIPE, blue[0], blue[1], blue[2], orange[0], orange[1], orange[2]);
        sp.material.pattern.transform = rotation_z((float)(-PI / 4.0));
        sp.transform = translation(0, 1, 0);
        s->spheres[0].common_attrs = sp;
        set_light(s, 0, 10, -10, 1, 1, 1);
        finish(s, 1, 1, 0, 0, 0, 0);
        set_camera(s, w, h, 1024, 1024, (float)(PI / 2.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0));
    } else if (!strcmp(name, "sphere_on_plane")) { /* samples/sphere_on_plane.zig */
        s->planes[0].common_attrs = default_attrs();
        s->planes[1].common_attrs = default_attrs();
        s->planes[1].common_attrs.transform = mul_matrix(translation(0, 0, 3), rotation_x((float)(PI / 2.0)));
        rt_attrs sp = default_attrs();
        sp.transform = translation(0, 1, 0.5f);
        set_color(&sp.material, 0.1f, 0.75f, 1);
        sp.material.diffuse = 0.7f;
        sp.material.specular = 0.3f;
        s->spheres[0].common_attrs = sp;
        set_light(s, 10, 10, -10, 1, 1, 1);
        finish(s, 1, 2, 0, 0, 0, 0);
        set_camera(s, w, h, 1024, 1024, (float)(PI / 4.0), point(0, 1.5f, -5), point(0, 1, 0), point(0, 1, 0));
    } else {
        free(s->spheres); free(s->planes); free(s->cubes); free(s->cylinders); free(s->cones); free(s->triangles);
        free(s);
        return NULL;
    }
    return s;
}

EXPORT const rt_world* oracle_scene_world(const oracle_scene* s) { return &s->world; }
EXPORT const rt_camera* oracle_scene_camera(const oracle_scene* s) { return &s->camera; }

EXPORT void oracle_sample_free(oracle_scene* s) {
    if (!s) return;
    free(s->spheres); free(s->planes); free(s->cubes); free(s->cylinders); free(s->cones); free(s->triangles);
    free(s);
}

/* Command-line driver: `oracle_cli <sample> [W H] > out.ppm` — P3 text exactly
 * as canvas.zig:42-50 prints it.  Used for timing and manual diffing only. */
#ifdef ORACLE_MAIN
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <sample> [W H] [libm 0|1]\n", argv[0]); return 2; }
    uint32_t w = argc > 3 ? (uint32_t)atoi(argv[2]) : 0, h = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
    if (argc > 4) oracle_set_libm(atoi(argv[4]));
    oracle_scene* s = oracle_sample(argv[1], w, h);
    if (!s) { fprintf(stderr, "unknown sample %s\n", argv[1]); return 2; }
    w = s->camera.horizontal_size; h = s->camera.vertical_size;
    float* canvas = (float*)malloc((size_t)w * h * 3 * sizeof(float));
    rt_ray_counts counts;
    oracle_render(&s->camera, &s->world, canvas, 1, 1, &counts);
    fprintf(stderr, "rays primary=%llu shadow=%llu secondary=%llu\n", (unsigned long long)counts.primary,
            (unsigned long long)counts.shadow, (unsigned long long)counts.secondary);
    printf("P3\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        printf("%d %d %d\n", oracle_ppm_byte(canvas[3 * i]), oracle_ppm_byte(canvas[3 * i + 1]), oracle_ppm_byte(canvas[3 * i + 2]));
    free(canvas);
    oracle_sample_free(s);
    return 0;
}
#endif
