/*
 * oracle/tsan_main.c — TEST INFRASTRUCTURE.  One ThreadSanitizer pass over the oracle's 8-thread
 * render: the reference's Group.intersect temporarily overwrites
 * subgroup.transform while eight threads share the world (src/group.zig:72-83,
 * src/camera.zig:89-99) — a latent data race for nested non-identity groups.  The oracle passes
 * the composed transform down instead (cpu_ref.c group_intersect), so its render must be race
 * free on exactly such a world.  `make -C oracle tsan` builds and runs this; exit code 0 and
 * no "WARNING: ThreadSanitizer" line is the result recorded in DESIGN.md.
 */
#include "scenes.c"

int main(void) {
    static rt_sphere inner_spheres[1], outer_spheres[1], world_spheres[1];
    static rt_cube inner_cubes[1];
    static rt_plane world_planes[1];
    static rt_group inner, outer;

    inner_spheres[0].common_attrs = default_attrs();
    inner_spheres[0].common_attrs.transform = translation(0.8f, 0.2f, 0);
    inner_spheres[0].common_attrs.material.reflective = 0.4f;
    inner_cubes[0].common_attrs = default_attrs();
    inner_cubes[0].common_attrs.transform = mul_matrix(translation(-1, 0, 0.5f), scaling(0.4f, 0.4f, 0.4f));
    memset(&inner, 0, sizeof inner);
    inner.transform = mul_matrix(rotation_y(0.6f), scaling(0.9f, 1.1f, 0.9f)); /* non-identity, nested */
    inner.n_spheres = 1; inner.spheres = inner_spheres;
    inner.n_cubes = 1; inner.cubes = inner_cubes;
    inner.bounding_box = bound_group(&inner);
    inner.has_bounding_box = 1;

    outer_spheres[0].common_attrs = default_attrs();
    outer_spheres[0].common_attrs.transform = translation(0, 1.5f, 0);
    outer_spheres[0].common_attrs.material.transparency = 0.7f;
    outer_spheres[0].common_attrs.material.refractive_index = 1.5f;
    memset(&outer, 0, sizeof outer);
    outer.transform = mul_matrix(translation(0.3f, 0.5f, 1), rotation_z(0.3f)); /* non-identity parent */
    outer.n_spheres = 1; outer.spheres = outer_spheres;
    outer.n_subgroups = 1; outer.subgroups = &inner;
    outer.bounding_box = bound_group(&outer);
    outer.has_bounding_box = 1;

    world_spheres[0].common_attrs = default_attrs();
    world_spheres[0].common_attrs.transform = translation(-2.5f, 1, 0);
    world_planes[0].common_attrs = default_attrs();
    world_planes[0].common_attrs.material.reflective = 0.3f;

    rt_world w;
    memset(&w, 0, sizeof w);
    tup_store(w.light_position, point(-10, 10, -10));
    w.light_intensity[0] = w.light_intensity[1] = w.light_intensity[2] = 1;
    w.n_spheres = 1; w.spheres = world_spheres;
    w.n_planes = 1; w.planes = world_planes;
    w.n_groups = 1; w.groups = &outer;

    rt_camera cam;
    float from[4], to[4], up[4];
    tup_store(from, point(0, 2.5f, -7));
    tup_store(to, point(0, 1, 0));
    tup_store(up, vector(0, 1, 0));
    oracle_camera(160, 96, (float)(PI / 3.0), from, to, up, &cam);
    float* canvas = (float*)malloc((size_t)160 * 96 * 3 * sizeof(float));
    rt_ray_counts counts;
    if (oracle_render(&cam, &w, canvas, 1, 1, &counts) != 0) return 1;
    double sum = 0;
    for (size_t i = 0; i < (size_t)160 * 96 * 3; ++i) sum += canvas[i];
    printf("tsan pass: 8 threads, nested non-identity groups, %llu primary / %llu shadow / %llu secondary rays, canvas sum %.3f\n",
           (unsigned long long)counts.primary, (unsigned long long)counts.shadow, (unsigned long long)counts.secondary, sum);
    free(canvas);
    return counts.secondary > 0 && sum > 0 ? 0 : 1;
}
