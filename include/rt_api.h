/*
 * rt_api.h — by-value drop-ins for the reference's public hot-path API, as a C ABI.
 *
 * Each entry point names the reference interface it replaces (file:line into
 * /root/reference).  A maintainer of the reference binds these with `extern fn`
 * declarations over `extern struct` copies of rt_scene.h and keeps
 * World/Camera/Shape construction, OBJ parsing and Group.bound in Zig.  Scene math that
 * feeds the device (inverses, composed group transforms) is recomputed inside from the
 * transforms passed in, with the reference's f32 operation order.
 *
 * Error behaviour: the reference's `render` returns a Zig error union that can only carry
 * allocation / thread-spawn failure (intersect errors are `catch unreachable`).  Here:
 * 0 on success, a negative RTX_E_* code otherwise, message via rtx_last_error().  There
 * is no CPU fallback: without a HIP device every rendering entry point fails with
 * RTX_E_NO_DEVICE.
 */
#ifndef RT_API_H
#define RT_API_H

#include "rt_scene.h"
#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the exported C ABI */

/* ---- the hot path ------------------------------------------------------------------ */

/* pub fn camera(h, v, fov, from, to, up) Camera — src/camera.zig:25-51 (host only). */
void rt_camera_init(uint32_t horizontal_size, uint32_t vertical_size, float field_of_view, const float from[4],
                    const float to[4], const float up[4], rt_camera* out);

/* pub fn rayForPixel(c, px, py) Ray — src/camera.zig:70-82 (host evaluation, for callers and tests). */
void rt_ray_for_pixel(const rt_camera* c, uint32_t px, uint32_t py, rt_ray* out);

/*
 * pub fn render(allocator, cam, world) !Canvas — src/camera.zig:84-105.
 * `canvas` is caller-owned, vertical_size*horizontal_size*3 f32, row-major (Canvas.pixels,
 * src/canvas.zig:15-28,52-54).  Like the reference, rows beyond 8*floor(vsize/8) stay
 * black (src/camera.zig:87-88).  Flattens the world, uploads it, renders — fanned out over every
 * visible HIP device like the reference's eight threads (rtx_multi_render; the environment variable
 * RT_GPUS caps the number; one device: rtx_render on device 0) — and copies the canvas back;
 * blocking.  `counts` may be NULL.
 */
int rt_render(const rt_camera* cam, const rt_world* world, float* canvas, rt_ray_counts* counts);

/* pub fn colorAt(world, r) Color — src/world.zig:81-83, for n rays at once. */
int rt_color_at(const rt_world* world, const rt_ray* rays, uint32_t n, float* rgb_out);

/* Canvas.toPPM — src/canvas.zig:42-50: P3 text into `buf` (size `cap`); returns bytes needed. */
uint64_t rt_canvas_to_ppm(const float* canvas, uint32_t width, uint32_t height, char* buf, uint64_t cap);

/* The 8-bit value toPPM prints for one channel (src/canvas.zig:45-48). */
int rt_ppm_byte(float c);

/* ---- flattening (host) ---------------------------------------------------------------- */

typedef struct rt_flat_scene rt_flat_scene; /* owns the arrays an rtx_scene_desc points into */

/* World -> rtx_scene_desc: insertion order, composed transforms, inverses, gates
 * (src/world.zig:38-47, src/group.zig:55-87,101-106). */
int rt_flatten(const rt_world* world, rt_flat_scene** out);
const rtx_scene_desc* rt_flat_desc(const rt_flat_scene* flat);
void rt_flat_free(rt_flat_scene* flat);

/* Camera -> rtx_camera (inverse precomputed, src/camera.zig:77-78). */
void rt_flatten_camera(const rt_camera* cam, rtx_camera* out);

/* ---- scene-construction helpers mirrored from the reference (host only) --------------- */

/* Triangle.init — src/triangle.zig:19-30 (fills p0..normal; common_attrs untouched). */
void rt_triangle_init(const float p0[4], const float p1[4], const float p2[4], rt_triangle* out);
/* Group.bound / bnd.boundGroup — src/group.zig:51-53, src/bounds.zig:85-107. */
void rt_bound_group(const rt_group* group, rt_bounds* out);
/* bnd.intersect — src/bounds.zig:149-158. */
int rt_bounds_intersect(const rt_bounds* b, const rt_ray* r);
/* mergeBounds / transformBounds — src/bounds.zig:109-112,119-133. */
void rt_bounds_merge(rt_bounds* a, const rt_bounds* b);
void rt_bounds_transform(const rt_bounds* b, const rt_mat4* m, rt_bounds* out);
/* Material{} / CommonShapeAttributes{} defaults — src/material.zig:10-37, src/shape.zig:20-23. */
void rt_default_material(rt_material* out);
void rt_default_attrs(rt_attrs* out);

/* mat.* — src/matrix.zig:43-222 */
void  rt_mat_identity(rt_mat4* out);
void  rt_mat_mul(const rt_mat4* a, const rt_mat4* b, rt_mat4* out);
void  rt_mat_mul_tuple(const rt_mat4* a, const float b[4], float out[4]);
void  rt_mat_transpose(const rt_mat4* a, rt_mat4* out);
float rt_mat_determinant(const rt_mat4* a);
float rt_mat_determinant_n(const float* m, int n);           /* n = 2, 3, 4; row-major n*n */
float rt_mat_minor_n(const float* m, int n, int row, int col);
float rt_mat_cofactor_n(const float* m, int n, int row, int col);
void  rt_mat_inverse(const rt_mat4* a, rt_mat4* out);
void  rt_mat_translation(float x, float y, float z, rt_mat4* out);
void  rt_mat_scaling(float x, float y, float z, rt_mat4* out);
void  rt_mat_rotation_x(float r, rt_mat4* out);
void  rt_mat_rotation_y(float r, rt_mat4* out);
void  rt_mat_rotation_z(float r, rt_mat4* out);
void  rt_mat_shearing(float xy, float xz, float yx, float yz, float zx, float zy, rt_mat4* out);

/* tup.* — src/tuple.zig:32-77 */
float rt_tup_dot(const float a[4], const float b[4]);
void  rt_tup_cross(const float a[4], const float b[4], float out[4]);
float rt_tup_magnitude(const float a[4]);
void  rt_tup_normalize(const float a[4], float out[4]);
void  rt_tup_reflect(const float incident[4], const float normal[4], float out[4]);
int   rt_tup_equal(const float a[4], const float b[4]);

/* std.math.pow / tan and @sin/@cos as the reference's toolchain evaluates them (rt_libm.hpp). */
float rt_pow(float x, float y);
float rt_sin(float x);
float rt_cos(float x);
float rt_tan(float x);

/* ---- OBJ ingest: obj.parseObjFile / ObjFile.toShapeGroup — src/obj_file.zig:45-62,77-147 (host only) ---- */

enum {
    RT_OBJ_INVALID_COMMAND = -10, /* ParserError.InvalidCommand (src/obj_file.zig:25-27,101,137,141,150) */
    RT_OBJ_BAD_NUMBER      = -11, /* std.fmt.parseFloat / parseInt failed (:105-107,115) */
    RT_OBJ_BAD_INDEX       = -12  /* vertex index out of range or a face with < 3 indices (a panic in the reference) */
};
typedef struct rt_obj rt_obj;
/* Parse `len` bytes of OBJ text (`v`, `f`, `g` only). */
int      rt_obj_parse(const char* text, uint64_t len, rt_obj** out);
uint32_t rt_obj_vertex_count(const rt_obj* obj);
uint32_t rt_obj_face_count(const rt_obj* obj);  /* triangles after fan triangulation */
uint32_t rt_obj_group_count(const rt_obj* obj); /* named groups that hold faces */
/* toShapeGroup; bound != 0 also runs Group.bound() (samples/teapot.zig:19).  The group is owned by `obj`. */
int      rt_obj_to_group(rt_obj* obj, int bound, const rt_group** out);
void     rt_obj_free(rt_obj* obj);

/* ---- the reference's sample scenes (samples/<name>.zig), built through the C++ mirror --- */

typedef struct rt_sample rt_sample;
/* width/height 0 -> the sample's shipped resolution.  NULL if the name is unknown. */
rt_sample*       rt_sample_create(const char* name, uint32_t width, uint32_t height);
/* The two scenes built around an OBJ mesh, from `len` bytes of OBJ text: "teapot"
 * (samples/teapot.zig:7-36; rt_sample_create("teapot") reads ./teapot.obj like the reference) and
 * "composite" (BASELINE.json configs[4]: teapot + glass_spheres' spheres + reflective_floor; not a
 * reference scene, frozen in rt_samples.cpp).  NULL on an unknown name or a parse error. */
rt_sample*       rt_sample_create_obj(const char* name, const char* obj_text, uint64_t len, uint32_t width, uint32_t height);
const rt_world*  rt_sample_world(const rt_sample* s);
const rt_camera* rt_sample_camera(const rt_sample* s);
void             rt_sample_destroy(rt_sample* s);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
