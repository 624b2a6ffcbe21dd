/*
 * rtx.h — C ABI of the MI355X renderer: the flattened, device-resident form of the hot
 * path `Camera.render -> World.colorAt`.
 *
 * The reference has no plugin/FFI interface; the narrowest waist its hot path sits
 * behind is
 *     pub fn render(allocator, cam: Camera, world: World) !Canvas   src/camera.zig:84-105
 *     pub fn colorAt(world: *const World, r: Ray) Color             src/world.zig:81-83
 *     pub fn rayForPixel(c: *const Camera, px, py) Ray              src/camera.zig:70-82
 * (all paths into /root/reference).  include/rt_api.h holds the by-value drop-ins for
 * those three; this header is the layer underneath, for callers that keep a scene
 * resident in HBM across frames: the scene is flattened once on the host into the
 * plain records below (precomputed inverses, composed group transforms, insertion
 * order), uploaded with rtx_scene_create, and rendered any number of times.
 *
 * Conventions: every function returns 0 on success or a negative RTX_E_* code and never
 * throws; rtx_last_error() gives the thread-local message; the caller owns every host
 * buffer it passes; the library owns all device memory; one render in flight per
 * rtx_scene and work set (rtx_opts.work_set).  Plain pointers and sizes only — no C++/torch types cross this boundary.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the exported C ABI */

enum {
    RTX_OK          = 0,
    RTX_E_INVALID   = -1, /* bad argument / inconsistent descriptor */
    RTX_E_NO_DEVICE = -2, /* no usable HIP device (the product path has no CPU fallback) */
    RTX_E_HIP       = -3, /* a HIP runtime call failed; see rtx_last_error() */
    RTX_E_NOMEM     = -4
};

/* Shape type tags, in the reference's insertion order (src/world.zig:39-44). */
enum {
    RTX_SPHERE = 0, RTX_PLANE = 1, RTX_CUBE = 2, RTX_CYLINDER = 3, RTX_CONE = 4, RTX_TRIANGLE = 5
};

enum {
    RTX_SHAPE_CLOSED       = 1u << 0, /* Cylinder/Cone.closed (src/cylinder.zig:12, src/cone.zig:12) */
    RTX_SHAPE_N1_INVISIBLE = 1u << 1  /* byte-identical twin cancels it in the n1/n2 container walk
                                         (std.meta.eql, src/intersection.zig:104,128) */
};

/*
 * One shape, in world-level insertion order (src/world.zig:39-45, src/group.zig:59-83):
 * index in the array == position in the reference's stable tie-break order.
 * `inverse` = mat.inverse(composed transform) where the composed transform is
 * mul(mul(G_outer, G_inner), S) exactly as src/group.zig:75,103-104 composes it.
 */
typedef struct rtx_shape {
    float    inverse[16]; /* row-major */
    uint32_t type;        /* RTX_* */
    uint32_t material;    /* index into rtx_scene_desc.materials */
    uint32_t flags;       /* RTX_SHAPE_* */
    uint32_t reserved;    /* twin class: 0, or k >= 1 for a shape that has byte-identical twins (std.meta.eql) behind OTHER
                             group boxes than its own — all members of one class carry the same k.  Which twins a ray
                             intersects then depends on the boxes it passes, and the cancelling of equal Intersections
                             (src/intersection.zig:126-137) is resolved per ray on the device.  At most 32 such shapes
                             per scene.  Twins behind the same boxes are settled by RTX_SHAPE_N1_INVISIBLE instead */
    float    minimum;     /* cylinder / cone */
    float    maximum;
    float    pad[2];
    float    p0[4];       /* triangle (src/triangle.zig:11-17) */
    float    e0[4];
    float    e1[4];
    float    normal[4];
} rtx_shape;

/* Material (src/material.zig:10-37); pattern < 0 means `pattern == null`. */
typedef struct rtx_material {
    float   color[3];
    float   ambient;
    float   diffuse;
    float   specular;
    float   shininess;
    float   reflective;
    float   refractive_index;
    float   transparency;
    int32_t pattern; /* index into rtx_scene_desc.patterns or -1 */
    int32_t reserved;
} rtx_material;

/* Pattern (src/pattern.zig:14-17) with its inverse precomputed (src/pattern.zig:29). */
typedef struct rtx_pattern {
    float    inverse[16];
    float    a[3];
    float    b[3];
    uint32_t color_map; /* RT_PATTERN_* from rt_scene.h */
    uint32_t reserved;
} rtx_pattern;

/*
 * One group gate (src/group.zig:55-58), in DFS order.  Node 0 is the world itself (no
 * box).  A ray that fails a node's box skips to node index `skip` (the first node after
 * the subtree); shapes [first_shape, first_shape + n_shapes) are the node's own members.
 */
typedef struct rtx_node {
    float    bounds_min[3];
    uint32_t has_bounds;
    float    bounds_max[3];
    uint32_t skip;
    uint32_t first_shape;
    uint32_t n_shapes;
    uint32_t reserved[2];
} rtx_node;

typedef struct rtx_scene_desc {
    uint32_t            n_shapes;
    const rtx_shape*    shapes;
    uint32_t            n_materials;
    const rtx_material* materials;
    uint32_t            n_patterns;
    const rtx_pattern*  patterns;
    uint32_t            n_nodes;
    const rtx_node*     nodes;
    float               light_position[4]; /* src/light.zig:5-8 */
    float               light_intensity[3];
    uint32_t            flags;             /* RTX_SCENE_* */
} rtx_scene_desc;

/* rtx_scene_desc.flags */
enum {
    /* Build the triangle BVHs on the device (LBVH: Morton sort + Karras hierarchy, rtx_lbvh.hip) instead
     * of on the host (binned SAH).  Same canvas either way — a tree only decides which triangles are
     * tested; the device build takes milliseconds where the host build of a 100 k - 400 k triangle mesh
     * takes 70 - 250 ms, walks a few per cent slower, and is what rtx_scene_refit needs. */
    RTX_SCENE_BVH_DEVICE = 1u << 0
};

/* Camera with its inverse precomputed on the host (src/camera.zig:17-23,77-78). */
typedef struct rtx_camera {
    uint32_t horizontal_size;
    uint32_t vertical_size;
    float    half_width;
    float    half_height;
    float    pixel_size;
    float    inverse_transform[16];
} rtx_camera;

typedef struct rtx_opts {
    uint32_t depth;        /* ray_depth_limit, src/world.zig:20; 0 means the default 5; max 5 */
    uint32_t rows_quirk;   /* 1: render only 8*floor(H/8) rows, the rest stay black (src/camera.zig:87-88) */
    uint32_t batch_pixels; /* pixels per wavefront batch; 0 = default */
    /* Row-strip sharding for one-process-per-GPU runs: this call renders the strips
     * s with s % strip_count == strip_index, a strip being strip_rows consecutive rows.
     * strip_count <= 1 renders the whole frame.  Output rows keep their place in the
     * full canvas unless compact_rows != 0 (then the rendered rows are packed in order). */
    uint32_t strip_rows;
    uint32_t strip_count;
    uint32_t strip_index;
    uint32_t compact_rows;
    uint32_t profile;      /* 1: time each kernel class with HIP events (adds syncs) */
    uint32_t no_bvh;       /* 1: test every triangle of a group linearly (A/B check of the triangle BVH) */
    uint32_t full_w;       /* 1: keep the w lane of every tuple operation even where the scene makes it an exact
                              0 / 1 (A/B check of the canonical-w kernels; same canvas, slower) */
    uint32_t lanes;        /* batches of a frame alternate between this many work-buffer sets on as many
                              streams, so one batch's kernels fill the tails of the other's: 1 or 2;
                              0 = 1.  Two lanes are opt-in: they win 1-8 % on 4K-8K mesh frames when the runtime
                              puts the two lane streams on different hardware queues and lose 24 % when it puts
                              them on one (which depends on the streams the process created before): measure
                              both in the calling process */
    uint32_t containers;   /* RTX_CONTAINERS_*: how prepareComputationsForRefraction's container walk compares entries */
    uint32_t real;         /* RTX_REAL_*: the arithmetic type of the frame */
    uint32_t all_shadows;  /* 1: traverse every isShadowed call, also where the light is behind the surface and the
                              answer cannot reach the colour (A/B check; same canvas, same counts but
                              rtx_stats.shadow_rays_skipped == 0, slower) */
    uint32_t work_set;     /* 0..2: which of the scene's independent sets of work buffers (ray queues, ray-tree nodes,
                              counters) this frame runs through.  Frames on different sets, queued on different streams,
                              may be in flight together — the deep, latency-bound levels of one frame then share the chip
                              with the next frame's level 0 (headline frame loop: 1.40 -> 1.21 ms per frame with two in
                              flight); frames on the same set must be ordered by the caller (one stream).  Needs lanes <= 1
                              and profile == 0 unless 0; every set is allocated on first use */
} rtx_opts;

/* rtx_opts.real.  F32 (default) is the reference's own arithmetic (src/tuple.zig:6: Tuple = @Vector(4, f32)) and
 * the parity mode.  F64 runs the same kernels instantiated over double — every computed quantity, the ray
 * queues and the ray-tree nodes; the scene records stay f32 — a reporting instantiation for the FP64 vector
 * roofline BASELINE.json's north_star asks about; its canvas differs from the reference's f32 arithmetic
 * where rounding decides (tests compare it with the oracle's own f64 build).  Plainest kernel variants only. */
enum { RTX_REAL_F32 = 0, RTX_REAL_F64 = 1 };

/* rtx_opts.containers.  V1 (default) is the reference as written: std.meta.eql on whole Intersections,
 * t included (src/intersection.zig:104,128), so an exit never removes its entry and n1 is the index of
 * the entry just before the hit.  BOOK compares shapes only — the algorithm of the book the reference
 * follows, and what the reference's shipped refractive images (samples/images/glass_spheres.ppm, the
 * glass cylinder of cylinder_world.ppm) were rendered with by an older revision: an entry
 * toggles its shape in and out of the list.  BOOK exists to report against those images; it runs the
 * plainest kernel variants. */
enum { RTX_CONTAINERS_V1 = 0, RTX_CONTAINERS_BOOK = 1 };

typedef struct rtx_stats {
    uint64_t rays_primary;   /* colorAtInternal calls at the root (src/world.zig:85) */
    uint64_t rays_shadow;    /* isShadowed calls (src/world.zig:97) */
    uint64_t rays_secondary; /* colorAtInternal calls below the root */
    uint64_t level_rays[6];  /* rays traced per depth level */
    uint64_t nodes;          /* interior ray-tree nodes stored for the combine pass */
    uint64_t level_hits[6];  /* hits per depth level (each one is an isShadowed call) */
    uint64_t hits_patterned; /* hits whose material has a pattern (Pattern.atShape evaluated) */
    uint64_t hits_transparent; /* hits on transparency != 0 above the depth limit (refraction set-up evaluated) */
    float    ms_total;       /* device time of the whole call (HIP events on the render stream) */
    float    ms_trace_shade; /* sum over launches of the trace+shade kernel (profile=1) */
    float    ms_combine;     /* sum over launches of the combine kernel (profile=1) */
    uint32_t launches_trace_shade;
    uint32_t launches_combine;
    uint32_t rows_rendered;
    uint32_t kernel_flags;   /* RTX_KERNEL_*: the k_trace_shade variant this frame's level kernels were */
    /* Event counts of the frame, closest-hit, predecessor and shadow traversals together (what the
     * roofline's algorithmic flops are priced from, SURVEY 8d's table): */
    uint64_t bvh_box_tests;      /* box tests of the triangle-BVH walks (new design; 16 flops like bnd.intersect) */
    uint64_t bvh_triangle_tests; /* Triangle.intersect calls on BVH leaves (src/triangle.zig:32-50) */
    uint64_t gate_tests;         /* bnd.intersect calls on group boxes (src/group.zig:58, src/bounds.zig:149-158) */
    uint64_t shape_tests[6];     /* Shape.intersect calls on linearly tested shapes, by RTX_* type */
    uint64_t shadow_rays_skipped; /* isShadowed calls counted in rays_shadow but not traversed: the light is behind the
                                    surface (src/material.zig:72-79), where lighting() returns the ambient term shadowed
                                    or not.  0 for a scene with a negative, -0 or non-finite colour / ambient / intensity */
} rtx_stats;

/* rtx_stats.kernel_flags: template flags of the trace+shade kernel a frame dispatched. */
enum {
    RTX_KERNEL_CANONICAL_W = 1u << 0, /* w lanes left out (every point w == 1, every vector w == 0 exactly) */
    RTX_KERNEL_BVH         = 1u << 1, /* some gate's triangle run is walked through its BVH */
    RTX_KERNEL_LDS_TABLES  = 1u << 2, /* every per-lane scene table staged in LDS */
    RTX_KERNEL_TRACK_N1    = 1u << 3, /* n1 predecessor tracked (some material is transparent) */
    RTX_KERNEL_LDS_HIT_RECORDS = 1u << 4, /* 16-byte hit records, materials and patterns staged in LDS; shape records in global memory */
    RTX_KERNEL_F64         = 1u << 5  /* the double-precision instantiation (rtx_opts.real = RTX_REAL_F64) */
};

typedef struct rtx_scene rtx_scene; /* opaque; owns the device copies */

/* Number of HIP devices visible, or a negative RTX_E_* code. */
int rtx_device_count(void);

/* Validate and upload a flattened scene to `device`. */
int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out);
void rtx_scene_destroy(rtx_scene* scene);

/*
 * The same scene after its shapes moved (new vertices, transforms, boxes, light; materials and patterns stay
 * as created: they decide the work buffers' layout):
 * `desc` must have the structure the scene was created with (same counts, same node ranges, every shape
 * of the same type AND with the same material index: which materials are in use decided at creation whether
 * hits can spawn one or two child rays, and the ray queues are sized for that — a changed index is rejected
 * with RTX_E_INVALID).  Records are uploaded again and the device-built trees are refitted — boxes
 * recomputed bottom-up over the unchanged topology — instead of rebuilt.  Needs RTX_SCENE_BVH_DEVICE at
 * creation.  No counterpart in the reference (its groups are rebuilt by value every frame).
 * Blocking, and it waits for the whole device first (hipDeviceSynchronize): frames still in flight on any
 * stream finish on the old records.  A descriptor that cannot be refitted (structure changed, a triangle that
 * cannot be bounded) is rejected before any device record changes; if an error strikes later (a HIP failure
 * half way), the scene is marked unusable and every later call on it fails until it is destroyed.
 */
int rtx_scene_refit(rtx_scene* scene, const rtx_scene_desc* desc);

/* Rows rtx_render would produce for these options (compact_rows layout). */
int rtx_rows_for(const rtx_camera* cam, const rtx_opts* opts, uint32_t* n_rows);

/*
 * render() — src/camera.zig:84-105.  Fills `rgb_out` (caller-owned host memory,
 * H*W*3 f32 row-major like Canvas.pixels, src/canvas.zig:15-28,52-54; or
 * n_rows*W*3 when compact_rows) — blocking.  `opts` and `stats` may be NULL.
 */
int rtx_render(rtx_scene* scene, const rtx_camera* cam, const rtx_opts* opts, float* rgb_out, rtx_stats* stats);

/*
 * Same, but the canvas stays on the device: `d_rgb_out` is a device pointer of the same
 * shape, `stream` a hipStream_t (NULL = the library's own stream).  The call returns after
 * the work is enqueued and stats (if requested) are read back, i.e. it synchronises only
 * when `stats` is non-NULL.  `d_rgb8_out`, if non-NULL, also receives the canvas quantised
 * as canvas.zig:45-47 does (round(clamp(c*255, 0, 255)) as u8, 3 bytes per pixel).
 */
int rtx_render_device(rtx_scene* scene, const rtx_camera* cam, const rtx_opts* opts, void* d_rgb_out,
                      void* d_rgb8_out, void* stream, rtx_stats* stats);

/*
 * rtx_render_device for callers that cannot allocate — code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference (whose render() allocates its
 * canvas and its per-ray lists, src/camera.zig:84-105, src/world.zig:86).  Same arguments and the same
 * canvas, bit for bit, as rtx_render_device without `stats`; the ray queues, ray-tree nodes and
 * counters live in a file-scope device array of 29 MiB that every GPU the library is loaded on holds
 * until the process ends, instead of in buffers grown with hipMalloc.  The call allocates and frees
 * nothing, on the host or on the device, and does not synchronise.
 * Limit: frames up to rtx_static_max_width() = 1024 pixels wide, any height (the frame is rendered in
 * 8-row bands of at most 8192 pixels); a wider camera is rejected with RTX_E_INVALID.  opts->lanes and
 * opts->profile are ignored (one stream, no per-launch events).
 * Every call on a GPU — from any rtx_scene, any thread — shares that GPU's array: calls that overlap
 * on one GPU race; calls ordered on one stream, or separated by a synchronisation, are fine.
 */
int rtx_render_device_static(rtx_scene* scene, const rtx_camera* cam, const rtx_opts* opts, void* d_rgb_out,
                             void* d_rgb8_out, void* stream);
uint32_t rtx_static_max_width(void);

/*
 * ---- one frame on several GPUs of this process ------------------------------------------
 *
 * render() fans out and joins by itself (src/camera.zig:84-105: eight threads over static row
 * bands); rtx_multi_* do the same over GPUs.  rtx_multi_create replicates the scene on `devices`
 * (NULL / 0 = every visible device; an index may appear more than once, which is how a one-GPU box
 * rehearses the path).  A render cuts the frame into n_devices * chunks_per_gpu interleaved sets of
 * 16-row strips; one host thread per GPU takes sets from a shared counter until none is left
 * (chunks_per_gpu = 1: each GPU renders one set; > 1: a GPU keeps up to min(chunks_per_gpu, 3) sets in
 * flight, each on a stream, chunk canvas and work set (rtx_opts.work_set) of its own, and takes another
 * only when one of them is done — dynamic stealing, where the reference's bands are static,
 * src/camera.zig:88,99; with opts->lanes = 2 one set at a time), renders each with
 * the single-GPU pipeline and puts its rows in place:
 *   rtx_multi_render         into caller-owned HOST canvases (f32 and / or u8, either may be NULL),
 *                            every GPU copying over its own PCIe link;
 *   rtx_multi_render_device  into canvases on the FIRST device of the list, peers copying over
 *                            their own xGMI link (one strided hipMemcpy2DAsync per set and canvas; peer
 *                            access is enabled in rtx_multi_create) — the canvas exchange.
 * The calling thread's current device is restored before each of these returns.
 * Both block until the frame is complete.  The canvas equals rtx_render's bit for bit for any
 * device list and chunk count.  `opts` as for rtx_render, except that strip_count / compact_rows
 * must be zero (the strips are this call's) and profile is ignored; `stats` (may be NULL) sums the
 * ray counts over all sets, ms_total is the wall time of the call.
 */
typedef struct rtx_multi rtx_multi;
int rtx_multi_create(const rtx_scene_desc* desc, const int* devices, uint32_t n_devices, rtx_multi** out);
void rtx_multi_destroy(rtx_multi* multi);
uint32_t rtx_multi_device_count(const rtx_multi* multi);
/* Directed peer-access links between the first device and the others that rtx_multi_create found switched on or
 * switched on itself (hipDeviceEnablePeerAccess): 2 * (distinct peers) when every peer can reach the root directly. */
uint32_t rtx_multi_peer_links(const rtx_multi* multi);
int rtx_multi_render(rtx_multi* multi, const rtx_camera* cam, const rtx_opts* opts, uint32_t chunks_per_gpu,
                     float* rgb_out, uint8_t* rgb8_out, rtx_stats* stats);
int rtx_multi_render_device(rtx_multi* multi, const rtx_camera* cam, const rtx_opts* opts, uint32_t chunks_per_gpu,
                            void* d_rgb_out, void* d_rgb8_out, rtx_stats* stats);

/* Global row of every local row a strip set renders (the compact_rows layout), host only: the
 * partition arithmetic of the multi-GPU paths.  rows_out may be NULL to query the count. */
int rtx_strip_row_map(const rtx_camera* cam, const rtx_opts* opts, uint32_t* rows_out, uint32_t cap, uint32_t* n_rows);

/*
 * With opts->profile = 1 every kernel launch is bracketed by HIP events on the render
 * stream.  A call that passes `stats` reads them back itself; calls that do not (the
 * asynchronous frame loop) leave them pending, and this collects everything recorded since
 * the last read into the ms_* / launches_* fields of `out` (synchronises the stream).
 */
int rtx_profile_collect(rtx_scene* scene, rtx_stats* out);

/* colorAt() — src/world.zig:81-83 — for n rays (origins/directions: n*4 f32; rgb: n*3 f32), host buffers. */
int rtx_color_at(rtx_scene* scene, const float* origins, const float* directions, uint32_t n, uint32_t depth,
                 float* rgb_out);

/*
 * Host-only self-check of the triangle BVHs rtx_scene_create would build for `desc` (new design,
 * no counterpart in the reference, whose groups are brute force: src/group.zig:55-87).  Validates
 * every tree (forward skips, each hittable triangle in exactly one leaf) and, for n rays (n*4 f32
 * each), compares the closest hit and the n1 predecessor the traversal code finds — the same
 * source the kernels compile — with brute force over the same triangles.  Needs no device.
 * Returns 0 and the counts, or RTX_E_INVALID with a message when a tree is malformed.
 */
int rtx_bvh_selftest(const rtx_scene_desc* desc, const float* origins, const float* directions, uint32_t n,
                     uint32_t* n_bvh_nodes, uint32_t* n_bvh_tris, uint32_t* mismatches);

/* Box tests and triangle tests made by the closest-hit walks of this thread's last rtx_bvh_selftest
 * (tree-quality statistics; host only). */
void rtx_bvh_selftest_work(uint64_t* box_tests, uint64_t* triangle_tests);

const char* rtx_last_error(void);

/* A short hash of the sources this library was built from (ray-tracer_amd/Makefile): lets measurements kept in files
 * say which build they were taken on. */
const char* rtx_build_id(void);

/* sizeof of the public structs as this library was compiled (0 rtx_shape, 1 rtx_material, 2 rtx_pattern, 3 rtx_node,
 * 4 rtx_scene_desc, 5 rtx_camera, 6 rtx_opts, 7 rtx_stats): lets a binding in another language check its mirrors. */
uint32_t rtx_struct_size(uint32_t which);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
