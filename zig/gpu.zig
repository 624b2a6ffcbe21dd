//! gpu.zig — binding of arieldon/ray-tracer (Zig 0.9.x dialect, as the reference is written) to
//! libraytracer_amd.so: a drop-in for `camera.render` (src/camera.zig:84-105) and `world.colorAt`
//! (src/world.zig:81-83) that runs the per-pixel loop on MI355X GPUs.
//!
//! Place this file in the reference's `src/`, export it from src/main.zig (`pub const gpu = @import("gpu.zig");`),
//! call `rt.gpu.render(allocator, camera, world)` where the samples call `rt.render(...)`, and link with
//!     exe.linkSystemLibrary("raytracer_amd"); exe.linkLibC(); exe.addLibPath("<repo>/ray-tracer_amd");
//!
//! NOT COMPILED in this repository's image (it has no Zig toolchain, SURVEY.md fact 3): the extern
//! structs below are field-for-field copies of include/rt_scene.h, whose layout the C ABI tests pin
//! (tests/test_abi.py); treat any compile error as a typo of mine, not as a property of the library.
const std = @import("std");
const cam = @import("camera.zig");
const cnv = @import("canvas.zig");
const con = @import("cone.zig");
const cub = @import("cube.zig");
const cyl = @import("cylinder.zig");
const grp = @import("group.zig");
const mat = @import("matrix.zig");
const mtl = @import("material.zig");
const pat = @import("pattern.zig");
const pln = @import("plane.zig");
const ray = @import("ray.zig");
const shp = @import("shape.zig");
const sph = @import("sphere.zig");
const tri = @import("triangle.zig");
const tup = @import("tuple.zig");
const wrd = @import("world.zig");

// ---- include/rt_scene.h ---------------------------------------------------------------------------

pub const CMat4 = extern struct { m: [4][4]f32 }; // mat.Matrix = [4]@Vector(4, f32), row-major
pub const CPattern = extern struct { a: [3]f32, b: [3]f32, transform: CMat4, color_map: i32 };
pub const CMaterial = extern struct {
    color: [3]f32,
    has_pattern: i32,
    pattern: CPattern,
    ambient: f32,
    diffuse: f32,
    specular: f32,
    shininess: f32,
    reflective: f32,
    refractive_index: f32,
    transparency: f32,
};
pub const CAttrs = extern struct { transform: CMat4, material: CMaterial };
pub const CSphere = extern struct { common_attrs: CAttrs };
pub const CPlane = extern struct { common_attrs: CAttrs };
pub const CCube = extern struct { common_attrs: CAttrs };
pub const CCylinder = extern struct { common_attrs: CAttrs, minimum: f32, maximum: f32, closed: i32 };
pub const CCone = extern struct { common_attrs: CAttrs, minimum: f32, maximum: f32, closed: i32 };
pub const CTriangle = extern struct {
    common_attrs: CAttrs,
    p0: [4]f32,
    p1: [4]f32,
    p2: [4]f32,
    e0: [4]f32,
    e1: [4]f32,
    normal: [4]f32,
};
pub const CBounds = extern struct { minimum: [4]f32, maximum: [4]f32 };
pub const CGroup = extern struct {
    transform: CMat4,
    has_bounding_box: i32,
    bounding_box: CBounds,
    n_spheres: u32,
    spheres: ?[*]const CSphere,
    n_planes: u32,
    planes: ?[*]const CPlane,
    n_cubes: u32,
    cubes: ?[*]const CCube,
    n_cylinders: u32,
    cylinders: ?[*]const CCylinder,
    n_cones: u32,
    cones: ?[*]const CCone,
    n_triangles: u32,
    triangles: ?[*]const CTriangle,
    n_subgroups: u32,
    subgroups: ?[*]const CGroup,
};
pub const CWorld = extern struct {
    light_position: [4]f32,
    light_intensity: [3]f32,
    n_spheres: u32,
    spheres: ?[*]const CSphere,
    n_planes: u32,
    planes: ?[*]const CPlane,
    n_cubes: u32,
    cubes: ?[*]const CCube,
    n_cylinders: u32,
    cylinders: ?[*]const CCylinder,
    n_cones: u32,
    cones: ?[*]const CCone,
    n_triangles: u32,
    triangles: ?[*]const CTriangle,
    n_groups: u32,
    groups: ?[*]const CGroup,
};
pub const CCamera = extern struct {
    horizontal_size: u32,
    vertical_size: u32,
    field_of_view: f32,
    half_width: f32,
    half_height: f32,
    pixel_size: f32,
    transform: CMat4,
};
pub const CRay = extern struct { origin: [4]f32, direction: [4]f32 };
pub const CRayCounts = extern struct { primary: u64, shadow: u64, secondary: u64 };

// ---- include/rt_api.h -----------------------------------------------------------------------------

extern fn rt_render(c: *const CCamera, w: *const CWorld, canvas: [*]f32, counts: ?*CRayCounts) c_int;
extern fn rt_color_at(w: *const CWorld, rays: [*]const CRay, n: u32, rgb_out: [*]f32) c_int;
extern fn rtx_last_error() [*:0]const u8;

pub const GpuError = error{ NoDevice, Hip, Invalid, OutOfMemory };

fn check(rc: c_int) GpuError!void {
    switch (rc) {
        0 => {},
        -2 => return GpuError.NoDevice,
        -3 => return GpuError.Hip,
        -4 => return GpuError.OutOfMemory,
        else => return GpuError.Invalid,
    }
}

// ---- field-for-field copies -----------------------------------------------------------------------

fn toCMat4(m: mat.Matrix) CMat4 {
    var out: CMat4 = undefined;
    for (m) |row, r| {
        const lanes: [4]f32 = row;
        out.m[r] = lanes;
    }
    return out;
}

fn toC4(t: tup.Tuple) [4]f32 {
    const lanes: [4]f32 = t;
    return lanes;
}

fn toC3(c: cnv.Color) [3]f32 {
    const lanes: [3]f32 = c;
    return lanes;
}

/// Pattern.color_map is a function pointer (src/pattern.zig:17); the library takes the tag of one of
/// the four functions the reference defines (RT_PATTERN_* in rt_scene.h).
fn patternTag(p: pat.Pattern) i32 {
    if (p.color_map == pat.stripe) return 0;
    if (p.color_map == pat.gradient) return 1;
    if (p.color_map == pat.ring) return 2;
    return 3; // pat.checker
}

fn toCMaterial(m: mtl.Material) CMaterial {
    var out = std.mem.zeroes(CMaterial);
    out.color = toC3(m.color);
    if (m.pattern) |p| {
        out.has_pattern = 1;
        out.pattern = .{ .a = toC3(p.a), .b = toC3(p.b), .transform = toCMat4(p.transform), .color_map = patternTag(p) };
    } else {
        out.pattern.transform = toCMat4(mat.identity);
    }
    out.ambient = m.ambient;
    out.diffuse = m.diffuse;
    out.specular = m.specular;
    out.shininess = m.shininess;
    out.reflective = m.reflective;
    out.refractive_index = m.refractive_index;
    out.transparency = m.transparency;
    return out;
}

fn toCAttrs(a: shp.CommonShapeAttributes) CAttrs {
    return .{ .transform = toCMat4(a.transform), .material = toCMaterial(a.material) };
}

fn copySpheres(a: std.mem.Allocator, xs: []const sph.Sphere) ![]CSphere {
    var out = try a.alloc(CSphere, xs.len);
    for (xs) |x, i| out[i] = .{ .common_attrs = toCAttrs(x.common_attrs) };
    return out;
}
fn copyPlanes(a: std.mem.Allocator, xs: []const pln.Plane) ![]CPlane {
    var out = try a.alloc(CPlane, xs.len);
    for (xs) |x, i| out[i] = .{ .common_attrs = toCAttrs(x.common_attrs) };
    return out;
}
fn copyCubes(a: std.mem.Allocator, xs: []const cub.Cube) ![]CCube {
    var out = try a.alloc(CCube, xs.len);
    for (xs) |x, i| out[i] = .{ .common_attrs = toCAttrs(x.common_attrs) };
    return out;
}
fn copyCylinders(a: std.mem.Allocator, xs: []const cyl.Cylinder) ![]CCylinder {
    var out = try a.alloc(CCylinder, xs.len);
    for (xs) |x, i| out[i] = .{
        .common_attrs = toCAttrs(x.common_attrs),
        .minimum = x.minimum,
        .maximum = x.maximum,
        .closed = @boolToInt(x.closed),
    };
    return out;
}
fn copyCones(a: std.mem.Allocator, xs: []const con.Cone) ![]CCone {
    var out = try a.alloc(CCone, xs.len);
    for (xs) |x, i| out[i] = .{
        .common_attrs = toCAttrs(x.common_attrs),
        .minimum = x.minimum,
        .maximum = x.maximum,
        .closed = @boolToInt(x.closed),
    };
    return out;
}
fn copyTriangles(a: std.mem.Allocator, xs: []const tri.Triangle) ![]CTriangle {
    var out = try a.alloc(CTriangle, xs.len);
    for (xs) |x, i| out[i] = .{
        .common_attrs = toCAttrs(x.common_attrs),
        .p0 = toC4(x.p0),
        .p1 = toC4(x.p1),
        .p2 = toC4(x.p2),
        .e0 = toC4(x.e0),
        .e1 = toC4(x.e1),
        .normal = toC4(x.normal),
    };
    return out;
}

fn toCGroup(a: std.mem.Allocator, g: grp.Group) std.mem.Allocator.Error!CGroup {
    var out = std.mem.zeroes(CGroup);
    out.transform = toCMat4(g.transform);
    if (g.bounding_box) |b| {
        out.has_bounding_box = 1;
        out.bounding_box = .{ .minimum = toC4(b.minimum), .maximum = toC4(b.maximum) };
    }
    const spheres = try copySpheres(a, g.spheres.items);
    const planes = try copyPlanes(a, g.planes.items);
    const cubes = try copyCubes(a, g.cubes.items);
    const cylinders = try copyCylinders(a, g.cylinders.items);
    const cones = try copyCones(a, g.cones.items);
    const triangles = try copyTriangles(a, g.triangles.items);
    var subgroups = try a.alloc(CGroup, g.subgroups.items.len);
    for (g.subgroups.items) |sub, i| subgroups[i] = try toCGroup(a, sub);
    out.n_spheres = @intCast(u32, spheres.len);
    out.spheres = spheres.ptr;
    out.n_planes = @intCast(u32, planes.len);
    out.planes = planes.ptr;
    out.n_cubes = @intCast(u32, cubes.len);
    out.cubes = cubes.ptr;
    out.n_cylinders = @intCast(u32, cylinders.len);
    out.cylinders = cylinders.ptr;
    out.n_cones = @intCast(u32, cones.len);
    out.cones = cones.ptr;
    out.n_triangles = @intCast(u32, triangles.len);
    out.triangles = triangles.ptr;
    out.n_subgroups = @intCast(u32, subgroups.len);
    out.subgroups = subgroups.ptr;
    return out;
}

/// World -> CWorld; every array lives in `a` (use an arena and free it after the call).
pub fn toCWorld(a: std.mem.Allocator, w: wrd.World) !CWorld {
    var out = std.mem.zeroes(CWorld);
    out.light_position = toC4(w.light.position);
    out.light_intensity = toC3(w.light.intensity);
    const spheres = try copySpheres(a, w.spheres);
    const planes = try copyPlanes(a, w.planes);
    const cubes = try copyCubes(a, w.cubes);
    const cylinders = try copyCylinders(a, w.cylinders);
    const cones = try copyCones(a, w.cones);
    const triangles = try copyTriangles(a, w.triangles);
    var groups = try a.alloc(CGroup, w.groups.len);
    for (w.groups) |g, i| groups[i] = try toCGroup(a, g);
    out.n_spheres = @intCast(u32, spheres.len);
    out.spheres = spheres.ptr;
    out.n_planes = @intCast(u32, planes.len);
    out.planes = planes.ptr;
    out.n_cubes = @intCast(u32, cubes.len);
    out.cubes = cubes.ptr;
    out.n_cylinders = @intCast(u32, cylinders.len);
    out.cylinders = cylinders.ptr;
    out.n_cones = @intCast(u32, cones.len);
    out.cones = cones.ptr;
    out.n_triangles = @intCast(u32, triangles.len);
    out.triangles = triangles.ptr;
    out.n_groups = @intCast(u32, groups.len);
    out.groups = groups.ptr;
    return out;
}

pub fn toCCamera(c: cam.Camera) CCamera {
    return .{
        .horizontal_size = c.horizontal_size,
        .vertical_size = c.vertical_size,
        .field_of_view = c.field_of_view,
        .half_width = c.half_width,
        .half_height = c.half_height,
        .pixel_size = c.pixel_size,
        .transform = toCMat4(c.transform),
    };
}

// ---- the drop-ins ---------------------------------------------------------------------------------

/// Drop-in for camera.render (src/camera.zig:84-105): same arguments, same caller-owned Canvas
/// ("Caller owns returned memory", :103), blocking.  The library fans out over every visible GPU
/// itself, as the reference fans out over eight threads.
pub fn render(allocator: std.mem.Allocator, c: cam.Camera, world: wrd.World) !cnv.Canvas {
    var image = try cnv.canvas(allocator, c.horizontal_size, c.vertical_size);
    errdefer image.deinit();
    var arena = std.heap.ArenaAllocator.init(allocator);
    defer arena.deinit();
    const cw = try toCWorld(arena.allocator(), world);
    const cc = toCCamera(c);
    // Color = @Vector(3, f32) is padded to 16 bytes: render into a tight f32 buffer, then copy.
    var rgb = try allocator.alloc(f32, image.pixels.len * 3);
    defer allocator.free(rgb);
    try check(rt_render(&cc, &cw, rgb.ptr, null));
    for (image.pixels) |*p, i| p.* = cnv.Color{ rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
    return image;
}

/// Drop-in for world.colorAt (src/world.zig:81-83), one ray.  (Batch rays through rt_color_at
/// directly where many are needed: every call uploads the world.)
pub fn colorAt(allocator: std.mem.Allocator, world: *const wrd.World, r: ray.Ray) !cnv.Color {
    var arena = std.heap.ArenaAllocator.init(allocator);
    defer arena.deinit();
    const cw = try toCWorld(arena.allocator(), world.*);
    const cr = CRay{ .origin = toC4(r.origin), .direction = toC4(r.direction) };
    var rgb: [3]f32 = undefined;
    try check(rt_color_at(&cw, @ptrCast([*]const CRay, &cr), 1, &rgb));
    return cnv.Color{ rgb[0], rgb[1], rgb[2] };
}

/// The library's message for the last failed call on this thread.
pub fn lastError() []const u8 {
    return std.mem.span(rtx_last_error());
}
