// rtx_multi.hip — one frame on several GPUs of one process, behind the C ABI (include/rtx.h).
//
// The reference's render() fans out and joins by itself: eight OS threads over static row bands
// (src/camera.zig:84-105).  Here the callee does the same over GPUs: the scene is replicated on
// every device (it is KBs to MBs), the frame is cut into interleaved sets of 16-row strips
// ("chunks": chunk k of n holds the strips s with s % n == k, so every chunk sees the same mix of
// cheap and expensive rows), one host thread per GPU takes chunks from a shared atomic counter until
// none is left — with more chunks than GPUs that is dynamic stealing: a GPU that finishes early takes
// the next chunk (a GPU keeps up to three chunks in flight, each on a stream, chunk canvas and work set of
// its own — a chunk of a split frame fills little of a chip per launch, and its six level launches cannot
// take less than their longest iterations — and claims another only when one of them is done) — renders
// each with the ordinary single-GPU pipeline (rtx_render_device, compact
// rows) and puts the rows where they belong: straight into the caller's host canvas over that GPU's
// own PCIe link, or into a canvas on the first device with peer copies (one strided 2-D copy per chunk
// and canvas over the peer's own xGMI link to the root: no ring, no staging; peer access is switched on
// once in rtx_multi_create where the hardware allows it) — the exchange step of the path.  A single
// process needs no collective library for peer-to-root copies; RCCL carries the same exchange where
// ranks are processes (bench.py, ray-tracer_amd/tiling.py).
//
// Every pixel is produced by the same chain of f32 operations whatever the decomposition, so the
// canvas equals the single-GPU canvas bit for bit (tests/test_gpu_parity.py).
//
// Host code only: this file launches no kernel of its own.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtx.h"

extern "C" void rtx_set_error(int code, const char* msg);  // rtx_device.hip: the thread-local message of rtx_last_error()

namespace {

constexpr uint32_t kStripRows = 16;

int fail(int code, const std::string& msg) {
    rtx_set_error(code, msg.c_str());
    return code;
}

// Restores the calling thread's current device when it goes out of scope: the entry points below move it around
// (hipSetDevice per GPU), and a caller — torch, say — must find the device it had selected.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

constexpr int kSlots = 3;  // chunks a GPU keeps in flight (rtx_opts.work_set has three sets)

struct Gpu {
    int device = 0;
    rtx_scene* scene = nullptr;
    hipStream_t stream[kSlots] = {};
    hipEvent_t done[kSlots] = {};      // the slot's last chunk has been rendered and copied out
    float* d_rgb[kSlots] = {};         // compact rows of one chunk
    uint8_t* d_rgb8[kSlots] = {};
    size_t cap_rows = 0, cap_width = 0;
};

}  // namespace

struct rtx_multi {
    std::vector<Gpu> gpus;
    uint32_t peer_links = 0;  // directed peer-access links to / from the first device that are switched on
};

namespace {

// Rows rendered at all (rows quirk, src/camera.zig:87-88).
uint32_t rendered_rows(const rtx_camera* cam, const rtx_opts* o) {
    return (o && o->rows_quirk) ? 8u * (cam->vertical_size / 8u) : cam->vertical_size;
}

int ensure_buffers(Gpu& g, size_t rows, size_t width) {
    if (rows <= g.cap_rows && width <= g.cap_width) return RTX_OK;
    // (only ever grown by the GPU's own worker before its first chunk of a frame is queued, or after a smaller frame whose
    // chunks rtx_multi_render has waited for: nothing is in flight on these buffers)
    for (int k = 0; k < kSlots; ++k) {
        if (g.d_rgb[k]) (void)hipFree(g.d_rgb[k]);
        if (g.d_rgb8[k]) (void)hipFree(g.d_rgb8[k]);
        g.d_rgb[k] = nullptr;
        g.d_rgb8[k] = nullptr;
    }
    g.cap_rows = g.cap_width = 0;
    const size_t n = (rows ? rows : 1) * width * 3;
    hipError_t e = hipSuccess;
    for (int k = 0; k < kSlots && e == hipSuccess; ++k) {
        e = hipMalloc(&g.d_rgb[k], n * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&g.d_rgb8[k], n);
    }
    if (e != hipSuccess) return fail(RTX_E_HIP, std::string("hipMalloc (multi-GPU chunk canvas): ") + hipGetErrorString(e));
    g.cap_rows = rows;
    g.cap_width = width;
    return RTX_OK;
}

struct Job {
    rtx_multi* m;
    const rtx_camera* cam;
    rtx_opts opts;          // the caller's options; strips are set per chunk
    uint32_t n_chunks;
    float* rgb_out;         // host (to_device == false) or device-0 (to_device == true) f32 canvas, may be NULL
    uint8_t* rgb8_out;      // the same for the u8 canvas, may be NULL
    bool to_device;
    bool want_stats;
    uint32_t slots;         // chunks a GPU keeps in flight (1 .. kSlots)
    uint32_t max_rows;      // rows of the largest chunk: the chunk canvases are sized once per frame, before anything is queued
    std::atomic<uint32_t> next{0};
};

struct Result {
    int rc = RTX_OK;
    std::string error;
    rtx_stats stats{};
    uint32_t chunks = 0;
};

void add_stats(rtx_stats& a, const rtx_stats& b) {
    a.rays_primary += b.rays_primary;
    a.rays_shadow += b.rays_shadow;
    a.rays_secondary += b.rays_secondary;
    a.nodes += b.nodes;
    a.hits_patterned += b.hits_patterned;
    a.hits_transparent += b.hits_transparent;
    a.bvh_box_tests += b.bvh_box_tests;
    a.bvh_triangle_tests += b.bvh_triangle_tests;
    a.gate_tests += b.gate_tests;
    for (int k = 0; k < 6; ++k) {
        a.level_rays[k] += b.level_rays[k];
        a.level_hits[k] += b.level_hits[k];
        a.shape_tests[k] += b.shape_tests[k];
    }
    a.shadow_rays_skipped += b.shadow_rays_skipped;
    a.rows_rendered += b.rows_rendered;
    a.kernel_flags |= b.kernel_flags;
}

// One GPU's host thread: take chunks until none is left.
void worker(Job* job, size_t gi, Result* res) {
    Gpu& g = job->m->gpus[gi];
    const uint32_t W = job->cam->horizontal_size;
    auto bail = [&](int rc, const std::string& what) {
        res->rc = rc;
        res->error = what;
    };
    hipError_t e = hipSetDevice(g.device);
    if (e != hipSuccess) return bail(RTX_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    int rc = ensure_buffers(g, job->max_rows, W);
    if (rc) return bail(rc, rtx_last_error());
    uint32_t taken = 0;  // chunks this GPU has queued
    for (;;) {
        // A slot is free again when its last chunk is done: only then does this GPU claim another chunk, so that the
        // chunks go to the GPUs that finish early (not to the threads that queue fastest).
        const uint32_t slot = taken % job->slots;
        if (taken >= job->slots) {
            e = hipEventSynchronize(g.done[slot]);
            if (e != hipSuccess) return bail(RTX_E_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
        }
        const uint32_t k = job->next.fetch_add(1u);
        if (k >= job->n_chunks) break;
        rtx_opts o = job->opts;
        o.strip_rows = kStripRows;
        o.strip_count = job->n_chunks;
        o.strip_index = k;
        o.compact_rows = 1;
        o.profile = 0;
        o.work_set = job->slots > 1 ? slot : 0u;
        uint32_t rows = 0;
        rc = rtx_rows_for(job->cam, &o, &rows);
        if (rc) return bail(rc, rtx_last_error());
        if (rows == 0) continue;  // more chunks than strips
        if (rows > g.cap_rows) return bail(RTX_E_INVALID, "chunk larger than the chunk canvas");
        hipStream_t st_k = g.stream[slot];
        rtx_stats st;
        rc = rtx_render_device(g.scene, job->cam, &o, g.d_rgb[slot], job->rgb8_out ? g.d_rgb8[slot] : nullptr, st_k, job->want_stats ? &st : nullptr);
        if (rc) return bail(rc, rtx_last_error());
        if (job->want_stats) add_stats(res->stats, st);
        // The chunk's strips back to their rows: local strip j is global strip k + j * n_chunks, so the chunk is one
        // 2-D copy per canvas — "rows" of one strip (16 * W * 3 elements) each, packed at the source and
        // n_chunks strips apart at the destination — plus one short copy when the frame's last strip is partial
        // and belongs to this chunk.
        const size_t strip_elems = (size_t)kStripRows * W * 3;
        const uint32_t full_strips = rows / kStripRows, tail_rows = rows % kStripRows;
        auto put = [&](void* dst_base, const void* src_base, size_t elt) -> hipError_t {
            char* dst = static_cast<char*>(dst_base) + (size_t)k * strip_elems * elt;
            const char* src = static_cast<const char*>(src_base);
            const hipMemcpyKind kind = job->to_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
            hipError_t err = hipSuccess;
            if (full_strips)
                err = hipMemcpy2DAsync(dst, (size_t)job->n_chunks * strip_elems * elt, src, strip_elems * elt, strip_elems * elt, full_strips, kind, st_k);
            if (err == hipSuccess && tail_rows)
                err = hipMemcpyAsync(dst + (size_t)full_strips * job->n_chunks * strip_elems * elt, src + (size_t)full_strips * strip_elems * elt,
                                     (size_t)tail_rows * W * 3 * elt, kind, st_k);
            return err;
        };
        if (job->rgb_out) {
            e = put(job->rgb_out, g.d_rgb[slot], sizeof(float));
            if (e != hipSuccess) return bail(RTX_E_HIP, std::string("canvas copy: ") + hipGetErrorString(e));
        }
        if (job->rgb8_out) {
            e = put(job->rgb8_out, g.d_rgb8[slot], 1);
            if (e != hipSuccess) return bail(RTX_E_HIP, std::string("canvas copy: ") + hipGetErrorString(e));
        }
        // The slot's buffers and work set are reused by the chunk this GPU takes `slots` chunks later: same stream, so
        // the copies above are ordered before that render.
        e = hipEventRecord(g.done[slot], st_k);
        if (e != hipSuccess) return bail(RTX_E_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
        res->chunks++;
        ++taken;
    }
    for (int k = 0; k < kSlots; ++k) {
        e = hipStreamSynchronize(g.stream[k]);
        if (e != hipSuccess) return bail(RTX_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
}

int multi_render(rtx_multi* m, const rtx_camera* cam, const rtx_opts* opts, uint32_t chunks_per_gpu, float* rgb_out, uint8_t* rgb8_out,
                 bool to_device, rtx_stats* stats) {
    if (!m || m->gpus.empty()) return fail(RTX_E_INVALID, "multi-GPU scene is NULL");
    if (!cam) return fail(RTX_E_INVALID, "camera is NULL");
    DeviceGuard guard;  // worker 0 runs on this thread and selects its GPU
    if (!rgb_out && !rgb8_out) return fail(RTX_E_INVALID, "no output canvas");
    if (opts && (opts->strip_count > 1 || opts->compact_rows)) return fail(RTX_E_INVALID, "strips are this call's to set: leave strip_count and compact_rows zero");
    uint32_t probe = 0;
    int rc = rtx_rows_for(cam, opts, &probe);  // validates the camera
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    Job job;
    job.m = m;
    job.cam = cam;
    std::memset(&job.opts, 0, sizeof job.opts);
    if (opts) job.opts = *opts;
    const size_t n_gpus = m->gpus.size();
    if (chunks_per_gpu == 0) chunks_per_gpu = 1;
    if (chunks_per_gpu > 64) return fail(RTX_E_INVALID, "chunks_per_gpu at most 64");
    job.n_chunks = (uint32_t)n_gpus * chunks_per_gpu;
    job.rgb_out = rgb_out;
    job.rgb8_out = rgb8_out;
    job.to_device = to_device;
    job.want_stats = stats != nullptr;
    // two lanes (rtx_opts.lanes) use the scene's work sets themselves; a counted frame reads its statistics back chunk by chunk
    job.slots = (job.opts.lanes > 1 || chunks_per_gpu < 2) ? 1u : (chunks_per_gpu < (uint32_t)kSlots ? chunks_per_gpu : (uint32_t)kSlots);
    job.max_rows = 0;
    for (uint32_t k = 0; k < job.n_chunks && k < 2u; ++k) {  // chunk 0 holds the most rows (and the partial last strip goes to a later one at most as large)
        rtx_opts o = job.opts;
        o.strip_rows = kStripRows;
        o.strip_count = job.n_chunks;
        o.strip_index = k;
        o.compact_rows = 1;
        uint32_t rows = 0;
        rc = rtx_rows_for(cam, &o, &rows);
        if (rc) return rc;
        if (rows > job.max_rows) job.max_rows = rows;
    }
    // Rows the reference never renders stay black (src/camera.zig:87-88, src/canvas.zig:23-25).
    const uint32_t W = cam->horizontal_size, H = cam->vertical_size, total = rendered_rows(cam, &job.opts);
    if (total < H) {
        const size_t off = (size_t)total * W * 3, count = (size_t)(H - total) * W * 3;
        if (to_device) {
            hipError_t e = hipSetDevice(m->gpus[0].device);
            if (e == hipSuccess && rgb_out) e = hipMemsetAsync(rgb_out + off, 0, count * sizeof(float), m->gpus[0].stream[0]);
            if (e == hipSuccess && rgb8_out) e = hipMemsetAsync(rgb8_out + off, 0, count, m->gpus[0].stream[0]);
            if (e != hipSuccess) return fail(RTX_E_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
        } else {
            if (rgb_out) std::memset(rgb_out + off, 0, count * sizeof(float));
            if (rgb8_out) std::memset(rgb8_out + off, 0, count);
        }
    }
    std::vector<Result> results(n_gpus);
    std::vector<std::thread> threads;
    threads.reserve(n_gpus);
    for (size_t g = 1; g < n_gpus; ++g) threads.emplace_back(worker, &job, g, &results[g]);
    worker(&job, 0, &results[0]);  // the calling thread serves the first GPU
    for (std::thread& t : threads) t.join();
    for (const Result& r : results)
        if (r.rc) return fail(r.rc, r.error);
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (const Result& r : results) add_stats(*stats, r.stats);
        stats->ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_multi_create(const rtx_scene_desc* desc, const int* devices, uint32_t n_devices, rtx_multi** out) {
    if (!out) return fail(RTX_E_INVALID, "out is NULL");
    *out = nullptr;
    const int visible = rtx_device_count();
    if (visible <= 0) return fail(RTX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    std::vector<int> devs;
    if (!devices || n_devices == 0) {
        for (int d = 0; d < visible; ++d) devs.push_back(d);
    } else {
        devs.assign(devices, devices + n_devices);
    }
    if (devs.size() > 64) return fail(RTX_E_INVALID, "at most 64 devices");
    DeviceGuard guard;
    rtx_multi* m = new (std::nothrow) rtx_multi();
    if (!m) return fail(RTX_E_NOMEM, "out of host memory");
    m->gpus.resize(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        Gpu& g = m->gpus[i];
        g.device = devs[i];
        int rc = rtx_scene_create(desc, g.device, &g.scene);  // validates the device index and the descriptor
        hipError_t e = hipSuccess;
        if (rc == RTX_OK) {
            e = hipSetDevice(g.device);
            for (int k = 0; k < kSlots && e == hipSuccess; ++k) {
                e = hipStreamCreateWithFlags(&g.stream[k], hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&g.done[k], hipEventDisableTiming);
            }
            if (e != hipSuccess) rc = fail(RTX_E_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
        }
        if (rc) {
            const std::string msg = rtx_last_error();
            rtx_multi_destroy(m);
            return fail(rc, msg);
        }
    }
    // Peers write into canvases on the first device (rtx_multi_render_device): switch peer access on once, in both
    // directions, where the hardware allows it (without it the runtime stages the copies through the host).
    // "Already enabled" is not an error — another rtx_multi, or the caller, may have done it.
    const int root = m->gpus[0].device;
    for (size_t i = 1; i < m->gpus.size(); ++i) {
        const int peer = m->gpus[i].device;
        if (peer == root) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int from = dir ? root : peer, to = dir ? peer : root;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) continue;
            if (hipSetDevice(from) != hipSuccess) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e == hipSuccess) m->peer_links++;
            else if (e == hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();
                m->peer_links++;
            }
        }
    }
    (void)hipGetLastError();
    *out = m;
    return RTX_OK;
}

void rtx_multi_destroy(rtx_multi* m) {
    if (!m) return;
    DeviceGuard guard;
    for (Gpu& g : m->gpus) {
        if (!g.scene && !g.stream[0] && !g.d_rgb[0]) continue;  // never set up (rtx_multi_create failed before it)
        (void)hipSetDevice(g.device);
        for (int k = 0; k < kSlots; ++k) {
            if (g.stream[k]) {
                (void)hipStreamSynchronize(g.stream[k]);
                (void)hipStreamDestroy(g.stream[k]);
            }
            if (g.done[k]) (void)hipEventDestroy(g.done[k]);
            if (g.d_rgb[k]) (void)hipFree(g.d_rgb[k]);
            if (g.d_rgb8[k]) (void)hipFree(g.d_rgb8[k]);
        }
        if (g.scene) rtx_scene_destroy(g.scene);
    }
    delete m;
}

uint32_t rtx_multi_device_count(const rtx_multi* m) { return m ? (uint32_t)m->gpus.size() : 0u; }
uint32_t rtx_multi_peer_links(const rtx_multi* m) { return m ? m->peer_links : 0u; }

int rtx_multi_render(rtx_multi* m, const rtx_camera* cam, const rtx_opts* opts, uint32_t chunks_per_gpu, float* rgb_out,
                     uint8_t* rgb8_out, rtx_stats* stats) {
    return multi_render(m, cam, opts, chunks_per_gpu, rgb_out, rgb8_out, false, stats);
}

int rtx_multi_render_device(rtx_multi* m, const rtx_camera* cam, const rtx_opts* opts, uint32_t chunks_per_gpu, void* d_rgb_out,
                            void* d_rgb8_out, rtx_stats* stats) {
    return multi_render(m, cam, opts, chunks_per_gpu, (float*)d_rgb_out, (uint8_t*)d_rgb8_out, true, stats);
}

#ifndef RTX_BUILD_ID
#define RTX_BUILD_ID "unknown"
#endif
const char* rtx_build_id(void) { return RTX_BUILD_ID; }

uint32_t rtx_struct_size(uint32_t which) {
    switch (which) {
    case 0: return (uint32_t)sizeof(rtx_shape);
    case 1: return (uint32_t)sizeof(rtx_material);
    case 2: return (uint32_t)sizeof(rtx_pattern);
    case 3: return (uint32_t)sizeof(rtx_node);
    case 4: return (uint32_t)sizeof(rtx_scene_desc);
    case 5: return (uint32_t)sizeof(rtx_camera);
    case 6: return (uint32_t)sizeof(rtx_opts);
    case 7: return (uint32_t)sizeof(rtx_stats);
    default: return 0u;
    }
}

int rtx_strip_row_map(const rtx_camera* cam, const rtx_opts* opts, uint32_t* rows_out, uint32_t cap, uint32_t* n_rows) {
    uint32_t rows = 0;
    int rc = rtx_rows_for(cam, opts, &rows);
    if (rc) return rc;
    if (!n_rows) return fail(RTX_E_INVALID, "n_rows is NULL");
    *n_rows = rows;
    if (!rows_out) return RTX_OK;
    if (cap < rows) return fail(RTX_E_INVALID, "row map needs " + std::to_string(rows) + " entries");
    const uint32_t total = rendered_rows(cam, opts);
    const uint32_t count = (opts && opts->strip_count > 1) ? opts->strip_count : 1u;
    const uint32_t strip = (opts && opts->strip_rows) ? opts->strip_rows : 8u;
    const uint32_t index = count > 1 ? opts->strip_index : 0u;
    uint32_t local = 0;
    if (count == 1) {
        for (uint32_t y = 0; y < total; ++y) rows_out[local++] = y;
        return RTX_OK;
    }
    for (uint64_t s = index; s * strip < total; s += count)
        for (uint32_t y = (uint32_t)(s * strip); y < total && y < (uint32_t)((s + 1) * strip); ++y) rows_out[local++] = y;
    return RTX_OK;
}

}  // extern "C"
