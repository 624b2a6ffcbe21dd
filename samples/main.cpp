// samples/main.cpp — what the reference's sample programs do (samples/<name>.zig: build the
// scene, render(), write <name>.ppm into the working directory), for every sample, as one
// program over the C++ facade (ray-tracer_amd/host/rt.hpp).  Rendering runs on the MI355X;
// there is no CPU fallback, so without a HIP device this exits non-zero with the library's
// message.
//
//     render_sample <name> [width height] [out.ppm] [--gpus N] [--real f32|f64] [--containers v1|book] [--rows-quirk on|off]
//
// The reference hard-codes all of these per sample (samples/teapot.zig:30-31, src/world.zig:20, src/camera.zig:87);
// here they are the switches SURVEY section 5 lists:
//   --gpus N           GPUs the frame is spread over (default: every visible one, like rt_render; rtx_multi_* with two
//                      strip sets per GPU when N > 1)
//   --real f32|f64     f32 (default) is the reference's arithmetic and the parity mode; f64 is the reporting instantiation
//   --containers v1|book   the n1/n2 container walk: the reference as written (default) or the book's (what the
//                      reference's shipped refractive images were rendered with)
//   --rows-quirk on|off    on (default): only 8 * floor(H / 8) rows are rendered, as src/camera.zig:87-88 does
// `teapot` and `composite` read ./teapot.obj like samples/teapot.zig:7 does.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../ray-tracer_amd/host/rt.hpp"

namespace {
int usage(const char* argv0) {
    std::fprintf(stderr, "usage: %s <sample> [width height] [out.ppm] [--gpus N] [--real f32|f64] [--containers v1|book] [--rows-quirk on|off]\n", argv0);
    return 2;
}
}  // namespace

int main(int argc, char** argv) {
    // switches first: what is left are the positional arguments
    std::vector<std::string> pos;
    int gpus = 0;
    rtx_opts opts{};
    opts.rows_quirk = 1;  // src/camera.zig:87-88
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--", 0) != 0) {
            pos.push_back(a);
            continue;
        }
        if (i + 1 >= argc) return usage(argv[0]);
        const std::string v = argv[++i];
        if (a == "--gpus") gpus = std::atoi(v.c_str());
        else if (a == "--real" && (v == "f32" || v == "f64")) opts.real = v == "f64" ? RTX_REAL_F64 : RTX_REAL_F32;
        else if (a == "--containers" && (v == "v1" || v == "book")) opts.containers = v == "book" ? RTX_CONTAINERS_BOOK : RTX_CONTAINERS_V1;
        else if (a == "--rows-quirk" && (v == "on" || v == "off")) opts.rows_quirk = v == "on" ? 1u : 0u;
        else return usage(argv[0]);
    }
    if (pos.empty() || gpus < 0) return usage(argv[0]);
    const std::string name = pos[0];
    const uint32_t width = pos.size() > 2 ? (uint32_t)std::strtoul(pos[1].c_str(), nullptr, 10) : 0u;
    const uint32_t height = pos.size() > 2 ? (uint32_t)std::strtoul(pos[2].c_str(), nullptr, 10) : 0u;
    const std::string out = pos.size() > 3 ? pos[3] : (pos.size() == 2 ? pos[1] : name + ".ppm");

    rt_sample* s = nullptr;
    if (name == "composite") {
        std::ifstream f("teapot.obj", std::ios::binary);
        std::stringstream text;
        text << f.rdbuf();
        const std::string t = text.str();
        if (f) s = rt_sample_create_obj("composite", t.data(), t.size(), width, height);
    } else {
        s = rt_sample_create(name.c_str(), width, height);
    }
    if (!s) {
        std::fprintf(stderr, "unknown sample '%s' (or ./teapot.obj missing / unparsable)\n", name.c_str());
        return 2;
    }
    const rt_camera* cam = rt_sample_camera(s);
    rt::Canvas canvas(cam->horizontal_size, cam->vertical_size);
    // What rt_render does (flatten, upload, render, copy back), spelled out so that the switches reach rtx_opts.
    rt_flat_scene* flat = nullptr;
    int rc = rt_flatten(rt_sample_world(s), &flat);
    rtx_camera fcam;
    rt_flatten_camera(cam, &fcam);
    rtx_stats stats{};
    if (rc == 0) {
        const int visible = rtx_device_count();
        if (gpus == 0 || gpus > visible) gpus = visible;
        if (gpus > 1) {
            rtx_multi* multi = nullptr;
            std::vector<int> devices;
            for (int d = 0; d < gpus; ++d) devices.push_back(d);
            rc = rtx_multi_create(rt_flat_desc(flat), devices.data(), (uint32_t)devices.size(), &multi);
            if (rc == 0) rc = rtx_multi_render(multi, &fcam, &opts, 2, canvas.pixels.data(), nullptr, &stats);
            rtx_multi_destroy(multi);
        } else {
            rtx_scene* scene = nullptr;
            rc = rtx_scene_create(rt_flat_desc(flat), 0, &scene);  // fails with RTX_E_NO_DEVICE when there is no GPU
            if (rc == 0) rc = rtx_render(scene, &fcam, &opts, canvas.pixels.data(), &stats);
            rtx_scene_destroy(scene);
        }
        rt_flat_free(flat);
    }
    rt_sample_destroy(s);
    if (rc != 0) {
        std::fprintf(stderr, "render failed (%d): %s\n", rc, rtx_last_error());
        return 1;
    }
    std::ofstream ppm(out, std::ios::binary);
    canvas.toPPM(ppm);
    std::printf("%s %ux%u -> %s  rays: %llu primary, %llu shadow, %llu secondary  (%d GPU%s, %s, containers %s, rows quirk %s)\n", name.c_str(),
                canvas.width, canvas.height, out.c_str(), (unsigned long long)stats.rays_primary, (unsigned long long)stats.rays_shadow,
                (unsigned long long)stats.rays_secondary, gpus, gpus == 1 ? "" : "s", opts.real == RTX_REAL_F64 ? "f64" : "f32",
                opts.containers == RTX_CONTAINERS_BOOK ? "book" : "v1", opts.rows_quirk ? "on" : "off");
    return ppm ? 0 : 1;
}
