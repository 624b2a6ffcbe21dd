// samples/main.cpp — what the reference's sample programs do (samples/<name>.zig: build the
// scene, render(), write <name>.ppm into the working directory), for every sample, as one
// program over the C++ facade (ray-tracer_amd/host/rt.hpp).  Rendering runs on the MI355X;
// there is no CPU fallback, so without a HIP device this exits non-zero with the library's
// message.
//
//     render_sample <name> [width height] [out.ppm]
//
// `teapot` and `composite` read ./teapot.obj like samples/teapot.zig:7 does.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../ray-tracer_amd/host/rt.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <sample> [width height] [out.ppm]\n", argv[0]);
        return 2;
    }
    const std::string name = argv[1];
    const uint32_t width = argc > 3 ? (uint32_t)std::strtoul(argv[2], nullptr, 10) : 0u;
    const uint32_t height = argc > 3 ? (uint32_t)std::strtoul(argv[3], nullptr, 10) : 0u;
    const std::string out = argc > 4 ? argv[4] : (argc == 3 ? argv[2] : name + ".ppm");

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
    rt_ray_counts counts{};
    const int rc = rt_render(cam, rt_sample_world(s), canvas.pixels.data(), &counts);
    rt_sample_destroy(s);
    if (rc != 0) {
        std::fprintf(stderr, "render failed (%d): %s\n", rc, rtx_last_error());
        return 1;
    }
    std::ofstream ppm(out, std::ios::binary);
    canvas.toPPM(ppm);
    std::printf("%s %ux%u -> %s  rays: %llu primary, %llu shadow, %llu secondary\n", name.c_str(), canvas.width,
                canvas.height, out.c_str(), (unsigned long long)counts.primary, (unsigned long long)counts.shadow,
                (unsigned long long)counts.secondary);
    return ppm ? 0 : 1;
}
