// main.cpp -- command-line driver on the render:: operators, in the shape of the reference's
// main() (/root/reference/source/main.cpp:17-88): load the scene, set up, push, render, report the
// same timing lines, write the image.  What the reference fixes at compile time
// (constants.h:11-27) is a flag here.
//
//   raymarch [--scene scene/objects.txt] [--width 1024] [--height 768] [--ssaa 3] [--iterations 1000]
//            [--frames 1] [--out out_gpu.png] [--raw out_gpu.rgba] [--exhaustive] [--interpreted]
//            [--kernel-cache DIR] [--shade-all] [--gpus N | --devices 0,1,...]
//
// The CPU renders of the reference's main() (render::CPU / render::OMP) are not part of this
// library; their lines are printed as "not built" so that the report keeps its shape.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "render.h"
#include "scene.h"

namespace {

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int usage(const char *argv0) {
    std::cout << "usage: " << argv0
              << " [--scene PATH] [--width W] [--height H] [--ssaa K] [--iterations N] [--frames N]\n"
                 "       [--out FILE.png] [--raw FILE.rgba] [--exhaustive] [--interpreted] [--kernel-cache DIR] [--shade-all]\n"
                 "       [--gpus N | --devices 0,1,...]   (the frame sharded over several GPUs; a device may be named twice)\n";
    return 2;
}

}  // namespace

int main(int argc, char **argv) {
    std::string scene_path = "scene/objects.txt", out_png = "out_gpu.png", out_raw;
    int frames = 1;
    bool exhaustive = false, interpreted = false, shade_all = false;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        const char *v = nullptr;
        if (a == "--exhaustive") exhaustive = true;
        else if (a == "--interpreted") interpreted = true;
        else if (a == "--shade-all") shade_all = true;
        else if ((v = value()) == nullptr) return usage(argv[0]);
        else if (a == "--scene") scene_path = v;
        else if (a == "--width") constants::params.width = (uint32_t)std::atoi(v);
        else if (a == "--height") constants::params.height = (uint32_t)std::atoi(v);
        else if (a == "--ssaa") constants::params.kernel = (uint32_t)std::atoi(v);
        else if (a == "--iterations") constants::params.iterations = std::atoi(v);
        else if (a == "--frames") frames = std::atoi(v);
        else if (a == "--gpus") { devices.clear(); for (int d = 0; d < std::atoi(v); d++) devices.push_back(d); }
        else if (a == "--devices") { devices.clear(); for (const char *c = v; *c;) { devices.push_back((int)std::strtol(c, const_cast<char **>(&c), 10)); if (*c == ',') c++; else if (*c) return usage(argv[0]); } }
        else if (a == "--out") out_png = v;
        else if (a == "--raw") out_raw = v;
        else if (a == "--kernel-cache") rm_set_kernel_cache_dir(v);
        else return usage(argv[0]);
    }
    if (frames < 1) frames = 1;
    const rm_params &p = constants::params;

    std::cout << "...Loading scene" << std::endl;
    scene::load(scene_path.c_str());

    std::cout << "...Setting up context" << std::endl;
    if (!devices.empty()) render::setup::devices(devices.data(), (unsigned)devices.size());
    render::setup::context();
    if (!render::ok()) return 1;
    const uint32_t n_ranks = render::multi() ? rm_multi_ranks(render::multi()) : 1u;
    for (uint32_t r = 0; r < n_ranks; r++) {
        rm_context *ctx = render::multi() ? rm_multi_context(render::multi(), r) : render::context();
        if (exhaustive) rm_set_evaluator(ctx, RM_EVAL_EXHAUSTIVE);
        if (interpreted) rm_set_kernel_mode(ctx, RM_KERNEL_INTERPRETED);
        if (shade_all) rm_set_shading(ctx, RM_SHADE_ALL);
    }

    std::cout << "...Compiling shaders" << std::endl;
    auto t0 = std::chrono::steady_clock::now();
    render::setup::shaders();
    const double compile_s = seconds_since(t0);
    if (!render::ok()) return 1;

    std::cout << "...Generating buffers" << std::endl;
    render::setup::buffers();

    std::cout << "...Rendering" << std::endl;
    std::cout << "Render with CPU (1 thread):\tnot built (the CPU path is the external baseline)" << std::endl;
    std::cout << "Render with OpenMP:\t\tnot built (the CPU path is the external baseline)" << std::endl;

    // The frame's image: page-locked memory from the library where it can be had (the GPU then writes the pixels into it
    // itself; render::GPU page-locks a caller's ordinary buffer on its own, which an unprivileged process may be refused)
    const size_t image_bytes = (size_t)p.width * p.height * 4;
    std::vector<unsigned char> pageable;
    void *locked = nullptr;
    if (rm_host_alloc(image_bytes, &locked) != RM_OK) {
        locked = nullptr;
        pageable.resize(image_bytes);
    }
    unsigned char *const image = locked ? static_cast<unsigned char *>(locked) : pageable.data();

    t0 = std::chrono::steady_clock::now();
    render::push();
    const double push_s = seconds_since(t0);
    if (!render::ok()) return 1;

    // the reference renders one frame and reports its time (main.cpp:68-73): that is the first frame here too, with
    // everything a first frame pays (buffers, natural tile order, tile timing); --frames N > 1 adds the steady rate
    t0 = std::chrono::steady_clock::now();
    render::GPU(image);
    const double first_s = seconds_since(t0);
    if (!render::ok()) return 1;
    double render_s = first_s;
    if (frames > 1) {
        t0 = std::chrono::steady_clock::now();
        for (int f = 1; f < frames; f++) render::GPU(image);
        render_s = seconds_since(t0) / (frames - 1);
        if (!render::ok()) return 1;
    }
    rm_counters c{};
    rm_get_counters(render::context(), &c);
    const double rays = (double)p.width * p.height * p.kernel * p.kernel * (1.0 + (double)scene::current().lights.size());

    std::cout << "Render with GPU:\t\t" << first_s << "s" << std::endl;
    std::cout << "Copy to GPU:\t\t\t" << push_s << "s" << std::endl;
    std::cout << "Render + Copy on GPU:\t\t" << first_s + push_s << "s" << std::endl;
    if (frames > 1) std::cout << "Render with GPU (later frames):\t" << render_s << "s per frame over " << frames - 1 << std::endl;
    std::cout << "Kernel compile:\t\t\t" << compile_s << "s" << std::endl;
    if (render::multi())
        std::cout << "Devices:\t\t\t" << n_ranks << " ranks, "
                  << (rm_multi_transport_in_use(render::multi()) == RM_MULTI_DIRECT ? "rows stored straight into the first device's frame"
                      : rm_multi_transport_in_use(render::multi()) == RM_MULTI_RCCL ? "shares gathered by RCCL send/recv" : "shares gathered by peer copies")
                  << std::endl;
    std::cout << "Trace kernel:\t\t\t" << c.kernel_ms * 1e-3 << "s (" << (c.compiled_kernel ? "compiled for the scene" : "record walker")
              << "), resolve " << c.resolve_ms * 1e-3 << "s" << std::endl;
    std::cout << "Throughput:\t\t\t" << rays / render_s / 1e6 << " Mrays/s (" << (uint64_t)rays << " rays per frame, " << frames
              << " frame" << (frames > 1 ? "s" : "") << ", " << (frames > 1 ? "later frames" : "the first frame") << ", host image included)" << std::endl;

    int status = 0;
    if (!out_png.empty()) {
        if (rm_write_png(out_png.c_str(), image, p.width, p.height) != RM_OK) {
            std::cout << "[Error] " << rm_last_error() << std::endl;
            status = 1;
        }
    }
    if (!out_raw.empty()) {
        if (FILE *f = std::fopen(out_raw.c_str(), "wb")) {
            std::fwrite(image, 1, image_bytes, f);
            std::fclose(f);
        } else {
            std::cout << "[Error] cannot write " << out_raw << std::endl;
            status = 1;
        }
    }

    render::destroy();
    if (locked) (void)rm_host_free(locked);
    return status;
}
