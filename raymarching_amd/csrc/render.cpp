// render.cpp -- render::setup / push / GPU / destroy on the C ABI.
//
// Where the reference creates a hidden GLFW window, compiles shader.comp at run
// time and allocates a texture plus three SSBOs (render.cpp:408-445), this
// back-end needs a HIP context, and compiles kernels for the loaded scene where the
// reference compiles its shader; buffers are sized on demand.  The three setup
// operators are kept so that a caller written against the reference runs unchanged.
#include "render.h"

#include <iostream>
#include <string>
#include <vector>

namespace constants {
rm_params params = { 1024u, 768u, 3u, 1000, 0.05f, 1e-3f, 1e-3f };
}

namespace render {

void unpin(void);

namespace {

rm_context *g_ctx = nullptr;     // one device (the reference's case), or rank 0's context of g_multi
rm_multi *g_multi = nullptr;     // setup::devices() named more than one rank: the frame is sharded over them
std::vector<int> g_devices = { 0 };
bool g_ok = true;
std::string g_error;
// The caller's image buffer, page-locked the first time GPU() sees it so that the read-back is a copy-engine
// transfer instead of a staged one (the reference's caller allocates the buffer once and reuses it, main.cpp:56).
// The registration is on the caller's memory: the image must stay allocated until render::destroy() or render::unpin()
// (a buffer freed and allocated again at the same address would otherwise be taken for the locked one).
void *g_pinned = nullptr;
size_t g_pinned_bytes = 0;

void *g_refused = nullptr;       // the last buffer that could not be locked: not tried again frame after frame
size_t g_refused_bytes = 0;

void pin(void *image, size_t bytes) {
    if (image == g_pinned && bytes <= g_pinned_bytes) return;   // the same buffer asked for more (float after 8-bit frames) is locked anew
    if (image == g_refused && bytes == g_refused_bytes) return;
    unpin();
    // a buffer that cannot be locked still works, more slowly; one that is page-locked already (rm_host_alloc) is refused
    // here too, and rendered into as fast as a registered one
    if (rm_host_register(image, bytes) == RM_OK) {
        g_pinned = image;
        g_pinned_bytes = bytes;
    } else {
        g_refused = image;
        g_refused_bytes = bytes;
    }
}

void report(int rc, const char *what) {
    g_ok = (rc == RM_OK);
    if (!g_ok) {
        g_error = std::string(what) + ": " + rm_last_error();
        std::cout << "[Error] " << g_error << std::endl;
    }
}

bool have_context(const char *what) {
    if (g_ctx) return true;
    g_ok = false;
    g_error = std::string(what) + ": render::setup::context() has not succeeded";
    std::cout << "[Error] " << g_error << std::endl;
    return false;
}

}  // namespace

void unpin(void) {
    if (g_pinned) (void)rm_host_unregister(g_pinned);
    g_pinned = nullptr;
    g_pinned_bytes = 0;
    g_refused = nullptr;
    g_refused_bytes = 0;
}

void setup::devices(const int *device_ids, unsigned count) {
    g_devices.assign(device_ids, device_ids + (device_ids ? count : 0u));
    if (g_devices.empty()) g_devices.push_back(0);
}

void setup::context(void) {
    if (g_multi) rm_multi_destroy(g_multi);
    else if (g_ctx) rm_destroy(g_ctx);
    g_multi = nullptr;
    g_ctx = nullptr;
    // a scene loaded already (the reference's main.cpp loads it first, main.cpp:22-32): its kernels start compiling now,
    // while the device is initialised
    if (scene::current().tree) {
        const std::vector<rm_node> nodes = scene::current().nodes();
        (void)rm_precompile_scene(nodes.data(), static_cast<uint32_t>(nodes.size()), RM_EVAL_LATTICE);
    }
    if (g_devices.size() == 1) return report(rm_create(g_devices[0], &g_ctx), "Failed to create HIP context");
    report(rm_multi_create(g_devices.data(), (uint32_t)g_devices.size(), RM_MULTI_AUTO, &g_multi), "Failed to create HIP contexts");
    if (g_multi) g_ctx = rm_multi_context(g_multi, 0);
}

// Where the reference compiles and links shader.comp (render.cpp:428-437): generates and
// compiles the kernels of the scene loaded so far (scene::load precedes the setup calls in the
// reference's main.cpp:22-32), so that push() finds them in the library's code cache and costs
// an upload only.  With no scene loaded yet there is nothing to compile; push() then does it.
void setup::shaders(void) {
    if (!have_context("setup::shaders")) return;
    if (!scene::current().tree) return;
    push();
}

// Where the reference creates its texture and SSBOs (render.cpp:439-445): sizes the device buffers of a frame.
void setup::buffers(void) {
    if (!have_context("setup::buffers")) return;
    int rc = g_multi ? rm_multi_set_params(g_multi, &constants::params) : rm_set_params(g_ctx, &constants::params);
    if (rc == RM_OK && !g_multi) rc = rm_reserve(g_ctx);   // the ranks of a sharded frame size their shares at the first frame
    report(rc, "Failed to allocate buffers");
}

void push(void) {
    if (!have_context("push")) return;
    int rc = g_multi ? rm_multi_set_params(g_multi, &constants::params) : rm_set_params(g_ctx, &constants::params);
    if (rc != RM_OK) return report(rc, "Failed to upload constants");
    const scene::Scene &sc = scene::current();
    const std::vector<rm_node> nodes = sc.nodes();
    const std::vector<float> lights = sc.light_positions();
    rc = g_multi ? rm_multi_set_scene(g_multi, nodes.data(), static_cast<uint32_t>(nodes.size()), lights.data(),
                                      static_cast<uint32_t>(lights.size() / 3), sc.camera->transform.m, sc.camera->focal)
                 : rm_set_scene(g_ctx, nodes.data(), static_cast<uint32_t>(nodes.size()), lights.data(),
                                static_cast<uint32_t>(lights.size() / 3), sc.camera->transform.m, sc.camera->focal);
    report(rc, "Failed to upload scene");
}

void GPU(unsigned char *image) {
    if (!have_context("GPU")) return;
    if (image) pin(image, (size_t)constants::params.width * constants::params.height * 4);
    report(g_multi ? rm_multi_render_rgba8(g_multi, image) : rm_render_rgba8(g_ctx, image), "Failed to render");
}

void GPU(float *image) {
    if (!have_context("GPU")) return;
    if (g_multi) {
        g_ok = false;
        g_error = "GPU(float*): a frame sharded over several devices is gathered as RGBA8 only";
        std::cout << "[Error] " << g_error << std::endl;
        return;
    }
    if (image) pin(image, (size_t)constants::params.width * constants::params.height * 4 * sizeof(float));
    report(rm_render_f32(g_ctx, image), "Failed to render");
}

void destroy(void) {
    unpin();
    if (g_multi) rm_multi_destroy(g_multi);
    else if (g_ctx) rm_destroy(g_ctx);
    g_multi = nullptr;
    g_ctx = nullptr;
    g_ok = true;
}

bool ok(void) { return g_ok; }
const char *error(void) { return g_error.c_str(); }
rm_context *context(void) { return g_ctx; }
rm_multi *multi(void) { return g_multi; }

}  // namespace render
