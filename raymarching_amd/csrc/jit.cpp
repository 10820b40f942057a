// jit.cpp -- compiles the kernel source generated for a scene with hiprtc and launches it.
//
// Same compiler flags as the ahead-of-time kernels (Makefile): -ffp-contract=off (one rounding
// per fp32 operation), no SLP packing, correctly rounded sqrt and divide.
#include "jit.h"

#include <hip/hiprtc.h>

#include <chrono>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace rmdev {

namespace {

using CodeObject = std::shared_ptr<const std::vector<char>>;
std::mutex g_cache_lock;
std::unordered_map<std::string, CodeObject> g_cache;   // source text -> gfx950 code object

CodeObject build_code_object(const std::string &source, std::string &log) {
    std::vector<const char *> texts, names;
    for (unsigned i = 0; i < kEmbeddedHeaderCount; i++) {
        texts.push_back(kEmbeddedHeaders[i].text);
        names.push_back(kEmbeddedHeaders[i].name);
    }
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, source.c_str(), "scene_kernels.hip", (int)texts.size(), texts.data(), names.data());
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return nullptr;
    }
    const char *options[] = { "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
                              "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt" };
    r = hiprtcCompileProgram(prog, (int)(sizeof options / sizeof options[0]), options);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string text(n, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &text[0]);
        log = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + text;
        (void)hiprtcDestroyProgram(&prog);
        return nullptr;
    }
    size_t size = 0;
    (void)hiprtcGetCodeSize(prog, &size);
    auto code = std::make_shared<std::vector<char>>(size);
    r = size ? hiprtcGetCode(prog, code->data()) : HIPRTC_ERROR_INTERNAL_ERROR;
    (void)hiprtcDestroyProgram(&prog);
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcGetCode: ") + hiprtcGetErrorString(r);
        return nullptr;
    }
    return code;
}

}  // namespace

void SceneModule::unload() {
    if (module) (void)hipModuleUnload(module);
    *this = SceneModule();
}

bool check_scene_source(const std::string &source, std::string &log) { return build_code_object(source, log) != nullptr; }

bool compile_scene_module(const std::string &source, uint32_t points_per_lane, uint32_t waves_per_group, SceneModule &out,
                          std::string &log) {
    out.unload();
    CodeObject code;
    {
        std::lock_guard<std::mutex> hold(g_cache_lock);
        auto it = g_cache.find(source);
        if (it != g_cache.end()) code = it->second;
    }
    double seconds = 0.0;
    if (!code) {
        const auto t0 = std::chrono::steady_clock::now();
        code = build_code_object(source, log);
        if (!code) return false;
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> hold(g_cache_lock);
        g_cache.emplace(source, code);
    }
    hipError_t e = hipModuleLoadData(&out.module, code->data());
    if (e != hipSuccess) {
        log = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        out.module = nullptr;
        return false;
    }
    // a module holds the plain kernel or its work-counting variant (named apart, so that profiles tell them apart)
    const bool counting = source.find("rm_trace_counting") != std::string::npos;
    const struct { hipFunction_t *fn; const char *name; } symbols[] = { { &out.trace, counting ? "rm_trace_counting" : "rm_trace" },
                                                                         { &out.eval_sdf, "rm_eval_sdf" } };
    for (const auto &sym : symbols) {
        e = hipModuleGetFunction(sym.fn, out.module, sym.name);
        if (e != hipSuccess) {
            log = std::string("hipModuleGetFunction(") + sym.name + "): " + hipGetErrorString(e);
            out.unload();
            return false;
        }
    }
    out.points_per_lane = points_per_lane;
    out.waves_per_group = waves_per_group;
    out.compile_seconds = seconds;
    return true;
}

namespace {

uint32_t lds_bytes_of(const SceneView &sc) { return (sc.lattice_words + sc.max_depth * 2u * 2u * 256u) * 4u; }

}  // namespace

hipError_t launch_trace_module(const SceneModule &m, const SceneView &sc, const FrameView &fr, Float4 *samples,
                               WorkCounters *counters, hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0) return hipSuccess;
    const uint32_t pts = m.points_per_lane;
    SceneView sc_arg = sc;
    FrameView fr_arg = fr;
    void *args[] = { &sc_arg, &fr_arg, &samples, &counters };
    const uint32_t side = m.waves_per_group == 4u ? 16u : 8u;
    return hipModuleLaunchKernel(m.trace, (sw + side * pts - 1u) / (side * pts), (sh + side - 1u) / side, 1, 64u * m.waves_per_group, 1, 1,
                                 lds_bytes_of(sc), stream, args, nullptr);
}

hipError_t launch_eval_sdf_module(const SceneModule &m, const SceneView &sc, const float *xyz, uint32_t n, float *sd,
                                  uint32_t *prim, float *sd_march, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    SceneView sc_arg = sc;
    void *args[] = { &sc_arg, &xyz, &n, &sd, &prim, &sd_march };
    return hipModuleLaunchKernel(m.eval_sdf, (n + 255u) / 256u, 1, 1, 256, 1, 1, lds_bytes_of(sc), stream, args, nullptr);
}

}  // namespace rmdev
