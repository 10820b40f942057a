// jit.h -- run-time compilation of scene-specialised kernels (hiprtc) and their launch.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_scene.h"

namespace rmdev {

struct EmbeddedHeader {
    const char *name;
    const char *text;
};
extern const EmbeddedHeader kEmbeddedHeaders[];
extern const unsigned kEmbeddedHeaderCount;

// One loaded code object with the kernels generated for one scene.
struct SceneModule {
    hipModule_t module = nullptr;
    hipModule_t more_modules[2] = { nullptr, nullptr };   // where the source's parts were compiled as programs of their own
    hipFunction_t trace = nullptr;
    hipFunction_t trace_exact = nullptr;    // the pass over the tiles `trace` flagged (null in work-counting modules: their walks are exact)
    hipFunction_t trace_refill = nullptr;   // null unless the source holds it
    hipFunction_t eval_sdf = nullptr;
    uint32_t points_per_lane = 0;   // of `trace`
    double compile_seconds = 0.0;   // 0 when the code came from the process-wide cache
    bool loaded() const { return module != nullptr; }
    void unload();
};

// Compiles `source` (see codegen.cpp) for gfx950 and loads it on the current device.  Code objects
// are cached per process by source text, so uploading the same scene again costs no compile.
// Returns false and fills `log` on failure.
bool compile_scene_module(const std::string &source, uint32_t points_per_lane, SceneModule &out, std::string &log);

// Starts the compile of `source` (its parts) in the background, where the compiler runs in processes of its own: a later
// compile_scene_module of the same source waits for it instead of starting its own.  For callers that know the scene before
// they have a device context (the compile then overlaps the device's initialisation).
void precompile_scene_source(const std::string &source);

// Second-level cache of code objects on disk.  Default: $RM_KERNEL_CACHE_DIR ("" or "off": none), else the first of
// $XDG_CACHE_HOME/raymarch_hip, $HOME/.cache/raymarch_hip, /tmp/raymarch_hip-<uid> that is (or can be made) a directory
// of the calling user's that nobody else can write; set_kernel_cache_dir overrides (null or "": none).  Files are keyed
// by the generated source and the device headers, written atomically, and verified against the full key on load.
void set_kernel_cache_dir(const char *dir);

// Small keyed blobs kept in the same directory (tile launch orders): false / no-op without a cache directory.
bool load_cached_blob(const std::string &kind, const std::string &key, std::vector<uint32_t> &words);
void store_cached_blob(const std::string &kind, const std::string &key, const std::vector<uint32_t> &words);

// How this process last obtained a code object: "rm_jit, a process of its own: <path>", "hiprtc in this process",
// "none: code object from the cache directory", or "none yet".  The library compiles with the program rm_jit beside it
// (jit_helper.cpp) when there is one, so that the kernels come from the ROCm release the library was built with whatever
// other copies of hiprtc the process holds; $RM_JIT_HELPER names another program, "" or "off" compiles in the process.
std::string kernel_compiler_used();

// The compile step alone (no device needed): true if `source` compiles for gfx950.
bool check_scene_source(const std::string &source, std::string &log);

// waves_per_side = 1, 2 or 3: workgroups of that many waves squared, tiles of 8 x that many samples per side.
// LDS: the lattice tables and, when the kernel writes pixels (fr.pixels_*), the tile's sample colours.
hipError_t launch_trace_module(const SceneModule &m, const SceneView &sc, const FrameView &fr, uint32_t waves_per_side, Sample *samples,
                               WorkCounters *counters, hipStream_t stream);
// The exact pass behind launch_trace_module over the same frame: one workgroup of the frame's shape per 64 tiles.  With
// out_f32 or out_rgba8 it also resolves the sample image into those pixels (one workgroup per k x k tiles then), in
// place of launch_resolve: for SSAA kernels that exact_launch_can_resolve accepts.
bool exact_launch_can_resolve(uint32_t kernel);
hipError_t launch_trace_exact_module(const SceneModule &m, const SceneView &sc, const FrameView &fr, uint32_t waves_per_side, Sample *samples,
                                     float *out_f32, unsigned char *out_rgba8, hipStream_t stream);
hipError_t launch_eval_sdf_module(const SceneModule &m, const SceneView &sc, const float *xyz, uint32_t n, float *sd,
                                  uint32_t *prim, float *sd_march, hipStream_t stream);

}  // namespace rmdev
