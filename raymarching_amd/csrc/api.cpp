// api.cpp -- the C ABI of include/raymarch_hip.h on top of the kernels.
//
// Takes the place of the reference's OpenGL host code: context/shader/buffer
// setup (render.cpp:408-445), scene + uniform upload (render.cpp:368-405,
// 447-466) and dispatch + read-back (render.cpp:468-475).
#include "../../include/raymarch_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <cstring>
#include <future>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "codegen.h"
#include "jit.h"
#include "kernels.h"
#include "program.h"
#include "scene.h"

// The static scratch of rm_enqueue_device_static (see find_static_scratch below): RM_STATIC_SCRATCH_BYTES for the SSAA sample
// image, then a word per tile (up to kStaticFlagTiles of them) for the tiles that the scene's optimistic kernel asks its exact
// kernel to render again (FrameView::tile_flags; zero when the code object is loaded, and the kernels leave them zero).
constexpr size_t kStaticFlagTiles = 4u << 20;
static __device__ __align__(256) unsigned char scratch[RM_STATIC_SCRATCH_BYTES + kStaticFlagTiles * sizeof(uint32_t)];

int rm_write_png_file(const char *path, const uint8_t *rgba, uint32_t width, uint32_t height);   // png.cpp

namespace {

thread_local std::string g_error;

int fail(rm_status code, const std::string &what) {
    g_error = what;
    return code;
}

}  // namespace

// for the library's other files (multi.cpp): the calling thread's rm_last_error()
int rm_internal_fail(int code, const std::string &what) { return fail(static_cast<rm_status>(code), what); }

namespace {

#define RM_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(RM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

template <typename T>
struct DeviceArray {
    T *ptr = nullptr;
    size_t capacity = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= capacity && ptr) return hipSuccess;
        if (ptr) {
            hipError_t e = hipFree(ptr);
            ptr = nullptr;
            capacity = 0;
            if (e != hipSuccess) return e;
        }
        if (n == 0) n = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
        if (e == hipSuccess) capacity = n;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        capacity = 0;
    }
};

// ---- the static scratch of rm_enqueue_device_static ---------------------------------------------
// Device memory the kernels need beyond the caller's inputs and outputs (the SSAA sample image that the trace
// kernel leaves for the resolve kernel), held in the code object instead of allocated: every GPU this library
// runs on carries its own copy of the array, at its own address, until the process ends.  `scratch` names
// the array on the host side only; the address the kernels get is the one hipGetSymbolAddress reports for
// the current device, looked up once per device in rm_create.
constexpr size_t kStaticScratchBytes = RM_STATIC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;
void *g_scratch_of_device[kMaxDevices] = {};
std::mutex g_scratch_lock;

// with `device` current
hipError_t find_static_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipSuccess;   // such a device renders through allocated buffers only
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_of_device[device]) return hipSuccess;
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e == hipSuccess) g_scratch_of_device[device] = ptr;
    return e;
}

void *static_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    return g_scratch_of_device[device];
}

// Page-locked host words for the tile tables on their way to and from the device: a copy from ordinary memory makes the
// runtime set up staging buffers of its own the first time (milliseconds, inside the first frame).
struct PinnedWords {
    uint32_t *ptr = nullptr;
    size_t capacity = 0;
    hipError_t reserve(size_t n) {
        if (n <= capacity && ptr) return hipSuccess;
        release();
        if (n == 0) n = 1;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(uint32_t), hipHostMallocDefault);
        if (e == hipSuccess) capacity = n;
        else ptr = nullptr;
        return e;
    }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        capacity = 0;
    }
};

}  // namespace

struct rm_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = { nullptr, nullptr, nullptr };

    bool have_params = false, have_scene = false, counting = false;
    int evaluator = RM_EVAL_LATTICE;
    int kernel_mode = RM_KERNEL_COMPILED;
    rmdev::SceneModule module;   // loaded iff the uploaded scene has its own kernels
    rmdev::SceneModule counting_module;   // the same with work counting, compiled at the first counted render
    rm_params params{};
    rmdev::HostProgram program;
    rmdev::SceneView scene{};

    DeviceArray<rmdev::Rec> d_recs;
    DeviceArray<rmdev::Float4> d_pa, d_pb, d_colors, d_lights;
    DeviceArray<rmdev::Sample> d_samples;
    DeviceArray<float> d_frame_f32;
    DeviceArray<unsigned char> d_frame_rgba8;
    DeviceArray<rmdev::WorkCounters> d_counters;
    DeviceArray<float> d_points, d_sd, d_camera;
    DeviceArray<uint32_t> d_tile_flags;   // FrameView::tile_flags: a word per tile, zero between launches
    DeviceArray<uint32_t> d_slot_stats;   // rm_set_slot_stats: four words per workgroup slot of the last render
    bool slot_stats = false;
    size_t slot_stats_slots = 0;
    DeviceArray<unsigned char> d_warm;   // target of the one-pixel launch that warms a freshly loaded kernel
    DeviceArray<uint32_t> d_prim, d_lattice, d_row_prim;
    DeviceArray<uint32_t> d_grid_cells, d_grid_entries;
    DeviceArray<unsigned long long> d_grid_total;
    DeviceArray<float> d_grid_bound;
    double grid_build_ms = 0.0;
    uint64_t grid_entries = 0, grid_cells = 0;

    // Cost-ordered tile launch: the first frame of a configuration (scene, parameters, row bands,
    // kernel shape) measures every tile's duration, later frames start the expensive tiles first.
    struct OrderKey {
        rm_params params;
        rm_rows rows;
        uint64_t scene_serial, band_list_serial;
        uint32_t points_per_lane, shading;
        uint32_t fused;   // the trace kernel writes pixels itself: its tile parts must hold whole pixels
        uint32_t waves_per_side;   // tile shape of the launch
    };
    int tile_scheduling = RM_TILES_BY_COST;
    // rm_set_lane_refill; the environment's RM_REFILL=WH (two digits, e.g. 44) presets it, for the profiling tools
    uint32_t refill_w_log2 = std::getenv("RM_REFILL") ? (uint32_t)(std::atoi(std::getenv("RM_REFILL")) / 10) : 0u;
    uint32_t refill_h_log2 = std::getenv("RM_REFILL") ? (uint32_t)(std::atoi(std::getenv("RM_REFILL")) % 10) : 0u;
    int shading = RM_SHADE_COLOURED;
    uint64_t scene_serial = 0;
    std::string scene_fingerprint;   // digest of the uploaded scene, camera and lights (keys cached tile orders)
    OrderKey order_key{};
    bool order_ready = false;
    bool order_stored = false;       // the order in use is in the cache directory (or came from it)
    std::vector<uint32_t> order_entries;   // the measured order, kept until it has been used again and stored
    bool costs_valid = false;        // d_tile_cost holds the durations a render of this context measured (rm_get_tile_costs)
    uint32_t measured_tiles_x = 0, measured_tiles_y = 0;
    DeviceArray<uint32_t> d_band_list;
    std::vector<uint32_t> band_list_host;
    uint32_t band_list_size = 0;   // 0: bands are first_band + b * band_stride
    bool rows_in_frame = false;    // rm_set_row_placement: device-buffer renders store every row at its place in a whole frame
    bool module_places_rows = false;   // the scene's kernels were compiled with RM_PLACE_ROWS
    uint64_t band_list_serial = 0;   // shape of the tile grid d_tile_cost holds
    DeviceArray<uint32_t> d_tile_order, d_tile_cost;
    PinnedWords h_tiles;   // staging of both
    uint32_t order_extra = 0;        // entries of d_tile_order beyond one per tile (quarters of the longest tiles)

    // the frame enqueued last and not yet booked by finish_bands
    struct Pending {
        bool active = false, measuring = false, by_cost = false, compiled = false, counted = false;
        bool splittable = false;   // the kernel takes tile-part entries (one sample per lane)
        uint32_t part_multiple = 1;   // a part's side in samples must be a multiple of this (whole pixels when the kernel writes pixels)
        uint32_t tile_side = 8;       // samples per tile side
        size_t n_tiles = 0;
    };
    Pending pending;
    hipStream_t own_stream = nullptr;   // created by rm_create; `stream` may be the caller's instead

    rm_counters last{};
};

namespace {

// Launch shape of the compiled kernels.  One sample per lane: with the scene in the instruction
// stream there is no per-record scalar work left to share between two samples, and a second
// sample would double the divergent lattice walks.  Eight waves per SIMD: 64 registers per lane cost the shipped scene
// 76 bytes of spills (none at 80 registers), and still every workload but the smallest gains from the waves that can
// take over while others wait -- ms per frame at 4 / 7 / 8 waves: shipped scene 0.510 / 0.496 / 0.491, 4K 9.03 / 8.77 /
// 8.68, Menger-6 close-up 14.1 / 13.3 / 12.9, 1024 objects 18.1 / 16.9 / 16.2, Menger-2 0.077 / 0.081 / 0.081.
// (Since the camera march is written apart from the light loop the eight-wave build spills one register, and the
// order still holds: shipped scene 0.476 / 0.460 / 0.447 at 6 / 7 / 8 waves.)
constexpr int kCompiledPts = 1, kCompiledWaves = 8;
// how many of the longest tiles are launched in 16 and in 4 parts; $RM_SPLIT_TILES=S,Q overrides the counts for sweeps
// (orders in the cache directory are not keyed by them: sweep with the cache off).  Parts cost waves; what they buy is a
// shorter wait for the frame's longest workgroup, which matters the less the shorter that workgroup runs.  Default
// frame, same GPU, ms per frame in flight / rank 0's eighth share alone: no parts 0.439 / 0.252, (2, 4) 0.440 / 0.207,
// (4, 16) 0.441 / 0.184, (8, 16) 0.442 / 0.184, (16, 32) 0.448 / 0.180, (32, 64) 0.454 / 0.181.
struct SplitCounts { size_t sixteenths = 4, quarters = 16; };
const SplitCounts &split_counts() {
    static const SplitCounts counts = [] {
        SplitCounts c;
        if (const char *env = std::getenv("RM_SPLIT_TILES")) {
            unsigned long s16 = 0, q = 0;
            if (std::sscanf(env, "%lu,%lu", &s16, &q) == 2 && s16 <= 4096 && q <= 4096) { c.sixteenths = s16; c.quarters = q; }
        }
        return c;
    }();
    return counts;
}
#define kSixteenthTiles (split_counts().sixteenths)
#define kQuarterTiles (split_counts().quarters)

// What a measured tile order depends on, as bytes: a digest of the scene as uploaded (records, tables, camera, lights), the
// parameters, the row bands and the launch shape.  Keys the tile orders kept in the kernel cache directory, so that
// the first frame of a configuration some earlier process has rendered starts in cost order without being timed.
std::string order_cache_key(const rm_context *ctx, const rm_context::OrderKey &key, const std::vector<uint32_t> &band_list) {
    std::string k = ctx->scene_fingerprint;
    rm_context::OrderKey plain = key;
    plain.scene_serial = plain.band_list_serial = 0;   // process-local counters
    k.append(reinterpret_cast<const char *>(&plain), sizeof plain);
    if (!band_list.empty()) k.append(reinterpret_cast<const char *>(band_list.data()), band_list.size() * sizeof(uint32_t));
    return k;
}

int check_ready(rm_context *ctx) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (!ctx->have_params) return fail(RM_ERR_STATE, "rm_set_params has not been called");
    if (!ctx->have_scene) return fail(RM_ERR_STATE, "rm_set_scene has not been called");
    return RM_OK;
}

rmdev::FrameView frame_view(const rm_params &p, const rm_rows &rows) {
    rmdev::FrameView f{};
    f.width = p.width; f.height = p.height; f.kernel = p.kernel; f.iterations = p.iterations;
    f.saturation = p.saturation; f.surface_eps = p.surface_eps; f.offset_eps = p.offset_eps;
    f.band_rows = rows.band_rows; f.first_band = rows.first_band;
    f.band_stride = rows.band_stride; f.band_count = rows.band_count;
    return f;
}

// Launch shape of a render.  Record walker: one wave of 8 x 8 samples per workgroup, or 2 x 2 waves with two samples per lane.
// Kernels generated for the scene: n x n waves per workgroup, n = 1 for the finest scheduling (a workgroup lasts as
// long as its slowest wave), 2 where large lattice tables are staged into LDS by every workgroup; they write the
// pixels themselves when a tile of 8n x 8n samples holds whole pixels, otherwise the trace kernel writes samples and
// the resolve kernel adds them up.  n = 3 would hold whole pixels for SSAA kernels 3, 6, 12 and 24 too, but nine-wave
// workgroups fill a CU's wave slots so unevenly that the shipped frame takes 0.82-0.99 ms instead of 0.58 with the
// resolve kernel (measured): that shape is used only where the sample image has nowhere to go (static scratch too small).
struct LaunchShape {
    bool compiled, fused, refill;
    uint32_t n_side, pts;
    size_t n_tiles;
};

LaunchShape launch_shape(const rm_context *ctx, rmdev::FrameView &fr, bool has_output, bool static_samples) {
    LaunchShape s{};
    s.compiled = ctx->module.loaded();
    const bool wide_runs = ctx->program.run_prims >= 64;
    const bool large_tables = ctx->program.lattice.size() > 1024;
    s.n_side = s.compiled ? (large_tables ? 2u : 1u) : (wide_runs ? 1u : 2u);
    if (s.compiled && has_output) {
        const uint32_t k = fr.kernel;
        if ((8u * s.n_side) % k == 0u) {
            s.fused = true;
        } else if (static_samples && 24u % k == 0u && rmdev::sample_texels(fr) * sizeof(rmdev::Sample) > kStaticScratchBytes) {
            s.n_side = 3u;
            s.fused = true;
        }
    }
    s.refill = s.compiled && ctx->refill_w_log2 >= 3u && ctx->refill_h_log2 >= 3u && ctx->module.trace_refill != nullptr &&
               fr.iterations > 0 && !static_samples;
    if (s.refill) {
        s.fused = false;
        s.n_side = 1u;
        fr.refill_w_log2 = ctx->refill_w_log2;
        fr.refill_h_log2 = ctx->refill_h_log2;
    }
    s.pts = s.compiled ? (uint32_t)kCompiledPts : (wide_runs ? 1u : 2u);
    const uint32_t sw = fr.width * fr.kernel, sh = fr.band_count * fr.band_rows * fr.kernel, side = 8u * s.n_side;
    s.n_tiles = (size_t)((sw + side * s.pts - 1u) / (side * s.pts)) * ((sh + side - 1u) / side);
    if (s.refill) s.n_tiles = (size_t)((sw + (1u << fr.refill_w_log2) - 1u) >> fr.refill_w_log2) * ((sh + (1u << fr.refill_h_log2) - 1u) >> fr.refill_h_log2);
    return s;
}

// what the tile order of such a launch depends on
rm_context::OrderKey order_key_of(const rm_context *ctx, const rm_rows &rows, const LaunchShape &shape, const rmdev::FrameView &fr) {
    rm_context::OrderKey key{};
    key.params = ctx->params;
    key.rows = rows;
    key.scene_serial = ctx->scene_serial;
    key.band_list_serial = ctx->band_list_size ? ctx->band_list_serial : 0;
    key.points_per_lane = shape.pts;
    key.shading = (uint32_t)ctx->shading;
    key.fused = shape.fused ? 1u : 0u;
    key.waves_per_side = shape.refill ? 100u + 16u * fr.refill_w_log2 + fr.refill_h_log2 : shape.n_side;
    return key;
}

// Trace + resolve of the given bands into device buffers; synchronises the stream.
// Puts trace + resolve of the given bands on the context's stream.  may_measure: this frame may run in
// natural tile order with every tile timed (its caller will call finish_bands before the next frame).
// A tile order an earlier process measured for exactly this configuration, from the cache directory.
bool load_tile_order(rm_context *ctx, const rm_context::OrderKey &key, size_t n_tiles) {
    std::vector<uint32_t> entries;
    if (!rmdev::load_cached_blob("tile-order:", order_cache_key(ctx, key, ctx->band_list_host), entries)) return false;
    if (entries.size() < n_tiles || entries.size() > n_tiles * 16u) return false;
    // every tile must be covered exactly: whole, or as its 4 quarters, or as its 16 sixteenths (a damaged file must
    // not leave pixels unrendered).  Whole tiles are ticked off in a bitmap that stays in cache (the entries come in
    // cost order, i.e. at random), the few split tiles are counted apart.
    std::vector<uint64_t> whole((n_tiles + 63u) / 64u, 0ull);
    std::vector<std::pair<uint32_t, uint32_t>> parts;   // (tile, part: 1-4 its quarters, 5-20 its sixteenths)
    size_t n_whole = 0;
    for (uint32_t e : entries) {
        const uint32_t tile = e & rmdev::kTileMask, part = e >> rmdev::kTileBits;
        if (tile >= n_tiles || part > 20u) return false;
        if (part == 0u) {
            const uint64_t bit = 1ull << (tile & 63u);
            if (whole[tile >> 6] & bit) return false;
            whole[tile >> 6] |= bit;
            n_whole++;
        } else {
            if (parts.size() >= 4096u) return false;   // far more than any order splits
            parts.emplace_back(tile, part);
        }
    }
    std::sort(parts.begin(), parts.end());
    size_t n_split = 0;
    for (size_t i = 0; i < parts.size();) {
        // sorted: a split tile's entries must read 1, 2, 3, 4 or 5, 6, ..., 20
        const uint32_t tile = parts[i].first, first = parts[i].second, count = first == 1u ? 4u : 16u;
        if ((first != 1u && first != 5u) || i + count > parts.size()) return false;
        size_t j = i;
        for (; j < i + count; j++)
            if (parts[j].first != tile || parts[j].second != first + (uint32_t)(j - i)) return false;
        if ((j < parts.size() && parts[j].first == tile) || (whole[tile >> 6] >> (tile & 63u) & 1ull)) return false;
        n_split++;
        i = j;
    }
    if (n_whole + n_split != n_tiles) return false;
    if (ctx->d_tile_order.reserve(entries.size()) != hipSuccess || ctx->h_tiles.reserve(entries.size()) != hipSuccess) return false;
    std::memcpy(ctx->h_tiles.ptr, entries.data(), entries.size() * sizeof(uint32_t));
    // ordered before the launch that follows on the same stream; the staging words stay untouched until that frame is waited for
    if (hipMemcpyAsync(ctx->d_tile_order.ptr, ctx->h_tiles.ptr, entries.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return false;
    ctx->order_extra = (uint32_t)(entries.size() - n_tiles);
    ctx->order_key = key;
    ctx->order_ready = true;
    ctx->order_stored = true;
    ctx->order_entries.clear();
    ctx->costs_valid = false;   // an order read from the cache comes without the durations behind it
    return true;
}

// The whole frame's tile order, if an earlier process left one for this configuration: read and checked when the
// buffers are reserved or the scene is uploaded, whichever comes last (milliseconds for a 4K frame's half a million
// tiles), so that the first frame starts in cost order at once.
int preload_tile_order(rm_context *ctx) {
    if (!ctx->have_scene || !ctx->have_params || ctx->tile_scheduling != RM_TILES_BY_COST || ctx->counting || ctx->band_list_size) return RM_OK;
    const rm_params &p = ctx->params;
    const rm_rows whole = { p.height, 0, 1, 1 };
    rmdev::FrameView fr = frame_view(p, whole);
    const LaunchShape shape = launch_shape(ctx, fr, true, false);
    if (shape.n_tiles <= 1) return RM_OK;
    const rm_context::OrderKey key = order_key_of(ctx, whole, shape, fr);
    if (!(ctx->order_ready && std::memcmp(&key, &ctx->order_key, sizeof key) == 0) && load_tile_order(ctx, key, shape.n_tiles))
        RM_HIP(hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

// Compiles and loads the kernels of the context's scene (RM_KERNEL_COMPILED; otherwise only unloads the last scene's) and
// launches each of them once.  A context that places rows (rm_set_row_placement) compiles them with RM_PLACE_ROWS.
int load_scene_module(rm_context *ctx) {
    ctx->module.unload();
    ctx->counting_module.unload();
    ctx->module_places_rows = false;
    if (ctx->kernel_mode == RM_KERNEL_COMPILED) {
        std::string log;
        const std::string source = std::string(ctx->rows_in_frame ? "#define RM_PLACE_ROWS 1\n" : "") +
                                   rmdev::generate_scene_source(ctx->program, kCompiledPts, kCompiledWaves, false, ctx->refill_w_log2 != 0u);
        if (!rmdev::compile_scene_module(source, kCompiledPts, ctx->module, log))
            return fail(RM_ERR_HIP, "compiling the scene's kernels failed: " + log);
        ctx->module_places_rows = ctx->rows_in_frame;
    }
    if (ctx->module.loaded()) {
        // The first launch of a freshly loaded kernel costs milliseconds (code upload, queue set-up): one pixel now, where
        // the reference compiles its shader, instead of inside the first frame.
        rmdev::FrameView fr{};
        fr.width = fr.height = fr.kernel = 1u;
        fr.iterations = 1;
        fr.surface_eps = fr.offset_eps = 1e-3f;
        fr.band_rows = fr.band_stride = fr.band_count = 1u;
        // into a buffer of the context's own: the GPU's static scratch may hold another context's frame in flight
        RM_HIP(ctx->d_warm.reserve(1024));   // (the word at 512 counts the tiles the exact kernel renders: rm_get_exact_tiles)
        unsigned char *const spare = ctx->d_warm.ptr;
        RM_HIP(hipMemsetAsync(spare, 0, 1024, ctx->stream));
        fr.tile_flags = reinterpret_cast<uint32_t *>(spare + 384);   // (zero: the exact kernel finds nothing to do, but has been launched once)
        RM_HIP(rmdev::launch_trace_module(ctx->module, ctx->scene, fr, 1u, reinterpret_cast<rmdev::Sample *>(spare), nullptr, ctx->stream));
        RM_HIP(rmdev::launch_trace_exact_module(ctx->module, ctx->scene, fr, 1u, reinterpret_cast<rmdev::Sample *>(spare), nullptr, spare + 256, ctx->stream));
        RM_HIP(rmdev::launch_resolve(fr, ctx->scene.colors, reinterpret_cast<rmdev::Sample *>(spare), nullptr, spare + 256, ctx->stream));
        RM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RM_OK;
}

// The words through which a scene's optimistic kernel asks for tiles to be rendered again (FrameView::tile_flags): at least
// `tiles` of them, zero.  The kernels leave them zero, so they are cleared only when the buffer is new.
int ensure_tile_flags(rm_context *ctx, size_t tiles) {
    if (ctx->d_tile_flags.ptr && ctx->d_tile_flags.capacity >= tiles) return RM_OK;
    RM_HIP(hipStreamSynchronize(ctx->stream));   // no launch may still be using the old words
    RM_HIP(ctx->d_tile_flags.reserve(tiles));
    RM_HIP(hipMemsetAsync(ctx->d_tile_flags.ptr, 0, ctx->d_tile_flags.capacity * sizeof(uint32_t), ctx->stream));
    return RM_OK;
}

// static_samples: the device's static scratch (rm_enqueue_device_static) -- the call then allocates nothing: the sample
// image goes there, tiles are never timed, and a configuration that would need a buffer the context does not hold is refused.
// in_frame: the outputs are whole frames and every row goes to its own place in them (FrameView::rows_in_frame).
int enqueue_bands(rm_context *ctx, const rm_rows &rows, float *d_f32, unsigned char *d_rgba8, bool may_measure,
                  rmdev::Sample *static_samples = nullptr, bool in_frame = false) {
    if (rows.band_rows == 0 || rows.band_count == 0 || rows.band_stride == 0)
        return fail(RM_ERR_INVALID, "rm_rows: band_rows, band_stride and band_count must be non-zero");
    RM_HIP(hipSetDevice(ctx->device));
    rmdev::FrameView fr = frame_view(ctx->params, rows);
    if (ctx->band_list_size) {
        if (rows.band_count != ctx->band_list_size)
            return fail(RM_ERR_INVALID, "rm_rows: band_count must equal the length of the band list set with rm_set_band_list");
        fr.band_list = ctx->d_band_list.ptr;
    }
    fr.skip_unlit = ctx->shading == RM_SHADE_COLOURED ? 1u : 0u;
    fr.rows_in_frame = in_frame ? 1u : 0u;
    if (in_frame && ctx->module.loaded() && !ctx->module_places_rows) return fail(RM_ERR_STATE, "the scene's kernels were compiled without row placement");
    const LaunchShape shape = launch_shape(ctx, fr, d_f32 || d_rgba8, static_samples != nullptr);
    const bool compiled = shape.compiled, fused = shape.fused, refill = shape.refill;
    const bool wide_runs = ctx->program.run_prims >= 64;
    const uint32_t n_side = shape.n_side;
    if (fused) {
        fr.pixels_f32 = d_f32;
        fr.pixels_rgba8 = d_rgba8;
    } else if (static_samples) {
        const size_t need = rmdev::sample_texels(fr) * sizeof(rmdev::Sample);
        if (need > kStaticScratchBytes)
            return fail(RM_ERR_INVALID, "rm_enqueue_device_static: this render needs " + std::to_string(need) + " bytes of scratch for its " +
                                            std::to_string(rmdev::sample_texels(fr)) + " SSAA samples, the static buffer holds " +
                                            std::to_string(kStaticScratchBytes) + " (render fewer rows per call, or use rm_enqueue_device)");
    } else {
        RM_HIP(ctx->d_samples.reserve(rmdev::sample_texels(fr)));
    }
    rmdev::Sample *const samples = static_samples ? static_samples : ctx->d_samples.ptr;
    rm_context::Pending &pend = ctx->pending;
    pend = rm_context::Pending{};
    pend.compiled = compiled;

    // ---- tile schedule of this launch
    const uint32_t pts = shape.pts;
    const uint32_t sw = fr.width * fr.kernel, sh = fr.band_count * fr.band_rows * fr.kernel;
    const uint32_t side = 8u * n_side;
    pend.n_tiles = shape.n_tiles;
    if (ctx->tile_scheduling == RM_TILES_BY_COST && !ctx->counting && pend.n_tiles > 1) {
        const rm_context::OrderKey key = order_key_of(ctx, rows, shape, fr);
        if (ctx->order_ready && std::memcmp(&key, &ctx->order_key, sizeof key) == 0) {
            fr.tile_order = ctx->d_tile_order.ptr;
            fr.extra_slots = ctx->order_extra;
            pend.by_cost = true;
        } else if (may_measure && load_tile_order(ctx, key, pend.n_tiles)) {
            fr.tile_order = ctx->d_tile_order.ptr;
            fr.extra_slots = ctx->order_extra;
            pend.by_cost = true;
        } else if (may_measure) {
            ctx->order_key = key;
            ctx->order_ready = false;
            ctx->costs_valid = false;
            pend.measuring = true;
            ctx->measured_tiles_x = refill ? (sw + (1u << fr.refill_w_log2) - 1u) >> fr.refill_w_log2 : (sw + side * pts - 1u) / (side * pts);
            ctx->measured_tiles_y = refill ? (sh + (1u << fr.refill_h_log2) - 1u) >> fr.refill_h_log2 : (sh + side - 1u) / side;
            RM_HIP(ctx->d_tile_cost.reserve(pend.n_tiles));
            pend.splittable = compiled && pts == 1u && !refill;
            pend.part_multiple = fused ? fr.kernel : 1u;
            pend.tile_side = side;
            RM_HIP(ctx->d_tile_order.reserve(pend.n_tiles + 15u * kSixteenthTiles + 3u * kQuarterTiles));
            fr.tile_cost = ctx->d_tile_cost.ptr;
        }
    }
    ctx->slot_stats_slots = 0;
    if (ctx->slot_stats && !static_samples && !refill) {
        const size_t slots = pend.n_tiles + (fr.tile_order ? fr.extra_slots : 0u);
        RM_HIP(ctx->d_slot_stats.reserve(4u * slots));
        RM_HIP(hipMemsetAsync(ctx->d_slot_stats.ptr, 0, 4u * slots * sizeof(uint32_t), ctx->stream));
        fr.slot_stats = ctx->d_slot_stats.ptr;
        ctx->slot_stats_slots = slots;
    }
    rmdev::WorkCounters *d_cnt = nullptr;
    if (ctx->counting && static_samples) return fail(RM_ERR_STATE, "rm_enqueue_device_static: work counting needs allocated buffers (rm_set_counting(ctx, 0) first)");
    if (ctx->counting) {
        RM_HIP(ctx->d_counters.reserve(1));
        d_cnt = ctx->d_counters.ptr;
        RM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(rmdev::WorkCounters), ctx->stream));
        pend.counted = true;
    }
    if (compiled && ctx->counting && !ctx->counting_module.loaded()) {
        std::string log;
        if (!rmdev::compile_scene_module(std::string(ctx->rows_in_frame ? "#define RM_PLACE_ROWS 1\n" : "") +
                                             rmdev::generate_scene_source(ctx->program, kCompiledPts, kCompiledWaves, true, ctx->refill_w_log2 != 0u), kCompiledPts,
                                         ctx->counting_module, log))
            return fail(RM_ERR_HIP, "compiling the scene's counting kernel failed: " + log);
    }
    // the scene's own kernel walks optimistically; its exact kernel follows it over the tiles it flags (rarely any)
    const bool optimistic = compiled && !ctx->counting && !refill && ctx->module.trace_exact != nullptr;
    if (optimistic && static_samples) {
        // the call that allocates nothing: the words live behind the sample image in the static scratch
        if (pend.n_tiles > kStaticFlagTiles)
            return fail(RM_ERR_INVALID, "rm_enqueue_device_static: this render has " + std::to_string(pend.n_tiles) + " tiles, the static buffer holds the flags of " +
                                            std::to_string(kStaticFlagTiles) + " (render fewer rows per call, or use rm_enqueue_device)");
        fr.tile_flags = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(static_scratch(ctx->device)) + kStaticScratchBytes);
    } else if (optimistic) {
        if (int rc = ensure_tile_flags(ctx, pend.n_tiles)) return rc;
        fr.tile_flags = ctx->d_tile_flags.ptr;
    }
    if (optimistic && ctx->d_warm.ptr) fr.exact_count = reinterpret_cast<uint32_t *>(ctx->d_warm.ptr + 512);
    RM_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    if (compiled) RM_HIP(rmdev::launch_trace_module(ctx->counting ? ctx->counting_module : ctx->module, ctx->scene, fr, n_side, samples, d_cnt, ctx->stream));
    else RM_HIP(rmdev::launch_trace(ctx->scene, fr, wide_runs, samples, d_cnt, ctx->stream));
    // behind it: the scene's exact kernel over the tiles the optimistic one marked, and the resolve of the sample image into
    // pixels where the trace kernel has not written them itself -- in one launch where the exact kernel can do both
    // (Its workgroups resolve blocks of 8 x 8 or 16 x 16 pixels, which reads the samples in shorter runs than the resolve
    // kernel's rows of 64 pixels: for the whole default frame the one launch is 6 us slower than the two -- 0.4905 against
    // 0.4848 ms a frame alone, 0.4513 against 0.4482 in flight -- for an eighth of it 7 % faster in flight, 0.0676 against
    // 0.0725 ms.  So: up to two million samples.)
    const size_t launch_samples = (size_t)fr.width * fr.kernel * fr.band_count * fr.band_rows * fr.kernel;
    const bool exact_resolves = optimistic && !fused && rmdev::exact_launch_can_resolve(fr.kernel) && launch_samples <= (1u << 21);
    if (optimistic && !exact_resolves) RM_HIP(rmdev::launch_trace_exact_module(ctx->module, ctx->scene, fr, n_side, samples, nullptr, nullptr, ctx->stream));
    RM_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    if (exact_resolves) RM_HIP(rmdev::launch_trace_exact_module(ctx->module, ctx->scene, fr, n_side, samples, d_f32, d_rgba8, ctx->stream));
    else if (!fused) RM_HIP(rmdev::launch_resolve(fr, ctx->scene.colors, samples, d_f32, d_rgba8, ctx->stream));
    RM_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    pend.active = true;
    return RM_OK;
}

// Waits for the stream and books the last enqueued frame: kernel times, tile order, work counters.
int finish_bands(rm_context *ctx) {
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    const rm_context::Pending pend = ctx->pending;
    ctx->pending = rm_context::Pending{};
    if (!pend.active) return RM_OK;
    float ms_trace = 0.0f, ms_resolve = 0.0f;
    RM_HIP(hipEventElapsedTime(&ms_trace, ctx->ev[0], ctx->ev[1]));
    RM_HIP(hipEventElapsedTime(&ms_resolve, ctx->ev[1], ctx->ev[2]));
    if (pend.measuring) {
        std::vector<uint32_t> order(pend.n_tiles);
        RM_HIP(ctx->h_tiles.reserve(pend.n_tiles + 15u * kSixteenthTiles + 3u * kQuarterTiles));
        RM_HIP(hipMemcpyAsync(ctx->h_tiles.ptr, ctx->d_tile_cost.ptr, pend.n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        RM_HIP(hipStreamSynchronize(ctx->stream));
        const std::vector<uint32_t> cost(ctx->h_tiles.ptr, ctx->h_tiles.ptr + pend.n_tiles);
        // by falling cost, ties in tile order: three stable counting passes over 11-bit digits of the inverted cost
        // (a comparison sort of the shipped frame's 110 592 tiles took 6 ms, ten frames' worth)
        {
            std::vector<uint32_t> from(pend.n_tiles), count(2048);
            std::iota(order.begin(), order.end(), 0u);
            for (int pass = 0; pass < 3; pass++) {
                const int shift = 11 * pass;
                std::fill(count.begin(), count.end(), 0u);
                for (uint32_t t : order) count[(~cost[t] >> shift) & 2047u]++;
                uint32_t at = 0;
                for (uint32_t &c : count) { const uint32_t n = c; c = at; at += n; }
                from.swap(order);
                for (uint32_t t : from) order[count[(~cost[t] >> shift) & 2047u]++] = t;
            }
        }
        // The frame cannot end before its longest tile does.  The tiles that ran longest -- a few dozen, where a few rays
        // take hundreds of expensive steps -- are listed as 16 or 4 workgroups, each of which waits only for its own
        // samples: measured on the shipped frame, 0.85 -> 0.62 ms alone on the GPU.  By rank, because the measured
        // durations of the longest tiles vary by 10 % from run to run; splitting more tiles (64 into sixteenths, or 64
        // more into quarters) costs more in extra waves than the shorter tail pays for.  A frame whose tiles all take
        // about as long has no tail to shorten: only tiles far above the median, and not far below the longest.
        std::vector<uint32_t> entries;
        entries.reserve(order.size() + 15u * kSixteenthTiles + 3u * kQuarterTiles);
        const double longest = order.empty() ? 0.0 : (double)cost[order[0]];
        const uint32_t median = order.empty() ? 0u : cost[order[order.size() / 2]];
        size_t rank = 0;
        for (uint32_t t : order) {
            const bool may = pend.splittable && rank < kSixteenthTiles + kQuarterTiles && cost[t] > 8u * median &&
                             (double)cost[t] >= 0.3 * longest && t <= rmdev::kTileMask;
            if (may && rank < kSixteenthTiles && (pend.tile_side / 4u) % pend.part_multiple == 0u) {
                for (uint32_t q = 0; q < 16u; q++) entries.push_back(t | ((q + 5u) << rmdev::kTileBits));
            } else if (may && (pend.tile_side / 2u) % pend.part_multiple == 0u) {
                for (uint32_t q = 0; q < 4u; q++) entries.push_back(t | ((q + 1u) << rmdev::kTileBits));
            } else {
                entries.push_back(t);
            }
            rank++;
        }
        std::memcpy(ctx->h_tiles.ptr, entries.data(), entries.size() * sizeof(uint32_t));
        RM_HIP(hipMemcpyAsync(ctx->d_tile_order.ptr, ctx->h_tiles.ptr, entries.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        RM_HIP(hipStreamSynchronize(ctx->stream));
        ctx->order_extra = (uint32_t)(entries.size() - order.size());
        ctx->order_ready = true;
        ctx->costs_valid = true;
        // Kept in memory for now: the order goes to the cache directory once a later frame has used it (below), so that a
        // caller who moves the camera or the scene every frame leaves no file per frame behind.
        ctx->order_stored = false;
        ctx->order_entries = std::move(entries);
    } else if (pend.by_cost && !ctx->order_stored && !ctx->order_entries.empty()) {
        // second frame of the configuration: worth keeping for later processes (the frame is finished; the write is
        // the wait's, not the render's)
        rmdev::store_cached_blob("tile-order:", order_cache_key(ctx, ctx->order_key, ctx->band_list_host), ctx->order_entries);
        ctx->order_stored = true;
        ctx->order_entries.clear();
        ctx->order_entries.shrink_to_fit();
    }
    ctx->last = rm_counters{};
    ctx->last.tiles_by_cost = pend.by_cost ? 1u : 0u;
    ctx->last.tile_parts = pend.by_cost ? ctx->order_extra : 0u;
    ctx->last.kernel_ms = ms_trace;
    ctx->last.resolve_ms = ms_resolve;
    ctx->last.prims = ctx->program.n_prims;
    ctx->last.lattice_prims = ctx->program.lattice_prims;
    ctx->last.flops_per_sdf = ctx->program.flops_per_sdf;
    ctx->last.compile_ms = ctx->module.compile_seconds * 1e3;
    ctx->last.compiled_kernel = pend.compiled ? 1u : 0u;
    if (pend.counted) {
        rmdev::WorkCounters h{};
        RM_HIP(hipMemcpy(&h, ctx->d_counters.ptr, sizeof h, hipMemcpyDeviceToHost));
        ctx->last.sdf_calls = h.sdf_calls;
        ctx->last.march_steps = h.march_steps;
        ctx->last.rays = h.rays;
        ctx->last.prim_evals = h.prim_evals;
        ctx->last.march_slots = h.march_slots;
        ctx->last.flops_executed = h.flops;
        ctx->last.flops_shared_first = h.flops_shared_first;
        for (int i = 0; i < 16; i++) ctx->last.probe[i] = h.probe[i];
    }
    return RM_OK;
}

// Trace + resolve of the given bands into device buffers; synchronises the stream.
int render_bands(rm_context *ctx, const rm_rows &rows, float *d_f32, unsigned char *d_rgba8, bool in_frame = false) {
    if (int rc = enqueue_bands(ctx, rows, d_f32, d_rgba8, true, nullptr, in_frame)) return rc;
    return finish_bands(ctx);
}

// most list entries a cell grid may hold; $RM_GRID_MAX_ENTRIES lowers it (tests of the fallback)
unsigned long long grid_entry_limit() {
    if (const char *env = std::getenv("RM_GRID_MAX_ENTRIES")) return std::strtoull(env, nullptr, 10);
    return 1ull << 28;
}

// Builds the cell grid over the root's long runs (device_scene.h: GridView) on the device; leaves ctx->scene.grid empty
// if the scene has no such runs.  The box: around the runs' primitives, a fifth of its longest side wider on every
// side (camera and shadow rays start and end near the scene); about 8192 cells per table row, sixteen million at most.
int build_grid(rm_context *ctx) {
    rmdev::GridView &g = ctx->scene.grid;
    g = rmdev::GridView{};
    ctx->grid_build_ms = 0.0;
    ctx->grid_entries = ctx->grid_cells = 0;
    const rmdev::HostProgram &hp = ctx->program;
    if (hp.grid_runs.empty() || ctx->evaluator != RM_EVAL_LATTICE || ctx->kernel_mode != RM_KERNEL_COMPILED) return RM_OK;
    rmdev::GridBuild b{};
    std::vector<rmdev::GridRun> runs = hp.grid_runs;
    std::stable_sort(runs.begin(), runs.end(), [](const rmdev::GridRun &x, const rmdev::GridRun &y) { return x.cls < y.cls; });
    size_t n_rows = 0;
    for (size_t i = 0; i < runs.size(); i++) {
        b.runs[i] = runs[i];
        n_rows += runs[i].count;
        if ((uint64_t)runs[i].first + runs[i].count > (1u << 14)) return RM_OK;   // a list entry names its table row in 14 bits: larger tables take the hull walk
    }
    b.n_runs = (uint32_t)runs.size();
    double side[3], longest = 0.0;
    for (int k = 0; k < 3; k++) {
        side[k] = (double)hp.grid_hi[k] - (double)hp.grid_lo[k];
        if (!(side[k] >= 0.0) || !std::isfinite(side[k])) return RM_OK;   // no box to speak of: the hull walk serves
        longest = std::max(longest, side[k]);
    }
    if (!(longest > 0.0)) return RM_OK;
    // How far the box reaches beyond the primitives, as a share of its longest side: points outside it take the hull walk
    // (dear), cells far from every primitive list many rows (dear too).  1024-object scene / its wide variant, ms per
    // frame: 0.06 43.5 / 7.0, 0.125 22.7 / 6.3, 0.2 20.6 / 6.4, 0.3 23.7 / 6.8, 0.6 34.2 / 9.1.  $RM_GRID_PAD for sweeps.
    double pad_share = 0.2;
    if (const char *env = std::getenv("RM_GRID_PAD")) pad_share = std::max(0.0, std::atof(env));
    const double pad = pad_share * longest;
    double volume = 1.0;
    for (int k = 0; k < 3; k++) {
        side[k] += 2.0 * pad;
        volume *= side[k];
    }
    // cells per table row: finer cells list fewer rows each, and their headers (8 bytes a cell) miss the caches more often.
    // Measured on the 1024-object scene / its wide variant with the box above, ms per frame: 2048 cells per row 10.27 / 2.95,
    // 4096 10.06 / 2.84, 8192 9.80 / 2.79, 16384 9.59 / 2.78 (16.8 M cells: 134 MB of headers, four times the build);
    // $RM_GRID_CELLS_PER_ROW overrides it for such sweeps
    double per_row = 8192.0;
    if (const char *env = std::getenv("RM_GRID_CELLS_PER_ROW")) per_row = std::max(1.0, std::atof(env));
    const double target = std::min(double(1u << 24), std::max(4096.0, per_row * (double)n_rows));
    const double cell = std::cbrt(volume / target);
    if (!(cell > 0.0) || !std::isfinite(cell)) return RM_OK;
    uint64_t n_cells = 1;
    for (int k = 0; k < 3; k++) {
        b.origin[k] = (float)((double)hp.grid_lo[k] - pad);
        b.dims[k] = (uint32_t)std::min(1024.0, std::max(1.0, std::ceil(side[k] / cell)));
        n_cells *= b.dims[k];
    }
    b.cell = (float)cell;
    const auto t0 = std::chrono::steady_clock::now();
    RM_HIP(ctx->d_grid_cells.reserve(n_cells * 2));
    RM_HIP(ctx->d_grid_bound.reserve(n_cells));
    RM_HIP(ctx->d_grid_total.reserve(1));
    b.pa = ctx->d_pa.ptr;
    b.pb = ctx->d_pb.ptr;
    b.cells = ctx->d_grid_cells.ptr;
    b.bound = ctx->d_grid_bound.ptr;
    b.total = ctx->d_grid_total.ptr;
    RM_HIP(hipMemsetAsync(b.total, 0, sizeof(unsigned long long), ctx->stream));
    RM_HIP(rmdev::launch_count_grid(b, ctx->stream));
    unsigned long long total = 0;   // 64 bits: 2^24 cells of a thousand coincident rows each would wrap a 32-bit sum
    RM_HIP(hipMemcpyAsync(&total, b.total, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    // a gigabyte of lists: leave such a scene to the hull walk (the cells' 32-bit offsets are only used below this)
    if (total > grid_entry_limit()) return RM_OK;
    RM_HIP(ctx->d_grid_entries.reserve(total + 2u));   // one word per entry
    b.entries = ctx->d_grid_entries.ptr;
    RM_HIP(rmdev::launch_fill_grid(b, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    g.cells = b.cells;
    g.entries = b.entries;
    for (int k = 0; k < 3; k++) {
        g.origin[k] = b.origin[k];
        g.dims[k] = b.dims[k];
    }
    g.cell = b.cell;
    g.inv_cell = 1.0f / b.cell;
    ctx->grid_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->grid_entries = total;
    ctx->grid_cells = n_cells;
    return RM_OK;
}

int eval_points(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim, float *host_sd_march);

}  // namespace

extern "C" {

const char *rm_last_error(void) { return g_error.c_str(); }

int rm_create(int device, rm_context **out) {
    if (!out) return fail(RM_ERR_INVALID, "rm_create: null output pointer");
    *out = nullptr;
    int n_devices = 0;
    RM_HIP(hipGetDeviceCount(&n_devices));
    if (device < 0 || device >= n_devices)
        return fail(RM_ERR_INVALID, "rm_create: no HIP device with ordinal " + std::to_string(device) + " (" +
                                        std::to_string(n_devices) + " visible)");
    RM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RM_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RM_ERR_HIP, std::string("rm_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    rm_context *ctx = new (std::nothrow) rm_context();
    if (!ctx) return fail(RM_ERR_INVALID, "rm_create: out of host memory");
    ctx->device = device;
    hipError_t e = find_static_scratch(device);
    if (e == hipSuccess) e = rmdev::preload_kernels();
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        rm_destroy(ctx);
        return fail(RM_ERR_HIP, std::string("rm_create: ") + hipGetErrorString(e));
    }
    *out = ctx;
    return RM_OK;
}

int rm_destroy(rm_context *ctx) {
    if (!ctx) return RM_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->d_recs.release(); ctx->d_pa.release(); ctx->d_pb.release(); ctx->d_colors.release();
    ctx->d_lights.release(); ctx->d_samples.release(); ctx->d_frame_f32.release();
    ctx->d_frame_rgba8.release(); ctx->d_counters.release(); ctx->d_points.release();
    ctx->module.unload();
    ctx->counting_module.unload();
    ctx->d_tile_order.release(); ctx->d_tile_cost.release(); ctx->d_band_list.release(); ctx->h_tiles.release();
    ctx->d_grid_cells.release(); ctx->d_grid_entries.release(); ctx->d_grid_total.release(); ctx->d_grid_bound.release();
    ctx->d_warm.release();
    ctx->d_slot_stats.release();
    ctx->d_tile_flags.release();
    ctx->d_sd.release(); ctx->d_prim.release(); ctx->d_lattice.release(); ctx->d_row_prim.release(); ctx->d_camera.release();
    for (int i = 0; i < 3; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RM_OK;
}

int rm_set_params(rm_context *ctx, const rm_params *params) {
    if (!ctx || !params) return fail(RM_ERR_INVALID, "rm_set_params: null argument");
    if (params->width == 0 || params->height == 0 || params->kernel == 0)
        return fail(RM_ERR_INVALID, "rm_set_params: width, height and kernel must be positive");
    if ((uint64_t)params->width * params->kernel > 0x7fffffffull || (uint64_t)params->height * params->kernel > 0x7fffffffull)
        return fail(RM_ERR_INVALID, "rm_set_params: image too large");
    ctx->params = *params;
    ctx->have_params = true;
    return RM_OK;
}

int rm_reserve(rm_context *ctx) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (!ctx->have_params) return fail(RM_ERR_STATE, "rm_set_params has not been called");
    RM_HIP(hipSetDevice(ctx->device));
    ctx->order_ready = false;   // the tile tables are cleared below: an order they held is read again, or measured again
    ctx->costs_valid = false;
    const rm_params &p = ctx->params;
    const size_t pixels = (size_t)p.width * p.height;
    RM_HIP(ctx->d_frame_rgba8.reserve(pixels * 4));
    RM_HIP(ctx->d_frame_f32.reserve(pixels * 4));
    const size_t samples = pixels * p.kernel * p.kernel;
    // the sample image, unless the scene's own kernel will write the pixels (not known before the scene is)
    const bool writes_pixels = ctx->have_scene && ctx->module.loaded() && (8u * (ctx->program.lattice.size() > 1024 ? 2u : 1u)) % p.kernel == 0u;
    if (!writes_pixels) RM_HIP(ctx->d_samples.reserve(samples));
    const size_t tiles = ((size_t)p.width * p.kernel + 7u) / 8u * (((size_t)p.height * p.kernel + 7u) / 8u);
    RM_HIP(ctx->d_tile_cost.reserve(tiles));
    if (int rc = ensure_tile_flags(ctx, tiles)) return rc;
    RM_HIP(ctx->d_tile_order.reserve(tiles + 15u * kSixteenthTiles + 3u * kQuarterTiles));
    RM_HIP(ctx->h_tiles.reserve(tiles + 15u * kSixteenthTiles + 3u * kQuarterTiles));
    // first use by the GPU is what makes fresh device memory resident (milliseconds for a frame's worth): now, not in the first frame
    RM_HIP(hipMemsetAsync(ctx->d_frame_rgba8.ptr, 0, pixels * 4, ctx->stream));
    RM_HIP(hipMemsetAsync(ctx->d_frame_f32.ptr, 0, pixels * 4 * sizeof(float), ctx->stream));
    if (!writes_pixels) RM_HIP(hipMemsetAsync(ctx->d_samples.ptr, 0, samples * sizeof(rmdev::Sample), ctx->stream));
    RM_HIP(hipMemsetAsync(ctx->d_tile_cost.ptr, 0, tiles * sizeof(uint32_t), ctx->stream));
    // ... and the first copy between page-locked host memory and the device sets up the copy path (6 ms measured).  Out of
    // the device first, directly behind the fills: a copy queued behind another copy rides on that one's engine, and the
    // first frame's copy of the tile costs -- behind kernels -- would then pay for the outward engine's set-up itself (ROCm
    // 7.2's runtime; the one bundled with PyTorch 2.10 does not care).
    RM_HIP(hipMemcpyAsync(ctx->h_tiles.ptr, ctx->d_tile_cost.ptr, tiles * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    std::memset(ctx->h_tiles.ptr, 0, tiles * sizeof(uint32_t));
    RM_HIP(hipMemcpyAsync(ctx->d_tile_order.ptr, ctx->h_tiles.ptr, tiles * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    return preload_tile_order(ctx);
}

int rm_set_scene(rm_context *ctx, const rm_node *nodes, uint32_t n_nodes, const float *lights_xyz, uint32_t n_lights,
                 const float transform[16], float focal) {
    if (!ctx || !transform) return fail(RM_ERR_INVALID, "rm_set_scene: null argument");
    if (n_lights && !lights_xyz) return fail(RM_ERR_INVALID, "rm_set_scene: lights pointer is null");
    std::string why;
    rmdev::HostProgram prog;
    if (!rmdev::compile_scene(nodes, n_nodes, ctx->evaluator == RM_EVAL_LATTICE, prog, why)) return fail(RM_ERR_INVALID, "rm_set_scene: " + why);
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(ctx->d_recs.reserve(prog.recs.size()));
    RM_HIP(ctx->d_pa.reserve(prog.pa.size()));
    RM_HIP(ctx->d_pb.reserve(prog.pb.size()));
    RM_HIP(ctx->d_colors.reserve(prog.colors.size()));
    RM_HIP(ctx->d_lights.reserve(n_lights));
    RM_HIP(ctx->d_lattice.reserve(prog.lattice.size()));
    RM_HIP(ctx->d_row_prim.reserve(prog.row_prim.size()));
    RM_HIP(ctx->d_camera.reserve(17));
    float camera[17];
    std::memcpy(camera, transform, 16 * sizeof(float));
    camera[16] = focal;
    std::vector<rmdev::Float4> lights(n_lights ? n_lights : 1, rmdev::Float4{ 0, 0, 0, 0 });
    for (uint32_t i = 0; i < n_lights; i++)
        lights[i] = rmdev::Float4{ lights_xyz[3 * i], lights_xyz[3 * i + 1], lights_xyz[3 * i + 2], 0.0f };
    RM_HIP(hipMemcpyAsync(ctx->d_recs.ptr, prog.recs.data(), prog.recs.size() * sizeof(rmdev::Rec), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_pa.ptr, prog.pa.data(), prog.pa.size() * sizeof(rmdev::Float4), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_pb.ptr, prog.pb.data(), prog.pb.size() * sizeof(rmdev::Float4), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_colors.ptr, prog.colors.data(), prog.colors.size() * sizeof(rmdev::Float4), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_lights.ptr, lights.data(), lights.size() * sizeof(rmdev::Float4), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_camera.ptr, camera, sizeof camera, hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_row_prim.ptr, prog.row_prim.data(), prog.row_prim.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipMemcpyAsync(ctx->d_lattice.ptr, prog.lattice.data(), prog.lattice.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this call
    ctx->program = std::move(prog);
    rmdev::SceneView &sv = ctx->scene;
    sv.recs = ctx->d_recs.ptr;
    sv.n_recs = static_cast<uint32_t>(ctx->program.recs.size()) - 1u;
    sv.max_depth = ctx->program.max_depth;
    sv.root_identity = ctx->program.root_identity;
    sv.pa = ctx->d_pa.ptr; sv.pb = ctx->d_pb.ptr; sv.row_prim = ctx->d_row_prim.ptr;
    sv.colors = ctx->d_colors.ptr; sv.lights = ctx->d_lights.ptr;
    sv.n_lights = n_lights; sv.n_prims = ctx->program.n_prims;
    {
        double magnitude = 0.0;
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (nodes[i].type == RM_LIST) continue;
            const rm_node &n = nodes[i];
            const double m = std::fabs((double)n.center[0]) + std::fabs((double)n.center[1]) + std::fabs((double)n.center[2]) +
                             std::max(std::fabs((double)n.size[0]), std::max(std::fabs((double)n.size[1]), std::fabs((double)n.size[2])));
            if (std::isfinite(m)) magnitude = std::max(magnitude, m);
        }
        sv.cap_slack = (float)(4e-6 * magnitude);
    }
    sv.lattice = ctx->d_lattice.ptr;
    sv.lattice_words = ctx->program.n_lattices ? static_cast<uint32_t>(ctx->program.lattice.size()) : 0u;
    sv.camera = ctx->d_camera.ptr;
    if (int rc = build_grid(ctx)) return rc;
    ctx->scene_serial++;
    {
        // Two 64-bit multiply-xor digests over the uploaded bytes, eight at a time (a Menger-6 scene has 54 MB of colours:
        // the bytes themselves would make every cache key that long).  A collision would load another scene's tile
        // order, which load_tile_order accepts only if it covers every tile exactly once: a slower frame, never a wrong one.
        uint64_t d0 = 0x9e3779b97f4a7c15ull, d1 = 0xc2b2ae3d27d4eb4full, total = 0;
        auto add = [&](const void *data, size_t bytes) {
            const unsigned char *b = static_cast<const unsigned char *>(data);
            total += bytes;
            for (size_t i = 0; i < bytes; i += 8) {
                uint64_t w = 0;
                std::memcpy(&w, b + i, bytes - i < 8 ? bytes - i : 8);
                d0 = (d0 ^ w) * 0x100000001b3ull;
                d0 ^= d0 >> 29;
                d1 = (d1 + w) * 0xff51afd7ed558ccdull;
                d1 ^= d1 >> 32;
            }
            d0 = (d0 ^ bytes) * 0x100000001b3ull;   // the parts' lengths count too
        };
        const rmdev::HostProgram &hp = ctx->program;
        add(hp.recs.data(), hp.recs.size() * sizeof(rmdev::Rec));
        add(hp.pa.data(), hp.pa.size() * sizeof(rmdev::Float4));
        add(hp.pb.data(), hp.pb.size() * sizeof(rmdev::Float4));
        add(hp.colors.data(), hp.colors.size() * sizeof(rmdev::Float4));
        add(hp.lattice.data(), hp.lattice.size() * sizeof(uint32_t));
        add(camera, sizeof camera);
        add(lights.data(), lights.size() * sizeof(rmdev::Float4));
        const int modes[2] = { ctx->kernel_mode, ctx->evaluator };
        add(modes, sizeof modes);
        const uint64_t digest[3] = { d0, d1, total };
        ctx->scene_fingerprint.assign(reinterpret_cast<const char *>(digest), sizeof digest);
    }
    ctx->scene.camera_sd_known = 0u;
    if (int rc = load_scene_module(ctx)) {
        ctx->have_scene = false;
        return rc;
    }
    ctx->have_scene = true;
    // the one evaluation all camera rays share (trace_kernel.h: march, first_sd), with the kernels just set up
    ctx->scene.camera_sd_known = 0u;
    bool finite_camera = true;
    for (int i = 0; i < 16; i++) finite_camera = finite_camera && std::isfinite(camera[i]);
    if (finite_camera) {
        float sd = 0.0f, sd_march = 0.0f;
        uint32_t prim = 0;
        if (int rc = eval_points(ctx, camera + 12, 1, &sd, &prim, &sd_march)) return rc;
        if (std::isfinite(sd_march)) {
            ctx->scene.camera_sd = sd_march;
            ctx->scene.camera_sd_known = 1u;
        }
    }
    return preload_tile_order(ctx);
}

int rm_render_device(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8) {
    if (int rc = check_ready(ctx)) return rc;
    if (!rows) return fail(RM_ERR_INVALID, "rm_render_device: null rows");
    return render_bands(ctx, *rows, device_f32, device_rgba8, ctx->rows_in_frame);
}

int rm_enqueue_device(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8) {
    if (int rc = check_ready(ctx)) return rc;
    if (!rows) return fail(RM_ERR_INVALID, "rm_enqueue_device: null rows");
    return enqueue_bands(ctx, *rows, device_f32, device_rgba8, false, nullptr, ctx->rows_in_frame);
}

int rm_enqueue_device_static(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8) {
    if (int rc = check_ready(ctx)) return rc;
    if (!rows) return fail(RM_ERR_INVALID, "rm_enqueue_device_static: null rows");
    void *ptr = static_scratch(ctx->device);
    if (!ptr) return fail(RM_ERR_STATE, "rm_enqueue_device_static: no static scratch on device " + std::to_string(ctx->device));
    return enqueue_bands(ctx, *rows, device_f32, device_rgba8, false, static_cast<rmdev::Sample *>(ptr), ctx->rows_in_frame);
}

int rm_set_row_placement(rm_context *ctx, int placement) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (placement != RM_ROWS_COMPACT && placement != RM_ROWS_IN_FRAME) return fail(RM_ERR_INVALID, "rm_set_row_placement: unknown placement");
    ctx->rows_in_frame = placement == RM_ROWS_IN_FRAME;
    // kernels compiled for the scene before the context placed rows do not know how to: once more, with the switch
    if (ctx->rows_in_frame && ctx->have_scene && ctx->module.loaded() && !ctx->module_places_rows) {
        RM_HIP(hipSetDevice(ctx->device));
        RM_HIP(hipStreamSynchronize(ctx->stream));
        if (int rc = load_scene_module(ctx)) {
            ctx->have_scene = false;
            return rc;
        }
    }
    return RM_OK;
}

size_t rm_static_scratch_bytes(void) { return kStaticScratchBytes; }

int rm_wait(rm_context *ctx) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    return finish_bands(ctx);
}

int rm_set_stream(rm_context *ctx, void *hip_stream) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pending = rm_context::Pending{};
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return RM_OK;
}

int rm_enqueue_host(rm_context *ctx, const rm_rows *rows, float *host_f32, uint8_t *host_rgba8) {
    if (int rc = check_ready(ctx)) return rc;
    if (!rows) return fail(RM_ERR_INVALID, "rm_enqueue_host: null rows");
    if (!host_f32 && !host_rgba8) return fail(RM_ERR_INVALID, "rm_enqueue_host: no output buffer");
    const size_t pixels = (size_t)rows->band_count * rows->band_rows * ctx->params.width;
    RM_HIP(hipSetDevice(ctx->device));
    if (host_f32) RM_HIP(ctx->d_frame_f32.reserve(pixels * 4));
    if (host_rgba8) RM_HIP(ctx->d_frame_rgba8.reserve(pixels * 4));
    if (int rc = enqueue_bands(ctx, *rows, host_f32 ? ctx->d_frame_f32.ptr : nullptr, host_rgba8 ? ctx->d_frame_rgba8.ptr : nullptr, false))
        return rc;
    // behind the kernels on the context's stream; with page-locked host memory (rm_host_alloc, rm_host_register) the
    // copies run on the copy engine while other streams' kernels have the CUs
    if (host_f32) RM_HIP(hipMemcpyAsync(host_f32, ctx->d_frame_f32.ptr, pixels * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (host_rgba8) RM_HIP(hipMemcpyAsync(host_rgba8, ctx->d_frame_rgba8.ptr, pixels * 4, hipMemcpyDeviceToHost, ctx->stream));
    return RM_OK;
}

int rm_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(RM_ERR_INVALID, "rm_host_alloc: null output pointer");
    *out = nullptr;
    RM_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RM_OK;
}

int rm_host_free(void *ptr) {
    if (ptr) RM_HIP(hipHostFree(ptr));
    return RM_OK;
}

int rm_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(RM_ERR_INVALID, "rm_host_register: null or empty range");
    RM_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return RM_OK;
}

int rm_host_unregister(void *ptr) {
    if (ptr) RM_HIP(hipHostUnregister(ptr));
    return RM_OK;
}

namespace {

// The whole frame into host memory, synchronously: the first render of a configuration times its tiles (device
// buffers), every render ends with the copy behind the kernels on the context's stream.
int render_host(rm_context *ctx, float *host_f32, uint8_t *host_rgba8) {
    const rm_rows whole = { ctx->params.height, 0, 1, 1 };
    const size_t pixels = (size_t)ctx->params.width * ctx->params.height;
    RM_HIP(hipSetDevice(ctx->device));
    // An 8-bit image in page-locked memory is written by the resolve pass itself where there is one (rows of 64 pixels, 256
    // bytes a wave: the copy's 3 MB cross the bus while the pass runs instead of behind it).  Not where the trace
    // kernel writes the pixels (2 x 2-pixel tiles: 8-byte pieces), and not for frames in flight (rm_enqueue_host copies:
    // kernels that wait for the bus hold the compute units other frames could use).  $RM_DIRECT_HOST=0: always copy.
    unsigned char *direct = nullptr;
    static const bool direct_allowed = [] { const char *e = std::getenv("RM_DIRECT_HOST"); return !(e && e[0] == '0'); }();
    const bool resolve_pass = !(ctx->kernel_mode == RM_KERNEL_COMPILED && ctx->module.loaded() &&
                                (8u * (ctx->program.lattice.size() > 1024 ? 2u : 1u)) % ctx->params.kernel == 0u);
    if (host_rgba8 && !host_f32 && direct_allowed && resolve_pass) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, host_rgba8) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
            direct = static_cast<unsigned char *>(at.devicePointer);
        else
            (void)hipGetLastError();   // pageable memory: not an error here
    }
    if (host_f32) RM_HIP(ctx->d_frame_f32.reserve(pixels * 4));
    if (host_rgba8 && !direct) RM_HIP(ctx->d_frame_rgba8.reserve(pixels * 4));
    if (int rc = enqueue_bands(ctx, whole, host_f32 ? ctx->d_frame_f32.ptr : nullptr, host_rgba8 ? (direct ? direct : ctx->d_frame_rgba8.ptr) : nullptr, true))
        return rc;
    if (host_f32) RM_HIP(hipMemcpyAsync(host_f32, ctx->d_frame_f32.ptr, pixels * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (host_rgba8 && !direct) RM_HIP(hipMemcpyAsync(host_rgba8, ctx->d_frame_rgba8.ptr, pixels * 4, hipMemcpyDeviceToHost, ctx->stream));
    return finish_bands(ctx);
}

}  // namespace

int rm_render_rgba8(rm_context *ctx, uint8_t *host_image) {
    if (int rc = check_ready(ctx)) return rc;
    if (!host_image) return fail(RM_ERR_INVALID, "rm_render_rgba8: null image");
    return render_host(ctx, nullptr, host_image);
}

int rm_render_f32(rm_context *ctx, float *host_image) {
    if (int rc = check_ready(ctx)) return rc;
    if (!host_image) return fail(RM_ERR_INVALID, "rm_render_f32: null image");
    return render_host(ctx, host_image, nullptr);
}

namespace {

int eval_points(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim, float *host_sd_march) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (!ctx->have_scene) return fail(RM_ERR_STATE, "rm_set_scene has not been called");
    if (n == 0) return RM_OK;
    if (!host_xyz || !host_sd || !host_prim) return fail(RM_ERR_INVALID, "rm_eval_sdf: null argument");
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(ctx->d_points.reserve((size_t)n * 3));
    RM_HIP(ctx->d_sd.reserve((size_t)n * 2));
    RM_HIP(ctx->d_prim.reserve(n));
    float *d_march = host_sd_march ? ctx->d_sd.ptr + n : nullptr;
    RM_HIP(hipMemcpyAsync(ctx->d_points.ptr, host_xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->module.loaded()) RM_HIP(rmdev::launch_eval_sdf_module(ctx->module, ctx->scene, ctx->d_points.ptr, n, ctx->d_sd.ptr, ctx->d_prim.ptr, d_march, ctx->stream));
    else RM_HIP(rmdev::launch_eval_sdf(ctx->scene, ctx->d_points.ptr, n, ctx->d_sd.ptr, ctx->d_prim.ptr, d_march, ctx->stream));
    RM_HIP(hipMemcpyAsync(host_sd, ctx->d_sd.ptr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(hipMemcpyAsync(host_prim, ctx->d_prim.ptr, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (host_sd_march) RM_HIP(hipMemcpyAsync(host_sd_march, d_march, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    return RM_OK;
}

}  // namespace

int rm_eval_sdf(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim) {
    return eval_points(ctx, host_xyz, n, host_sd, host_prim, nullptr);
}

int rm_eval_sdf_both(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim, float *host_sd_march) {
    if (!host_sd_march) return fail(RM_ERR_INVALID, "rm_eval_sdf_both: null argument");
    return eval_points(ctx, host_xyz, n, host_sd, host_prim, host_sd_march);
}

int rm_kernel_source(const rm_node *nodes, uint32_t n_nodes, int evaluator, int counting, char *buffer, size_t capacity, size_t *length) {
    if (!length) return fail(RM_ERR_INVALID, "rm_kernel_source: null length pointer");
    std::string why;
    rmdev::HostProgram prog;
    if (!rmdev::compile_scene(nodes, n_nodes, evaluator == RM_EVAL_LATTICE, prog, why)) return fail(RM_ERR_INVALID, "rm_kernel_source: " + why);
    const std::string src = rmdev::generate_scene_source(prog, kCompiledPts, kCompiledWaves, counting != 0, std::getenv("RM_REFILL") != nullptr);
    *length = src.size();
    if (buffer && capacity) {
        const size_t n = src.size() < capacity - 1 ? src.size() : capacity - 1;
        std::memcpy(buffer, src.data(), n);
        buffer[n] = '\0';
    }
    return RM_OK;
}

int rm_precompile_scene(const rm_node *nodes, uint32_t n_nodes, int evaluator) {
    if (!nodes || n_nodes == 0) return fail(RM_ERR_INVALID, "rm_precompile_scene: no nodes");
    // Everything in the background, the scene's host half too (tables and source of a Menger-6 scene take a quarter of a
    // second): the caller goes on to create its context.  A scene that does not compile is rm_set_scene's to report.
    static std::mutex lock;
    static std::vector<std::future<void>> running;   // (waited for when the process ends: made late, so destroyed early)
    std::vector<rm_node> copy(nodes, nodes + n_nodes);
    const bool refill = std::getenv("RM_REFILL") != nullptr;
    std::lock_guard<std::mutex> hold(lock);
    for (size_t i = 0; i < running.size();) {
        if (running[i].wait_for(std::chrono::seconds(0)) == std::future_status::ready) running.erase(running.begin() + (long)i);
        else i++;
    }
    if (running.size() >= 8) return RM_OK;   // enough announced scenes on their way
    running.push_back(std::async(std::launch::async, [copy = std::move(copy), evaluator, refill] {
        std::string why;
        rmdev::HostProgram prog;
        if (!rmdev::compile_scene(copy.data(), (uint32_t)copy.size(), evaluator == RM_EVAL_LATTICE, prog, why)) return;
        rmdev::precompile_scene_source(rmdev::generate_scene_source(prog, kCompiledPts, kCompiledWaves, false, refill));
    }));
    return RM_OK;
}

int rm_set_kernel_cache_dir(const char *directory) {
    rmdev::set_kernel_cache_dir(directory);
    return RM_OK;
}

const char *rm_kernel_compiler(void) {
    static thread_local std::string text;
    text = rmdev::kernel_compiler_used();
    return text.c_str();
}

int rm_kernel_source_compiles(const char *source) {
    if (!source) return fail(RM_ERR_INVALID, "rm_kernel_source_compiles: null source");
    std::string log;
    if (!rmdev::check_scene_source(source, log)) return fail(RM_ERR_HIP, log);
    return RM_OK;
}

int rm_set_evaluator(rm_context *ctx, int evaluator) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (evaluator != RM_EVAL_EXHAUSTIVE && evaluator != RM_EVAL_LATTICE)
        return fail(RM_ERR_INVALID, "rm_set_evaluator: unknown evaluator");
    ctx->evaluator = evaluator;
    return RM_OK;
}

int rm_set_kernel_mode(rm_context *ctx, int mode) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (mode != RM_KERNEL_INTERPRETED && mode != RM_KERNEL_COMPILED) return fail(RM_ERR_INVALID, "rm_set_kernel_mode: unknown mode");
    ctx->kernel_mode = mode;
    return RM_OK;
}

int rm_set_tile_scheduling(rm_context *ctx, int mode) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (mode != RM_TILES_NATURAL && mode != RM_TILES_BY_COST) return fail(RM_ERR_INVALID, "rm_set_tile_scheduling: unknown mode");
    ctx->tile_scheduling = mode;
    ctx->order_ready = false;
    ctx->costs_valid = false;
    return RM_OK;
}

int rm_set_lane_refill(rm_context *ctx, uint32_t tile_w_log2, uint32_t tile_h_log2) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    const bool off = tile_w_log2 == 0u && tile_h_log2 == 0u;
    if (!off && (tile_w_log2 < 3u || tile_h_log2 < 3u || tile_w_log2 > 8u || tile_h_log2 > 8u))
        return fail(RM_ERR_INVALID, "rm_set_lane_refill: tile sides are 2^3 .. 2^8 samples (0, 0 turns it off)");
    ctx->refill_w_log2 = tile_w_log2;
    ctx->refill_h_log2 = tile_h_log2;
    return RM_OK;
}

int rm_set_shading(rm_context *ctx, int mode) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    if (mode != RM_SHADE_ALL && mode != RM_SHADE_COLOURED) return fail(RM_ERR_INVALID, "rm_set_shading: unknown mode");
    ctx->shading = mode;
    return RM_OK;
}

int rm_set_band_list(rm_context *ctx, const uint32_t *image_bands, uint32_t count) {
    if (!ctx || (count && !image_bands)) return fail(RM_ERR_INVALID, "rm_set_band_list: null argument");
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    if (count) {
        RM_HIP(ctx->d_band_list.reserve(count));
        RM_HIP(hipMemcpy(ctx->d_band_list.ptr, image_bands, count * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    ctx->band_list_host.assign(image_bands, image_bands + count);
    ctx->band_list_size = count;
    ctx->band_list_serial++;
    return RM_OK;
}

int rm_get_tile_costs(rm_context *ctx, uint32_t *ticks, size_t capacity, uint32_t *tiles_x, uint32_t *tiles_y) {
    if (!ctx || !tiles_x || !tiles_y) return fail(RM_ERR_INVALID, "rm_get_tile_costs: null argument");
    if (!ctx->costs_valid)
        return fail(RM_ERR_STATE, "rm_get_tile_costs: no render of this context has timed the tiles of its current configuration "
                                  "(an order read from the cache directory comes without durations)");
    *tiles_x = ctx->measured_tiles_x;
    *tiles_y = ctx->measured_tiles_y;
    const size_t n = (size_t)ctx->measured_tiles_x * ctx->measured_tiles_y;
    if (ticks) {
        if (capacity < n) return fail(RM_ERR_INVALID, "rm_get_tile_costs: capacity too small");
        RM_HIP(hipSetDevice(ctx->device));
        RM_HIP(hipMemcpy(ticks, ctx->d_tile_cost.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return RM_OK;
}

int rm_set_slot_stats(rm_context *ctx, int enabled) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    ctx->slot_stats = enabled != 0;
    return RM_OK;
}

int rm_get_slot_stats(rm_context *ctx, uint32_t *words, size_t capacity_slots, size_t *n_slots) {
    if (!ctx || !n_slots) return fail(RM_ERR_INVALID, "rm_get_slot_stats: null argument");
    *n_slots = ctx->slot_stats_slots;
    if (words) {
        if (capacity_slots < ctx->slot_stats_slots) return fail(RM_ERR_INVALID, "rm_get_slot_stats: capacity too small");
        RM_HIP(hipSetDevice(ctx->device));
        RM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->slot_stats_slots)
            RM_HIP(hipMemcpy(words, ctx->d_slot_stats.ptr, 4u * ctx->slot_stats_slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return RM_OK;
}

int rm_get_exact_tiles(rm_context *ctx, uint64_t *tiles) {
    if (!ctx || !tiles) return fail(RM_ERR_INVALID, "rm_get_exact_tiles: null argument");
    *tiles = 0;
    if (!ctx->d_warm.ptr) return RM_OK;
    RM_HIP(hipSetDevice(ctx->device));
    RM_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t word = 0;
    RM_HIP(hipMemcpy(&word, ctx->d_warm.ptr + 512, sizeof(word), hipMemcpyDeviceToHost));
    *tiles = word;
    return RM_OK;
}

int rm_set_counting(rm_context *ctx, int enabled) {
    if (!ctx) return fail(RM_ERR_INVALID, "null context");
    ctx->counting = enabled != 0;
    return RM_OK;
}

int rm_get_counters(rm_context *ctx, rm_counters *out) {
    if (!ctx || !out) return fail(RM_ERR_INVALID, "rm_get_counters: null argument");
    *out = ctx->last;
    return RM_OK;
}

void *rm_stream(rm_context *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

int rm_write_png(const char *path, const uint8_t *rgba, uint32_t width, uint32_t height) {
    const int rc = rm_write_png_file(path, rgba, width, height);
    if (rc == RM_ERR_INVALID) return fail(RM_ERR_INVALID, "rm_write_png: null path or image, or an empty image");
    if (rc != RM_OK) return fail(RM_ERR_IO, std::string("rm_write_png: cannot write ") + path);
    return RM_OK;
}

// ---- host scene library ----------------------------------------------------

struct rm_scene {
    scene::Scene s;
};

int rm_scene_load(const char *path, rm_scene **out) {
    if (!path || !out) return fail(RM_ERR_INVALID, "rm_scene_load: null argument");
    rm_scene *sc = new (std::nothrow) rm_scene();
    if (!sc) return fail(RM_ERR_INVALID, "rm_scene_load: out of host memory");
    sc->s.load(path);  // a missing file leaves the scene empty, as scene::load does
    *out = sc;
    return RM_OK;
}

int rm_scene_parse(const char *text, rm_scene **out) {
    if (!text || !out) return fail(RM_ERR_INVALID, "rm_scene_parse: null argument");
    rm_scene *sc = new (std::nothrow) rm_scene();
    if (!sc) return fail(RM_ERR_INVALID, "rm_scene_parse: out of host memory");
    sc->s.parse(text);
    *out = sc;
    return RM_OK;
}

void rm_scene_free(rm_scene *scene) { delete scene; }

uint32_t rm_scene_node_count(const rm_scene *scene) {
    return scene ? static_cast<uint32_t>(scene->s.nodes().size()) : 0;
}

uint32_t rm_scene_light_count(const rm_scene *scene) {
    return scene ? static_cast<uint32_t>(scene->s.lights.size()) : 0;
}

int rm_scene_nodes(const rm_scene *scene, rm_node *out, uint32_t capacity) {
    if (!scene || !out) return fail(RM_ERR_INVALID, "rm_scene_nodes: null argument");
    const std::vector<rm_node> nodes = scene->s.nodes();
    if (capacity < nodes.size()) return fail(RM_ERR_INVALID, "rm_scene_nodes: capacity too small");
    std::memcpy(out, nodes.data(), nodes.size() * sizeof(rm_node));
    return RM_OK;
}

int rm_scene_lights(const rm_scene *scene, float *out_xyz, uint32_t capacity) {
    if (!scene || (!out_xyz && capacity)) return fail(RM_ERR_INVALID, "rm_scene_lights: null argument");
    const std::vector<float> l = scene->s.light_positions();
    if ((size_t)capacity * 3 < l.size()) return fail(RM_ERR_INVALID, "rm_scene_lights: capacity too small");
    if (!l.empty()) std::memcpy(out_xyz, l.data(), l.size() * sizeof(float));
    return RM_OK;
}

int rm_scene_camera(const rm_scene *scene, float transform[16], float *focal) {
    if (!scene || !transform || !focal) return fail(RM_ERR_INVALID, "rm_scene_camera: null argument");
    std::memcpy(transform, scene->s.camera->transform.m, 16 * sizeof(float));
    *focal = scene->s.camera->focal;
    return RM_OK;
}

int rm_set_scene_from(rm_context *ctx, const rm_scene *scene) {
    if (!ctx || !scene) return fail(RM_ERR_INVALID, "rm_set_scene_from: null argument");
    const std::vector<rm_node> nodes = scene->s.nodes();
    const std::vector<float> lights = scene->s.light_positions();
    return rm_set_scene(ctx, nodes.data(), static_cast<uint32_t>(nodes.size()), lights.data(),
                        static_cast<uint32_t>(lights.size() / 3), scene->s.camera->transform.m,
                        scene->s.camera->focal);
}

}  // extern "C"
