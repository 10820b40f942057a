// multi.cpp -- one frame over several GPUs from one process: rm_multi_* of include/raymarch_hip.h.
//
// The reference's render::GPU drives one device (/root/reference/source/render.cpp:468-475).  Pixels are independent
// (render.cpp:122-129), so a frame shards by rows with one exchange step: rank r (one rm_context per listed device)
// renders image bands r, r + n, r + 2n, ... of RM_MULTI_BAND_ROWS rows into a compact buffer on its own device --
// narrow cyclic bands, because cost per row differs by more than ten times between sky and sponge and interleaving
// gives every rank a statistically equal share with all shares the same size -- and the shares are gathered on rank
// 0's device: grouped ncclSend / ncclRecv (RCCL over xGMI: every peer has its own link to rank 0) followed by a
// de-interleaving 2-D copy per rank, or, where the device list repeats a device (RCCL refuses that), 2-D peer copies
// that de-interleave on the way.  Or no exchange step at all (RM_MULTI_DIRECT): every rank's kernels store their rows
// at their place in the frame on rank 0's device -- stores to a peer's memory travel over the rank's own xGMI link to
// rank 0 as the pixels are finished -- and rank 0's stream only waits for the ranks' kernels.
// Built on the public C ABI only: a rank is an ordinary context.
#include "../../include/raymarch_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

int rm_internal_fail(int code, const std::string &what);   // api.cpp: sets rm_last_error() of the calling thread

namespace {

#define RM_MHIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return rm_internal_fail(RM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// librccl, loaded on demand: a process that renders on one GPU, or gathers by copies, never needs it.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

const Rccl &rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        // the SONAME first: a process that already holds an RCCL (PyTorch ships one) must use that one
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            lib.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib.handle) break;
        }
        if (!lib.handle) {
            lib.why = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(lib.handle, name);
            if (!p) { ok = false; lib.why = std::string("librccl lacks ") + name; }
            return p;
        };
        lib.CommInitAll = reinterpret_cast<decltype(lib.CommInitAll)>(sym("ncclCommInitAll"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.Send = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));
        lib.Recv = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) lib.handle = nullptr;
    });
    return lib;
}

#define RM_NCCL(expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess) return rm_internal_fail(RM_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

struct Rank {
    int device = 0;
    rm_context *ctx = nullptr;
    unsigned char *shard = nullptr;   // this rank's bands, compact, on its own device (rank 0: inside `stack`)
    hipEvent_t rendered = nullptr;    // its kernels are done
    ncclComm_t comm = nullptr;
    bool ordered = false;             // a synchronous render has left this rank a tile order for the configuration
    bool queued = false;              // the rank's last frame was queued (not synchronous)
};

}  // namespace

struct rm_multi {
    std::vector<Rank> ranks;
    int transport = RM_MULTI_COPY;
    bool have_params = false, have_scene = false;
    rm_params params{};
    // geometry of the split
    uint32_t bands_per_rank = 0;
    size_t shard_bytes = 0, band_bytes = 0;
    // on rank 0's device
    unsigned char *stack = nullptr;   // RCCL: the ranks' shares, rank-major (copies gather without it)
    unsigned char *frame = nullptr;   // the assembled frame
    size_t stack_bytes = 0, frame_bytes = 0;
    size_t shard_capacity = 0;
    hipEvent_t gathered = nullptr;
    bool in_flight = false;           // rm_multi_enqueue_* has queued a frame that rm_multi_wait has not waited for
};

namespace {

void release_buffers(rm_multi *m) {
    for (size_t r = 0; r < m->ranks.size(); r++) {
        Rank &k = m->ranks[r];
        if (!k.shard) continue;
        (void)hipSetDevice(k.device);
        if (k.shard && !(m->stack && r == 0)) (void)hipFree(k.shard);
        k.shard = nullptr;
    }
    if (m->stack || m->frame) (void)hipSetDevice(m->ranks[0].device);
    if (m->stack) (void)hipFree(m->stack);
    if (m->frame) (void)hipFree(m->frame);
    m->stack = m->frame = nullptr;
    m->stack_bytes = m->frame_bytes = m->shard_capacity = 0;
}

// sizes the split and the buffers for the current parameters
int plan(rm_multi *m) {
    const rm_params &p = m->params;
    const uint32_t n = (uint32_t)m->ranks.size();
    const uint32_t n_bands = (p.height + RM_MULTI_BAND_ROWS - 1u) / RM_MULTI_BAND_ROWS;
    m->bands_per_rank = (n_bands + n - 1u) / n;
    m->band_bytes = (size_t)RM_MULTI_BAND_ROWS * p.width * 4u;
    m->shard_bytes = (size_t)m->bands_per_rank * m->band_bytes;
    const size_t frame_bytes = (size_t)p.width * p.height * 4u;
    const bool shards = m->transport != RM_MULTI_DIRECT;   // direct stores need the frame only
    if ((!shards || m->shard_bytes <= m->shard_capacity) && frame_bytes <= m->frame_bytes) return RM_OK;
    release_buffers(m);
    RM_MHIP(hipSetDevice(m->ranks[0].device));
    RM_MHIP(hipMalloc(reinterpret_cast<void **>(&m->frame), frame_bytes));
    m->frame_bytes = frame_bytes;
    if (!shards) return RM_OK;
    if (m->transport == RM_MULTI_RCCL) {
        m->stack_bytes = m->shard_bytes * n;
        RM_MHIP(hipMalloc(reinterpret_cast<void **>(&m->stack), m->stack_bytes));
        m->ranks[0].shard = m->stack;   // rank 0 renders where the others' shares arrive
    }
    for (uint32_t r = 0; r < n; r++) {
        Rank &k = m->ranks[r];
        if (k.shard) continue;
        RM_MHIP(hipSetDevice(k.device));
        RM_MHIP(hipMalloc(reinterpret_cast<void **>(&k.shard), m->shard_bytes));
    }
    m->shard_capacity = m->shard_bytes;
    return RM_OK;
}

// Rank r's share, which lies at `src` on rank 0's device or on its own, into the frame on rank 0's device: band b of
// the share is image band r + b n.  One 2-D copy for the bands that lie wholly in the image, one more for a last band
// that the image's height cuts short; bands below the image are left out.  On rank 0's stream.
int scatter_share(rm_multi *m, uint32_t r, const unsigned char *src, hipStream_t stream) {
    const rm_params &p = m->params;
    const uint32_t n = (uint32_t)m->ranks.size();
    const size_t row_bytes = (size_t)p.width * 4u;
    uint32_t whole = 0, cut_rows = 0;
    for (uint32_t b = 0; b < m->bands_per_rank; b++) {
        const uint64_t first_row = (uint64_t)(r + (uint64_t)b * n) * RM_MULTI_BAND_ROWS;
        if (first_row + RM_MULTI_BAND_ROWS <= p.height) whole++;
        else if (first_row < p.height) cut_rows = (uint32_t)(p.height - first_row);   // at most one such band, the share's last in the image
    }
    unsigned char *dst = m->frame + (size_t)r * m->band_bytes;
    const size_t dst_pitch = (size_t)n * m->band_bytes;
    if (whole) RM_MHIP(hipMemcpy2DAsync(dst, dst_pitch, src, m->band_bytes, m->band_bytes, whole, hipMemcpyDeviceToDevice, stream));
    if (cut_rows)
        RM_MHIP(hipMemcpyAsync(dst + (size_t)whole * dst_pitch, src + (size_t)whole * m->band_bytes, cut_rows * row_bytes, hipMemcpyDeviceToDevice, stream));
    return RM_OK;
}

// Queues the frame on every rank and its exchange on rank 0's stream.  `direct`: where the ranks' kernels store the frame
// when the transport is RM_MULTI_DIRECT (on rank 0's device; null: the object's own frame).  Afterwards the frame lies
// there, or with the other transports in the object's own frame.
int render_frame(rm_multi *m, unsigned char *direct) {
    if (!m->have_params) return rm_internal_fail(RM_ERR_STATE, "rm_multi_set_params has not been called");
    if (!m->have_scene) return rm_internal_fail(RM_ERR_STATE, "rm_multi_set_scene has not been called");
    if (int rc = plan(m)) return rc;
    const uint32_t n = (uint32_t)m->ranks.size();
    hipStream_t stream0 = static_cast<hipStream_t>(rm_stream(m->ranks[0].ctx));
    const bool in_place = m->transport == RM_MULTI_DIRECT;
    if (in_place && !direct) direct = m->frame;
    // ---- every rank renders its bands
    for (uint32_t r = 0; r < n; r++) {
        Rank &k = m->ranks[r];
        const rm_rows rows = { RM_MULTI_BAND_ROWS, r, n, m->bands_per_rank };
        unsigned char *const out = in_place ? direct : k.shard;
        if (!k.ordered) {
            // the first frame of a configuration: synchronous, so that the context times its tiles (or reads their order
            // from the cache directory); queued frames never measure
            if (int rc = rm_render_device(k.ctx, &rows, nullptr, out)) return rc;
            k.ordered = true;
            k.queued = false;
        } else {
            if (int rc = rm_enqueue_device(k.ctx, &rows, nullptr, out)) return rc;
            k.queued = true;
        }
        RM_MHIP(hipSetDevice(k.device));
        RM_MHIP(hipEventRecord(k.rendered, static_cast<hipStream_t>(rm_stream(k.ctx))));
    }
    // ---- the exchange: shares to rank 0's device, de-interleaved into the frame
    RM_MHIP(hipSetDevice(m->ranks[0].device));
    if (in_place) {
        // the rows are where they belong once the ranks' kernels have ended
        for (uint32_t r = 1; r < n; r++) RM_MHIP(hipStreamWaitEvent(stream0, m->ranks[r].rendered, 0));
    } else if (m->transport == RM_MULTI_RCCL) {
        const Rccl &nc = rccl();
        RM_NCCL(nc.GroupStart());
        for (uint32_t r = 1; r < n; r++) {
            // each call is ordered behind the work queued on the stream it is given: the sender's kernels, rank 0's own
            RM_NCCL(nc.Send(m->ranks[r].shard, m->shard_bytes, ncclUint8, 0, m->ranks[r].comm, static_cast<hipStream_t>(rm_stream(m->ranks[r].ctx))));
            RM_NCCL(nc.Recv(m->stack + (size_t)r * m->shard_bytes, m->shard_bytes, ncclUint8, (int)r, m->ranks[0].comm, stream0));
        }
        RM_NCCL(nc.GroupEnd());
        RM_MHIP(hipSetDevice(m->ranks[0].device));
        for (uint32_t r = 0; r < n; r++)
            if (int rc = scatter_share(m, r, m->stack + (size_t)r * m->shard_bytes, stream0)) return rc;
    } else {
        for (uint32_t r = 0; r < n; r++) {
            if (r) RM_MHIP(hipStreamWaitEvent(stream0, m->ranks[r].rendered, 0));
            if (int rc = scatter_share(m, r, m->ranks[r].shard, stream0)) return rc;
        }
    }
    RM_MHIP(hipEventRecord(m->gathered, stream0));
    return RM_OK;
}

// every rank's frame is booked (kernel times, counters), the exchange is done
int finish_frame(rm_multi *m) {
    int first = RM_OK;
    for (Rank &k : m->ranks) {
        const int rc = rm_wait(k.ctx);
        if (rc != RM_OK && first == RM_OK) first = rc;
    }
    if (first != RM_OK) return first;
    // A queued frame that found no tile order for its configuration (a setting changed on the rank's context since its
    // last synchronous frame) ran in natural order: the rank's next frame is synchronous again and times its tiles.
    for (Rank &k : m->ranks) {
        rm_counters c{};
        if (k.queued && rm_get_counters(k.ctx, &c) == RM_OK && c.tiles_by_cost == 0u) k.ordered = false;
    }
    RM_MHIP(hipSetDevice(m->ranks[0].device));
    RM_MHIP(hipEventSynchronize(m->gathered));
    return RM_OK;
}

}  // namespace

extern "C" {

int rm_multi_create(const int *devices, uint32_t n_devices, int transport, rm_multi **out) {
    if (!out) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_create: null output pointer");
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64u) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_create: 1 to 64 devices");
    if (transport != RM_MULTI_AUTO && transport != RM_MULTI_RCCL && transport != RM_MULTI_COPY && transport != RM_MULTI_DIRECT)
        return rm_internal_fail(RM_ERR_INVALID, "rm_multi_create: unknown transport");
    const bool distinct = std::set<int>(devices, devices + n_devices).size() == n_devices;
    if (transport == RM_MULTI_RCCL && !distinct)
        return rm_internal_fail(RM_ERR_INVALID, "rm_multi_create: RCCL takes one rank per device; this list names a device twice (use RM_MULTI_COPY)");
    if (transport == RM_MULTI_AUTO) {
        // direct stores wherever every rank's device can write rank 0's memory; RCCL otherwise
        bool peers = true;
        for (uint32_t r = 1; r < n_devices && peers; r++) {
            int can = 0;
            if (devices[r] != devices[0] && (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) != hipSuccess || !can)) peers = false;
        }
        (void)hipGetLastError();
        transport = peers ? RM_MULTI_DIRECT : (distinct ? RM_MULTI_RCCL : RM_MULTI_COPY);
    }
    if (transport == RM_MULTI_RCCL && n_devices == 1u) transport = RM_MULTI_COPY;   // nothing to exchange
    rm_multi *m = new (std::nothrow) rm_multi();
    if (!m) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_create: out of host memory");
    m->transport = transport;
    m->ranks.resize(n_devices);
    for (uint32_t r = 0; r < n_devices; r++) m->ranks[r].device = devices[r];
    auto undo = [&](int rc) {
        const std::string why = rm_last_error();
        rm_multi_destroy(m);
        return rm_internal_fail(rc, why);
    };
    for (uint32_t r = 0; r < n_devices; r++) {
        Rank &k = m->ranks[r];
        if (int rc = rm_create(k.device, &k.ctx)) return undo(rc);
        if (hipSetDevice(k.device) != hipSuccess || hipEventCreateWithFlags(&k.rendered, hipEventDisableTiming) != hipSuccess) {
            rm_internal_fail(RM_ERR_HIP, "rm_multi_create: hipEventCreate failed on device " + std::to_string(k.device));
            return undo(RM_ERR_HIP);
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&m->gathered, hipEventDisableTiming) != hipSuccess) {
        rm_internal_fail(RM_ERR_HIP, "rm_multi_create: hipEventCreate failed on device " + std::to_string(devices[0]));
        return undo(RM_ERR_HIP);
    }
    if (transport == RM_MULTI_DIRECT) {
        // every rank's kernels write the frame on rank 0's device
        for (uint32_t r = 0; r < n_devices; r++) {
            if (int rc = rm_set_row_placement(m->ranks[r].ctx, RM_ROWS_IN_FRAME)) return undo(rc);
            if (devices[r] == devices[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) != hipSuccess || !can) {
                rm_internal_fail(RM_ERR_HIP, "rm_multi_create: device " + std::to_string(devices[r]) + " cannot write device " +
                                                 std::to_string(devices[0]) + "'s memory (no peer access)");
                return undo(RM_ERR_HIP);
            }
            hipError_t e = hipSetDevice(devices[r]);
            if (e == hipSuccess) e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                rm_internal_fail(RM_ERR_HIP, std::string("rm_multi_create: hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                return undo(RM_ERR_HIP);
            }
            (void)hipGetLastError();
        }
    } else if (transport == RM_MULTI_COPY) {
        // rank 0's device reads the other devices' memory in the gathering copies
        for (uint32_t r = 1; r < n_devices; r++) {
            if (devices[r] == devices[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[r]) != hipSuccess || !can) {
                rm_internal_fail(RM_ERR_HIP, "rm_multi_create: device " + std::to_string(devices[0]) + " cannot read device " +
                                                 std::to_string(devices[r]) + "'s memory (no peer access)");
                return undo(RM_ERR_HIP);
            }
            const hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                rm_internal_fail(RM_ERR_HIP, std::string("rm_multi_create: hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                return undo(RM_ERR_HIP);
            }
            (void)hipGetLastError();
        }
    } else {
        const Rccl &nc = rccl();
        if (!nc.handle) {
            rm_internal_fail(RM_ERR_HIP, "rm_multi_create: " + nc.why);
            return undo(RM_ERR_HIP);
        }
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        const ncclResult_t res = nc.CommInitAll(comms.data(), (int)n_devices, devices);
        if (res != ncclSuccess) {
            rm_internal_fail(RM_ERR_HIP, std::string("rm_multi_create: ncclCommInitAll: ") + nc.GetErrorString(res));
            return undo(RM_ERR_HIP);
        }
        for (uint32_t r = 0; r < n_devices; r++) m->ranks[r].comm = comms[r];
    }
    *out = m;
    return RM_OK;
}

int rm_multi_destroy(rm_multi *m) {
    if (!m) return RM_OK;
    if (m->in_flight) (void)rm_multi_wait(m);
    for (Rank &k : m->ranks)
        if (k.ctx) (void)rm_wait(k.ctx);
    for (Rank &k : m->ranks)
        if (k.comm && rccl().handle) (void)rccl().CommDestroy(k.comm);
    release_buffers(m);
    for (Rank &k : m->ranks) {
        if (!k.ctx) continue;   // a rank whose device could not be opened (rm_multi_create undoing itself)
        (void)hipSetDevice(k.device);
        if (k.rendered) (void)hipEventDestroy(k.rendered);
        (void)rm_destroy(k.ctx);
    }
    if (m->gathered) {
        (void)hipSetDevice(m->ranks[0].device);
        (void)hipEventDestroy(m->gathered);
    }
    delete m;
    return RM_OK;
}

int rm_multi_set_params(rm_multi *m, const rm_params *params) {
    if (!m || !params) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_set_params: null argument");
    if (int rc = rm_multi_wait(m)) return rc;
    for (Rank &k : m->ranks) {
        if (int rc = rm_set_params(k.ctx, params)) return rc;
        k.ordered = false;
    }
    m->params = *params;
    m->have_params = true;
    return RM_OK;
}

int rm_multi_set_scene(rm_multi *m, const rm_node *nodes, uint32_t n_nodes, const float *lights_xyz, uint32_t n_lights,
                       const float transform[16], float focal) {
    if (!m) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_set_scene: null argument");
    if (int rc = rm_multi_wait(m)) return rc;
    m->have_scene = false;
    for (Rank &k : m->ranks) {
        // the first rank compiles the scene's kernels; the others find the code object in the process's cache
        if (int rc = rm_set_scene(k.ctx, nodes, n_nodes, lights_xyz, n_lights, transform, focal)) return rc;
        k.ordered = false;
    }
    m->have_scene = true;
    return RM_OK;
}

int rm_multi_set_scene_from(rm_multi *m, const rm_scene *scene) {
    if (!m || !scene) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_set_scene_from: null argument");
    const uint32_t n_nodes = rm_scene_node_count(scene), n_lights = rm_scene_light_count(scene);
    std::vector<rm_node> nodes(n_nodes ? n_nodes : 1);
    std::vector<float> lights(n_lights ? 3u * n_lights : 3u);
    float transform[16], focal = 0.0f;
    if (int rc = rm_scene_nodes(scene, nodes.data(), n_nodes)) return rc;
    if (int rc = rm_scene_lights(scene, lights.data(), n_lights)) return rc;
    if (int rc = rm_scene_camera(scene, transform, &focal)) return rc;
    return rm_multi_set_scene(m, nodes.data(), n_nodes, lights.data(), n_lights, transform, focal);
}

int rm_multi_enqueue_device(rm_multi *m, uint8_t *device_rgba8) {
    if (!m || !device_rgba8) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_enqueue_device: null argument");
    if (m->in_flight) return rm_internal_fail(RM_ERR_STATE, "rm_multi_enqueue_device: a frame is in flight (rm_multi_wait first)");
    if (int rc = render_frame(m, device_rgba8)) return rc;
    hipStream_t stream0 = static_cast<hipStream_t>(rm_stream(m->ranks[0].ctx));
    if (m->transport != RM_MULTI_DIRECT) {
        RM_MHIP(hipMemcpyAsync(device_rgba8, m->frame, (size_t)m->params.width * m->params.height * 4u, hipMemcpyDeviceToDevice, stream0));
        RM_MHIP(hipEventRecord(m->gathered, stream0));
    }
    m->in_flight = true;
    return RM_OK;
}

int rm_multi_enqueue_rgba8(rm_multi *m, uint8_t *host_image) {
    if (!m || !host_image) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_enqueue_rgba8: null argument");
    if (m->in_flight) return rm_internal_fail(RM_ERR_STATE, "rm_multi_enqueue_rgba8: a frame is in flight (rm_multi_wait first)");
    if (int rc = render_frame(m, nullptr)) return rc;
    hipStream_t stream0 = static_cast<hipStream_t>(rm_stream(m->ranks[0].ctx));
    RM_MHIP(hipMemcpyAsync(host_image, m->frame, (size_t)m->params.width * m->params.height * 4u, hipMemcpyDeviceToHost, stream0));
    RM_MHIP(hipEventRecord(m->gathered, stream0));
    m->in_flight = true;
    return RM_OK;
}

int rm_multi_wait(rm_multi *m) {
    if (!m) return rm_internal_fail(RM_ERR_INVALID, "rm_multi_wait: null argument");
    if (!m->in_flight) return RM_OK;
    m->in_flight = false;
    return finish_frame(m);
}

int rm_multi_render_device(rm_multi *m, uint8_t *device_rgba8) {
    if (int rc = rm_multi_enqueue_device(m, device_rgba8)) return rc;
    return rm_multi_wait(m);
}

int rm_multi_render_rgba8(rm_multi *m, uint8_t *host_image) {
    if (int rc = rm_multi_enqueue_rgba8(m, host_image)) return rc;
    return rm_multi_wait(m);
}

rm_context *rm_multi_context(rm_multi *m, uint32_t rank) { return (m && rank < m->ranks.size()) ? m->ranks[rank].ctx : nullptr; }
uint32_t rm_multi_ranks(const rm_multi *m) { return m ? (uint32_t)m->ranks.size() : 0u; }
int rm_multi_transport_in_use(const rm_multi *m) { return m ? m->transport : RM_MULTI_AUTO; }

}  // extern "C"
