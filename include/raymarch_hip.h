/*
 * raymarch_hip.h -- C ABI of the MI355X (gfx950) sphere-tracing renderer.
 *
 * This is the boundary that replaces the reference's OpenGL compute dispatch
 * (glad/GLFW context, shader compile, SSBO upload, glDispatchCompute,
 * glGetTexImage).  Plain pointers and sizes only.  Every entry point returns
 * RM_OK (0) or a negative rm_status; rm_last_error() gives the message of the
 * calling thread's last failure.  The reference's operators are all void and
 * print "[Error] ..." (render.cpp:417-425); the C++ facade in
 * raymarching_amd/csrc/render.h keeps that behaviour on top of these codes.
 *
 * A context is not re-entrant: calls on one context are serialised by the
 * caller (the reference keeps all of this state in namespace-scope globals,
 * render.cpp:32-46).  Different contexts may be used from different threads.
 */
#ifndef RAYMARCH_HIP_H
#define RAYMARCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rm_status {
    RM_OK = 0,
    RM_ERR_INVALID = -1,   /* bad argument / malformed scene description   */
    RM_ERR_HIP = -2,       /* a HIP runtime call failed (message has which) */
    RM_ERR_STATE = -3,     /* call order: no scene / no params set yet      */
    RM_ERR_IO = -4         /* file could not be read / written              */
} rm_status;

/* Body::Type, /root/reference/source/include/body.h:10-15 */
typedef enum rm_body_type { RM_LIST = 0, RM_SPHERE = 1, RM_BOX = 2, RM_CROSS = 3 } rm_body_type;
/* Body::Mode, /root/reference/source/include/body.h:17-22 */
typedef enum rm_list_mode { RM_UNION = 0, RM_COMPLEMENT = 1, RM_INTERSECTION = 2, RM_DIFFERENCE = 3 } rm_list_mode;

/*
 * One node of the CSG tree.  The scene is the depth-first pre-order listing of
 * scene::tree: the root list first, every list followed by its child_count
 * children (recursively).  Replaces the fixed-stride tables built by
 * render::shader::genscene / packbody (render.cpp:246-274, 294-366):
 * no 1024-bodies / 1024-nodes / 64-lists limits.
 */
typedef struct rm_node {
    uint32_t type;         /* rm_body_type                                     */
    uint32_t mode;         /* lists: rm_list_mode                              */
    uint32_t child_count;  /* lists: number of direct children                 */
    float    center[3];    /* primitives: position                             */
    float    size[3];      /* box, cross: full edge lengths; sphere: size[0] = radius */
    float    color[3];     /* primitives: colour                               */
} rm_node;

/*
 * Render constants: the uniforms of render::pushuniforms (render.cpp:368-391),
 * i.e. /root/reference/source/include/constants.h:11-27 made run-time values.
 */
typedef struct rm_params {
    uint32_t width;        /* constants::width            */
    uint32_t height;       /* constants::height           */
    uint32_t kernel;       /* constants::SSAA::kernel     */
    int32_t  iterations;   /* constants::iterations       */
    float    saturation;   /* constants::saturation       */
    float    surface_eps;  /* constants::precision::surface */
    float    offset_eps;   /* constants::precision::offset  */
} rm_params;

/*
 * Row bands of the image rendered by one call: rows are grouped in bands of
 * band_rows; the call renders bands first_band, first_band + band_stride, ...
 * (band_count of them) into a compact buffer of band_count * band_rows rows.
 * {height, 0, 1, 1} is the whole frame.  Used to shard one frame over GPUs.
 * With a band list set (rm_set_band_list) band b of the call is image band list[b] instead, and
 * band_count must equal the list's length; first_band and band_stride are then ignored.
 */
typedef struct rm_rows {
    uint32_t band_rows;
    uint32_t first_band;
    uint32_t band_stride;
    uint32_t band_count;
} rm_rows;

/* Work counters of the last render (filled only when counting is enabled). */
typedef struct rm_counters {
    uint64_t sdf_calls;    /* scene SDF evaluations (march steps + 6 per gradient) */
    uint64_t march_steps;  /* iterations of the sphere-tracing loop                 */
    uint64_t rays;         /* primary + shadow rays                                 */
    uint64_t prims;        /* primitives in the uploaded scene                      */
    uint64_t prim_evals;   /* primitive distance evaluations actually executed      */
    uint64_t lattice_prims;/* primitives of the scene covered by recognised lattices */
    uint64_t flops_per_sdf;/* algorithmic FLOPs of one exhaustive scene SDF call:
                              11 per sphere, 9 per box, 15 per cross                  */
    double   kernel_ms;    /* device time of the trace kernel(s), HIP events        */
    double   resolve_ms;   /* device time of the resolve kernel (0 where the trace kernel writes
                              the pixels itself: compiled kernels, SSAA kernel 1, 2, 4 or 8) */
    double   compile_ms;   /* time rm_set_scene spent compiling the scene's kernels
                              (0: interpreted mode, or code object found in the cache) */
    uint32_t compiled_kernel; /* 1 if the last render ran the scene's own kernel      */
    uint32_t tiles_by_cost; /* 1 if the last render launched its tiles in cost order             */
    uint64_t march_slots;  /* march-loop iterations x samples of the lanes that sat through them,
                              live or idle: march_steps / march_slots = lane utilisation of
                              the sphere-tracing loop                                   */
    uint64_t flops_executed; /* algorithmic FLOPs of the primitive evaluations the kernel executed
                              (11 per sphere, 9 per box, 15 per cross; evaluations skipped by a
                              bound or a lattice are not counted), march steps excluded  */
    uint64_t tile_parts;   /* workgroups of the last render beyond one per tile: in cost order the few
                              tiles that run longest are launched as 4 or 16 parts           */
    uint64_t probe[16];    /* diagnostic (counting renders of compiled kernels): lane occupancy of code regions as
                              pairs (lanes that executed region r x weight, 64 x weight per wave that went through it):
                              r = 0..3 lattice level 0, 1, 2, deeper (weight 1 per visit); 4 lattice row cells (weight =
                              cells of the row); 5 gradient taps (lanes whose sample is shaded); 6: not a pair -- hull tests and rows folded by
                              clustered runs, per wave; 7 scene SDF calls of the
                              march loop */
    uint64_t flops_shared_first; /* the part of flops_executed spent on the camera rays' first evaluation -- the same
                              for every ray, found once at rm_set_scene -- which renders that do not count skip */
} rm_counters;

typedef struct rm_context rm_context;

/* Replaces render::setup::context / shaders / buffers (render.cpp:408-445):
 * selects the HIP device, creates the stream, checks the kernels are loadable. */
int rm_create(int device, rm_context **out);

/* Replaces render::destroy (render.cpp:477-479). */
int rm_destroy(rm_context *ctx);

/* Replaces render::pushuniforms (render.cpp:368-391). */
int rm_set_params(rm_context *ctx, const rm_params *params);

/* Replaces render::setup::buffers (render.cpp:439-445): sizes the device buffers whole-frame renders of the current
 * parameters need (frame, SSAA samples, tile tables), so that the first frame does not pay for them, and -- with a scene
 * set -- reads the whole frame's tile order if an earlier process left one in the cache directory (rm_set_scene does
 * the same once the parameters are known, so either order of the two calls serves).  Optional: every render sizes and
 * loads what it lacks.  Not while a frame is queued (rm_enqueue_*): the buffers are cleared. */
int rm_reserve(rm_context *ctx);

/* Replaces render::push's scene part: genscene + genlights + 3x pushssbo and the
 * camera uniforms (render.cpp:393-404, 447-466).  transform is the camera's
 * 4x4 matrix, column-major (packmatrix, render.cpp:281-286). */
int rm_set_scene(rm_context *ctx, const rm_node *nodes, uint32_t n_nodes,
                 const float *lights_xyz, uint32_t n_lights,
                 const float transform[16], float focal);

/* Replaces render::GPU (render.cpp:468-475): renders the whole frame and
 * copies it into the caller's host buffer, width*height*4 bytes, row 0 = top,
 * RGBA8 = round(clamp(c,0,1)*255), alpha 255.  An image in page-locked memory
 * (rm_host_alloc, rm_host_register) arrives faster; where a resolve pass follows
 * the trace kernel it is written by that pass itself instead of copied. */
int rm_render_rgba8(rm_context *ctx, uint8_t *host_image);

/* Same frame as float RGBA (the CPU operators' Image2D<float4>, render.cpp:119),
 * width*height*4 floats. */
int rm_render_f32(rm_context *ctx, float *host_image);

/*
 * render::GPU's read-back (glGetTexImage, render.cpp:474) without the wait: queues the kernels of the given row bands
 * and, behind them on the context's stream, the copy of the finished rows into the caller's HOST buffers (either may
 * be NULL; rows * width * 4 floats / bytes), and returns; rm_wait completes the frame.  With page-locked host memory
 * -- from rm_host_alloc, or the caller's own buffer passed through rm_host_register once -- the copy runs on the copy
 * engine while other contexts' kernels use the GPU, so frames in flight on several contexts deliver host images at
 * nearly the device-resident rate; with ordinary pageable memory the copy is staged by the runtime and slower.
 * rm_render_rgba8 / rm_render_f32 are this plus rm_wait for the whole frame (and time the tiles of a new configuration).
 */
int rm_enqueue_host(rm_context *ctx, const rm_rows *rows, float *host_f32, uint8_t *host_rgba8);
/* Page-locked host memory for the frames (hipHostMalloc / hipHostFree behind a C signature, for callers without HIP
 * headers), and page-locking of a buffer the caller owns (hipHostRegister / hipHostUnregister; unregister before freeing). */
int rm_host_alloc(size_t bytes, void **out);
int rm_host_free(void *ptr);
int rm_host_register(void *ptr, size_t bytes);
int rm_host_unregister(void *ptr);

/* Renders the given row bands into caller-owned DEVICE buffers (either may be
 * NULL): device_f32 holds rows*width*4 floats, device_rgba8 rows*width*4 bytes,
 * rows = band_count*band_rows.  Work is queued on the context's stream; the
 * call returns after the stream has been synchronised. */
int rm_render_device(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8);

/* Deals arbitrary image bands to this context (e.g. by measured cost, see rm_get_tile_costs): the
 * device-buffer renders that follow draw band image_bands[b] into rows b * band_rows .. of their
 * buffers.  count = 0 returns to the arithmetic bands of rm_rows.  Bands beyond the image are left
 * black.  No counterpart in the reference (one GPU). */
int rm_set_band_list(rm_context *ctx, const uint32_t *image_bands, uint32_t count);

/* Where the device-buffer renders (rm_render_device, rm_enqueue_device, rm_enqueue_device_static) store their rows.
 *   RM_ROWS_COMPACT   (default) a compact image of the call's bands, as described at rm_rows;
 *   RM_ROWS_IN_FRAME  the buffers are whole frames (height * width * 4 floats / bytes) and every rendered row is stored
 *                     at its own place in them; nothing else is written, rows of bands below the image not at all.
 *                     Calls with different bands -- on contexts of different GPUs, into memory all of them can write --
 *                     then fill one frame between them, with no gathering copy afterwards.
 * The host-image calls are not affected.  No counterpart in the reference (one GPU). */
typedef enum rm_row_placement { RM_ROWS_COMPACT = 0, RM_ROWS_IN_FRAME = 1 } rm_row_placement;
int rm_set_row_placement(rm_context *ctx, int placement);

/*
 * The same without waiting (the reference's render::GPU blocks in glGetTexImage, render.cpp:474;
 * these exist so that a caller can keep several frames in flight, e.g. two contexts on two
 * streams, or overlap the render with a collective that gathers the previous frame).
 *   rm_enqueue_device  queues trace + resolve on the context's stream and returns.  A context
 *                      renders one frame at a time through one scratch buffer: frames queued on
 *                      it run in order.  Tile costs are not measured by queued frames; they use
 *                      the order a synchronous render of the same configuration left behind.
 *   rm_wait            waits for the stream and books the last queued frame (rm_get_counters).
 *   rm_set_stream      makes the context work on the caller's hipStream_t (NULL: back to its own),
 *                      so that its kernels are ordered with the caller's other work on it.
 */
int rm_enqueue_device(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8);
int rm_wait(rm_context *ctx);
int rm_set_stream(rm_context *ctx, void *hip_stream);

/*
 * rm_enqueue_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The call itself allocates nothing
 * (no hipMalloc / hipFree, no allocating library routine): what the kernels need beyond the caller's buffers --
 * the SSAA sample image that the trace kernel leaves for the resolve kernel, 8 bytes per sample, wherever the
 * trace kernel does not write the pixels itself -- lives in a static device array of RM_STATIC_SCRATCH_BYTES
 * (plus 16 MiB behind it: one word per tile, up to 4 194 304 tiles, through which a scene's own kernel asks for
 * tiles to be rendered again by its exact kernel; renders of more tiles are refused)
 * that every GPU the library runs on holds from the first rm_create on it until the process ends.
 *   - A render whose sample image (band_count * band_rows * kernel rows of width * kernel samples) exceeds
 *     the array is refused with RM_ERR_INVALID (render fewer rows per call) -- unless the scene has kernels of
 *     its own (RM_KERNEL_COMPILED) and the SSAA kernel is 3, 6, 12 or 24: the trace kernel then runs workgroups
 *     of 3 x 3 waves that hold whole pixels and write them themselves, slower but without a sample image.
 *   - Tiles run in the order an earlier rm_render_device of the same configuration measured, if this context
 *     holds one, in row-major order otherwise; this call never times tiles.  With work counting on it is
 *     refused (RM_ERR_STATE).
 *   - All contexts on a GPU share that GPU's array: calls that overlap on one GPU (two contexts on two
 *     streams) race; queue them on one stream.  Nothing in it needs to start at zero: every call overwrites
 *     the part it reads.
 * Pair with rm_wait like rm_enqueue_device.  Same frame as rm_render_device, bit for bit.
 */
#define RM_STATIC_SCRATCH_BYTES (64u << 20)
int rm_enqueue_device_static(rm_context *ctx, const rm_rows *rows, float *device_f32, uint8_t *device_rgba8);
size_t rm_static_scratch_bytes(void);

/* Scene SDF at n host points (xyz interleaved): signed distance and the index
 * (depth-first primitive order) of the primitive whose colour wins.  Exposes
 * the device evaluator of Body::List::SDF (body.cpp:66-111) for parity tests. */
int rm_eval_sdf(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim);

/* The same, plus the value of the distance-only evaluator that the march loop, the gradient taps
 * and the shadow rays use (it skips primitives that provably cannot change the result: hulls of
 * long runs, rejected spheres, bounded nested lists, lattice crosses out of range).  The two
 * distances must be equal bit for bit; parity tests check that at arbitrary points. */
int rm_eval_sdf_both(rm_context *ctx, const float *host_xyz, uint32_t n, float *host_sd, uint32_t *host_prim,
                     float *host_sd_march);

/*
 * How runs of primitives are evaluated (no counterpart in the reference, whose two back-ends
 * always visit every primitive).  Both give bit-identical frames.
 *   RM_EVAL_EXHAUSTIVE  every primitive of the scene is evaluated at every SDF call;
 *   RM_EVAL_LATTICE     (default) lists of cubic crosses that form a complete Menger lattice,
 *                       as Body::MengerSponge generates (body.cpp:114-156), are recognised on
 *                       upload and evaluated exactly by visiting only the crosses that can win;
 *                       other long runs of primitives get bounding hulls (spatially sorted
 *                       groups of 8 and 64) that rule out groups which cannot win.
 * Takes effect at the next rm_set_scene.
 */
typedef enum rm_evaluator { RM_EVAL_EXHAUSTIVE = 0, RM_EVAL_LATTICE = 1 } rm_evaluator;
int rm_set_evaluator(rm_context *ctx, int evaluator);

/*
 * Which kernels render the scene.  The reference compiles its compute shader when the program
 * starts (render::setup::shaders, render.cpp:428-437) and reads the scene from buffers; here
 *   RM_KERNEL_COMPILED     (default) rm_set_scene generates HIP source for the uploaded scene --
 *                          every primitive a line of straight-line code with its numbers as
 *                          literals -- compiles it for gfx950 with hiprtc and renders with that
 *                          kernel; code objects are cached per process by source text.
 *                          rm_set_scene fails if the compile fails (no silent fallback);
 *   RM_KERNEL_INTERPRETED  one ahead-of-time kernel walks the scene's record stream.  The mode
 *                          for geometry that changes every frame (camera, lights and parameters
 *                          are data in both modes and never cause a compile).
 * Both give bit-identical frames.  With work counting enabled each mode runs its own counting
 * variant (compiled lazily at the first counted render), so prim_evals and flops_executed are
 * those of the mode: the compiled kernels skip work the interpreted one does.  Takes effect
 * at the next rm_set_scene.
 */
typedef enum rm_kernel_mode { RM_KERNEL_INTERPRETED = 0, RM_KERNEL_COMPILED = 1 } rm_kernel_mode;
int rm_set_kernel_mode(rm_context *ctx, int mode);

/*
 * Order in which the trace kernel's tiles (one workgroup each: 8x8 or 16x16 samples) are started (no counterpart in the
 * reference; the frame is the same either way).
 *   RM_TILES_BY_COST  (default) the first render of a configuration -- scene upload, parameters,
 *                     row bands -- times every tile; later renders of the same configuration
 *                     start the expensive tiles first, which hides the long-running tiles of
 *                     dense regions behind the cheap ones (the cost of one host-side sort after
 *                     that first render);
 *   RM_TILES_NATURAL  row-major order, nothing measured.
 */
typedef enum rm_tile_scheduling { RM_TILES_NATURAL = 0, RM_TILES_BY_COST = 1 } rm_tile_scheduling;
int rm_set_tile_scheduling(rm_context *ctx, int mode);
/* The durations the last tile-timing render measured, in 10 ns ticks, row-major over the tile grid
 * of that render's sample image (tiles_x * tiles_y entries; ticks may be NULL to ask for the shape). */
int rm_get_tile_costs(rm_context *ctx, uint32_t *ticks, size_t capacity, uint32_t *tiles_x, uint32_t *tiles_y);

/*
 * How much of the shading work after the camera ray -- the six gradient taps and a shadow ray per
 * light (scene::raymarch, scene.cpp:26-32) -- is done where it cannot change the frame.  No counterpart
 * in the reference; the frame is the same either way.
 *   RM_SHADE_COLOURED  (default) a sample whose surface colour is (0, 0, 0) -- the wall of `Bounds`
 *                      behind the scene, 62 % of the shipped frame -- skips all of it: scene::raymarch
 *                      returns lighting * colour with the lighting clamped into [saturation, 1]
 *                      (scene.cpp:60), which is zero for such a sample whatever the rays give
 *                      (applies while saturation >= 0).  And a shadow ray stops marching once it is
 *                      safely past its light: scene::shadow (scene.cpp:45-50) only asks whether
 *                      the ray ended before the light, and from there on it cannot come back;
 *   RM_SHADE_ALL       every sample and every ray does the reference's full work: the setting under
 *                      which the work counters (sdf_calls, march_steps, rays) equal the reference's
 *                      own counts.
 */
typedef enum rm_shading { RM_SHADE_ALL = 0, RM_SHADE_COLOURED = 1 } rm_shading;
int rm_set_shading(rm_context *ctx, int mode);

/*
 * Lane refill (no counterpart in the reference, whose compute shader runs a pixel's samples and rays one after the
 * other in one invocation, shader.comp:353-365; the frame is the same either way).  Off by default.  When on, the
 * scene's own kernels (RM_KERNEL_COMPILED; takes effect at the next rm_set_scene) give every wave a tile of
 * 2^tile_w_log2 x 2^tile_h_log2 samples (3..8 each), and a lane whose ray has ended goes on at once -- with its
 * sample's gradient taps, its next shadow ray, or the tile's next sample -- instead of waiting for the slowest of
 * the wave's 64 rays; surface colours are found 64 at a time from a queue in LDS.  Measured on MI355X this keeps
 * 95 % of the march loop's lanes busy (88 % without) and is slower all the same, 0.92-0.99 ms against 0.58 ms per
 * shipped frame: lanes of one wave at different stages of their samples share less of the evaluator's wave-wide
 * culling, and every turn of the loop runs some lane's change of stage with most lanes masked off.  Kept for study
 * and as the measured answer; SSAA samples go through the resolve kernel in this mode.  0, 0 turns it off.
 */
int rm_set_lane_refill(rm_context *ctx, uint32_t tile_w_log2, uint32_t tile_h_log2);

/*
 * The compiled mode's two steps without a device or a context (build checks, inspection):
 *   rm_kernel_source           writes the HIP source rm_set_scene would generate for the scene into
 *                              `buffer` (NUL-terminated, truncated to `capacity`) and its full length,
 *                              without the NUL, into *length; buffer may be NULL to ask for the length;
 *   rm_kernel_source_compiles  RM_OK if the source compiles for gfx950 with hiprtc; otherwise the
 *                              compiler's log is in rm_last_error().
 */
int rm_kernel_source(const rm_node *nodes, uint32_t n_nodes, int evaluator, int counting, char *buffer, size_t capacity,
                     size_t *length);
int rm_kernel_source_compiles(const char *source);
/* Starts compiling the kernels rm_set_scene will want for this scene (default settings: RM_KERNEL_COMPILED, compact rows)
 * and returns at once; needs no context.  A later rm_set_scene of the same nodes with the same evaluator waits for these
 * compiles instead of starting its own (a scene that does not compile is reported there, not here).  For callers that know the scene before they have a context: called ahead of
 * rm_create, the compile (0.65 s for the shipped scene) runs while the device is initialised (0.12 s).  Does nothing where the
 * compiler runs inside the calling process (see rm_kernel_compiler).  No counterpart in the reference. */
int rm_precompile_scene(const rm_node *nodes, uint32_t n_nodes, int evaluator);

/* How this process last obtained the code object of a scene's kernels (a text owned by the library, valid until the calling
 * thread's next call of this function): "rm_jit, a process of its own: <path>" -- the library compiles with the program
 * rm_jit that is installed beside it, a process of its own linked against the ROCm release the library was built with, so
 * that the kernels do not depend on which other copy of hiprtc the calling process has loaded (PyTorch bundles an older
 * one under the same SONAME; its kernels measure 3-6 % slower); "hiprtc in this process" where that program is missing
 * or cannot run ($RM_JIT_HELPER names another program; "" or "off" keeps the compiler in the process); "none: code object
 * from the cache directory"; "none yet".  No counterpart in the reference, whose driver compiles its shader itself. */
const char *rm_kernel_compiler(void);
/* Process-wide: where compiled code objects and measured tile orders are kept so that later processes load them
 * instead of compiling (0.3-1.3 s per scene) or timing the tiles again.  ON BY DEFAULT: without this call the library
 * uses $RM_KERNEL_CACHE_DIR, else $XDG_CACHE_HOME/raymarch_hip, $HOME/.cache/raymarch_hip, /tmp/raymarch_hip-<uid>
 * (created 0700, used only if owned by the caller); NULL, "" here, or RM_KERNEL_CACHE_DIR=off in the environment,
 * turn it off.  `directory` must exist and be writable.  A file is used only if the key stored in it (source and
 * device headers; scene digest, parameters and launch shape for a tile order) equals the current one.  A tile order
 * is written once a second frame of its configuration has used it, never by a configuration rendered once; the
 * directory is kept below $RM_KERNEL_CACHE_MAX_MB (default 256) by deleting the files unused for longest. */
int rm_set_kernel_cache_dir(const char *directory);

/*
 * Diagnostic (no counterpart in the reference; the frame is the same): with slot statistics on, every workgroup of a
 * render leaves four words -- its tile-order entry (tile number in the low 27 bits; above them 0 for a whole tile,
 * 1-4 a quarter, 5-20 a sixteenth of it), its duration in 10 ns ticks, the trip counts of its camera march (low 16
 * bits) and of its shadow marches (high 16 bits; the longest ray of its wave each time, capped at 65535), and the low
 * 32 bits of the 100 MHz clock at its start -- in launch order.  Tells which workgroups a frame waits
 * for at its end, and whether they were marching a camera ray or shadow rays.  words: 4 * capacity_slots uint32
 * (NULL to ask for *n_slots, the number of workgroups of the last render).  Not with rm_enqueue_device_static.
 */
int rm_set_slot_stats(rm_context *ctx, int enabled);
int rm_get_slot_stats(rm_context *ctx, uint32_t *words, size_t capacity_slots, size_t *n_slots);

/*
 * Diagnostic (no counterpart in the reference): the kernel generated for a scene with recognised lattices decides a
 * point's distance to a lattice from the crosses round the nearest centres and, where that is not enough (deep inside
 * material, where rays as good as never go), marks the point's tile; a second kernel of the scene, launched behind it
 * in every render, renders the marked tiles again with the full walk.  *tiles: how many tiles that second kernel has
 * rendered in this context since the last rm_set_scene (0 for frames that never needed it).  Waits for the context's
 * stream.
 */
int rm_get_exact_tiles(rm_context *ctx, uint64_t *tiles);

/* Enable (1) / disable (0) per-ray work counting in subsequent renders. */
int rm_set_counting(rm_context *ctx, int enabled);
int rm_get_counters(rm_context *ctx, rm_counters *out);

/* The context's HIP stream (hipStream_t) for callers that time with HIP events. */
void *rm_stream(rm_context *ctx);

const char *rm_last_error(void);

/* ---- host scene library (scene::load and the generators), C view ---------- */

typedef struct rm_scene rm_scene;

/* scene::load (scene.cpp:92-190): parses an objects.txt file into a tree.
 * A missing file gives an empty scene, as the reference does. */
int rm_scene_load(const char *path, rm_scene **out);
int rm_scene_parse(const char *text, rm_scene **out);
void rm_scene_free(rm_scene *scene);

uint32_t rm_scene_node_count(const rm_scene *scene);
uint32_t rm_scene_light_count(const rm_scene *scene);
/* Copies the pre-order node list / light positions / camera of the scene. */
int rm_scene_nodes(const rm_scene *scene, rm_node *out, uint32_t capacity);
int rm_scene_lights(const rm_scene *scene, float *out_xyz, uint32_t capacity);
int rm_scene_camera(const rm_scene *scene, float transform[16], float *focal);

/* rm_set_scene from a loaded scene. */
int rm_set_scene_from(rm_context *ctx, const rm_scene *scene);

/* ---- one frame over several GPUs of a node, one process ---------------------------------------------------
 *
 * The reference's render::GPU (render.cpp:468-475) drives one device.  An rm_multi owns one rendering context
 * per entry of `devices` (its ranks), deals the frame's row bands to them cyclically (band b of RM_MULTI_BAND_ROWS
 * rows belongs to rank b mod n; every pixel is independent, render.cpp:122-129) and has the frame end up on rank 0's
 * device, from where it goes to the caller: either every rank stores its rows there itself, or every rank renders its
 * share into its own device memory and the shares are gathered -- the one exchange step of the path.  The frame
 * equals the frame one context renders, bit for bit.
 *   transport  RM_MULTI_RCCL   grouped ncclSend / ncclRecv of every rank's share to rank 0 over xGMI (librccl is
 *                              loaded when the first such object is created; one communicator per rank from
 *                              ncclCommInitAll), then a de-interleaving copy on rank 0's device;
 *              RM_MULTI_COPY   peer copies (hipMemcpy2DAsync between the devices' memories), which also serve a
 *                              device list that names a GPU more than once -- RCCL refuses two ranks on one
 *                              device -- so that the whole path can be exercised on a single GPU;
 *              RM_MULTI_DIRECT no exchange step: every rank's kernels store their rows at their place in the frame on
 *                              rank 0's device (rm_set_row_placement on every rank; stores into a peer's memory,
 *                              which need peer access from every device to the first) -- for rm_multi_render_device
 *                              straight into the caller's buffer, which must then be memory of rank 0's device
 *                              that the other devices can write (hipMalloc there); rank 0's stream only waits for
 *                              the ranks' kernels.  Serves repeated devices too;
 *              RM_MULTI_AUTO   direct stores where every device can access the first one's memory, else RCCL when all
 *                              devices differ, else copies.
 * Calls on one rm_multi are serialised by the caller, like calls on a context.  The first frame of a configuration
 * renders rank after rank (each times its tiles); later frames run on all ranks at once.
 */
typedef struct rm_multi rm_multi;
typedef enum rm_multi_transport { RM_MULTI_AUTO = 0, RM_MULTI_RCCL = 1, RM_MULTI_COPY = 2, RM_MULTI_DIRECT = 3 } rm_multi_transport;
#define RM_MULTI_BAND_ROWS 4u
int rm_multi_create(const int *devices, uint32_t n_devices, int transport, rm_multi **out);
int rm_multi_destroy(rm_multi *m);
/* rm_set_params / rm_set_scene / rm_set_scene_from on every rank (render::push, render.cpp:447-466). */
int rm_multi_set_params(rm_multi *m, const rm_params *params);
int rm_multi_set_scene(rm_multi *m, const rm_node *nodes, uint32_t n_nodes, const float *lights_xyz, uint32_t n_lights,
                       const float transform[16], float focal);
int rm_multi_set_scene_from(rm_multi *m, const rm_scene *scene);
/* render::GPU (render.cpp:468-475) on all ranks: the whole frame, RGBA8, into the caller's host image
 * (width * height * 4 bytes, row 0 = top), or into device memory on rank 0's device. */
int rm_multi_render_rgba8(rm_multi *m, uint8_t *host_image);
int rm_multi_render_device(rm_multi *m, uint8_t *device_rgba8);
/* The same without waiting (no counterpart in the reference, whose render::GPU blocks): the frame's kernels, exchange and
 * final copy are queued on the ranks' streams and the call returns; rm_multi_wait waits for them.  One frame at a time per
 * rm_multi: several rm_multi objects over the same devices keep several frames in flight (a rank's longest tiles then
 * overlap the next frame's start).  Each object's first frame of a configuration still renders synchronously, rank after
 * rank (it times the tiles).  host_image should be page-locked (rm_host_alloc) for the copy to overlap anything. */
int rm_multi_enqueue_rgba8(rm_multi *m, uint8_t *host_image);
int rm_multi_enqueue_device(rm_multi *m, uint8_t *device_rgba8);
int rm_multi_wait(rm_multi *m);
/* Rank r's context (settings such as rm_set_kernel_mode or rm_set_shading before the scene is set, counters after a
 * frame), the number of ranks, and the transport in use (RM_MULTI_RCCL, RM_MULTI_COPY or RM_MULTI_DIRECT). */
rm_context *rm_multi_context(rm_multi *m, uint32_t rank);
uint32_t rm_multi_ranks(const rm_multi *m);
int rm_multi_transport_in_use(const rm_multi *m);

/* Writes an RGBA8 image as a PNG file (stored/deflate-less zlib stream). */
int rm_write_png(const char *path, const uint8_t *rgba, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif
