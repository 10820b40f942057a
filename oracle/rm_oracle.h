/*
 * rm_oracle.h -- CPU restatement of the reference renderer's CPU path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the timed
 * CPU baseline.  Nothing in the product library (raymarching_amd/) may include,
 * link or call it; only tests/, the smoke check and the benchmark's
 * cpu_baseline leg do.
 *
 * Parity status: the reference cannot be built in this tree (its vector-math and
 * image submodules, msu-graphics-group/LiteMath and nothings/stb, are empty
 * directories with no pinned revision), so per-operation parity is UNPINNED;
 * the end-to-end pin is the reference's committed render out_cpu.png
 * (tests/golden/out_cpu.png), which tests/test_oracle_golden.py compares
 * against.  LiteMath's published definitions are restated in rm_oracle.c.
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Body::Type, /root/reference/source/include/body.h:10-15 */
enum { RMO_LIST = 0, RMO_SPHERE = 1, RMO_BOX = 2, RMO_CROSS = 3 };
/* Body::Mode, /root/reference/source/include/body.h:17-22 */
enum { RMO_UNION = 0, RMO_COMPLEMENT = 1, RMO_INTERSECTION = 2, RMO_DIFFERENCE = 3 };

typedef struct rmo_scene rmo_scene;

/* Run-time replacement of /root/reference/source/include/constants.h:11-27. */
typedef struct rmo_params {
    uint32_t width;        /* constants::width            (1024)  */
    uint32_t height;       /* constants::height           (768)   */
    uint32_t kernel;       /* constants::SSAA::kernel     (3)     */
    int32_t  iterations;   /* constants::iterations       (1000)  */
    float    saturation;   /* constants::saturation       (0.05f) */
    float    surface_eps;  /* constants::precision::surface (1e-3f) */
    float    offset_eps;   /* constants::precision::offset  (1e-3f) */
    int32_t  normalize_div;/* 0: a*(1/len) (LiteMath), 1: a/len          */
    uint32_t variant;      /* 0; tests only: bit 0 cross = exact median, bit 1 lerp = (1-t)u + tv, bit 2 fused dot products */
} rmo_params;

typedef struct rmo_counters {
    uint64_t sdf_calls;    /* scene::SDF invocations                     */
    uint64_t prim_evals;   /* Sphere/Box/Cross::SDF invocations          */
    uint64_t march_steps;  /* iterations of the loop in scene::surface   */
    uint64_t rays;         /* primary + shadow rays                      */
} rmo_counters;

/* Flat record of one primitive in depth-first (evaluation) order. */
typedef struct rmo_flat_prim {
    uint32_t type;      /* RMO_SPHERE / RMO_BOX / RMO_CROSS            */
    uint32_t list;      /* index of the owning list in DFS order, root = 0 */
    float    center[3];
    float    size[3];   /* sphere: radius in size[0]                   */
    float    color[3];
} rmo_flat_prim;

/* One tree node in depth-first pre-order (root list first; each list is
 * followed by its child_count children). */
typedef struct rmo_node {
    uint32_t type, mode, child_count;
    float    center[3];
    float    size[3];   /* sphere: radius in size[0] */
    float    color[3];
} rmo_node;

void        rmo_default_params(rmo_params *p);

/* Builds a scene from an explicit node list (any nesting, any list mode). */
rmo_scene  *rmo_scene_from_nodes(const rmo_node *nodes, uint32_t n_nodes,
                                 const float *lights_xyz, uint32_t n_lights,
                                 const float cam_position[3], const float cam_direction[3],
                                 const float cam_up[3], float cam_fov);

rmo_scene  *rmo_scene_load(const char *path);        /* scene::load      */
rmo_scene  *rmo_scene_parse(const char *text);       /* same, from memory */
void        rmo_scene_free(rmo_scene *s);

uint32_t    rmo_scene_prim_count(const rmo_scene *s);
uint32_t    rmo_scene_list_count(const rmo_scene *s);
uint32_t    rmo_scene_light_count(const rmo_scene *s);
void        rmo_scene_flat_prims(const rmo_scene *s, rmo_flat_prim *out);
void        rmo_scene_list_modes(const rmo_scene *s, uint32_t *modes, int32_t *parents);
void        rmo_scene_lights(const rmo_scene *s, float *xyz);
/* 16 floats column-major + focal; Object::Camera::update */
void        rmo_scene_camera(const rmo_scene *s, float *transform16, float *focal);

/* scene::SDF at n points (xyz interleaved): writes SD and the winning colour. */
void        rmo_sdf(const rmo_scene *s, const float *xyz, uint32_t n,
                    float *sd_out, float *rgb_out);

/*
 * render::CPU / render::OMP over the rows row0, row0+row_step, ... < row1.
 * out is a full width*height*4 float framebuffer (row 0 = top); only the
 * selected rows are written.  threads <= 1 runs the serial loop.
 */
void        rmo_render(const rmo_scene *s, const rmo_params *p,
                       uint32_t row0, uint32_t row1, uint32_t row_step,
                       int threads, float *out, rmo_counters *counters);

/* render::pixel at n arbitrary pixels, xy = (x, y) pairs, dealt to `threads` OpenMP threads;
 * out = n x RGBA floats in list order. */
void        rmo_render_pixels(const rmo_scene *s, const rmo_params *p, const uint32_t *xy,
                              uint32_t n, int threads, float *out);

/* One pixel's k*k sample colours in loop order (i outer, j inner): rgb*k*k. */
void        rmo_pixel_samples(const rmo_scene *s, const rmo_params *p,
                              uint32_t px, uint32_t py, float *rgb_out);

/* float RGBA -> RGBA8 as the reference's saved PNG: round(clamp(c,0,1)*255). */
void        rmo_quantize(const float *rgba, uint64_t n_pixels, uint8_t *out);

int         rmo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
