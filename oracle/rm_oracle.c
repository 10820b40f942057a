/*
 * rm_oracle.c -- CPU restatement of the reference renderer's CPU path, plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see rm_oracle.h).  Build: oracle/Makefile
 * (-O2 -ffp-contract=off -fopenmp, baseline x86-64 SSE2: every fp32 operation is
 * rounded once, no fused multiply-add, IEEE sqrt and divide -- the arithmetic of
 * the reference's "-std=c++11 -fopenmp -O2" build, /root/reference/Makefile:14-20).
 *
 * Each function cites the reference lines it follows.  The vector helpers
 * restate the published definitions of msu-graphics-group/LiteMath (absent
 * submodule, no pinned revision):
 *   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z          (left to right)
 *   length(a)     = sqrt(dot(a,a))
 *   normalize(a)  = a * (1/length(a))
 *   lerp(u,v,t)   = u + t*(v-u)
 *   clamp(u,a,b)  = min(max(a,u),b)
 *   cross(a,b)    = (a.y*b.z-a.z*b.y, a.z*b.x-a.x*b.z, a.x*b.y-a.y*b.x)
 *   M*v           = v0*col0 + v1*col1 + v2*col2 + v3*col3 (left to right)
 *   min/max       = std::min/std::max (operator< only; ties keep the first)
 */
#include "rm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline f3 f3_divs(f3 a, float s) { return f3_make(a.x / s, a.y / s, a.z / s); }
/* Deliberately WRONG forms of three operations, selected by rmo_params.variant: the tests use them to show that
 * the reference's committed render tells the forms used here from these (a pin that every variant passed would pin
 * nothing).  Thread-local because the scene SDF has no parameter block; set per pixel from the parameters. */
enum { VARIANT_CROSS_MEDIAN = 1, VARIANT_LERP_WEIGHTED = 2, VARIANT_DOT_FUSED = 4 };
static _Thread_local uint32_t tl_variant;

static inline float f3_dot(f3 a, f3 b) {
    if (tl_variant & VARIANT_DOT_FUSED) return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));   /* what -ffp-contract=fast would make of it */
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
static inline float f3_length(f3 a) { return sqrtf(f3_dot(a, a)); }
static inline f3 f3_cross(f3 a, f3 b) {
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float f_lerp(float u, float v, float t) {
    if (tl_variant & VARIANT_LERP_WEIGHTED) return (1.0f - t) * u + t * v;
    return u + t * (v - u);
}

/* std::min / std::max on floats */
static inline float f_min(float a, float b) { return (b < a) ? b : a; }
static inline float f_max(float a, float b) { return (a < b) ? b : a; }

static inline f3 f3_normalize(f3 a, int use_div) {
    if (use_div) return f3_divs(a, f3_length(a));
    float inv = 1.0f / f3_length(a);
    return f3_scale(a, inv);
}

/* ------------------------------------------------------------------ tree */

typedef struct rmo_body {
    int    type;            /* RMO_LIST / SPHERE / BOX / CROSS            */
    int    mode;            /* lists only                                 */
    f3     position;
    f3     size;            /* box, cross: full edge lengths              */
    float  radius;          /* sphere                                     */
    f3     color;
    struct rmo_body **kids; /* lists only                                 */
    uint32_t nkids, cap;
} rmo_body;

/* Body::Surface, /root/reference/source/include/body.h:24-27 */
typedef struct { float SD; f3 color; } surface_t;

struct rmo_scene {
    rmo_body *tree;         /* scene::tree, root UNION list               */
    f3       *lights;       /* scene::lights (positions; colour is unused) */
    uint32_t  nlights, caplights;
    /* Object::Camera fields, /root/reference/source/include/object.h:27-41 */
    f3    cam_position, cam_direction, cam_up;
    float cam_fov;
    float transform[16];    /* column-major                               */
    float focal;
};

static rmo_body *body_new(int type) {
    rmo_body *b = (rmo_body *)calloc(1, sizeof(rmo_body));
    b->type = type;
    return b;
}

static void list_append(rmo_body *list, rmo_body *kid) {
    if (list->nkids == list->cap) {
        list->cap = list->cap ? list->cap * 2 : 8;
        list->kids = (rmo_body **)realloc(list->kids, list->cap * sizeof(rmo_body *));
    }
    list->kids[list->nkids++] = kid;
}

static void body_free(rmo_body *b) {
    if (!b) return;
    for (uint32_t i = 0; i < b->nkids; i++) body_free(b->kids[i]);
    free(b->kids);
    free(b);
}

static rmo_body *sphere_new(f3 pos, float radius, f3 color) {
    rmo_body *b = body_new(RMO_SPHERE);
    b->position = pos; b->radius = radius; b->color = color;
    return b;
}
static rmo_body *box_new(int type, f3 pos, f3 size, f3 color) {
    rmo_body *b = body_new(type);
    b->position = pos; b->size = size; b->color = color;
    return b;
}
static rmo_body *list_new(int mode) {
    rmo_body *b = body_new(RMO_LIST);
    b->mode = mode;
    return b;
}

/* generateMengerSponge, /root/reference/source/body.cpp:114-147.
 * The 20 sub-cube offsets in units of d, in the reference's call order:
 * 8 "front" (z = -d), 8 "back" (z = +d), 4 "middle" (z = 0). */
static const signed char k_menger_offsets[20][3] = {
    { 1, -1, -1 }, { 0, -1, -1 }, { -1, -1, -1 }, { 1, 1, -1 },
    { 0, 1, -1 }, { -1, 1, -1 }, { -1, 0, -1 }, { 1, 0, -1 },
    { 1, -1, 1 }, { 0, -1, 1 }, { -1, -1, 1 }, { 1, 1, 1 },
    { 0, 1, 1 }, { -1, 1, 1 }, { -1, 0, 1 }, { 1, 0, 1 },
    { -1, -1, 0 }, { 1, -1, 0 }, { -1, 1, 0 }, { 1, 1, 0 },
};

static void menger_generate(rmo_body *result, f3 position, float size, int iterations, f3 color) {
    float d = size / 3;
    list_append(result, box_new(RMO_CROSS, position, f3_make(d, d, d), color));
    if (iterations >= 2) {
        for (int n = 0; n < 20; n++) {
            /* float3(d,-d,-d) etc.: each component is +d, -d or literal 0 */
            f3 off = f3_make(k_menger_offsets[n][0] * d, k_menger_offsets[n][1] * d,
                             k_menger_offsets[n][2] * d);
            if (k_menger_offsets[n][0] == 0) off.x = 0.0f;
            if (k_menger_offsets[n][1] == 0) off.y = 0.0f;
            if (k_menger_offsets[n][2] == 0) off.z = 0.0f;
            menger_generate(result, f3_add(position, off), d, iterations - 1, color);
        }
    }
}

/* Body::MengerSponge, /root/reference/source/body.cpp:149-156 */
static rmo_body *menger_sponge(f3 position, float size, int iterations, f3 color) {
    rmo_body *result = list_new(RMO_DIFFERENCE);
    list_append(result, box_new(RMO_BOX, position, f3_make(size, size, size), color));
    menger_generate(result, position, size, iterations, color);
    return result;
}

/* Body::DeathStar, /root/reference/source/body.cpp:159-169 */
static rmo_body *death_star(f3 position, float radius, f3 color) {
    rmo_body *result = list_new(RMO_DIFFERENCE);
    list_append(result, sphere_new(position, radius, color));
    f3 diffposition = f3_make(position.x + 1.5f * radius, position.y, position.z);
    list_append(result, sphere_new(diffposition, radius, color));
    return result;
}

/* ------------------------------------------------------------------ SDF */

static __thread rmo_counters tl_counters;

/* Sphere::SDF body.cpp:32-35, Box::SDF :41-45, Cross::SDF :51-57 */
static inline float prim_sdf(const rmo_body *b, f3 p) {
    tl_counters.prim_evals++;
    if (b->type == RMO_SPHERE) {
        return f3_length(f3_sub(b->position, p)) - b->radius;
    }
    f3 d = f3_sub(p, b->position);
    d = f3_make(fabsf(d.x) - b->size.x / 2, fabsf(d.y) - b->size.y / 2, fabsf(d.z) - b->size.z / 2);
    if (b->type == RMO_BOX) {
        return f_max(f_max(d.x, d.y), d.z);
    }
    float dmin = f_min(f_min(d.x, d.y), d.z);
    float dmax = f_max(f_max(d.x, d.y), d.z);
    if (tl_variant & VARIANT_CROSS_MEDIAN)   /* the exact median, which the reference's rounded sum is not */
        return f_max(f_min(d.x, d.y), f_min(f_max(d.x, d.y), d.z));
    return d.x + d.y + d.z - dmin - dmax;
}

static surface_t body_sdf(const rmo_body *b, f3 p);

/* List::SDF, /root/reference/source/body.cpp:66-111 */
static surface_t list_sdf(const rmo_body *l, f3 p) {
    surface_t s;
    if (l->nkids == 0) {
        s.SD = INFINITY; s.color = f3_make(0, 0, 0);
        return s;
    }
    s = body_sdf(l->kids[0], p);
    if (l->mode == RMO_COMPLEMENT) s.SD = -s.SD;
    for (uint32_t i = 1; i < l->nkids; i++) {
        surface_t c = body_sdf(l->kids[i], p);
        switch (l->mode) {
        case RMO_UNION:        if (c.SD < s.SD) s = c; break;                 /* min(s, c)  */
        case RMO_COMPLEMENT:   c.SD = -c.SD; if (c.SD < s.SD) s = c; break;   /* min(s, -c) */
        case RMO_INTERSECTION: if (s.SD < c.SD) s = c; break;                 /* max(s, c)  */
        case RMO_DIFFERENCE:   c.SD = -c.SD; if (s.SD < c.SD) s = c; break;   /* max(s, -c) */
        default: break;
        }
    }
    return s;
}

static surface_t body_sdf(const rmo_body *b, f3 p) {
    if (b->type == RMO_LIST) return list_sdf(b, p);
    surface_t s;
    s.SD = prim_sdf(b, p);
    s.color = b->color;
    return s;
}

/* scene::SDF, /root/reference/source/scene.cpp:65-67 */
static inline surface_t scene_sdf(const rmo_scene *sc, f3 p) {
    tl_counters.sdf_calls++;
    return list_sdf(sc->tree, p);
}

/* scene::surface, /root/reference/source/scene.cpp:34-42 */
static surface_t scene_surface(const rmo_scene *sc, const rmo_params *pr, f3 *position, f3 ray) {
    surface_t s;
    s.SD = 0.0f; s.color = f3_make(0, 0, 0);
    for (int it = 0; it < pr->iterations; it++) {
        s = scene_sdf(sc, *position);
        tl_counters.march_steps++;
        *position = f3_add(*position, f3_scale(ray, s.SD));
        if (s.SD < pr->surface_eps) break;
    }
    return s;
}

/* scene::grad, /root/reference/source/scene.cpp:70-89 */
static f3 scene_grad(const rmo_scene *sc, f3 p) {
    const float h = 1e-3f;
    f3 dx = f3_make(h, 0.0f, 0.0f), dy = f3_make(0.0f, h, 0.0f), dz = f3_make(0.0f, 0.0f, h);
    float dfdx = scene_sdf(sc, f3_add(p, dx)).SD - scene_sdf(sc, f3_sub(p, dx)).SD;
    float dfdy = scene_sdf(sc, f3_add(p, dy)).SD - scene_sdf(sc, f3_sub(p, dy)).SD;
    float dfdz = scene_sdf(sc, f3_add(p, dz)).SD - scene_sdf(sc, f3_sub(p, dz)).SD;
    return f3_divs(f3_make(dfdx, dfdy, dfdz), 2 * h);
}

/* scene::shadow, /root/reference/source/scene.cpp:45-50 */
static int scene_shadow(const rmo_scene *sc, const rmo_params *pr, f3 light, f3 position, f3 normal) {
    f3 ray = f3_normalize(f3_sub(light, position), pr->normalize_div);
    position = f3_add(position, f3_scale(normal, pr->surface_eps + pr->offset_eps));
    tl_counters.rays++;
    scene_surface(sc, pr, &position, ray);
    return f3_dot(f3_sub(light, position), ray) > 0;
}

/* scene::lighting, /root/reference/source/scene.cpp:53-62 */
static float scene_lighting(const rmo_scene *sc, const rmo_params *pr, f3 position, f3 normal) {
    float lighting = 0.0f;
    for (uint32_t i = 0; i < sc->nlights; i++) {
        f3 light = sc->lights[i];
        if (!scene_shadow(sc, pr, light, position, normal))
            lighting += f3_dot(normal, f3_normalize(f3_sub(light, position), pr->normalize_div));
    }
    return f_min(f_max(pr->saturation, lighting), 1.0f);
}

/* scene::raymarch, /root/reference/source/scene.cpp:26-32 */
static f3 scene_raymarch(const rmo_scene *sc, const rmo_params *pr, f3 position, f3 ray) {
    tl_counters.rays++;
    surface_t s = scene_surface(sc, pr, &position, ray);
    f3 normal = f3_normalize(scene_grad(sc, position), pr->normalize_div);
    float light = scene_lighting(sc, pr, position, normal);
    return f3_scale(s.color, light);
}

/* -------------------------------------------------------------- camera */

static void mat_mul_vec4(const float *m, const float *v, float *r) {
    for (int i = 0; i < 4; i++)
        r[i] = v[0] * m[i] + v[1] * m[4 + i] + v[2] * m[8 + i] + v[3] * m[12 + i];
}

/* Object::Camera::view, /root/reference/source/object.cpp:38-42 */
static f3 camera_view(const rmo_scene *sc, f3 v, int offset) {
    float e[4] = { v.x, v.y, v.z, offset ? 1.0f : 0.0f }, r[4];
    mat_mul_vec4(sc->transform, e, r);
    return f3_make(r[0], r[1], r[2]);
}

/* Object::Camera::update, /root/reference/source/object.cpp:23-36.
 * (Camera geometry always used LiteMath's normalize, multiply-by-reciprocal.) */
static void camera_update(rmo_scene *sc) {
    f3 right = f3_normalize(f3_cross(sc->cam_direction, sc->cam_up), 0);
    f3 up = f3_normalize(f3_cross(right, sc->cam_direction), 0);
    f3 forward = f3_normalize(sc->cam_direction, 0);
    float *t = sc->transform;
    t[0] = right.x;  t[1] = right.y;  t[2] = right.z;  t[3] = 0.0f;
    t[4] = up.x;     t[5] = up.y;     t[6] = up.z;     t[7] = 0.0f;
    t[8] = -forward.x; t[9] = -forward.y; t[10] = -forward.z; t[11] = 0.0f;
    t[12] = sc->cam_position.x; t[13] = sc->cam_position.y; t[14] = sc->cam_position.z; t[15] = 1.0f;
    /* DEG_TO_RAD = float(pi)/180.0f; the unqualified tan() on a float argument
     * resolves to the double overload, its result is narrowed by the float multiply. */
    const float deg_to_rad = 3.14159265358979323846f / 180.0f;
    sc->focal = 2.0f * (float)tan((double)(sc->cam_fov * deg_to_rad / 2));
}

/* --------------------------------------------------------------- loader */

static int next_token(const char **cur, char *buf, size_t cap) {
    const char *c = *cur;
    while (*c && isspace((unsigned char)*c)) c++;
    if (!*c) { *cur = c; return 0; }
    size_t n = 0;
    while (*c && !isspace((unsigned char)*c)) { if (n + 1 < cap) buf[n++] = *c; c++; }
    buf[n] = 0;
    *cur = c;
    return 1;
}

static float next_float(const char **cur) {
    char tok[128];
    if (!next_token(cur, tok, sizeof tok)) return 0.0f;
    return strtof(tok, NULL);
}

static f3 next_f3(const char **cur) {
    f3 v;
    v.x = next_float(cur); v.y = next_float(cur); v.z = next_float(cur);
    return v;
}

static rmo_scene *scene_new(void) {
    rmo_scene *sc = (rmo_scene *)calloc(1, sizeof(rmo_scene));
    sc->tree = list_new(RMO_UNION);
    /* Camera defaults, /root/reference/source/include/object.h:35-38 */
    sc->cam_position = f3_make(0, 0, 0);
    sc->cam_direction = f3_make(0, 0, -1);
    sc->cam_up = f3_make(0, 1, 0);
    sc->cam_fov = 90.0f;
    return sc;
}

/* One line of the scene file; scene::load loop body, scene.cpp:100-186 */
static void scene_line(rmo_scene *sc, const char *line, f3 *color) {
    const char *cur = line;
    char cmd[64];
    if (!next_token(&cur, cmd, sizeof cmd)) return;
    rmo_body *obj = NULL;
    if (!strcmp(cmd, "Sphere")) {
        f3 pos = next_f3(&cur);
        float radius = next_float(&cur);
        obj = sphere_new(pos, radius, *color);
    } else if (!strcmp(cmd, "Box")) {
        f3 pos = next_f3(&cur);
        f3 size = next_f3(&cur);
        obj = box_new(RMO_BOX, pos, size, *color);
    } else if (!strcmp(cmd, "Bounds")) {
        float size = next_float(&cur);
        obj = list_new(RMO_COMPLEMENT);
        list_append(obj, box_new(RMO_BOX, f3_make(0, 0, 0), f3_make(size, size, size), f3_make(0, 0, 0)));
    } else if (!strcmp(cmd, "Cross")) {
        f3 pos = next_f3(&cur);
        f3 size = next_f3(&cur);
        obj = box_new(RMO_CROSS, pos, size, *color);
    } else if (!strcmp(cmd, "DeathStar")) {
        f3 pos = next_f3(&cur);
        float radius = next_float(&cur);
        obj = death_star(pos, radius, *color);
    } else if (!strcmp(cmd, "MengerSponge")) {
        f3 pos = next_f3(&cur);
        float size = next_float(&cur);
        char tok[64];
        int iterations = 0;
        if (next_token(&cur, tok, sizeof tok)) iterations = (int)strtol(tok, NULL, 10);
        obj = menger_sponge(pos, size, iterations, *color);
    }
    if (obj) { list_append(sc->tree, obj); return; }

    if (!strcmp(cmd, "Light")) {
        f3 pos = next_f3(&cur);
        if (sc->nlights == sc->caplights) {
            sc->caplights = sc->caplights ? sc->caplights * 2 : 4;
            sc->lights = (f3 *)realloc(sc->lights, sc->caplights * sizeof(f3));
        }
        sc->lights[sc->nlights++] = pos;
    } else if (!strcmp(cmd, "Camera")) {
        char sub[64];
        if (!next_token(&cur, sub, sizeof sub)) return;
        if (!strcmp(sub, "Position")) sc->cam_position = next_f3(&cur);
        else if (!strcmp(sub, "Direction")) sc->cam_direction = next_f3(&cur);
        else if (!strcmp(sub, "Up")) sc->cam_up = next_f3(&cur);
        else if (!strcmp(sub, "FOV")) sc->cam_fov = next_float(&cur);
    } else if (!strcmp(cmd, "Color")) {
        *color = next_f3(&cur);
    }
}

rmo_scene *rmo_scene_parse(const char *text) {
    rmo_scene *sc = scene_new();
    f3 color = f3_make(1.0f, 1.0f, 1.0f);  /* scene.cpp:99 */
    const char *c = text;
    while (*c) {
        const char *e = strchr(c, '\n');
        size_t n = e ? (size_t)(e - c) : strlen(c);
        char *line = (char *)malloc(n + 1);
        memcpy(line, c, n);
        line[n] = 0;
        scene_line(sc, line, &color);
        free(line);
        c += n;
        if (*c == '\n') c++;
    }
    camera_update(sc);  /* scene.cpp:189 */
    return sc;
}

/* scene::load, /root/reference/source/scene.cpp:92-190.  A missing file gives
 * an empty scene, as in the reference (the ifstream simply yields no lines). */
rmo_scene *rmo_scene_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return rmo_scene_parse("");
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = (char *)malloc((size_t)n + 1);
    size_t got = fread(text, 1, (size_t)n, f);
    text[got] = 0;
    fclose(f);
    rmo_scene *sc = rmo_scene_parse(text);
    free(text);
    return sc;
}

static rmo_body *node_build(const rmo_node *nodes, uint32_t n, uint32_t *cur) {
    if (*cur >= n) return NULL;
    const rmo_node *nd = &nodes[(*cur)++];
    f3 c = f3_make(nd->center[0], nd->center[1], nd->center[2]);
    f3 col = f3_make(nd->color[0], nd->color[1], nd->color[2]);
    if (nd->type == RMO_SPHERE) return sphere_new(c, nd->size[0], col);
    if (nd->type == RMO_BOX || nd->type == RMO_CROSS)
        return box_new((int)nd->type, c, f3_make(nd->size[0], nd->size[1], nd->size[2]), col);
    rmo_body *l = list_new((int)nd->mode);
    for (uint32_t i = 0; i < nd->child_count; i++) {
        rmo_body *k = node_build(nodes, n, cur);
        if (k) list_append(l, k);
    }
    return l;
}

rmo_scene *rmo_scene_from_nodes(const rmo_node *nodes, uint32_t n_nodes, const float *lights_xyz,
                                uint32_t n_lights, const float cam_position[3],
                                const float cam_direction[3], const float cam_up[3], float cam_fov) {
    rmo_scene *sc = scene_new();
    if (n_nodes && nodes[0].type == RMO_LIST) {
        uint32_t cur = 0;
        body_free(sc->tree);
        sc->tree = node_build(nodes, n_nodes, &cur);
    }
    for (uint32_t i = 0; i < n_lights; i++) {
        if (sc->nlights == sc->caplights) {
            sc->caplights = sc->caplights ? sc->caplights * 2 : 4;
            sc->lights = (f3 *)realloc(sc->lights, sc->caplights * sizeof(f3));
        }
        sc->lights[sc->nlights++] = f3_make(lights_xyz[3 * i], lights_xyz[3 * i + 1], lights_xyz[3 * i + 2]);
    }
    sc->cam_position = f3_make(cam_position[0], cam_position[1], cam_position[2]);
    sc->cam_direction = f3_make(cam_direction[0], cam_direction[1], cam_direction[2]);
    sc->cam_up = f3_make(cam_up[0], cam_up[1], cam_up[2]);
    sc->cam_fov = cam_fov;
    camera_update(sc);
    return sc;
}

void rmo_scene_free(rmo_scene *s) {
    if (!s) return;
    body_free(s->tree);
    free(s->lights);
    free(s);
}

/* ----------------------------------------------------------- inspection */

static void count_rec(const rmo_body *b, uint32_t *prims, uint32_t *lists) {
    if (b->type == RMO_LIST) {
        (*lists)++;
        for (uint32_t i = 0; i < b->nkids; i++) count_rec(b->kids[i], prims, lists);
    } else (*prims)++;
}

uint32_t rmo_scene_prim_count(const rmo_scene *s) {
    uint32_t p = 0, l = 0; count_rec(s->tree, &p, &l); return p;
}
uint32_t rmo_scene_list_count(const rmo_scene *s) {
    uint32_t p = 0, l = 0; count_rec(s->tree, &p, &l); return l;
}
uint32_t rmo_scene_light_count(const rmo_scene *s) { return s->nlights; }

static void flat_rec(const rmo_body *b, uint32_t list_id, uint32_t *next_list,
                     rmo_flat_prim *out, uint32_t *n) {
    for (uint32_t i = 0; i < b->nkids; i++) {
        const rmo_body *k = b->kids[i];
        if (k->type == RMO_LIST) {
            uint32_t id = (*next_list)++;
            flat_rec(k, id, next_list, out, n);
        } else {
            rmo_flat_prim *o = &out[(*n)++];
            o->type = (uint32_t)k->type;
            o->list = list_id;
            o->center[0] = k->position.x; o->center[1] = k->position.y; o->center[2] = k->position.z;
            if (k->type == RMO_SPHERE) { o->size[0] = k->radius; o->size[1] = 0; o->size[2] = 0; }
            else { o->size[0] = k->size.x; o->size[1] = k->size.y; o->size[2] = k->size.z; }
            o->color[0] = k->color.x; o->color[1] = k->color.y; o->color[2] = k->color.z;
        }
    }
}

void rmo_scene_flat_prims(const rmo_scene *s, rmo_flat_prim *out) {
    uint32_t next_list = 1, n = 0;
    flat_rec(s->tree, 0, &next_list, out, &n);
}

static void modes_rec(const rmo_body *b, int32_t parent, uint32_t *next, uint32_t *modes, int32_t *parents) {
    uint32_t id = (*next)++;
    modes[id] = (uint32_t)b->mode;
    parents[id] = parent;
    for (uint32_t i = 0; i < b->nkids; i++)
        if (b->kids[i]->type == RMO_LIST) modes_rec(b->kids[i], (int32_t)id, next, modes, parents);
}

void rmo_scene_list_modes(const rmo_scene *s, uint32_t *modes, int32_t *parents) {
    uint32_t next = 0;
    modes_rec(s->tree, -1, &next, modes, parents);
}

void rmo_scene_lights(const rmo_scene *s, float *xyz) {
    for (uint32_t i = 0; i < s->nlights; i++) {
        xyz[3 * i] = s->lights[i].x; xyz[3 * i + 1] = s->lights[i].y; xyz[3 * i + 2] = s->lights[i].z;
    }
}

void rmo_scene_camera(const rmo_scene *s, float *transform16, float *focal) {
    memcpy(transform16, s->transform, sizeof s->transform);
    *focal = s->focal;
}

void rmo_sdf(const rmo_scene *s, const float *xyz, uint32_t n, float *sd_out, float *rgb_out) {
    for (uint32_t i = 0; i < n; i++) {
        surface_t r = scene_sdf(s, f3_make(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        sd_out[i] = r.SD;
        if (rgb_out) { rgb_out[3 * i] = r.color.x; rgb_out[3 * i + 1] = r.color.y; rgb_out[3 * i + 2] = r.color.z; }
    }
}

/* --------------------------------------------------------------- render */

void rmo_default_params(rmo_params *p) {
    p->width = 1024; p->height = 768; p->kernel = 3; p->iterations = 1000;
    p->saturation = 0.05f; p->surface_eps = 1e-3f; p->offset_eps = 1e-3f;
    p->normalize_div = 0;
    p->variant = 0;
}

/* render::pixel, /root/reference/source/render.cpp:82-120.
 * samples (optional) receives the k*k per-sample colours in loop order. */
static void render_pixel(const rmo_scene *sc, const rmo_params *pr, int px, int py,
                         float *rgba_out, float *samples) {
    tl_variant = pr->variant;
    const float AR = (float)pr->width / (float)pr->height;
    float w = sc->focal;
    float h = w / AR;
    float s1x = -w / 2, s1y = h / 2;    /* screen top left corner      */
    float s2x = w / 2, s2y = -h / 2;    /* screen bottom right corner  */
    float psx = (float)1 / (float)pr->width, psy = (float)1 / (float)pr->height;

    float uv1x = (float)px * psx, uv1y = (float)py * psy;
    float uv2x = (float)(px + 1) * psx, uv2y = (float)(py + 1) * psy;

    float p1x = f_lerp(s1x, s2x, uv1x), p1y = f_lerp(s1y, s2y, uv1y);
    float p2x = f_lerp(s1x, s2x, uv2x), p2y = f_lerp(s1y, s2y, uv2y);

    f3 position = camera_view(sc, f3_make(0.0f, 0.0f, 0.0f), 1);

    f3 total = f3_make(0.0f, 0.0f, 0.0f);
    const int k = (int)pr->kernel;
    for (int i = 0; i < k; i++) {
        for (int j = 0; j < k; j++) {
            float uvx = (float)(i + 1) / (float)k, uvy = (float)(j + 1) / (float)k;
            float x = f_lerp(p1x, p2x, uvx);
            float y = f_lerp(p1y, p2y, uvy);
            float z = -1.0f;
            f3 ray = f3_normalize(f3_make(x, y, z), pr->normalize_div);
            ray = camera_view(sc, ray, 0);
            f3 color = scene_raymarch(sc, pr, position, ray);
            if (samples) {
                samples[3 * (i * k + j)] = color.x;
                samples[3 * (i * k + j) + 1] = color.y;
                samples[3 * (i * k + j) + 2] = color.z;
            }
            total = f3_add(total, color);
        }
    }
    f3 color = f3_divs(total, (float)(k * k));
    if (rgba_out) { rgba_out[0] = color.x; rgba_out[1] = color.y; rgba_out[2] = color.z; rgba_out[3] = 1.0f; }
}

static void counters_add(rmo_counters *dst, const rmo_counters *src) {
    dst->sdf_calls += src->sdf_calls; dst->prim_evals += src->prim_evals;
    dst->march_steps += src->march_steps; dst->rays += src->rays;
}

/* render::CPU (render.cpp:122-129) and render::OMP (:131-139): the row loop,
 * serial or "#pragma omp parallel for" with the default schedule. */
void rmo_render(const rmo_scene *s, const rmo_params *p, uint32_t row0, uint32_t row1,
                uint32_t row_step, int threads, float *out, rmo_counters *counters) {
    rmo_counters total;
    memset(&total, 0, sizeof total);
    if (row1 > p->height) row1 = p->height;
    if (row_step == 0) row_step = 1;
    int nrows = row1 > row0 ? (int)((row1 - row0 + row_step - 1) / row_step) : 0;
    if (threads <= 1) {
        memset(&tl_counters, 0, sizeof tl_counters);
        for (int r = 0; r < nrows; r++) {
            int pi = (int)(row0 + (uint32_t)r * row_step);
            for (int pj = 0; pj < (int)p->width; pj++)
                render_pixel(s, p, pj, pi, out + 4 * ((size_t)pi * p->width + (size_t)pj), NULL);
        }
        counters_add(&total, &tl_counters);
    } else {
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
        {
            memset(&tl_counters, 0, sizeof tl_counters);
#pragma omp for
            for (int r = 0; r < nrows; r++) {
                int pi = (int)(row0 + (uint32_t)r * row_step);
                for (int pj = 0; pj < (int)p->width; pj++)
                    render_pixel(s, p, pj, pi, out + 4 * ((size_t)pi * p->width + (size_t)pj), NULL);
            }
#pragma omp critical
            counters_add(&total, &tl_counters);
        }
#else
        (void)threads;
#endif
    }
    if (counters) *counters = total;
}

/* The same pixels, any subset of them: render::pixel (render.cpp:82-120) at n given (x, y) coordinates,
 * dealt to the threads one at a time (tests check sparse rows of large frames with it; pixels are
 * independent, so the split does not change a bit of any of them).  out: n x RGBA. */
void rmo_render_pixels(const rmo_scene *s, const rmo_params *p, const uint32_t *xy, uint32_t n, int threads, float *out) {
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int i = 0; i < (int)n; i++)
        render_pixel(s, p, (int)xy[2 * i], (int)xy[2 * i + 1], out + 4 * (size_t)i, NULL);
}

void rmo_pixel_samples(const rmo_scene *s, const rmo_params *p, uint32_t px, uint32_t py, float *rgb_out) {
    render_pixel(s, p, (int)px, (int)py, NULL, rgb_out);
}

/* The PNG the reference saved (LiteImage SaveImage, gamma 1) holds
 * round-to-nearest values: the ambient floor 0.05 is stored as 13 = round(12.75). */
void rmo_quantize(const float *rgba, uint64_t n_pixels, uint8_t *out) {
    for (uint64_t i = 0; i < n_pixels * 4; i++) {
        float c = rgba[i];
        c = f_min(f_max(0.0f, c), 1.0f);
        out[i] = (uint8_t)(int)(c * 255.0f + 0.5f);
    }
}

int rmo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
