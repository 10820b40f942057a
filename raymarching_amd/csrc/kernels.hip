// kernels.hip -- gfx950 kernels of the sphere-tracing render path.
//
// One SSAA sample per lane.  A lane does what scene::raymarch does for one
// sample (/root/reference/source/scene.cpp:26-62): primary march, colour of the
// hit, central-difference normal, one shadow march per light, clamped Lambert
// sum.  A second kernel averages each pixel's kernel x kernel samples in the
// reference's accumulation order (render.cpp:103-119) and quantises.
//
// Arithmetic contract: every fp32 operation below is a single correctly rounded
// IEEE operation in the reference's order -- compiled with -ffp-contract=off
// (no fused multiply-add) and hipcc's default correctly rounded sqrt / divide --
// so the float framebuffer is bit-identical to the CPU path's (signs of zeros
// and NaN payloads aside).
//
// The scene is a flat program (device_scene.h).  All lanes of a wave walk the
// same program at the same time, so operations and primitive rows are fetched
// through wave-uniform addresses (scalar loads, operands in SGPRs); only the
// sample position lives in vector registers.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace rmdev {

namespace {

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{ x, y, z }; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// normalize(a) = a * (1 / sqrt(dot(a, a)))
__device__ __forceinline__ V3 normalize3(V3 a) {
    const float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * inv;
}

// ---- primitive distances (body.cpp:32-35, 41-45, 51-57) --------------------

template <uint32_t KIND>
__device__ __forceinline__ float prim_distance(const Float4 a, const Float4 b, float px, float py, float pz) {
    if (KIND == K_SPHERE) {
        const float dx = a.x - px, dy = a.y - py, dz = a.z - pz;
        return __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - a.w;
    }
    const float hy = (KIND == K_BOX_ISO || KIND == K_CROSS_ISO) ? a.w : b.x;
    const float hz = (KIND == K_BOX_ISO || KIND == K_CROSS_ISO) ? a.w : b.y;
    const float dx = __builtin_fabsf(px - a.x) - a.w;
    const float dy = __builtin_fabsf(py - a.y) - hy;
    const float dz = __builtin_fabsf(pz - a.z) - hz;
    if (KIND == K_BOX || KIND == K_BOX_ISO) {
        return __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
    }
    const float dmin = __builtin_fminf(__builtin_fminf(dx, dy), dz);
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
    return dx + dy + dz - dmin - dmax;
}

// One RUN: fold `count` primitives of one kind into (acc, acc_prim).
template <uint32_t KIND, bool NEG, bool MAX, bool TRACK>
__device__ __forceinline__ void fold_run(const Float4 *__restrict__ pa, const Float4 *__restrict__ pb,
                                         uint32_t first, uint32_t count, float px, float py, float pz,
                                         float &acc, uint32_t &acc_prim) {
    constexpr bool kNeedB = (KIND == K_BOX || KIND == K_CROSS);
#pragma unroll 4
    for (uint32_t i = first; i < first + count; ++i) {
        const Float4 a = pa[i];
        Float4 b = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (kNeedB) b = pb[i];
        float v = prim_distance<KIND>(a, b, px, py, pz);
        if (NEG) v = -v;
        if (TRACK) {
            // std::min / std::max on Surface: the accumulator survives ties
            const bool take = MAX ? (acc < v) : (v < acc);
            acc = take ? v : acc;
            acc_prim = take ? i : acc_prim;
        } else {
            acc = MAX ? __builtin_fmaxf(acc, v) : __builtin_fminf(acc, v);
        }
    }
}

template <uint32_t KIND, bool TRACK>
__device__ __forceinline__ void fold_run_kind(const SceneView &sc, uint32_t fold, uint32_t first, uint32_t count,
                                              float px, float py, float pz, float &acc, uint32_t &acc_prim) {
    switch (fold & (FOLD_NEG | FOLD_MAX)) {
    case 0: fold_run<KIND, false, false, TRACK>(sc.pa, sc.pb, first, count, px, py, pz, acc, acc_prim); break;
    case FOLD_NEG: fold_run<KIND, true, false, TRACK>(sc.pa, sc.pb, first, count, px, py, pz, acc, acc_prim); break;
    case FOLD_MAX: fold_run<KIND, false, true, TRACK>(sc.pa, sc.pb, first, count, px, py, pz, acc, acc_prim); break;
    default: fold_run<KIND, true, true, TRACK>(sc.pa, sc.pb, first, count, px, py, pz, acc, acc_prim); break;
    }
}

// ---- LATTICE: a complete Menger lattice of cubic crosses, folded with max(s, -cross) ----------
//
// The brute-force fold visits every cross c and keeps max(s, -c(p)), c(p) = dx + dy + dz - min - max
// with d = |p - centre| - half per axis (body.cpp:51-57).  max() over floats is order-independent,
// so the result is the largest -c(p) of the lattice, whichever way it is found.  -c(p) can exceed
// the running value `run` only if the median of (dx, dy, dz) is below -run, i.e. only if at least
// two of the three axis terms are below tau = -run (+ a bound on the rounding of the five-term
// sum).  Per level and axis the coordinates are sorted and evenly spaced, so the indices whose axis
// term is below tau form one short contiguous range around p, found by scanning outwards from an
// index estimate with the exact axis term.  Only crosses with two axes in range are evaluated --
// with the exact reference formula, in the reference's operand order -- so the maximum, and
// therefore every bit of the fold's result, equals the brute-force one.  Levels are visited
// coarse to fine; a level whose half-width cannot beat `run` ends the walk (-c(p) <= half).
__device__ __forceinline__ float cross_value(float dx, float dy, float dz) {
    const float dmin = __builtin_fminf(__builtin_fminf(dx, dy), dz);
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
    return dx + dy + dz - dmin - dmax;
}

__device__ __forceinline__ float axis_term(const uint32_t *tab, int i, float p, float h) {
    return __builtin_fabsf(p - __uint_as_float(tab[i])) - h;
}

// All i in [0, n) with axis_term(i) < tau: [lo, hi], empty if lo > hi.
__device__ __forceinline__ void axis_range(const uint32_t *tab, int n, float p, float h, float inv_spacing, float tau,
                                           int &lo, int &hi) {
    float u = (p - __uint_as_float(tab[0])) * inv_spacing + 0.5f;   // index estimate; the scan below is exact
    u = __builtin_fminf(__builtin_fmaxf(u, 0.0f), (float)(n - 1));
    const int i0 = (int)u;
    lo = n;
    hi = -1;
    for (int i = i0; i >= 0; --i) {
        if (axis_term(tab, i, p, h) < tau) {
            lo = i;
            if (hi < 0) hi = i;
        } else if (i < i0) {
            break;
        }
    }
    for (int i = i0 + 1; i < n; ++i) {
        if (!(axis_term(tab, i, p, h) < tau)) break;
        hi = i;
        if (lo > i) lo = i;
    }
}

template <bool TRACK, bool COUNT>
__device__ __forceinline__ void lattice_fold(const uint32_t *w, uint32_t levels, uint32_t first_row, float px, float py,
                                             float pz, float &acc, uint32_t &acc_prim, uint32_t &n_evals) {
    const float bound = __builtin_fabsf(px) + __builtin_fabsf(py) + __builtin_fabsf(pz) + __uint_as_float(w[2]);
    const float eps = 2e-6f * bound;   // > 8x the worst rounding of dx + dy + dz - min - max at this magnitude
    float run = acc;
    int n = 1;
    for (uint32_t l = 0; l < levels; ++l, n *= 3) {
        const float h = __uint_as_float(w[kLatHalf + l]);
        if (!(run < h * 1.00001f + eps)) break;   // no cross this fine can beat the running value
        const float tau = eps - run;
        const float inv = __uint_as_float(w[kLatInvSpacing + l]);
        const uint32_t *X = w + w[kLatOffset + l];
        const uint32_t *Y = X + n;
        const uint32_t *Z = Y + n;
        const uint32_t *ONES = Z + n;
        int lox, hix, loy, hiy, loz, hiz;
        axis_range(X, n, px, h, inv, tau, lox, hix);
        axis_range(Y, n, py, h, inv, tau, loy, hiy);
        axis_range(Z, n, pz, h, inv, tau, loz, hiz);
        if (COUNT) n_evals += 3;
        const bool bx = lox <= hix, by = loy <= hiy, bz = loz <= hiz;
        if (bx && by) {
            for (int i = lox; i <= hix; ++i)
                for (int j = loy; j <= hiy; ++j) {
                    const uint32_t oi = ONES[i], oj = ONES[j];
                    if (oi & oj) continue;   // two zero offsets at one subdivision step: no such sub-cube
                    const float dx = axis_term(X, i, px, h), dy = axis_term(Y, j, py, h);
                    for (int k = 0; k < n; ++k) {
                        const float v = -cross_value(dx, dy, axis_term(Z, k, pz, h));
                        run = (ONES[k] & (oi | oj)) ? run : __builtin_fmaxf(run, v);
                    }
                    if (COUNT) n_evals += n;
                }
        }
        if (bx && bz) {
            for (int i = lox; i <= hix; ++i)
                for (int k = loz; k <= hiz; ++k) {
                    const uint32_t oi = ONES[i], ok = ONES[k];
                    if (oi & ok) continue;
                    const float dx = axis_term(X, i, px, h), dz = axis_term(Z, k, pz, h);
                    for (int j = 0; j < n; ++j) {
                        const float v = -cross_value(dx, axis_term(Y, j, py, h), dz);
                        run = (ONES[j] & (oi | ok)) ? run : __builtin_fmaxf(run, v);
                    }
                    if (COUNT) n_evals += n;
                }
        }
        if (by && bz) {
            for (int j = loy; j <= hiy; ++j)
                for (int k = loz; k <= hiz; ++k) {
                    const uint32_t oj = ONES[j], ok = ONES[k];
                    if (oj & ok) continue;
                    const float dy = axis_term(Y, j, py, h), dz = axis_term(Z, k, pz, h);
                    for (int i = 0; i < n; ++i) {
                        const float v = -cross_value(axis_term(X, i, px, h), dy, dz);
                        run = (ONES[i] & (oj | ok)) ? run : __builtin_fmaxf(run, v);
                    }
                    if (COUNT) n_evals += n;
                }
        }
    }
    if (TRACK) acc_prim = (acc < run) ? first_row : acc_prim;   // all crosses of a lattice share one colour
    acc = run;
}

// Body::List::SDF for the whole tree (body.cpp:66-111) at one point per lane.
// TRACK = false: distance only (march steps, gradient taps, shadow rays).
// TRACK = true : also the depth-first index of the primitive whose colour the
//                resulting Surface carries.
template <bool TRACK, bool COUNT>
__device__ __forceinline__ void scene_distance(const SceneView &sc, const uint32_t *lattice, float px, float py, float pz,
                                               float &out_sd, uint32_t &out_prim, uint32_t &n_evals) {
    const float inf = __builtin_inff();
    float acc = inf;              // an empty list is +inf / colour 0
    uint32_t acc_prim = kNoPrim;
    float parked[kMaxDepth];
    uint32_t parked_prim[kMaxDepth];
#pragma unroll
    for (uint32_t d = 0; d < kMaxDepth; ++d) {
        parked[d] = inf;
        parked_prim[d] = kNoPrim;
    }
    const Op *__restrict__ ops = sc.ops;
    for (uint32_t pc = 0;; ++pc) {
        const Op op = ops[pc];
        if (op.code == OP_END) break;
        if (op.code == OP_RUN) {
            const uint32_t kind = op.arg & 0xffu, fold = op.arg >> 8;
            if (fold & FOLD_FIRST) {
                // the first child replaces the accumulator: start from the fold's identity
                acc = (fold & FOLD_MAX) ? -inf : inf;
                acc_prim = kNoPrim;
            }
            if (COUNT) n_evals += op.count;
            switch (kind) {
            case K_SPHERE: fold_run_kind<K_SPHERE, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_BOX: fold_run_kind<K_BOX, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_CROSS: fold_run_kind<K_CROSS, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_BOX_ISO: fold_run_kind<K_BOX_ISO, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            default: fold_run_kind<K_CROSS_ISO, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            }
        } else if (op.code == OP_LATTICE) {
            lattice_fold<TRACK, COUNT>(lattice + (op.arg >> 12), op.arg & 0xfu, op.first, px, py, pz, acc, acc_prim, n_evals);
        } else if (op.code == OP_PUSH) {
            const uint32_t slot = op.arg;
#pragma unroll
            for (uint32_t d = 0; d < kMaxDepth; ++d) {
                if (d == slot) {
                    parked[d] = acc;
                    parked_prim[d] = acc_prim;
                }
            }
            acc = inf;
            acc_prim = kNoPrim;
        } else {  // OP_POP: the finished list is one child of the parked accumulator
            const uint32_t slot = op.arg & 0xffu, fold = op.arg >> 8;
            float v = (fold & FOLD_NEG) ? -acc : acc;
            const uint32_t v_prim = acc_prim;
            float up = inf;
            uint32_t up_prim = kNoPrim;
#pragma unroll
            for (uint32_t d = 0; d < kMaxDepth; ++d) {
                if (d == slot) {
                    up = parked[d];
                    up_prim = parked_prim[d];
                }
            }
            if (fold & FOLD_FIRST) {
                up = (fold & FOLD_MAX) ? -inf : inf;
                up_prim = kNoPrim;
            }
            const bool take = (fold & FOLD_MAX) ? (up < v) : (v < up);
            acc = take ? v : up;
            acc_prim = take ? v_prim : up_prim;
        }
    }
    out_sd = acc;
    out_prim = acc_prim;
}

template <bool COUNT>
__device__ __forceinline__ float scene_sd(const SceneView &sc, const uint32_t *lattice, V3 p, uint32_t &n_evals) {
    float sd;
    uint32_t prim;
    scene_distance<false, COUNT>(sc, lattice, p.x, p.y, p.z, sd, prim, n_evals);
    return sd;
}

extern __shared__ uint32_t lds_lattice[];

// Every workgroup keeps the lattice tables in LDS (a few KB; 0 words when the scene has none).
__device__ __forceinline__ const uint32_t *stage_lattice(const SceneView &sc) {
    for (uint32_t i = threadIdx.x; i < sc.lattice_words; i += blockDim.x) lds_lattice[i] = sc.lattice[i];
    __syncthreads();
    return lds_lattice;
}

// scene::surface (scene.cpp:34-42): advances p along ray; `at` is the point of
// the last SDF evaluation (whose Surface the reference returns).
template <bool COUNT>
__device__ __forceinline__ void march(const SceneView &sc, const uint32_t *lattice, const FrameView &fr, V3 &p, V3 ray,
                                      V3 &at, uint32_t &n_sdf, uint32_t &n_evals) {
    for (int it = 0; it < fr.iterations; ++it) {
        at = p;
        const float sd = scene_sd<COUNT>(sc, lattice, p, n_evals);
        if (COUNT) n_sdf++;
        p = p + ray * sd;
        if (sd < fr.surface_eps) break;
    }
}

__device__ __forceinline__ V3 camera_view(const float *m, V3 v, float w) {
    return v3(v.x * m[0] + v.y * m[4] + v.z * m[8] + w * m[12], v.x * m[1] + v.y * m[5] + v.z * m[9] + w * m[13],
              v.x * m[2] + v.y * m[6] + v.z * m[10] + w * m[14]);
}

__device__ __forceinline__ float lerp1(float u, float v, float t) { return u + t * (v - u); }

// Block = 16 x 16 sub-samples, each wave an 8 x 8 quadrant.
template <bool COUNT>
__global__ __launch_bounds__(256) void trace_kernel(const SceneView sc, const FrameView fr, Float4 *__restrict__ samples,
                                                     WorkCounters *__restrict__ counters) {
    __shared__ unsigned long long block_counts[4];
    if (COUNT) {
        if (threadIdx.x < 4) block_counts[threadIdx.x] = 0;
    }
    const uint32_t *lattice = stage_lattice(sc);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    const uint32_t sx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t sy = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    uint32_t n_sdf = 0, n_steps = 0, n_rays = 0, n_evals = 0;

    if (sx < sw && sy < sh) {
        const uint32_t pj = sx / k, i = sx - pj * k;    // pixel column, sub-sample column (outer loop index)
        const uint32_t ly = sy / k, j = sy - ly * k;    // local pixel row, sub-sample row (inner loop index)
        const uint32_t band = ly / fr.band_rows;
        const uint32_t pi = (fr.first_band + band * fr.band_stride) * fr.band_rows + (ly - band * fr.band_rows);
        V3 color = v3(0.0f, 0.0f, 0.0f);
        if (pi < fr.height) {
            // ---- render::pixel, render.cpp:82-111
            const float AR = (float)fr.width / (float)fr.height;
            const float w = sc.focal;
            const float h = w / AR;
            const float s1x = -w / 2, s1y = h / 2;
            const float s2x = w / 2, s2y = -h / 2;
            const float psx = 1.0f / (float)fr.width, psy = 1.0f / (float)fr.height;
            const float uv1x = (float)pj * psx, uv1y = (float)pi * psy;
            const float uv2x = (float)(pj + 1u) * psx, uv2y = (float)(pi + 1u) * psy;
            const float p1x = lerp1(s1x, s2x, uv1x), p1y = lerp1(s1y, s2y, uv1y);
            const float p2x = lerp1(s1x, s2x, uv2x), p2y = lerp1(s1y, s2y, uv2y);
            const V3 origin = camera_view(sc.cam, v3(0.0f, 0.0f, 0.0f), 1.0f);
            const float uvx = (float)(i + 1u) / (float)k, uvy = (float)(j + 1u) / (float)k;
            const float x = lerp1(p1x, p2x, uvx);
            const float y = lerp1(p1y, p2y, uvy);
            V3 ray = normalize3(v3(x, y, -1.0f));
            ray = camera_view(sc.cam, ray, 0.0f);

            // ---- scene::raymarch, scene.cpp:26-32
            V3 p = origin, at = origin;
            uint32_t steps0 = n_sdf;
            march<COUNT>(sc, lattice, fr, p, ray, at, n_sdf, n_evals);
            if (COUNT) { n_steps += n_sdf - steps0; n_rays++; }
            uint32_t prim = kNoPrim;
            if (fr.iterations > 0) {
                float sd_unused;
                uint32_t evals_unused = 0;   // the colour pass repeats the last march evaluation: not extra work of the algorithm
                scene_distance<true, false>(sc, lattice, at.x, at.y, at.z, sd_unused, prim, evals_unused);
            }
            // scene::grad, scene.cpp:70-89
            const float hh = 1e-3f;
            float g[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll 1
            for (int axis = 0; axis < 3; ++axis) {   // one evaluator instance for the six taps
                const V3 e = v3(axis == 0 ? hh : 0.0f, axis == 1 ? hh : 0.0f, axis == 2 ? hh : 0.0f);
                float tap[2];
#pragma unroll 1
                for (int side = 0; side < 2; ++side) {
                    const V3 q = side == 0 ? v3(p.x + e.x, p.y + e.y, p.z + e.z) : v3(p.x - e.x, p.y - e.y, p.z - e.z);
                    const float sd = scene_sd<COUNT>(sc, lattice, q, n_evals);
                    tap[0] = side == 0 ? sd : tap[0];
                    tap[1] = sd;
                }
                const float diff = tap[0] - tap[1];
                g[0] = axis == 0 ? diff : g[0];
                g[1] = axis == 1 ? diff : g[1];
                g[2] = axis == 2 ? diff : g[2];
            }
            if (COUNT) n_sdf += 6;
            const float two_h = 2 * hh;
            const V3 normal = normalize3(v3(g[0] / two_h, g[1] / two_h, g[2] / two_h));

            // scene::lighting + scene::shadow, scene.cpp:45-62
            float lighting = 0.0f;
            const float lift = fr.surface_eps + fr.offset_eps;
            for (uint32_t l = 0; l < sc.n_lights; ++l) {
                const Float4 lp = sc.lights[l];
                const V3 light = v3(lp.x, lp.y, lp.z);
                const V3 to_light = normalize3(light - p);
                V3 q = p + normal * lift;
                V3 q_at = q;
                steps0 = n_sdf;
                march<COUNT>(sc, lattice, fr, q, to_light, q_at, n_sdf, n_evals);
                if (COUNT) { n_steps += n_sdf - steps0; n_rays++; }
                const bool shadowed = dot3(light - q, to_light) > 0;
                if (!shadowed) lighting += dot3(normal, to_light);
            }
            lighting = __builtin_fminf(__builtin_fmaxf(fr.saturation, lighting), 1.0f);
            V3 base = v3(0.0f, 0.0f, 0.0f);
            if (prim != kNoPrim) {
                const Float4 c = sc.colors[prim];
                base = v3(c.x, c.y, c.z);
            }
            color = v3(lighting * base.x, lighting * base.y, lighting * base.z);
        }
        samples[(size_t)sy * sw + sx] = Float4{ color.x, color.y, color.z, 0.0f };
    }
    if (COUNT) {
        atomicAdd(&block_counts[0], (unsigned long long)n_sdf);
        atomicAdd(&block_counts[1], (unsigned long long)n_steps);
        atomicAdd(&block_counts[2], (unsigned long long)n_rays);
        atomicAdd(&block_counts[3], (unsigned long long)n_evals);
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&counters->sdf_calls, block_counts[0]);
            atomicAdd(&counters->march_steps, block_counts[1]);
            atomicAdd(&counters->rays, block_counts[2]);
            atomicAdd(&counters->prim_evals, block_counts[3]);
        }
    }
}

// render.cpp:103-119: total += colour over i (outer, x) then j (inner, y);
// pixel = total / (k*k), alpha 1.  RGBA8 = round(clamp(c, 0, 1) * 255).
__global__ __launch_bounds__(256) void resolve_kernel(const FrameView fr, const Float4 *__restrict__ samples,
                                                       float *__restrict__ out_f32, unsigned char *__restrict__ out_rgba8) {
    const uint32_t rows = fr.band_count * fr.band_rows;
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= fr.width || y >= rows) return;
    const uint32_t k = fr.kernel;
    const size_t sw = (size_t)fr.width * k;
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
    for (uint32_t i = 0; i < k; ++i) {
        for (uint32_t j = 0; j < k; ++j) {
            const Float4 s = samples[((size_t)y * k + j) * sw + (size_t)x * k + i];
            tx += s.x;
            ty += s.y;
            tz += s.z;
        }
    }
    const float n = (float)(k * k);
    const float r = tx / n, g = ty / n, b = tz / n;
    const size_t o = ((size_t)y * fr.width + x) * 4;
    if (out_f32) {
        *reinterpret_cast<float4 *>(out_f32 + o) = make_float4(r, g, b, 1.0f);
    }
    if (out_rgba8) {
        const float cr = __builtin_fminf(__builtin_fmaxf(0.0f, r), 1.0f);
        const float cg = __builtin_fminf(__builtin_fmaxf(0.0f, g), 1.0f);
        const float cb = __builtin_fminf(__builtin_fmaxf(0.0f, b), 1.0f);
        const uint32_t qr = (uint32_t)(int)(cr * 255.0f + 0.5f);
        const uint32_t qg = (uint32_t)(int)(cg * 255.0f + 0.5f);
        const uint32_t qb = (uint32_t)(int)(cb * 255.0f + 0.5f);
        *reinterpret_cast<uint32_t *>(out_rgba8 + o) = qr | (qg << 8) | (qb << 16) | (255u << 24);
    }
}

__global__ __launch_bounds__(256) void eval_sdf_kernel(const SceneView sc, const float *__restrict__ xyz, uint32_t n,
                                                        float *__restrict__ sd, uint32_t *__restrict__ prim) {
    const uint32_t *lattice = stage_lattice(sc);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    float d;
    uint32_t p, evals = 0;
    scene_distance<true, false>(sc, lattice, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], d, p, evals);
    sd[t] = d;
    prim[t] = p;
}

}  // namespace

size_t sample_texels(const FrameView &fr) {
    return (size_t)fr.band_count * fr.band_rows * fr.kernel * (size_t)fr.width * fr.kernel;
}

hipError_t launch_trace(const SceneView &sc, const FrameView &fr, Float4 *samples, WorkCounters *counters,
                        hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0) return hipSuccess;
    const dim3 grid((sw + 15u) / 16u, (sh + 15u) / 16u, 1u);
    if (counters) {
        hipLaunchKernelGGL(trace_kernel<true>, grid, dim3(256), sc.lattice_words * 4u, stream, sc, fr, samples, counters);
    } else {
        hipLaunchKernelGGL(trace_kernel<false>, grid, dim3(256), sc.lattice_words * 4u, stream, sc, fr, samples, counters);
    }
    return hipGetLastError();
}

hipError_t launch_resolve(const FrameView &fr, const Float4 *samples, float *out_f32, unsigned char *out_rgba8,
                          hipStream_t stream) {
    const uint32_t rows = fr.band_count * fr.band_rows;
    if (fr.width == 0 || rows == 0) return hipSuccess;
    const dim3 grid((fr.width + 63u) / 64u, (rows + 3u) / 4u, 1u);
    hipLaunchKernelGGL(resolve_kernel, grid, dim3(256), 0, stream, fr, samples, out_f32, out_rgba8);
    return hipGetLastError();
}

hipError_t launch_eval_sdf(const SceneView &sc, const float *xyz, uint32_t n, float *sd, uint32_t *prim,
                           hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(eval_sdf_kernel, dim3((n + 255u) / 256u), dim3(256), sc.lattice_words * 4u, stream, sc, xyz, n, sd, prim);
    return hipGetLastError();
}

}  // namespace rmdev
