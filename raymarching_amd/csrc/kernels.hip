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

// Body::List::SDF for the whole tree (body.cpp:66-111) at one point per lane.
// TRACK = false: distance only (march steps, gradient taps, shadow rays).
// TRACK = true : also the depth-first index of the primitive whose colour the
//                resulting Surface carries.
template <bool TRACK>
__device__ __forceinline__ void scene_distance(const SceneView &sc, float px, float py, float pz, float &out_sd,
                                               uint32_t &out_prim) {
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
            switch (kind) {
            case K_SPHERE: fold_run_kind<K_SPHERE, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_BOX: fold_run_kind<K_BOX, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_CROSS: fold_run_kind<K_CROSS, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            case K_BOX_ISO: fold_run_kind<K_BOX_ISO, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            default: fold_run_kind<K_CROSS_ISO, TRACK>(sc, fold, op.first, op.count, px, py, pz, acc, acc_prim); break;
            }
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

__device__ __forceinline__ float scene_sd(const SceneView &sc, V3 p) {
    float sd;
    uint32_t prim;
    scene_distance<false>(sc, p.x, p.y, p.z, sd, prim);
    return sd;
}

// scene::surface (scene.cpp:34-42): advances p along ray; `at` is the point of
// the last SDF evaluation (whose Surface the reference returns).
template <bool COUNT>
__device__ __forceinline__ void march(const SceneView &sc, const FrameView &fr, V3 &p, V3 ray, V3 &at,
                                      uint32_t &n_sdf) {
    for (int it = 0; it < fr.iterations; ++it) {
        at = p;
        const float sd = scene_sd(sc, p);
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
    __shared__ unsigned long long block_counts[3];
    if (COUNT) {
        if (threadIdx.x < 3) block_counts[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    const uint32_t sx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t sy = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    uint32_t n_sdf = 0, n_steps = 0, n_rays = 0;

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
            march<COUNT>(sc, fr, p, ray, at, n_sdf);
            if (COUNT) { n_steps += n_sdf - steps0; n_rays++; }
            uint32_t prim = kNoPrim;
            if (fr.iterations > 0) {
                float sd_unused;
                scene_distance<true>(sc, at.x, at.y, at.z, sd_unused, prim);
            }
            // scene::grad, scene.cpp:70-89
            const float hh = 1e-3f;
            const float gx = scene_sd(sc, v3(p.x + hh, p.y, p.z)) - scene_sd(sc, v3(p.x - hh, p.y, p.z));
            const float gy = scene_sd(sc, v3(p.x, p.y + hh, p.z)) - scene_sd(sc, v3(p.x, p.y - hh, p.z));
            const float gz = scene_sd(sc, v3(p.x, p.y, p.z + hh)) - scene_sd(sc, v3(p.x, p.y, p.z - hh));
            if (COUNT) n_sdf += 6;
            const float two_h = 2 * hh;
            const V3 normal = normalize3(v3(gx / two_h, gy / two_h, gz / two_h));

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
                march<COUNT>(sc, fr, q, to_light, q_at, n_sdf);
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
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&counters->sdf_calls, block_counts[0]);
            atomicAdd(&counters->march_steps, block_counts[1]);
            atomicAdd(&counters->rays, block_counts[2]);
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
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    float d;
    uint32_t p;
    scene_distance<true>(sc, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], d, p);
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
        hipLaunchKernelGGL(trace_kernel<true>, grid, dim3(256), 0, stream, sc, fr, samples, counters);
    } else {
        hipLaunchKernelGGL(trace_kernel<false>, grid, dim3(256), 0, stream, sc, fr, samples, counters);
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
    hipLaunchKernelGGL(eval_sdf_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, sc, xyz, n, sd, prim);
    return hipGetLastError();
}

}  // namespace rmdev
