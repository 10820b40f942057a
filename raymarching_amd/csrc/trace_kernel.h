// trace_kernel.h -- the render kernel body, generic over the scene evaluator `Scene`
// (InterpretedScene, or the straight-line evaluator generated for one scene):
//     Scene::sd<NP, COUNT>(sc, lds, points, distances, work, cap)  distance only
//     Scene::surface<NP>(sc, lds, points, distances, primitives)   distance + colour-carrying primitive
#pragma once

#include "device_core.h"

namespace rmdev {

// scene::surface (scene.cpp:34-42) for the lane's two samples in lockstep: advances p[s]
// along ray[s] while sample s is live; at[s] is the point of its last SDF evaluation (whose
// Surface the reference returns).
//
// to_light / past_light: a shadow ray only has to tell whether it stops before its light (scene.cpp:45-50:
// shadowed iff dot(light - end, ray) > 0).  to_light[s] starts as dot(light - p, ray) and goes down by every
// step's length, so it follows that dot product up to the roundings of the steps; once it is below
// -past_light[s] the ray is beyond the light for good and the rest of its march cannot change the answer:
// every later step but the last moves it further on (sd >= surface_eps > 0), and the last one cannot take it
// back by more than the roundings, because the scene SDF is 1-Lipschitz (see `cap` below), so after a step
// of length sd its value is at least -roundings.  past_light covers both: 1.8e-7 of the coordinates'
// magnitude per step for the roundings of p (three components, half an ulp each), 3e-5 of it for the SDF.
// Camera rays pass to_light = +inf.
template <class Scene, int kPts, bool COUNT>
__device__ __forceinline__ void march(const SceneView &sc, const Lds &lds, const FrameView &fr, V3 (&p)[kPts],
                                      const V3 (&ray)[kPts], V3 (&at)[kPts], const bool (&live)[kPts], float (&to_light)[kPts],
                                      const float (&past_light)[kPts], LaneCounts &cnt) {
    bool done[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        done[s] = !live[s];
        if (COUNT && live[s]) cnt.rays++;
    }
    // Upper bound of the next evaluation's result, for the scene walkers that can use one (see `capped` in
    // device_core.h).  The scene SDF is 1-Lipschitz -- spheres in the Euclidean norm, boxes and crosses in
    // the maximum norm, which is smaller, and min, max and negation keep the property -- so after the
    // step p' = p + ray * sd, |p' - p| = |sd| (1 + 1e-7), the value at p' is at most sd + |sd|; 1e-5
    // relative and 4e-6 of the coordinates' magnitude cover the roundings of the step, of |ray| = 1 and
    // of the evaluations on both sides (each below 1e-6 of the magnitudes involved).
    float cap[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) cap[s] = __builtin_inff();
    for (int it = 0; it < fr.iterations; ++it) {
        bool all_done = true;
#pragma unroll
        for (int s = 0; s < kPts; ++s) all_done = all_done && done[s];
        if (all_done) break;
        float sd[kPts];
        Work work[kPts] = {};
        Scene::template sd<kPts, COUNT>(sc, lds, p, sd, work, cap);
        probe<COUNT>(kProbeSceneCall, 1u);
        if (COUNT) {
            // the wave sits through this iteration with all of its lanes, whichever of them still march (with one
            // sample per lane the loop is left per lane): the first lane that is still here books the 64 slots
            const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
            if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(here)) cnt.slots += 64u * kPts;
        }
#pragma unroll
        for (int s = 0; s < kPts; ++s) {
            if (!done[s]) {
                at[s] = p[s];
                p[s] = p[s] + ray[s] * sd[s];
                cap[s] = sd[s] + __builtin_fabsf(sd[s]) * 1.00001f +
                         4e-6f * (__builtin_fabsf(p[s].x) + __builtin_fabsf(p[s].y) + __builtin_fabsf(p[s].z) + __builtin_fabsf(sd[s]));
                to_light[s] -= sd[s];
                done[s] = sd[s] < fr.surface_eps || to_light[s] < -past_light[s];
                if (COUNT) {
                    cnt.sdf++;
                    cnt.steps++;
                    cnt.evals += work[s].evals;
                    cnt.flops += work[s].flops;
                }
            }
        }
    }
}

// kPts = 2: block = 32 x 16 sub-samples, each wave a 16 x 8 tile, each lane two horizontally
// adjacent samples that march in lockstep -- halves the scalar-unit cost per sample, which bounds
// scenes of few records; a sample that finishes early idles until its partner is done.
// kPts = 1: block = 16 x 16, wave = 8 x 8, one sample per lane -- for scenes whose cost is in long
// primitive runs (vector-ALU bound), where idling would cost more than the scalar unit saves.
// MIN_WAVES: waves per SIMD the register allocation must allow.
template <class Scene, int kPts, bool COUNT, int kWaves = 4>
__device__ __forceinline__ void trace_body(const SceneView &sc, const FrameView &fr, Sample *__restrict__ samples,
                                           WorkCounters *__restrict__ counters) {
    __shared__ unsigned long long block_counts[6];
    if (COUNT) {
        if (threadIdx.x < 6) block_counts[threadIdx.x] = 0;
        if (threadIdx.x < 16) probe_words()[threadIdx.x] = 0;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    // kWaves = 4: a workgroup is 2 x 2 waves of 8 x 8 lanes; kWaves = 1: one wave
    constexpr uint32_t kSide = kWaves == 4 ? 2u : 1u;
    const uint32_t slot = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t n_tiles = gridDim.x * ((sh + 8u * kSide - 1u) / (8u * kSide));   // gridDim.x = tiles per row
    if (slot >= n_tiles + (fr.tile_order ? fr.extra_slots : 0u)) return;            // the last grid row's surplus
    const Lds lds = stage_lds(sc);
    const unsigned long long started = fr.tile_cost ? wall_clock64() : 0ull;
    const uint32_t entry = fr.tile_order ? ((ConstIndex)fr.tile_order)[slot] : slot;
    const uint32_t tile = entry & kTileMask, part = (kWaves == 1 && kPts == 1) ? entry >> kTileBits : 0u;
    const uint32_t tile_y = tile / gridDim.x, tile_x = tile - tile_y * gridDim.x;
    // a whole tile: lane = 8 * row + column; quarter q: lanes 0..15 = 4 * row + column in the quarter
    // part 1..4: quarter part - 1 (4 x 4 samples, lanes 0..15); part 5..20: sixteenth part - 5 (2 x 2 samples, lanes 0..3)
    const uint32_t six = part - 5u;
    const uint32_t col = part == 0u ? (lane & 7u) : (part < 5u ? ((part - 1u) & 1u) * 4u + (lane & 3u) : (six & 3u) * 2u + (lane & 1u));
    const uint32_t row = part == 0u ? (lane >> 3) : (part < 5u ? ((part - 1u) >> 1) * 4u + ((lane >> 2) & 3u) : (six >> 2) * 2u + ((lane >> 1) & 1u));
    const bool in_part = part == 0u || lane < (part < 5u ? 16u : 4u);
    const uint32_t sx0 = tile_x * (8u * kSide * kPts) + (kWaves == 4 ? (wave & 1u) : 0u) * (8u * kPts) + col * (uint32_t)kPts;
    const uint32_t sy = tile_y * (8u * kSide) + (kWaves == 4 ? (wave >> 1) : 0u) * 8u + row;
    LaneCounts cnt = { 0, 0, 0, 0, 0, 0 };

    // ---- render::pixel, render.cpp:82-111: the camera ray of each sample
    const float AR = (float)fr.width / (float)fr.height;
    const ConstFloats cam = (ConstFloats)sc.camera;   // 16 matrix floats, then the focal length
    const float w = cam[16];
    const float h = w / AR;
    const float s1x = -w / 2, s1y = h / 2;
    const float s2x = w / 2, s2y = -h / 2;
    const float psx = 1.0f / (float)fr.width, psy = 1.0f / (float)fr.height;
    const V3 origin = camera_view(cam, v3(0.0f, 0.0f, 0.0f), 1.0f);
    const uint32_t ly = sy / k, j = sy - ly * k;   // local pixel row, sub-sample row (inner loop index)
    const uint32_t band = ly / fr.band_rows;
    const uint32_t image_band = fr.band_list ? ((ConstIndex)fr.band_list)[band < fr.band_count ? band : fr.band_count - 1u]
                                             : fr.first_band + band * fr.band_stride;
    const uint32_t pi = image_band * fr.band_rows + (ly - band * fr.band_rows);
    const float uv1y = (float)pi * psy, uv2y = (float)(pi + 1u) * psy;
    const float p1y = lerp1(s1y, s2y, uv1y), p2y = lerp1(s1y, s2y, uv2y);
    const float uvy = (float)(j + 1u) / (float)k;
    const float y = lerp1(p1y, p2y, uvy);

    // live: the sample exists; lit: its colour can depend on the lighting (decided after the camera ray)
    bool stored[kPts], live[kPts], lit[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        stored[s] = in_part && sx0 + (uint32_t)s < sw && sy < sh;
        live[s] = stored[s] && pi < fr.height;
        lit[s] = live[s];
    }

    // ---- scene::raymarch, scene.cpp:26-32.  Ray 0 is the camera ray; ray 1 + l is the shadow
    // ray to light l (scene::lighting / scene::shadow, scene.cpp:45-62).  One march instance.
    V3 hit[kPts], normal[kPts], p[kPts], ray[kPts], at[kPts];
    float lighting[kPts], to_light[kPts], past_light[kPts];
    uint32_t prim[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        hit[s] = origin;
        normal[s] = v3(0.0f, 0.0f, 0.0f);
        lighting[s] = 0.0f;
        prim[s] = kNoPrim;
    }
    const float lift = fr.surface_eps + fr.offset_eps;
#pragma unroll 1
    for (uint32_t r = 0; r <= sc.n_lights; ++r) {
        V3 light = v3(0.0f, 0.0f, 0.0f);
        if (r == 0) {
#pragma unroll
            for (int s = 0; s < kPts; ++s) {
                const uint32_t sx = sx0 + (uint32_t)s;
                const uint32_t pj = sx / k, i = sx - pj * k;   // pixel column, sub-sample column (outer loop index)
                const float uv1x = (float)pj * psx, uv2x = (float)(pj + 1u) * psx;
                const float p1x = lerp1(s1x, s2x, uv1x), p2x = lerp1(s1x, s2x, uv2x);
                const float uvx = (float)(i + 1u) / (float)k;
                const float x = lerp1(p1x, p2x, uvx);
                p[s] = origin;
                ray[s] = camera_view(cam, normalize3(v3(x, y, -1.0f)), 0.0f);
                to_light[s] = __builtin_inff();
                past_light[s] = 0.0f;
            }
        } else {
            const RowWords lp = ((ConstRows)sc.lights)[r - 1];
            light = v3(lp[0], lp[1], lp[2]);
#pragma unroll
            for (int s = 0; s < kPts; ++s) {
                ray[s] = normalize3(light - hit[s]);
                p[s] = hit[s] + normal[s] * lift;
                to_light[s] = fr.skip_unlit ? dot3(light - p[s], ray[s]) : __builtin_inff();
                const float scale = __builtin_fabsf(light.x) + __builtin_fabsf(light.y) + __builtin_fabsf(light.z) +
                                    __builtin_fabsf(p[s].x) + __builtin_fabsf(p[s].y) + __builtin_fabsf(p[s].z) + __builtin_fabsf(to_light[s]);
                past_light[s] = 0.01f + (1e-4f + 4e-7f * (float)fr.iterations) * scale;
            }
        }
#pragma unroll
        for (int s = 0; s < kPts; ++s) at[s] = p[s];
        march<Scene, kPts, COUNT>(sc, lds, fr, p, ray, at, lit, to_light, past_light, cnt);
        if (r == 0) {
#pragma unroll
            for (int s = 0; s < kPts; ++s) hit[s] = p[s];
            // the Surface scene::surface returned: colour of the last evaluation
            if (fr.iterations > 0) {
                float sd_unused[kPts];
                Scene::template surface<kPts>(sc, lds, at, sd_unused, prim);
            }
            // scene::raymarch returns light * colour (scene.cpp:31) with light clamped into [saturation, 1]
            // (scene.cpp:60): where the surface has no colour (the wall behind the scene, 62 % of the shipped
            // frame) the product is zero whatever the normal and the shadow rays turn out to be, so those
            // samples sit out the rest, and a wave of nothing else is finished.
            if (fr.skip_unlit && fr.saturation >= 0.0f) {
                bool any = false;
#pragma unroll
                for (int s = 0; s < kPts; ++s) {
                    if (lit[s]) {
                        bool coloured = false;
                        if (prim[s] != kNoPrim) {
                            const Float4 c = sc.colors[prim[s]];
                            coloured = c.x != 0.0f || c.y != 0.0f || c.z != 0.0f;
                        }
                        lit[s] = coloured;
                    }
                    any = any || lit[s];
                }
                if (__builtin_amdgcn_ballot_w64(any) == 0) break;
            }
            // scene::grad, scene.cpp:70-89: six taps, one evaluator instance
            const float hh = 1e-3f;
            float g[kPts][3] = {};
#pragma unroll 1
            for (int axis = 0; axis < 3; ++axis) {
                const V3 e = v3(axis == 0 ? hh : 0.0f, axis == 1 ? hh : 0.0f, axis == 2 ? hh : 0.0f);
                float plus[kPts] = {};
#pragma unroll 1
                for (int side = 0; side < 2; ++side) {
                    V3 q[kPts];
#pragma unroll
                    for (int s = 0; s < kPts; ++s)
                        q[s] = side == 0 ? v3(hit[s].x + e.x, hit[s].y + e.y, hit[s].z + e.z)
                                         : v3(hit[s].x - e.x, hit[s].y - e.y, hit[s].z - e.z);
                    float sd[kPts];
                    Work work[kPts] = {};
                    float no_cap[kPts];
#pragma unroll
                    for (int s = 0; s < kPts; ++s) no_cap[s] = __builtin_inff();
                    Scene::template sd<kPts, COUNT>(sc, lds, q, sd, work, no_cap);
                    if (COUNT) {
                        bool any_lit = false;
#pragma unroll
                        for (int s = 0; s < kPts; ++s) any_lit = any_lit || lit[s];
                        const unsigned long long lit_lanes = __builtin_amdgcn_ballot_w64(any_lit);
                        if ((threadIdx.x & 63u) == 0u) {
                            atomicAdd(&probe_words()[2 * kProbeGradTap], (unsigned long long)__builtin_popcountll(lit_lanes));
                            atomicAdd(&probe_words()[2 * kProbeGradTap + 1], 64ull);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < kPts; ++s) {
                        if (COUNT && lit[s]) {
                            cnt.sdf++;
                            cnt.evals += work[s].evals;
                            cnt.flops += work[s].flops;
                        }
                        const float diff = plus[s] - sd[s];
                        plus[s] = side == 0 ? sd[s] : plus[s];
                        if (side == 1) {
                            g[s][0] = axis == 0 ? diff : g[s][0];
                            g[s][1] = axis == 1 ? diff : g[s][1];
                            g[s][2] = axis == 2 ? diff : g[s][2];
                        }
                    }
                }
            }
            const float two_h = 2 * hh;
#pragma unroll
            for (int s = 0; s < kPts; ++s) normal[s] = normalize3(v3(g[s][0] / two_h, g[s][1] / two_h, g[s][2] / two_h));
        } else {
#pragma unroll
            for (int s = 0; s < kPts; ++s) {
                const bool shadowed = dot3(light - p[s], ray[s]) > 0;
                if (!shadowed) lighting[s] += dot3(normal[s], ray[s]);
            }
        }
    }

    // Whole pixels in the workgroup (see FrameView::pixels_f32): the samples meet in LDS and the lane of every
    // pixel's first sample adds them up in the reference's order (render.cpp:103-112: i, the column, outer; j inner).
    const bool fused = kWaves == 1 && kPts == 1 && (fr.pixels_f32 || fr.pixels_rgba8);
    __shared__ float tile_rgb[kWaves == 1 && kPts == 1 ? 3 * 64 : 1];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        if (!stored[s] && !fused) continue;
        // scene::lighting's clamp (scene.cpp:60); the product with the colour is formed where the samples are added up
        Sample sample_out = { 0.0f, kNoPrim };
        if (live[s]) sample_out = Sample{ __builtin_fminf(__builtin_fmaxf(fr.saturation, lighting[s]), 1.0f), prim[s] };
        if (kWaves == 1 && kPts == 1 && fused) {
            const V3 color = sample_colour(sc.colors, sample_out);
            if (in_part) {
                float *mine = tile_rgb + 3u * (row * 8u + col);
                mine[0] = color.x;
                mine[1] = color.y;
                mine[2] = color.z;
            }
            __syncthreads();
            if (stored[s] && sx0 % k == 0u && sy % k == 0u) {
                float tx = 0.0f, ty = 0.0f, tz = 0.0f;
                for (uint32_t i = 0; i < k; ++i) {
                    for (uint32_t j = 0; j < k; ++j) {
                        const float *c = tile_rgb + 3u * ((row + j) * 8u + col + i);
                        tx += c[0];
                        ty += c[1];
                        tz += c[2];
                    }
                }
                store_pixel(tx, ty, tz, k, fr.width, sx0 / k, sy / k, fr.pixels_f32, fr.pixels_rgba8);
            }
        } else {
            samples[(size_t)sy * sw + sx0 + (uint32_t)s] = sample_out;
        }
    }

    if (fr.tile_cost) {
        // the workgroup's duration as its last wave sees it (waves of a workgroup start together)
        __syncthreads();
        if (threadIdx.x == 0) fr.tile_cost[tile] = (uint32_t)(wall_clock64() - started);
    }

    if (COUNT) {
        atomicAdd(&block_counts[0], (unsigned long long)cnt.sdf);
        atomicAdd(&block_counts[1], (unsigned long long)cnt.steps);
        atomicAdd(&block_counts[2], (unsigned long long)cnt.rays);
        atomicAdd(&block_counts[3], (unsigned long long)cnt.evals);
        atomicAdd(&block_counts[4], (unsigned long long)cnt.slots);
        atomicAdd(&block_counts[5], (unsigned long long)cnt.flops);
        __syncthreads();
        if (threadIdx.x < 16 && probe_words()[threadIdx.x]) atomicAdd(&counters->probe[threadIdx.x], probe_words()[threadIdx.x]);
        if (threadIdx.x == 0) {
            atomicAdd(&counters->sdf_calls, block_counts[0]);
            atomicAdd(&counters->march_steps, block_counts[1]);
            atomicAdd(&counters->rays, block_counts[2]);
            atomicAdd(&counters->prim_evals, block_counts[3]);
            atomicAdd(&counters->march_slots, block_counts[4]);
            atomicAdd(&counters->flops, block_counts[5]);
        }
    }
}

template <class Scene>
__device__ __forceinline__ void eval_sdf_body(const SceneView &sc, const float *__restrict__ xyz, uint32_t n,
                                              float *__restrict__ sd, uint32_t *__restrict__ prim, float *__restrict__ sd_march) {
    const Lds lds = stage_lds(sc);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t tt = t < n ? t : n - 1;   // every lane walks the scene (nested lists park through LDS)
    const V3 p[1] = { v3(xyz[3 * tt], xyz[3 * tt + 1], xyz[3 * tt + 2]) };
    float d[1];
    uint32_t pr[1];
    Scene::template surface<1>(sc, lds, p, d, pr);
    if (t < n) {
        sd[t] = d[0];
        prim[t] = pr[0];
    }
    if (sd_march) {   // the distance-only walker the march loop uses (bounds, rejection tests and all)
        float dm[1];
        Work work[1] = {};
        const float no_cap[1] = { __builtin_inff() };
        Scene::template sd<1, false>(sc, lds, p, dm, work, no_cap);
        if (t < n) sd_march[t] = dm[0];
    }
}


}  // namespace rmdev
