// trace_kernel.h -- the render kernel body, generic over the scene evaluator `Scene`
// (InterpretedScene, or the straight-line evaluator generated for one scene):
//     Scene::sd<NP, COUNT>(sc, lds, points, distances, work, cap, colourless)  distance only; colourless[n]: the value is
//                                                                  that of a surface known to have colour (0, 0, 0)
//     Scene::surface<NP>(sc, lds, points, distances, primitives)   distance + colour-carrying primitive
#pragma once

#include "device_core.h"

#ifndef RM_SPREAD_SHADOWS
#define RM_SPREAD_SHADOWS 1   // 0: the shadow rays of a tile part's samples march one light after the other on the sample's lane
#endif

namespace rmdev {

// the value lane `from` of the wave holds; `from` must be an executing lane (ds_bpermute returns 0 for the others), so
// this is called only where the whole wave executes
__device__ __forceinline__ float from_lane(float v, uint32_t from) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(from << 2), __float_as_int(v)));
}
__device__ __forceinline__ V3 from_lane(V3 v, uint32_t from) { return v3(from_lane(v.x, from), from_lane(v.y, from), from_lane(v.z, from)); }

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
// Camera rays have no light to pass: SHADOW = false leaves the test out (to_light and past_light are not read).
// Returns the number of times the calling lane went round the loop.
template <class Scene, int kPts, bool COUNT, bool SHADOW>
__device__ __forceinline__ int march(const SceneView &sc, const Lds &lds, const FrameView &fr, V3 (&p)[kPts],
                                      const V3 (&ray)[kPts], V3 (&at)[kPts], const bool (&live)[kPts], float (&to_light)[kPts],
                                      const float (&past_light)[kPts], LaneCounts &cnt, bool (&colourless)[kPts], int iterations,
                                      bool first_is_shared) {
    bool done[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        done[s] = !live[s];
        colourless[s] = false;
        if (COUNT && live[s]) cnt.rays++;
    }
    // Upper bound of the next evaluation's result, for the scene walkers that can use one (see `capped` in
    // device_core.h).  The scene SDF is 1-Lipschitz -- spheres in the Euclidean norm, boxes and crosses in
    // the maximum norm, which is smaller, and min, max and negation keep the property -- so after the
    // step p' = p + ray * sd, |p' - p| = |sd| (1 + 1e-7), the value at p' is at most sd + |sd|; 1e-5
    // relative and 4e-6 of the coordinates' magnitude cover the roundings of the step, of |ray| = 1 and
    // of the evaluations on both sides (each below 1e-6 of the magnitudes involved: the point's coordinates and, through
    // SceneView::cap_slack, the centre and size of whichever primitive decides the value, however far or large).
    float cap[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) cap[s] = __builtin_inff();
    int it = 0;
    for (; it < iterations; ++it) {
        bool all_done = true;
#pragma unroll
        for (int s = 0; s < kPts; ++s) all_done = all_done && done[s];
        if (all_done) break;
        float sd[kPts];
        Work work[kPts] = {};
        bool colourless_here[kPts];
        Scene::template sd<kPts, COUNT>(sc, lds, p, sd, work, cap, colourless_here);
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
                colourless[s] = colourless_here[s];   // of the ray's last evaluation, in the end
                at[s] = p[s];
                p[s] = p[s] + ray[s] * sd[s];
                cap[s] = sd[s] + __builtin_fabsf(sd[s]) * 1.00001f +
                         4e-6f * (__builtin_fabsf(p[s].x) + __builtin_fabsf(p[s].y) + __builtin_fabsf(p[s].z) + __builtin_fabsf(sd[s])) + sc.cap_slack;
                if (SHADOW) to_light[s] -= sd[s];
                done[s] = sd[s] < fr.surface_eps || (SHADOW && to_light[s] < -past_light[s]);
                if (COUNT) {
                    cnt.sdf++;
                    cnt.steps++;
                    cnt.evals += work[s].evals;
                    cnt.flops += work[s].flops;
                    if (it == 0 && first_is_shared) cnt.shared_first_flops += work[s].flops;
                }
            }
        }
    }
    return it;
}

// scene::lighting / scene::shadow, scene.cpp:45-62: the shadow ray from a hit to a light, and what lets its march stop
// once it is past the light (see march)
__device__ __forceinline__ void shadow_ray(V3 light, V3 hit, V3 normal, float lift, const FrameView &fr, V3 &p, V3 &ray, float &to_light,
                                           float &past_light) {
    ray = normalize3(light - hit);
    p = hit + normal * lift;
    to_light = fr.skip_unlit ? dot3(light - p, ray) : __builtin_inff();
    const float scale = __builtin_fabsf(light.x) + __builtin_fabsf(light.y) + __builtin_fabsf(light.z) +
                        __builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + __builtin_fabsf(to_light);
    past_light = 0.01f + (1e-4f + 4e-7f * (float)fr.iterations) * scale;
}

// A workgroup is n x n waves (n = 1, 2 or 3, from the launch) and renders one tile of 8n x 8n samples (16n x 8n with two
// samples per lane), each wave an 8 x 8 block of it.
// kPts = 2: each lane two horizontally adjacent samples that march in lockstep -- halves the scalar-unit cost per
// sample, which bounds scenes of few records; a sample that finishes early idles until its partner is done.
// kPts = 1: one sample per lane -- for scenes whose cost is in long primitive runs or lattice walks (vector-ALU bound),
// where idling would cost more than the scalar unit saves.
//
// Pixels from the trace kernel (FrameView::pixels_*): when the SSAA kernel k divides the tile's side, a tile holds whole
// pixels; its samples meet in LDS and the wave that finishes last adds them up and writes the pixels (k = 1, 2, 4, 8:
// one wave per tile; k = 3, 6, 12, 24: 3 x 3 waves).  No wave waits for another: each books its arrival with an LDS
// atomic after its samples are in LDS and ends, unless it was the last.
//
// Tile parts (tile_order entries above kTileBits): quarter q of a tile is its (q & 1, q >> 1) sub-tile of half the
// side, sixteenth q the (q & 3, q >> 2) sub-tile of a quarter of the side; the workgroup's waves split the sub-tile
// among them as they split a tile, so a wave renders 4 x 4 or 2 x 2 samples on its first 16 or 4 lanes and waits
// only for those.
//
// Shadow rays of a part, spread over the wave: a part exists because its samples' rays are the longest of the frame,
// and the shadow rays of one sample (scene::lighting's loop, scene.cpp:52-62) do not depend on each other.  With B
// samples on the first B lanes and G = min(64 / B, lights) groups of B lanes, lane g * B + s marches the ray of
// sample s to light `first + g` of the round, all G at once, and the sample's own lane then adds the contributions
// up in the order of the lights, as the reference's loop does.  The serial path of a sample goes from 1 + L marches
// to 1 + ceil(L / G); every ray is the same ray, so the frame and the work counters do not change.
//
// forced_tile (kNoForcedTile: none): the workgroup renders that whole tile, whatever its place in the grid (the exact
// kernel's second pass over flagged tiles, trace_exact_body).
constexpr uint32_t kNoForcedTile = 0xffffffffu;

template <class Scene, int kPts, bool COUNT>
__device__ __forceinline__ void trace_body(const SceneView &sc, const FrameView &fr, Sample *__restrict__ samples,
                                           WorkCounters *__restrict__ counters, uint32_t forced_tile = kNoForcedTile) {
    __shared__ uint32_t arrived;
    if (threadIdx.x == 0) {
        arrived = 0u;
        if (Scene::kOptimistic) *exact_wanted_word() = 0u;
    }
    if (COUNT) {
        if (threadIdx.x < 16) probe_words()[threadIdx.x] = 0;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t n_side = n_waves == 1u ? 1u : (n_waves == 4u ? 2u : 3u);   // waves per tile side
    const uint32_t wave_x = n_side == 1u ? 0u : wave % n_side, wave_y = n_side == 1u ? 0u : wave / n_side;
    const uint32_t tile_side = 8u * n_side;                                  // samples per tile side (x: times kPts)
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    const uint32_t tiles_x = (sw + tile_side * (uint32_t)kPts - 1u) / (tile_side * (uint32_t)kPts);   // = gridDim.x of a frame's launch
    const uint32_t slot = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t n_tiles = tiles_x * ((sh + tile_side - 1u) / tile_side);
    if (forced_tile == kNoForcedTile && slot >= n_tiles + (fr.tile_order ? fr.extra_slots : 0u)) return;   // the last grid row's surplus
    const Lds lds = stage_lds(sc);
    const unsigned long long started = (fr.tile_cost || fr.slot_stats) ? wall_clock64() : 0ull;
    int camera_turns = 0, shadow_turns = 0;
    const uint32_t entry = forced_tile != kNoForcedTile ? forced_tile : (fr.tile_order ? ((ConstIndex)fr.tile_order)[slot] : slot);
    const uint32_t tile = entry & kTileMask, part = kPts == 1 ? entry >> kTileBits : 0u;
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    // the part's place in the tile, the wave's block in the part, the lane's sample in the block (all in samples)
    const uint32_t split_log2 = part == 0u ? 0u : (part < 5u ? 1u : 2u);      // parts per tile side: 1, 2 or 4
    const uint32_t q = part == 0u ? 0u : (part < 5u ? part - 1u : part - 5u);
    const uint32_t block = 8u >> split_log2;                                  // the wave's block is block x block samples
    const uint32_t part_side = tile_side >> split_log2;
    const uint32_t part_x = (q & ((1u << split_log2) - 1u)) * part_side, part_y = (q >> split_log2) * part_side;
    const uint32_t col = part_x + wave_x * block + (lane & (block - 1u));
    const uint32_t row = part_y + wave_y * block + ((lane >> (3u - split_log2)) & (block - 1u));
    const bool in_part = lane < block * block;
    // shadow rays spread over the wave's idle lanes (wave-uniform; see above)
    const uint32_t part_lanes = block * block;                                // B: 16 or 4 in a part
    const bool spread = RM_SPREAD_SHADOWS != 0 && kPts == 1 && part != 0u && sc.n_lights >= 2u;
    const uint32_t groups = spread ? (64u / part_lanes < sc.n_lights ? 64u / part_lanes : sc.n_lights) : 1u;   // G
    const uint32_t rounds = spread ? (sc.n_lights + groups - 1u) / groups : sc.n_lights;
    // the lane of the sample whose shadow rays this lane marches, and which of the round's lights it takes: found anew
    // where they are needed (the empty asm keeps the compiler from holding them in registers across the marches)
    const auto lane_now = [] {
        uint32_t l = threadIdx.x & 63u;
        asm volatile("" : "+v"(l));
        return l;
    };
    const uint32_t sx0 = (tile_x * tile_side + col) * (uint32_t)kPts;
    const uint32_t sy = tile_y * tile_side + row;
    LaneCounts cnt = { 0, 0, 0, 0, 0, 0, 0 };

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

    // ---- scene::raymarch, scene.cpp:26-32: the camera ray, then one shadow ray per light
    // (scene::lighting / scene::shadow, scene.cpp:45-62).
    V3 hit[kPts], normal[kPts], p[kPts], ray[kPts], at[kPts];
    float lighting[kPts], to_light[kPts] = {}, past_light[kPts] = {};
    uint32_t prim[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        hit[s] = origin;
        normal[s] = v3(0.0f, 0.0f, 0.0f);
        lighting[s] = 0.0f;
        prim[s] = kNoPrim;
    }
    const float lift = fr.surface_eps + fr.offset_eps;
    // The camera ray and what follows it at the hit (colour, normal), then the shadow rays.  Two march instances: the
    // camera rays need neither the test against a light nor the cap of a shadow ray's first step.
    do {
        bool colourless[kPts];   // the ray ended on a surface the distance evaluator knows to have colour (0, 0, 0)
        {
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
            }
#pragma unroll
            for (int s = 0; s < kPts; ++s) at[s] = p[s];
            int iterations = fr.iterations;
            const bool first_is_shared = sc.camera_sd_known != 0u && sc.camera_sd >= fr.surface_eps && iterations > 0;
            if (!COUNT && first_is_shared) {
                // Every camera ray starts with the same evaluation at the camera's position: its value was found when the scene
                // was uploaded (SceneView::camera_sd); not below surface_eps, so no ray ends there, and the step is taken without it.
#pragma unroll
                for (int s = 0; s < kPts; ++s) p[s] = p[s] + ray[s] * sc.camera_sd;
                iterations--;
            }
            camera_turns = march<Scene, kPts, COUNT, false>(sc, lds, fr, p, ray, at, lit, to_light, past_light, cnt, colourless, iterations,
                                                            first_is_shared);
        }
        {
#pragma unroll
            for (int s = 0; s < kPts; ++s) hit[s] = p[s];
            // the Surface scene::surface returned: colour of the last evaluation -- from the colour-tracking walk, unless
            // the distance walk has told already that there is none (the wall behind the scene: most waves of most frames)
            bool unknown = false;
#pragma unroll
            for (int s = 0; s < kPts; ++s) unknown = unknown || (live[s] && !colourless[s]);
            if (fr.iterations > 0 && __builtin_amdgcn_ballot_w64(unknown) != 0) {
                float sd_unused[kPts];
                Scene::template surface<kPts>(sc, lds, at, sd_unused, prim);
            }
#pragma unroll
            for (int s = 0; s < kPts; ++s) prim[s] = colourless[s] ? kNoPrim : prim[s];
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
                    bool colourless_unused[kPts];
                    Scene::template sd<kPts, COUNT>(sc, lds, q, sd, work, no_cap, colourless_unused);
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
            if (spread) {
                // every lane takes the hit, the normal and the lit flag of the sample whose shadow rays it will march
                // (lanes of the part itself: their own)
                const uint32_t source = lane_now() & (part_lanes - 1u);
                hit[0] = from_lane(hit[0], source);
                normal[0] = from_lane(normal[0], source);
                lit[0] = from_lane(lit[0] ? 1.0f : 0.0f, source) != 0.0f;
            }
        }
#pragma unroll 1
        for (uint32_t r = 1; r <= rounds; ++r) {
            V3 light = v3(0.0f, 0.0f, 0.0f);
            bool marching[kPts];   // the rays of this round
#pragma unroll
            for (int s = 0; s < kPts; ++s) marching[s] = lit[s];
            if (spread) {
                // the lane's own light of this round (hit, normal and lit are its source sample's by now)
                const uint32_t group = lane_now() / part_lanes, mine = (r - 1u) * groups + group;
                const Float4 lp = sc.lights[mine < sc.n_lights ? mine : sc.n_lights - 1u];
                const V3 own = v3(lp.x, lp.y, lp.z);
                marching[0] = lit[0] && group < groups && mine < sc.n_lights;
                shadow_ray(own, hit[0], normal[0], lift, fr, p[0], ray[0], to_light[0], past_light[0]);
            } else {
                const RowWords lp = ((ConstRows)sc.lights)[r - 1];
                light = v3(lp[0], lp[1], lp[2]);
#pragma unroll
                for (int s = 0; s < kPts; ++s) shadow_ray(light, hit[s], normal[s], lift, fr, p[s], ray[s], to_light[s], past_light[s]);
            }
#pragma unroll
            for (int s = 0; s < kPts; ++s) at[s] = p[s];
            bool colourless_unused[kPts];
            shadow_turns += march<Scene, kPts, COUNT, true>(sc, lds, fr, p, ray, at, marching, to_light, past_light, cnt, colourless_unused,
                                                            fr.iterations, false);
            if (spread) {
                const uint32_t now = lane_now(), source = now & (part_lanes - 1u), mine = (r - 1u) * groups + now / part_lanes;
                const Float4 lp = sc.lights[mine < sc.n_lights ? mine : sc.n_lights - 1u];
                const float unshadowed = dot3(v3(lp.x, lp.y, lp.z) - p[0], ray[0]) > 0 ? 0.0f : 1.0f;
                const float towards = dot3(normal[0], ray[0]);
                // the sample's lane adds its rays' contributions in the order of the lights (scene.cpp:55-59)
                for (uint32_t g = 0; g < groups && (r - 1u) * groups + g < sc.n_lights; ++g) {
                    const float open = from_lane(unshadowed, g * part_lanes + source);
                    const float term = from_lane(towards, g * part_lanes + source);
                    if (in_part && open != 0.0f) lighting[0] += term;
                }
            } else {
#pragma unroll
                for (int s = 0; s < kPts; ++s) {
                    const bool shadowed = dot3(light - p[s], ray[s]) > 0;
                    if (!shadowed) lighting[s] += dot3(normal[s], ray[s]);
                }
            }
        }
    } while (false);

    // Whole pixels in the workgroup (see FrameView::pixels_f32): the samples meet in LDS and the wave that arrives last
    // adds them up, every pixel in the reference's order (render.cpp:103-112: i, the column, outer; j inner).
    const bool fused = kPts == 1 && (fr.pixels_f32 || fr.pixels_rgba8);
    float *tile_rgb = reinterpret_cast<float *>(lds.park);   // [tile_side x tile_side][3], behind the lattice tables
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        if (!stored[s] && !fused) continue;
        // scene::lighting's clamp (scene.cpp:60); the product with the colour is formed where the samples are added up
        Sample sample_out = { 0.0f, kNoPrim };
        if (live[s]) sample_out = Sample{ __builtin_fminf(__builtin_fmaxf(fr.saturation, lighting[s]), 1.0f), prim[s] };
        if (fused) {
            if (in_part) {
                const V3 color = sample_colour(sc.colors, sample_out);
                float *mine = tile_rgb + 3u * (row * tile_side + col);
                mine[0] = color.x;
                mine[1] = color.y;
                mine[2] = color.z;
            }
        } else {
            samples[(size_t)sy * sw + sx0 + (uint32_t)s] = sample_out;
        }
    }

    if (COUNT) {
        // per wave: its lanes' sums, then one atomic per counter
        unsigned long long sums[7] = { cnt.sdf, cnt.steps, cnt.rays, cnt.evals, cnt.slots, cnt.flops, cnt.shared_first_flops };
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            unsigned long long v = sums[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            sums[i] = v;
        }
        if (lane == 0) {
            atomicAdd(&counters->sdf_calls, sums[0]);
            atomicAdd(&counters->march_steps, sums[1]);
            atomicAdd(&counters->rays, sums[2]);
            atomicAdd(&counters->prim_evals, sums[3]);
            atomicAdd(&counters->march_slots, sums[4]);
            atomicAdd(&counters->flops, sums[5]);
            atomicAdd(&counters->flops_shared_first, sums[6]);
        }
    }

    // The wave's samples are in LDS (or on their way to memory): it books its arrival and ends, unless it is the
    // workgroup's last, which writes the pixels and the tile's duration.
    if (Scene::kOptimistic) {
        // some walk of this workgroup went on with a lower bound (every wave that finds the word set says so: the wave that
        // set it is among them): the exact kernel renders the tile again
#ifdef RM_FLAG_EVERY_TILE   // (tests: the exact kernel renders the whole frame over the optimistic one's)
        if (fr.tile_flags && lane == 0u) fr.tile_flags[tile] = 1u;
#else
        if (fr.tile_flags && *exact_wanted_word() != 0u && lane == 0u) fr.tile_flags[tile] = 1u;
#endif
    }
    if (fr.slot_stats) {
        // the wave's longest marches (a lane leaves the loop when its own ray is done), booked by the first wave only
        int cam = camera_turns, shd = shadow_turns;
        for (int off = 32; off > 0; off >>= 1) {
            const int c2 = __shfl_down(cam, off, 64), s2 = __shfl_down(shd, off, 64);
            cam = c2 > cam ? c2 : cam;
            shd = s2 > shd ? s2 : shd;
        }
        if (threadIdx.x == 0) {
            uint32_t *out = fr.slot_stats + 4u * slot;
            out[0] = entry;
            out[1] = (uint32_t)(wall_clock64() - started);
            out[2] = (uint32_t)(cam > 65535 ? 65535 : cam) | (uint32_t)(shd > 65535 ? 65535 : shd) << 16;
            out[3] = (uint32_t)started;
        }
    }
    if (fr.tile_cost == nullptr && !fused && !COUNT) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t before = 0u;
    if (lane == 0u) before = __hip_atomic_fetch_add(&arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (before + 1u != n_waves) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (fused) {
        const uint32_t pixels = part_side / k;                 // part_side is a multiple of k (the host lists no other parts)
        for (uint32_t i = lane; i < pixels * pixels; i += 64u) {
            const uint32_t px = i % pixels, py = i / pixels;
            const uint32_t lx = part_x + px * k, lrow = part_y + py * k;          // the pixel's first sample, in the tile
            const uint32_t gx = tile_x * tile_side + lx, gy = tile_y * tile_side + lrow;
            if (gx >= sw || gy >= sh) continue;
            float tx = 0.0f, ty = 0.0f, tz = 0.0f;
            for (uint32_t ii = 0; ii < k; ++ii) {
                for (uint32_t jj = 0; jj < k; ++jj) {
                    const float *c = tile_rgb + 3u * ((lrow + jj) * tile_side + lx + ii);
                    tx += c[0];
                    ty += c[1];
                    tz += c[2];
                }
            }
            const uint32_t out_y = output_row(fr, gy / k);
            if (out_y != kNoOutputRow) store_pixel(tx, ty, tz, k, fr.width, gx / k, out_y, fr.pixels_f32, fr.pixels_rgba8);
        }
    }
    if (fr.tile_cost && lane == 0u) fr.tile_cost[tile] = (uint32_t)(wall_clock64() - started);
    if (COUNT) {
        if (lane < 16u && probe_words()[lane]) atomicAdd(&counters->probe[lane], probe_words()[lane]);
    }
}

// The exact kernel's pass behind an optimistic launch: a workgroup looks at up to 64 tiles of the launch's sample image and
// renders those whose word in FrameView::tile_flags is set, whole and in natural order, with a scene whose every walk is
// exact; the words are cleared on the way.  Launched with the frame's workgroup shape; almost always every workgroup
// finds nothing to render.
// Its tiles are 64 in a row of the tile numbering (workgroup b: tiles 64 b .. 64 b + 63), or, where the launch also
// resolves the sample image into pixels (out_f32 or out_rgba8 given; SSAA kernels that do not divide the tile's side, k x k
// <= 64), the k x k tiles that hold the samples of its (tile side)^2 pixels -- workgroup (bx, by): tiles bx k .. bx k + k - 1
// by by k .. by k + k - 1 -- one pixel per thread (render.cpp:103-119: the colours of the pixel's k x k samples are added up
// over i (outer, x) then j (inner, y), divided by k x k).  One launch then takes the place of two.
template <class Scene, int kPts>
__device__ __forceinline__ void trace_exact_body(const SceneView &sc, const FrameView &fr, Sample *__restrict__ samples,
                                                 float *__restrict__ out_f32, unsigned char *__restrict__ out_rgba8) {
    __shared__ unsigned long long flagged;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t tile_side = 8u * (n_waves == 1u ? 1u : (n_waves == 4u ? 2u : 3u));
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k, rows = fr.band_count * fr.band_rows, sh = rows * k;
    const uint32_t tiles_x = (sw + tile_side * (uint32_t)kPts - 1u) / (tile_side * (uint32_t)kPts), tiles_y = (sh + tile_side - 1u) / tile_side;
    const bool resolving = out_f32 != nullptr || out_rgba8 != nullptr;
    // the i-th of this workgroup's tiles (kNoForcedTile: there is no such tile)
    const auto tile_of = [&](uint32_t i) -> uint32_t {
        if (!resolving) return blockIdx.x * 64u + i < tiles_x * tiles_y ? blockIdx.x * 64u + i : kNoForcedTile;
        const uint32_t tx = blockIdx.x * k + i % k, ty = blockIdx.y * k + i / k;
        return (i < k * k && tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : kNoForcedTile;
    };
    if (threadIdx.x < 64u) {
        const uint32_t t = tile_of(threadIdx.x);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(t != kNoForcedTile && fr.tile_flags[t] != 0u);
        if (threadIdx.x == 0u) flagged = mask;
    }
    __syncthreads();
    unsigned long long todo = flagged;   // the same for every wave
    for (; todo != 0ull; todo &= todo - 1ull) {
        const uint32_t tile = tile_of((uint32_t)__builtin_ctzll(todo));
        if (threadIdx.x == 0u) {
            fr.tile_flags[tile] = 0u;
            if (fr.exact_count) atomicAdd(fr.exact_count, 1u);
        }
        trace_body<Scene, kPts, false>(sc, fr, samples, nullptr, tile);
        __syncthreads();   // the tile's LDS (tables, sample colours, counters) is free for the next one; its samples are written
    }
    if (!resolving) return;
    const size_t row_samples = sw;
    for (uint32_t q = 0; q < (uint32_t)kPts; ++q) {
        const uint32_t x = (blockIdx.x * (uint32_t)kPts + q) * tile_side + threadIdx.x % tile_side, y = blockIdx.y * tile_side + threadIdx.x / tile_side;
        if (x >= fr.width || y >= rows) continue;
        const uint32_t out_y = output_row(fr, y);
        if (out_y == kNoOutputRow) continue;
        float tx = 0.0f, ty = 0.0f, tz = 0.0f;
        for (uint32_t i = 0; i < k; ++i) {
            for (uint32_t j = 0; j < k; ++j) {
                const V3 c = sample_colour(sc.colors, samples[((size_t)y * k + j) * row_samples + (size_t)x * k + i]);
                tx += c.x;
                ty += c.y;
                tz += c.z;
            }
        }
        store_pixel(tx, ty, tz, k, fr.width, x, out_y, out_f32, out_rgba8);
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
        bool colourless_unused[1];
        Scene::template sd<1, false>(sc, lds, p, dm, work, no_cap, colourless_unused);
        if (t < n) sd_march[t] = dm[0];
    }
}


}  // namespace rmdev
