// trace_refill.h -- the render kernel body in which no lane waits for its neighbours' rays: a wave owns a tile of
// samples and every lane, as soon as its ray has ended, goes on with its sample's next ray or takes the tile's
// next sample (the reference's compute shader is the opposite: one pixel per invocation, its samples and rays one
// after the other, /root/reference/source/shaders/shader.comp:353-365).
//
// Per lane a small state machine over what scene::raymarch does for one sample (scene.cpp:26-62), every
// state one evaluation of the scene's distance per turn of the loop:
//     CAMERA      one step of the camera ray (scene::surface, scene.cpp:34-42)
//     TAP 0..5    one tap of scene::grad (scene.cpp:70-89)
//     SHADOW l    one step of the shadow ray to light l (scene::shadow, scene.cpp:45-50)
// so that one call of the distance evaluator per turn serves all lanes, whatever they are at.  The per-sample
// arithmetic and its order are those of trace_body (trace_kernel.h); only who computes what when differs.
//
// The colour of the surface (the primitive scene::surface's last evaluation came from) needs the colour-tracking
// evaluator, another piece of code: a lane whose camera ray has ended parks the point of that evaluation in an LDS
// queue and goes on; whenever 64 points are parked (and at the end) the whole wave runs the tracking evaluator
// once, a point per lane.  Samples on a surface that the distance evaluator already knows to be colourless
// (Scene::sd's `colourless`) are finished on the spot when colourless samples skip their shading (FrameView::skip_unlit).
#pragma once

#include "trace_kernel.h"

namespace rmdev {

constexpr int kLaneIdle = -1, kLaneCamera = 0, kLaneTap = 1, kLaneShadow = 7;   // TAP t = 1 + t, SHADOW l = 7 + l
constexpr uint32_t kParkedMax = 128;                                           // LDS queue of points awaiting their colour

// a tile is 2^w_log2 x 2^h_log2 samples; its samples are handed out block by block of 8 x 8, row-major inside
__device__ __forceinline__ void tile_sample(uint32_t index, uint32_t w_log2, uint32_t &x, uint32_t &y) {
    const uint32_t block = index >> 6, inside = index & 63u;
    const uint32_t blocks_x_log2 = w_log2 - 3u;
    x = ((block & ((1u << blocks_x_log2) - 1u)) << 3) + (inside & 7u);
    y = ((block >> blocks_x_log2) << 3) + (inside >> 3);
}

template <class Scene, bool COUNT>
__device__ __forceinline__ void trace_refill_body(const SceneView &sc, const FrameView &fr, Sample *__restrict__ samples,
                                                  WorkCounters *__restrict__ counters) {
    if (COUNT) {
        if (threadIdx.x < 16) probe_words()[threadIdx.x] = 0;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    const uint32_t w_log2 = fr.refill_w_log2, h_log2 = fr.refill_h_log2;
    const uint32_t slot = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t n_tiles = gridDim.x * ((sh + (1u << h_log2) - 1u) >> h_log2);   // gridDim.x = tiles per row
    if (slot >= n_tiles) return;                                                   // the last grid row's surplus
    const Lds lds = stage_lds(sc);
    const unsigned long long started = fr.tile_cost ? wall_clock64() : 0ull;
    const uint32_t tile = (fr.tile_order ? ((ConstIndex)fr.tile_order)[slot] : slot) & kTileMask;
    const uint32_t tile_y = tile / gridDim.x, tile_x = tile - tile_y * gridDim.x;
    const uint32_t total = 1u << (w_log2 + h_log2);
    float *parked = reinterpret_cast<float *>(lds.park);   // [kParkedMax][x, y, z, sample address]
    LaneCounts cnt = { 0, 0, 0, 0, 0, 0, 0 };

    // ---- render::pixel, render.cpp:82-101: what all samples share
    const float AR = (float)fr.width / (float)fr.height;
    const ConstFloats cam = (ConstFloats)sc.camera;   // 16 matrix floats, then the focal length
    const float w = cam[16];
    const float h = w / AR;
    const float s1x = -w / 2, s1y = h / 2;
    const float s2x = w / 2, s2y = -h / 2;
    const float psx = 1.0f / (float)fr.width, psy = 1.0f / (float)fr.height;
    const V3 origin = camera_view(cam, v3(0.0f, 0.0f, 0.0f), 1.0f);
    const float lift = fr.surface_eps + fr.offset_eps;
    const float hh = 1e-3f, two_h = 2 * hh;
    const bool early_out = fr.skip_unlit != 0u && fr.saturation >= 0.0f;

    // ---- the lane's sample and ray
    int phase = kLaneIdle;
    uint32_t address = 0;        // of the sample in `samples`
    int steps = 0;               // evaluations of the current ray so far
    V3 p = origin, ray = origin, hit = origin, nrm = v3(0.0f, 0.0f, 0.0f);   // nrm: the gradient while it is tapped, then the normal
    float plus = 0.0f, lighting = 0.0f, to_light = 0.0f, past_light = 0.0f, cap = 0.0f;
    uint32_t next = 0;           // wave-uniform: samples of the tile handed out so far
    uint32_t parked_first = 0, parked_count = 0;   // wave-uniform: the queue

    for (;;) {
        // ---- idle lanes take the tile's next samples
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(phase == kLaneIdle);
        if (idle != 0ull && next < total) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t index = next + rank;
            next += (uint32_t)__builtin_popcountll(idle);
            if (phase == kLaneIdle && index < total) {
                uint32_t lx, ly_;
                tile_sample(index, w_log2, lx, ly_);
                const uint32_t sx = (tile_x << w_log2) + lx, sy = (tile_y << h_log2) + ly_;
                if (sx < sw && sy < sh) {
                    address = sy * sw + sx;
                    const uint32_t ly = sy / k, j = sy - ly * k;   // local pixel row, sub-sample row (inner loop index)
                    const uint32_t band = ly / fr.band_rows;
                    const uint32_t image_band = fr.band_list ? ((ConstIndex)fr.band_list)[band < fr.band_count ? band : fr.band_count - 1u]
                                                             : fr.first_band + band * fr.band_stride;
                    const uint32_t pi = image_band * fr.band_rows + (ly - band * fr.band_rows);
                    if (pi >= fr.height) {
                        samples[address] = Sample{ 0.0f, kNoPrim };   // a band beyond the image: black
                    } else {
                        const float uv1y = (float)pi * psy, uv2y = (float)(pi + 1u) * psy;
                        const float p1y = lerp1(s1y, s2y, uv1y), p2y = lerp1(s1y, s2y, uv2y);
                        const float uvy = (float)(j + 1u) / (float)k;
                        const float y = lerp1(p1y, p2y, uvy);
                        const uint32_t pj = sx / k, i = sx - pj * k;   // pixel column, sub-sample column (outer loop index)
                        const float uv1x = (float)pj * psx, uv2x = (float)(pj + 1u) * psx;
                        const float p1x = lerp1(s1x, s2x, uv1x), p2x = lerp1(s1x, s2x, uv2x);
                        const float uvx = (float)(i + 1u) / (float)k;
                        const float x = lerp1(p1x, p2x, uvx);
                        p = origin;
                        ray = camera_view(cam, normalize3(v3(x, y, -1.0f)), 0.0f);
                        to_light = __builtin_inff();
                        past_light = 0.0f;
                        cap = __builtin_inff();
                        lighting = 0.0f;
                        steps = 0;
                        phase = kLaneCamera;
                        if (COUNT) cnt.rays++;
                    }
                }
            }
        }
        const unsigned long long busy = __builtin_amdgcn_ballot_w64(phase != kLaneIdle);

        // ---- the colour of parked points: when a wave's worth is waiting, or nothing else is left to do
        if (parked_count >= 64u || (busy == 0ull && next >= total && parked_count != 0u)) {
            const uint32_t n = parked_count < 64u ? parked_count : 64u;
            if (lane < n) {
                const float *e = parked + 4u * ((parked_first + lane) & (kParkedMax - 1u));
                const V3 at[1] = { v3(e[0], e[1], e[2]) };
                const uint32_t where = __float_as_uint(e[3]);
                float sd_unused[1];
                uint32_t prim[1];
                Scene::template surface<1>(sc, lds, at, sd_unused, prim);
                samples[where].prim = prim[0];
            }
            parked_first = (parked_first + n) & (kParkedMax - 1u);
            parked_count -= n;
            continue;
        }
        if (busy == 0ull) {
            if (next >= total) break;
            continue;
        }

        bool parks = false;   // the lane's camera ray ended this turn on a surface whose colour is yet to be found
        V3 q[1] = { p };
        if (phase != kLaneIdle) {
            // ---- one evaluation of the scene's distance where the lane's state wants it
            const bool tapping = phase >= kLaneTap && phase < kLaneShadow;
            if (tapping) {
                const int t = phase - kLaneTap, axis = t >> 1;
                const float e = (t & 1) ? -hh : hh;
                // scene::grad's points: hit + (h, 0, 0), hit - (h, 0, 0), ... (adding the zeros as the reference does)
                const V3 d = v3(axis == 0 ? hh : 0.0f, axis == 1 ? hh : 0.0f, axis == 2 ? hh : 0.0f);
                q[0] = e > 0.0f ? v3(hit.x + d.x, hit.y + d.y, hit.z + d.z) : v3(hit.x - d.x, hit.y - d.y, hit.z - d.z);
            }
            float sd[1];
            Work work[1] = {};
            const float cap_in[1] = { tapping ? __builtin_inff() : cap };
            bool colourless[1];
            Scene::template sd<1, COUNT>(sc, lds, q, sd, work, cap_in, colourless);
            probe<COUNT>(kProbeSceneCall, 1u);
            if (COUNT) {
                cnt.sdf++;
                cnt.evals += work[0].evals;
                cnt.flops += work[0].flops;
                const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
                if (lane == (uint32_t)__builtin_ctzll(here)) cnt.slots += 64u;
            }

            bool start_shadows = false;   // the sample's normal is known: set out for the first light
            int light_index = 0;
            if (tapping) {
                const int t = phase - kLaneTap, axis = t >> 1;
                if ((t & 1) == 0) {
                    plus = sd[0];
                } else {
                    const float diff = plus - sd[0];
                    nrm.x = axis == 0 ? diff : nrm.x;
                    nrm.y = axis == 1 ? diff : nrm.y;
                    nrm.z = axis == 2 ? diff : nrm.z;
                }
                phase++;
                if (phase == kLaneShadow) {
                    nrm = normalize3(v3(nrm.x / two_h, nrm.y / two_h, nrm.z / two_h));
                    start_shadows = true;
                }
            } else {
                // scene::surface's step (scene.cpp:37-39) and the ends a ray can come to
                if (COUNT) cnt.steps++;
                p = p + ray * sd[0];
                cap = sd[0] + __builtin_fabsf(sd[0]) * 1.00001f +
                      4e-6f * (__builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + __builtin_fabsf(sd[0])) + sc.cap_slack;
                to_light -= sd[0];
                steps++;
                const bool ended = sd[0] < fr.surface_eps || to_light < -past_light || steps >= fr.iterations;
                if (ended && phase == kLaneCamera) {
                    hit = p;
                    if (early_out && colourless[0]) {
                        // light * (0, 0, 0): nothing the normal and the shadow rays could change (scene.cpp:31, :60)
                        samples[address] = Sample{ 0.0f, kNoPrim };
                        phase = kLaneIdle;
                    } else {
                        // the Surface scene::surface returned belongs to the point of this last evaluation: parked below
                        parks = true;
                        nrm = v3(0.0f, 0.0f, 0.0f);
                        phase = kLaneTap;
                    }
                } else if (ended) {
                    const int l = phase - kLaneShadow;
                    const RowWords lp = ((ConstRows)sc.lights)[l];
                    const V3 light = v3(lp[0], lp[1], lp[2]);
                    const bool shadowed = dot3(light - p, ray) > 0;
                    if (!shadowed) lighting += dot3(nrm, ray);
                    start_shadows = true;
                    light_index = l + 1;
                }
            }
            if (start_shadows) {
                if ((uint32_t)light_index < sc.n_lights) {
                    const RowWords lp = ((ConstRows)sc.lights)[light_index];
                    const V3 light = v3(lp[0], lp[1], lp[2]);
                    ray = normalize3(light - hit);
                    p = hit + nrm * lift;
                    to_light = fr.skip_unlit ? dot3(light - p, ray) : __builtin_inff();
                    const float scale = __builtin_fabsf(light.x) + __builtin_fabsf(light.y) + __builtin_fabsf(light.z) +
                                        __builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + __builtin_fabsf(to_light);
                    past_light = 0.01f + (1e-4f + 4e-7f * (float)fr.iterations) * scale;
                    cap = __builtin_inff();
                    steps = 0;
                    phase = kLaneShadow + light_index;
                    if (COUNT) cnt.rays++;
                } else {
                    // scene::lighting's clamp (scene.cpp:60); the product with the colour is formed where the samples are added up
                    samples[address].light = __builtin_fminf(__builtin_fmaxf(fr.saturation, lighting), 1.0f);
                    phase = kLaneIdle;
                }
            }
        }
        // ---- points parked this turn join the queue (at most 63 were waiting, at most 64 join)
        const unsigned long long parking = __builtin_amdgcn_ballot_w64(parks);
        if (parking != 0ull) {
            if (parks) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(parking >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)parking, 0u));
                float *e = parked + 4u * ((parked_first + parked_count + rank) & (kParkedMax - 1u));
                e[0] = q[0].x;
                e[1] = q[0].y;
                e[2] = q[0].z;
                e[3] = __uint_as_float(address);
            }
            parked_count += (uint32_t)__builtin_popcountll(parking);
        }
    }

    if (fr.tile_cost && lane == 0u) fr.tile_cost[tile] = (uint32_t)(wall_clock64() - started);
    if (COUNT) {
        unsigned long long sums[6] = { cnt.sdf, cnt.steps, cnt.rays, cnt.evals, cnt.slots, cnt.flops };
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            unsigned long long v = sums[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            sums[i] = v;
        }
        if (lane == 0u) {
            atomicAdd(&counters->sdf_calls, sums[0]);
            atomicAdd(&counters->march_steps, sums[1]);
            atomicAdd(&counters->rays, sums[2]);
            atomicAdd(&counters->prim_evals, sums[3]);
            atomicAdd(&counters->march_slots, sums[4]);
            atomicAdd(&counters->flops, sums[5]);
        }
        if (lane < 16u && probe_words()[lane]) atomicAdd(&counters->probe[lane], probe_words()[lane]);
    }
}

}  // namespace rmdev
