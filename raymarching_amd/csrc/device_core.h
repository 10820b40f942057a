// device_core.h -- device-side building blocks of the sphere-tracing kernels: vector helpers,
// primitive distances (body.cpp:32-57), folds, the exact LATTICE evaluator and the LDS layout.
// Included by the ahead-of-time kernels (kernels.hip) and, as text, by the kernels that are
// specialised for one scene at upload time (jit.cpp); it therefore includes nothing but
// device_scene.h.
#pragma once

#include "device_scene.h"

namespace rmdev {

constexpr int kMaxPts = 2;   // most points per lane and scene walk (sizes the LDS parking area)

struct V3 {
    float x, y, z;
};

typedef uint32_t RecWords __attribute__((ext_vector_type(8)));   // one Rec as a single 32-byte scalar load

// Scene tables are never written while a kernel runs: reading them through the constant address
// space keeps their wave-uniform loads on the scalar unit whatever else the kernel does.
typedef const __attribute__((address_space(4))) RecWords *ConstRecs;
typedef float RowWords __attribute__((ext_vector_type(4)));     // one Float4 table row
typedef const __attribute__((address_space(4))) RowWords *ConstRows;
typedef const __attribute__((address_space(4))) uint32_t *ConstIndex;
typedef const __attribute__((address_space(4))) float *ConstFloats;

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

// ---- LDS: lattice tables, then the parking area of nested lists (record walker) or the tile's sample colours
// (kernels generated for a scene, which keep nested lists in registers, when they write pixels themselves) -------

extern __shared__ uint32_t lds_words[];

struct Lds {
    const uint32_t *lattice;
    uint32_t *park;   // record walker: [slot][sd | prim][point][thread]; generated kernels: [sample of the tile][rgb]
};

__device__ __forceinline__ Lds stage_lds(const SceneView &sc) {
    for (uint32_t i = threadIdx.x; i < sc.lattice_words; i += blockDim.x) lds_words[i] = sc.lattice[i];
    __syncthreads();
    return Lds{ lds_words, lds_words + sc.lattice_words };
}

// dynamic LDS of the record-walking kernels
__host__ __device__ inline uint32_t lds_bytes(const SceneView &sc) {
    return (sc.lattice_words + sc.max_depth * 2u * (uint32_t)kMaxPts * 256u) * 4u;
}

// ---- occupancy probes (work-counting kernel variants only) ------------------------------------
// Region r books, per wave that executes it, (lanes executing it) x weight into probe[2r] and 64 x weight into
// probe[2r + 1]: their ratio is the region's lane utilisation, weight = the region's cost in instructions or cells.
constexpr int kProbeLevel0 = 0, kProbeLevel1 = 1, kProbeLevel2 = 2, kProbeLevelDeep = 3, kProbeRowCells = 4, kProbeGradTap = 5,
              kProbeClusteredRuns = 6, kProbeSceneCall = 7;

__device__ __forceinline__ unsigned long long *probe_words() {
    __shared__ unsigned long long words[16];
    return words;
}

template <bool COUNT>
__device__ __forceinline__ void probe(int region, uint32_t weight) {
    if (COUNT) {
        const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(here)) {
            atomicAdd(&probe_words()[2 * region], (unsigned long long)__builtin_popcountll(here) * weight);
            atomicAdd(&probe_words()[2 * region + 1], 64ull * weight);
        }
    }
}

// ---- primitive distances (body.cpp:32-35, 41-45, 51-57) --------------------

__device__ __forceinline__ float cross_value(float dx, float dy, float dz) {
    const float dmin = __builtin_fminf(__builtin_fminf(dx, dy), dz);
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
    return dx + dy + dz - dmin - dmax;
}

template <uint32_t KIND>
__device__ __forceinline__ float prim_distance(const Float4 a, const Float4 b, V3 p) {
    if (KIND == K_SPHERE) {
        const float dx = a.x - p.x, dy = a.y - p.y, dz = a.z - p.z;
        return __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - a.w;
    }
    const float hy = (KIND == K_BOX_ISO || KIND == K_CROSS_ISO) ? a.w : b.x;
    const float hz = (KIND == K_BOX_ISO || KIND == K_CROSS_ISO) ? a.w : b.y;
    const float dx = __builtin_fabsf(p.x - a.x) - a.w;
    const float dy = __builtin_fabsf(p.y - a.y) - hy;
    const float dz = __builtin_fabsf(p.z - a.z) - hz;
    if (KIND == K_BOX || KIND == K_BOX_ISO) return __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
    return cross_value(dx, dy, dz);
}

template <bool NEG, bool MAX>
__device__ __forceinline__ float fold_sd(float acc, float v) {
    return MAX ? __builtin_fmaxf(acc, NEG ? -v : v) : __builtin_fminf(acc, NEG ? -v : v);
}

// True if folding the sphere value v = sqrt(d2) - r into `acc` cannot change `acc`, decided without
// the square root (correctly rounded, it costs about as much as the rest of the sphere twice over).
// With S = sqrt(d2) and T = r - acc (negated folds) or acc + r:
//   min(acc, v), max(acc, -v)   change acc only if S < T:  no change if T < 0 or d2 > T^2 (1 + 1e-5)
//   min(acc, -v), max(acc, v)   change acc only if S > T:  no change if T > 0 and d2 (1 + 1e-5) < T^2
// The 1e-5 margin covers the roundings of T, T^2, d2's factor, S and S - r (each below 2^-23
// relative), the 1e-30 the loss of precision of T^2 or d2 near underflow; NaNs, infinite T^2 and
// everything near the boundary fail the test and take the exact path, so the fold's result is that
// of the plain formula bit for bit.
template <bool NEG, bool MAX>
__device__ __forceinline__ bool sphere_cannot_change(float acc, float d2, float r) {
    const float t = NEG ? r - acc : acc + r;
    const float t2 = t * t;
    if (NEG == MAX) return t < 0.0f || d2 > __builtin_fmaf(t2, 1.00001f, 1e-30f);
    return t > 0.0f && __builtin_fmaf(d2, 1.00001f, 1e-30f) < t2;
}

__device__ __forceinline__ float sphere_d2(V3 p, float cx, float cy, float cz) {
    const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
    return dx * dx + dy * dy + dz * dz;
}

// True unless folding the child value v into `acc` provably leaves `acc` as it is (NaNs count
// as "can change").
template <bool NEG, bool MAX>
__device__ __forceinline__ bool value_can_change(float acc, float v) {
    const float c = NEG ? -v : v;
    return MAX ? !(c <= acc) : !(c >= acc);
}

// scene::raymarch's result for one sample (scene.cpp:31): light * colour of the surface's primitive; no surface,
// or a sample outside the image: (0, 0, 0).
__device__ __forceinline__ V3 sample_colour(const Float4 *__restrict__ colors, Sample s) {
    V3 base = v3(0.0f, 0.0f, 0.0f);
    if (s.prim != kNoPrim) {
        const Float4 c = colors[s.prim];
        base = v3(c.x, c.y, c.z);
    }
    return v3(s.light * base.x, s.light * base.y, s.light * base.z);
}

// The output row at which row `ly` of the launch's band image is stored (FrameView::rows_in_frame): `ly` itself in a compact
// band image, the row's place in the frame otherwise -- kNoOutputRow for a row of a band below the image.
constexpr uint32_t kNoOutputRow = 0xffffffffu;
__device__ __forceinline__ uint32_t placed_row(const FrameView &fr, uint32_t ly) {
    if (fr.rows_in_frame == 0u) return ly;
    const uint32_t band = ly / fr.band_rows;
    const uint32_t image_band = fr.band_list ? ((ConstIndex)fr.band_list)[band < fr.band_count ? band : fr.band_count - 1u]
                                             : fr.first_band + band * fr.band_stride;
    const uint32_t pi = image_band * fr.band_rows + (ly - band * fr.band_rows);
    return pi < fr.height ? pi : kNoOutputRow;
}
// The same for the kernels generated for a scene, which honour FrameView::rows_in_frame only when they are compiled with
// RM_PLACE_ROWS (a context that places rows compiles its kernels so): the band plan's words kept until the pixels are
// stored cost the 1024-object scene's kernel three spilled registers and 1 % of its time, and most contexts never place rows.
__device__ __forceinline__ uint32_t output_row(const FrameView &fr, uint32_t ly) {
#ifdef RM_PLACE_ROWS
    return placed_row(fr, ly);
#else
    return ly;
#endif
}

// render.cpp:113-119 and the image writer's rule: pixel = total / (k * k), alpha 1; RGBA8 = round(clamp(c, 0, 1) * 255).
// (x, y): pixel in the output image of `width` pixels per row (y from output_row).
__device__ __forceinline__ void store_pixel(float tx, float ty, float tz, uint32_t k, uint32_t width, uint32_t x, uint32_t y,
                                            float *__restrict__ out_f32, unsigned char *__restrict__ out_rgba8) {
    const float n = (float)(k * k);
    const float r = tx / n, g = ty / n, b = tz / n;
    const size_t o = ((size_t)y * width + x) * 4;
    if (out_f32) {
        typedef float Out4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<Out4 *>(out_f32 + o) = Out4{ r, g, b, 1.0f };
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

// Marks a branch whose body must stay in the branch (stops the compiler from turning a cheap-looking
// body into unconditional code with selects).
__device__ __forceinline__ void keep_in_branch() { asm volatile(""); }

// For the folds that a sphere can change only when S = |c - p| is small (see sphere_cannot_change):
// true if the largest coordinate difference m alone proves S >= T.  S is computed from the same
// differences, so S >= m (1 - 2^-22) whatever the roundings of the squares, sums and root; with
// T' = (acc + r or r - acc) rounded, m >= T' (1 + 1e-6) gives S >= T.  T' <= 0 passes as it must
// (the sphere's value is at least -r); NaNs fail.  Seven instructions against the thirteen of the
// squared test, and enough for most spheres that are far from p.
template <bool NEG>
__device__ __forceinline__ bool sphere_far(float acc, float dx, float dy, float dz, float r) {
    const float t = NEG ? r - acc : acc + r;
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(dx), __builtin_fabsf(dy)), __builtin_fabsf(dz));
    return m >= t * 1.000001f;
}

template <bool NEG, bool MAX>
__device__ __forceinline__ float fold_sphere(float acc, V3 p, float cx, float cy, float cz, float r) {
    // Wave-uniform branches: taken, every lane folds (the fold is exact for all of them).  The
    // empty asm keeps the compiler from hoisting the square root out of the branch.
    const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
    if (NEG == MAX) {
        if (__builtin_amdgcn_ballot_w64(!sphere_far<NEG>(acc, dx, dy, dz, r)) == 0) return acc;
    }
    float d2 = dx * dx + dy * dy + dz * dz;
    if (__builtin_amdgcn_ballot_w64(!sphere_cannot_change<NEG, MAX>(acc, d2, r)) != 0) {
        asm volatile("" : "+v"(d2));
        acc = fold_sd<NEG, MAX>(acc, __builtin_sqrtf(d2) - r);
    }
    return acc;
}

// Wave-uniform: can folding a list that the sphere (c, r) bounds change some lane's accumulator?  (The list's value is at
// least, or at most, the sphere's; see codegen.cpp: Emitter::list.)  The largest-coordinate test first, the squared
// distances only if some lane is not ruled out by it.
template <bool NEG, bool MAX, int NP>
__device__ __forceinline__ bool sphere_bound_wanted(const V3 (&p)[NP], const float (&acc)[NP], float cx, float cy, float cz, float r) {
    if (NEG == MAX) {
        bool near = false;
#pragma unroll
        for (int n = 0; n < NP; ++n) near = near || !sphere_far<NEG>(acc[n], cx - p[n].x, cy - p[n].y, cz - p[n].z, r);
        if (__builtin_amdgcn_ballot_w64(near) == 0) return false;
        keep_in_branch();
    }
    bool need = false;
#pragma unroll
    for (int n = 0; n < NP; ++n) need = need || !sphere_cannot_change<NEG, MAX>(acc[n], sphere_d2(p[n], cx, cy, cz), r);
    return __builtin_amdgcn_ballot_w64(need) != 0;
}

// One test in front of several children of a union (values folded with min): the children's bounding spheres and
// boxes all lie in the box (centre c, half sizes H), and every one of their values is at least the largest-coordinate
// excess m of p over that box -- a sphere's |p - centre| - r and a box's max(|d| - h) both are, in exact arithmetic.
// The values the children would fold are rounded: each is at least m - 2^-20 (|m| + H + the child's own size), and m as
// computed here is within 2^-22 (|m| + H) of the exact one.  The host passes H enlarged by 3e-6 of (the largest H + the
// largest child size), and 3e-6 |m| is taken off here, so `false` proves that no child can lower any lane's accumulator
// (NaNs answer `true`).  Per wave, like the other bound tests: if one lane needs the children, all evaluate them.
template <int NP>
__device__ __forceinline__ bool group_wanted(const V3 (&p)[NP], const float (&acc)[NP], float cx, float cy, float cz, float hx, float hy,
                                             float hz) {
    bool need = false;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p[n].x - cx) - hx, __builtin_fabsf(p[n].y - cy) - hy),
                                        __builtin_fabsf(p[n].z - cz) - hz);
        need = need || !(__builtin_fmaf(__builtin_fabsf(m), -3e-6f, m) >= acc[n]);
    }
    return __builtin_amdgcn_ballot_w64(need) != 0;
}

// Folds one child value per point into the accumulators (see device_scene.h).
template <int NP, bool TRACK>
__device__ __forceinline__ void fold_value(uint32_t flags, const float (&v)[NP], const uint32_t (&vprim)[NP],
                                           float (&acc)[NP], uint32_t (&accp)[NP]) {
    const bool neg = (flags & FOLD_NEG) != 0, mx = (flags & FOLD_MAX) != 0;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const float c = neg ? -v[n] : v[n];
        if (TRACK) {
            // std::min / std::max on Surface: of two equal distances the earlier child survives
            // (its depth-first index is lower), whatever order the children are folded in
            const bool take = (mx ? (acc[n] < c) : (c < acc[n])) || (acc[n] == c && vprim[n] < accp[n]);
            acc[n] = take ? c : acc[n];
            accp[n] = take ? vprim[n] : accp[n];
        } else {
            acc[n] = mx ? __builtin_fmaxf(acc[n], c) : __builtin_fminf(acc[n], c);
        }
    }
}

// One RUN: `count` table rows of one kind, one fold.
template <uint32_t KIND, bool NEG, bool MAX, int NP, bool TRACK>
__device__ __forceinline__ void fold_run(ConstRows pa, ConstRows pb, ConstIndex row_prim, uint32_t first,
                                         uint32_t count, const V3 (&p)[NP], float (&acc)[NP], uint32_t (&accp)[NP]) {
    constexpr bool kNeedB = (KIND == K_BOX || KIND == K_CROSS);
#pragma unroll 4
    for (uint32_t i = first; i < first + count; ++i) {
        const RowWords ra = pa[i];
        const Float4 a = { ra[0], ra[1], ra[2], ra[3] };
        Float4 b = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (kNeedB) {
            const RowWords rb = pb[i];
            b.x = rb[0];
            b.y = rb[1];
        }
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            if (KIND == K_SPHERE && !TRACK) {
                acc[n] = fold_sphere<NEG, MAX>(acc[n], p[n], a.x, a.y, a.z, a.w);
                continue;
            }
            float v = prim_distance<KIND>(a, b, p[n]);
            if (NEG) v = -v;
            if (TRACK) {
                const uint32_t prim = row_prim[i];
                const bool take = (MAX ? (acc[n] < v) : (v < acc[n])) || (acc[n] == v && prim < accp[n]);
                acc[n] = take ? v : acc[n];
                accp[n] = take ? prim : accp[n];
            } else {
                acc[n] = MAX ? __builtin_fmaxf(acc[n], v) : __builtin_fminf(acc[n], v);
            }
        }
    }
}

template <uint32_t KIND, int NP, bool TRACK>
__device__ __forceinline__ void fold_run_kind(const SceneView &sc, uint32_t flags, uint32_t first, uint32_t count,
                                              const V3 (&p)[NP], float (&acc)[NP], uint32_t (&accp)[NP]) {
    switch (flags & (FOLD_NEG | FOLD_MAX)) {
    case 0: fold_run<KIND, false, false, NP, TRACK>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, first, count, p, acc, accp); break;
    case FOLD_NEG: fold_run<KIND, true, false, NP, TRACK>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, first, count, p, acc, accp); break;
    case FOLD_MAX: fold_run<KIND, false, true, NP, TRACK>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, first, count, p, acc, accp); break;
    default: fold_run<KIND, true, true, NP, TRACK>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, first, count, p, acc, accp); break;
    }
}

// ---- clustered RUN: hulls rule out groups of rows --------------------------------------------
//
// Folds that keep small values of the child distance v -- min(s, v) and max(s, -v) -- are left
// unchanged by every child with v >= need, need = s or -s.  A hull H contains its rows' primitives
// (per axis: [c - h, c + h] of every row lies inside the hull's interval; spheres: inside the hull
// sphere), so in exact arithmetic v_row(p) >= v_H(p): sphere distances by the triangle inequality,
// box and cross distances because each axis term |p - c| - h can only grow when the interval
// shrinks and max / median are monotone.  The host pads the hulls by 1e-6 relative, and the test
// leaves eps = 2e-6 x (|p| + the run's magnitude bound), more than 8 times the rounding of either
// side's few operations at that magnitude (spheres: the margins of sphere_cannot_change).  A
// hull that fails the test, NaNs included, has its rows evaluated with the plain formula, so the
// fold's result is that of visiting every row, bit for bit.
template <uint32_t KIND, bool NEG>
__device__ __forceinline__ bool hull_needs_rows(const RowWords ra, const RowWords rb, V3 p, float acc, float eps) {
    const float need = NEG ? -acc : acc;
    if (KIND == K_SPHERE) return !sphere_cannot_change<false, false>(need, sphere_d2(p, ra[0], ra[1], ra[2]), ra[3]);
    const float dx = __builtin_fabsf(p.x - ra[0]) - ra[3];
    const float dy = __builtin_fabsf(p.y - ra[1]) - rb[0];
    const float dz = __builtin_fabsf(p.z - ra[2]) - rb[1];
    const float v = (KIND == K_BOX || KIND == K_BOX_ISO) ? __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz) : cross_value(dx, dy, dz);
    return !(v - eps >= need);
}

// Crosses are three infinite bars (along z: max(dx, dy); along y: max(dx, dz); along x: max(dy, dz);
// the cross distance, the median of the three axis terms, is the smallest of the three), so a hull
// in space bounds them poorly.  A run of crosses is listed three times instead, once per bar
// direction, each listing sorted and grouped by the two coordinates that place that bar, with flat
// hulls (ra = centre a, centre b, half a, half b) that bound the bar's distance from below.  A cross
// that could still win is close to p with at least one of its bars, hence inside a group that the
// listing of that bar direction does not rule out, and is evaluated there with the plain formula;
// evaluating it from two listings is harmless (min and max are idempotent).
template <int DIR, bool NEG>
__device__ __forceinline__ bool bar_hull_needs_rows(const RowWords ra, V3 p, float acc, float eps) {
    const float need = NEG ? -acc : acc;
    const float pa_ = DIR == 2 ? p.y : p.x, pb_ = DIR == 0 ? p.y : p.z;   // DIR 0: (x, y), 1: (x, z), 2: (y, z)
    const float da = __builtin_fabsf(pa_ - ra[0]) - ra[2];
    const float db = __builtin_fabsf(pb_ - ra[1]) - ra[3];
    return !(__builtin_fmaxf(da, db) - eps >= need);
}

// `cap`: an upper bound of the value the WHOLE scene will have at p, known to the caller (+inf if
// none), usable by min-folds at the root of the scene only: a row whose distance is at least cap
// cannot be the scene's minimum, which lies strictly below cap, so a group that only holds such rows
// may be skipped even while the accumulator is still larger than cap.
template <bool NEG>
__device__ __forceinline__ float capped(float acc, float cap) {
    return NEG ? acc : __builtin_fminf(acc, cap);
}

// One LDS word per wave, for the exchanges among the lanes of a wave that are executing (a ballot tells who
// is there, but a value can only be gathered from lanes that are: float max / min atomics on this word).
__device__ __forceinline__ float *wave_slot() {
    __shared__ float slots[16];   // workgroups have up to 9 waves
    return &slots[threadIdx.x >> 6];
}

// Largest v among the executing lanes (NaNs must have been replaced by the caller).  Every lane performs the same
// three LDS operations in program order.
__device__ __forceinline__ float wave_max_of_executing(float v) {
    float *slot = wave_slot();
    __hip_atomic_store(slot, -__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The executing lanes' points, enclosed: the first such lane's point and the largest distance (1-norm, which is
// at least the Euclidean and the maximum norm) of any of them from it.  Wave-uniform.
struct WaveBall {
    V3 centre;
    float reach;
};

template <int NP>
__device__ __forceinline__ WaveBall wave_ball(const V3 (&p)[NP]) {
    const int first = __builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
    WaveBall ball;
    ball.centre = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0].x), first)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0].y), first)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0].z), first)));
    float spread = 0.0f;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const float d = __builtin_fabsf(p[n].x - ball.centre.x) + __builtin_fabsf(p[n].y - ball.centre.y) + __builtin_fabsf(p[n].z - ball.centre.z);
        spread = d <= spread ? spread : d;   // a NaN distance stays (and then reaches everything)
    }
    ball.reach = wave_max_of_executing(spread == spread ? spread : __builtin_inff());
    return ball;
}

// Distance of a hull row from a point, as the hull tests above compute it (a lower bound of its rows' distances).
template <uint32_t KIND, int DIR>
__device__ __forceinline__ float hull_value(const RowWords ra, const RowWords rb, V3 c) {
    if (DIR >= 0) {
        const float pa_ = DIR == 2 ? c.y : c.x, pb_ = DIR == 0 ? c.y : c.z;
        return __builtin_fmaxf(__builtin_fabsf(pa_ - ra[0]) - ra[2], __builtin_fabsf(pb_ - ra[1]) - ra[3]);
    }
    if (KIND == K_SPHERE) return __builtin_sqrtf(sphere_d2(c, ra[0], ra[1], ra[2])) * 0.999999f - ra[3];
    const float dx = __builtin_fabsf(c.x - ra[0]) - ra[3];
    const float dy = __builtin_fabsf(c.y - ra[1]) - rb[0];
    const float dz = __builtin_fabsf(c.z - ra[2]) - rb[1];
    return __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
}

// What a row's distance must stay below to change some executing lane's accumulator: the largest `need` of the
// wave's points (+inf if one of them is a NaN).
template <bool NEG, int NP>
__device__ __forceinline__ float listing_need(const float (&acc)[NP], const float (&cap)[NP]) {
    float need = -__builtin_inff();
    bool unordered = false;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const float c = capped<NEG>(acc[n], cap[n]);
        const float mine = NEG ? -c : c;
        unordered = unordered || mine != mine;
        need = __builtin_fmaxf(need, mine);
    }
    return wave_max_of_executing(unordered ? __builtin_inff() : need);
}

// One listing: rows [base, base + count), then a hull per kClusterSize rows.  DIR < 0: hulls of the run's own kind; DIR = 0..2: flat bar hulls.
//
// Broad phase, once per wave instead of once per lane: the hulls are dealt out to the lanes, 64 at a time, and each
// lane tests ITS hull against the ball around all of the wave's points -- hull distances are 1-Lipschitz, so at
// every point of the wave a hull is worth at least its distance from the ball's centre minus the ball's reach, and
// a hull whose bound is not below the largest `need` of the wave (need = what a row's distance must stay below to
// change that lane's accumulator) is ruled out for every lane.  The survivors (a ballot) are then tested lane by
// lane with the exact test and, where some lane needs them, have their rows folded with the plain formula, so the
// fold's result is that of visiting every row, bit for bit.  Margins: 1e-5 of the reach for its own roundings,
// eps (at the centre) for the hull distance's, as in the per-lane test.
template <uint32_t KIND, int DIR, bool NEG, bool MAX, int NP>
__device__ __forceinline__ void fold_listing(ConstRows pa, ConstRows pb, uint32_t base, uint32_t count, const WaveBall &ball, float centre_eps,
                                             const V3 (&p)[NP], float (&acc)[NP], const float (&cap)[NP], const float (&eps)[NP],
                                             uint32_t &tested, uint32_t &evaluated) {
    uint32_t no_prim[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) no_prim[n] = kNoPrim;
    const uint32_t n_clusters = (count + kClusterSize - 1u) / kClusterSize;
    const uint32_t cfirst = base + count;
    // the hulls are dealt to the lanes that are executing (lanes whose ray has ended are not here to test theirs)
    const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
    const uint32_t n_here = (uint32_t)__builtin_popcountll(here);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
    const float wave_need = listing_need<NEG, NP>(acc, cap);
    const float bound_slack = ball.reach * 1.00001f + centre_eps;
    for (uint32_t g0 = 0; g0 < n_clusters; g0 += n_here) {
        const uint32_t g = g0 + rank;
        const bool mine = g < n_clusters;
        const RowWords ha = pa[cfirst + (mine ? g : 0u)];
        RowWords hb = ha;
        if (DIR < 0 && KIND != K_SPHERE) hb = pb[cfirst + (mine ? g : 0u)];
        float bound = hull_value<KIND, DIR>(ha, hb, ball.centre) - bound_slack;
        bound = bound == bound ? bound : -__builtin_inff();   // a hull that cannot be placed is visited, first
        unsigned long long survivors = __builtin_amdgcn_ballot_w64(mine && !(bound >= wave_need));
        tested++;
        for (; survivors != 0ull; survivors &= survivors - 1ull) {
            // the surviving lane's rank among the executing lanes is its hull's place in this round
            const unsigned long long below = (survivors & (0ull - survivors)) - 1ull;
            const uint32_t c = g0 + (uint32_t)__builtin_popcountll(here & below);
            const RowWords ca = pa[cfirst + c];
            RowWords cb = ca;
            if (DIR < 0) cb = pb[cfirst + c];
            bool wanted = false;
#pragma unroll
            for (int n = 0; n < NP; ++n)
                wanted = wanted || (DIR < 0 ? hull_needs_rows<KIND, NEG>(ca, cb, p[n], capped<NEG>(acc[n], cap[n]), eps[n])
                                            : bar_hull_needs_rows<(DIR < 0 ? 0 : DIR), NEG>(ca, p[n], capped<NEG>(acc[n], cap[n]), eps[n]));
            tested++;
            if (__builtin_amdgcn_ballot_w64(wanted) == 0) continue;
            const uint32_t i0 = c * kClusterSize;
            const uint32_t rows = i0 + kClusterSize < count ? kClusterSize : count - i0;
            fold_run<KIND, NEG, MAX, NP, false>(pa, pb, (ConstIndex)0, base + i0, rows, p, acc, no_prim);
            evaluated += rows;
        }
    }
}

template <uint32_t KIND, bool NEG, bool MAX, int NP, bool COUNT>
__device__ __forceinline__ void fold_run_clustered(ConstRows pa, ConstRows pb, uint32_t first, uint32_t count,
                                                   float magnitude, const V3 (&p)[NP], float (&acc)[NP], const float (&cap)[NP],
                                                   Work (&work)[NP]) {
    static_assert(NEG == MAX, "hulls give lower bounds: only folds that keep small distances can use them");
    float eps[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n)
        eps[n] = 2e-6f * (__builtin_fabsf(p[n].x) + __builtin_fabsf(p[n].y) + __builtin_fabsf(p[n].z) + magnitude);
    uint32_t tested = 0, evaluated = 0;
    const WaveBall ball = wave_ball<NP>(p);
    const float centre_eps = 2e-6f * (__builtin_fabsf(ball.centre.x) + __builtin_fabsf(ball.centre.y) + __builtin_fabsf(ball.centre.z) + magnitude);
    if (KIND == K_CROSS || KIND == K_CROSS_ISO) {
        const uint32_t n_clusters = (count + kClusterSize - 1u) / kClusterSize;
        const uint32_t stride = count + n_clusters;
        fold_listing<KIND, 0, NEG, MAX, NP>(pa, pb, first, count, ball, centre_eps, p, acc, cap, eps, tested, evaluated);
        fold_listing<KIND, 1, NEG, MAX, NP>(pa, pb, first + stride, count, ball, centre_eps, p, acc, cap, eps, tested, evaluated);
        fold_listing<KIND, 2, NEG, MAX, NP>(pa, pb, first + 2u * stride, count, ball, centre_eps, p, acc, cap, eps, tested, evaluated);
    } else {
        fold_listing<KIND, -1, NEG, MAX, NP>(pa, pb, first, count, ball, centre_eps, p, acc, cap, eps, tested, evaluated);
    }
    if (COUNT) {
        // per wave: hull tests (dealt rounds and survivors' exact tests) and rows folded
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
            atomicAdd(&probe_words()[2 * kProbeClusteredRuns], (unsigned long long)tested);
            atomicAdd(&probe_words()[2 * kProbeClusteredRuns + 1], (unsigned long long)evaluated);
        }
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            work[n].evals += tested + evaluated;
            work[n].flops += (tested + evaluated) * flops_of_kind(KIND);
        }
    }
}

// ---- cell grid: the root's long runs, row lists per cell (device_scene.h: GridView) -------------
// min(acc, v) over the rows listed for the cell that holds p -- every row that can be the smallest there -- with the
// plain formulas.  Each lane walks its own cell's list (its own trip count, its own gathers).  outside[n]: p[n] is not in
// the grid's box (or not a number); its accumulator is left as it is, for the hull walk to finish.
__device__ __forceinline__ bool grid_cell_of(const GridView &g, const V3 p, uint32_t &cell) {
    const float fx = (p.x - g.origin[0]) * g.inv_cell, fy = (p.y - g.origin[1]) * g.inv_cell, fz = (p.z - g.origin[2]) * g.inv_cell;
    const bool inside = fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)g.dims[0] && fy < (float)g.dims[1] && fz < (float)g.dims[2];
    cell = inside ? ((uint32_t)fz * g.dims[1] + (uint32_t)fy) * g.dims[0] + (uint32_t)fx : 0u;
    return inside;
}

template <int NP, bool COUNT>
__device__ __forceinline__ void grid_fold(const GridView &g, const Float4 *__restrict__ pa, const Float4 *__restrict__ pb, const V3 (&p)[NP],
                                          float (&acc)[NP], bool (&outside)[NP], Work (&work)[NP]) {
    typedef uint32_t Pair __attribute__((ext_vector_type(2)));
    typedef float Float2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        uint32_t cell;
        outside[n] = !grid_cell_of(g, p[n], cell);
        if (outside[n]) continue;
        const Pair head = reinterpret_cast<const Pair *>(g.cells)[cell];   // where the other entries are, the first entry
        const uint32_t *e = g.entries + head[0];
        float a = acc[n];
        const V3 q = p[n];
        uint32_t evals = 0, flops = 0;
        // The entries come by rising lower bound: the first one whose bound is not below the running minimum ends the walk
        // (its row's distance, and every later row's, is at least that bound everywhere in the cell).  A minimum that
        // is not a number ends nothing: the end mark does.
        uint32_t en = head[1];
        _Pragma("nounroll") for (uint32_t i = 0; en != kGridListEnd && !(__uint_as_float(en & 0xffff0000u) >= a); ++i) {
            const uint32_t row = en & 0x3fffu, cls = (en >> 14) & 3u;
            en = e[i];   // (there is one: the end mark at the latest)
            const Float4 ra = pa[row];
            if (cls == kGridClassSphere) {
                const float d2 = sphere_d2(q, ra.x, ra.y, ra.z);
                if (!sphere_cannot_change<false, false>(a, d2, ra.w)) a = __builtin_fminf(a, __builtin_sqrtf(d2) - ra.w);
                if (COUNT) flops += kFlopsSphere;
            } else {
                const Float2 half = *reinterpret_cast<const Float2 *>(pb + row);   // the two half sizes a box or cross keeps there
                const Float4 rb = { half[0], half[1], 0.0f, 0.0f };
                if (cls == kGridClassBox) {
                    a = __builtin_fminf(a, prim_distance<K_BOX>(ra, rb, q));
                    if (COUNT) flops += kFlopsBox;
                } else {
                    a = __builtin_fminf(a, prim_distance<K_CROSS>(ra, rb, q));
                    if (COUNT) flops += kFlopsCross;
                }
            }
            if (COUNT) evals++;
        }
        acc[n] = a;
        if (COUNT) {
            work[n].evals += evals;
            work[n].flops += flops;
        }
    }
}

// The same walk with the surface: which row is the smallest at p.  Of equal distances the row of the lower depth-first
// index wins, whatever the order of the walk (fold_run's rule), so the walk ends only at a bound above the running minimum:
// a row whose bound equals it can still tie and win by index.
template <int NP>
__device__ __forceinline__ void grid_surface(const GridView &g, const Float4 *__restrict__ pa, const Float4 *__restrict__ pb,
                                             const uint32_t *__restrict__ row_prim, const V3 (&p)[NP], float (&acc)[NP], uint32_t (&accp)[NP],
                                             bool (&outside)[NP]) {
    typedef uint32_t Pair __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        uint32_t cell;
        outside[n] = !grid_cell_of(g, p[n], cell);
        if (outside[n]) continue;
        const Pair head = reinterpret_cast<const Pair *>(g.cells)[cell];
        const uint32_t *e = g.entries + head[0];
        float a = acc[n];
        uint32_t q = accp[n];
        uint32_t en = head[1];
        _Pragma("nounroll") for (uint32_t i = 0; en != kGridListEnd && !(__uint_as_float(en & 0xffff0000u) > a); ++i) {
            const uint32_t row = en & 0x3fffu, cls = (en >> 14) & 3u;
            en = e[i];
            const Float4 ra = pa[row], rb = pb[row];
            const float v = cls == kGridClassSphere ? prim_distance<K_SPHERE>(ra, rb, p[n])
                                                    : (cls == kGridClassBox ? prim_distance<K_BOX>(ra, rb, p[n]) : prim_distance<K_CROSS>(ra, rb, p[n]));
            const uint32_t prim = row_prim[row];
            const bool take = v < a || (a == v && prim < q);
            a = take ? v : a;
            q = take ? prim : q;
        }
        acc[n] = a;
        accp[n] = q;
    }
}

// interpreter entry: fold known at run time (one of the two a clustered run can have)
template <uint32_t KIND, int NP, bool COUNT>
__device__ __forceinline__ void fold_run_clustered_kind(const SceneView &sc, const RecWords &r, const V3 (&p)[NP],
                                                        float (&acc)[NP], Work (&work)[NP]) {
    float no_cap[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) no_cap[n] = __builtin_inff();
    if (r[6] & FOLD_MAX)
        fold_run_clustered<KIND, true, true, NP, COUNT>((ConstRows)sc.pa, (ConstRows)sc.pb, r[0], r[1], __uint_as_float(r[5]), p, acc, no_cap, work);
    else
        fold_run_clustered<KIND, false, false, NP, COUNT>((ConstRows)sc.pa, (ConstRows)sc.pb, r[0], r[1], __uint_as_float(r[5]), p, acc, no_cap, work);
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

// One row of crosses: two axis terms fixed (d1, d2 in axis order), the third axis running over all
// n centres of the level.  AXIS = the running axis; the operands keep their x, y, z places in the
// reference's sum.  `fixed_ones` = digit masks of the two fixed indices, or-ed.  (Reading the row's
// tables through the scalar unit instead of LDS was measured slower: 1.15 vs 1.05 ms per default frame.)
// What the cell contributes to the row's maximum: -cross(p), or minus infinity for a cell that names no sub-cube.
template <int AXIS>
__device__ __forceinline__ float lattice_cell_value(float coord, uint32_t ones, float pt, float h, float d1, float d2, uint32_t fixed_ones) {
    const float t = __builtin_fabsf(pt - coord) - h;
    const float v = -(AXIS == 0 ? cross_value(t, d1, d2) : (AXIS == 1 ? cross_value(d1, t, d2) : cross_value(d1, d2, t)));
    return (ones & fixed_ones) ? -__builtin_inff() : v;
}

template <int AXIS>
__device__ __forceinline__ float lattice_cell(float coord, uint32_t ones, float pt, float h, float d1, float d2,
                                              uint32_t fixed_ones, float run) {
    return __builtin_fmaxf(run, lattice_cell_value<AXIS>(coord, ones, pt, h, d1, d2, fixed_ones));
}

// The row's cells are folded among themselves first (from minus infinity: every operand of those maxima is a fresh
// arithmetic result or a constant, which spares the compiler a canonicalising copy of the running value per cell),
// then into the running value; a maximum does not depend on the order.
template <int AXIS>
__device__ __forceinline__ float lattice_row(const float *T, const uint32_t *ONES, int n, float pt, float h, float d1, float d2,
                                             uint32_t fixed_ones, float run) {
    if (n == 1) return lattice_cell<AXIS>(T[0], ONES[0], pt, h, d1, d2, fixed_ones, run);
    // n is a power of 3: three table entries are read together, then evaluated, so the loop waits for
    // LDS once per three crosses (batches of nine measured slower: 1.19 vs 1.07 ms per default frame)
    float best = -__builtin_inff();
    for (int k = 0; k < n; k += 3) {
        const float c0 = T[k], c1 = T[k + 1], c2 = T[k + 2];
        const uint32_t o0 = ONES[k], o1 = ONES[k + 1], o2 = ONES[k + 2];
        best = __builtin_fmaxf(best, lattice_cell_value<AXIS>(c0, o0, pt, h, d1, d2, fixed_ones));
        best = __builtin_fmaxf(best, lattice_cell_value<AXIS>(c1, o1, pt, h, d1, d2, fixed_ones));
        best = __builtin_fmaxf(best, lattice_cell_value<AXIS>(c2, o2, pt, h, d1, d2, fixed_ones));
    }
    return __builtin_fmaxf(run, best);
}

// Rows of the lanes that have one (`active`), evaluated either by every such lane for itself
// (lattice_row: all rows at once, n cells deep) or, when few lanes have a long row, one row after
// the other with the row's n cells dealt out to all lanes that are executing: the owner's operands
// are broadcast, every lane folds its share of the cells, and the partial maxima meet in one LDS
// word per wave (a float max is exact and order-free, so the row's result does not depend on who
// evaluated which cell).  Must be called from control flow that is uniform among the executing
// lanes; lanes that have left the walk earlier simply are not among the workers.

// what handing one row round costs, in cells (the broadcast of the owner's operands, the LDS exchange of the maxima)
#ifndef RM_ROW_COOP_COST
#define RM_ROW_COOP_COST 6
#endif

template <int AXIS, bool COUNT = false>
__device__ __forceinline__ float lattice_rows(bool active, const float *T, const uint32_t *ONES, int n, float pt, float h, float d1,
                                              float d2, uint32_t fixed_ones, float run) {
    // short rows: every owner walks its own, and a plain branch skips the walk when no lane has one
    if (!COUNT && n < 27) return active ? lattice_row<AXIS>(T, ONES, n, pt, h, d1, d2, fixed_ones, run) : run;
    const unsigned long long rows = __builtin_amdgcn_ballot_w64(active);
    if (rows == 0) return run;
    if (COUNT) {
        const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(here)) {
            atomicAdd(&probe_words()[2 * kProbeRowCells], (unsigned long long)__builtin_popcountll(rows) * (uint32_t)n);
            atomicAdd(&probe_words()[2 * kProbeRowCells + 1], 64ull * (uint32_t)n);
        }
    }
    if (n >= 27) {
        const unsigned long long workers = __builtin_amdgcn_ballot_w64(true);
        const int n_workers = __builtin_popcountll(workers), n_rows = __builtin_popcountll(rows);
        // cells per lane: n when every owner walks its own row; per row a share of n plus the exchange otherwise
        if (n_rows * ((n + n_workers - 1) / n_workers + RM_ROW_COOP_COST) < n) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(workers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)workers, 0u));
            const int lane = (int)(threadIdx.x & 63u);
            float *slot = wave_slot();
            for (unsigned long long m = rows; m != 0; m &= m - 1) {
                const int owner = __builtin_ctzll(m);
                const float o_pt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pt), owner));
                const float o_d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1), owner));
                const float o_d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), owner));
                const uint32_t o_fixed = (uint32_t)__builtin_amdgcn_readlane((int)fixed_ones, owner);
                float best = -__builtin_inff();
                for (int k = rank; k < n; k += n_workers) best = lattice_cell<AXIS>(T[k], ONES[k], o_pt, h, o_d1, o_d2, o_fixed, best);
                // every lane performs the same three LDS operations in program order (no lane-dependent
                // branch around them, which the compiler could lay out as two paths in either order)
                __hip_atomic_store(slot, -__builtin_inff(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(slot, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const float row_max = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                run = lane == owner ? __builtin_fmaxf(run, row_max) : run;
            }
            return run;
        }
    }
    if (active) run = lattice_row<AXIS>(T, ONES, n, pt, h, d1, d2, fixed_ones, run);
    return run;
}

// index of the centre nearest to p (exact enough: see the caller's 0.45-spacing condition)
__device__ __forceinline__ int nearest_index(float p, float origin, float inv_spacing, int last) {
    const float u = (p - origin) * inv_spacing + 0.5f;
    return (int)__builtin_fminf(__builtin_fmaxf(u, 0.0f), (float)last);
}

// One level of the lattice by ranges: on every axis the centres whose axis term is below tau, then every row whose two fixed
// indices lie in their ranges (and name a row), all n cells of it.  The rare walk of points deep inside material, written
// for size, not for speed: one body for the three directions a row can run in, the direction a loop variable.  The cells'
// values are the reference's: x runs (t + d1) + d2, y runs (d1 + t) + d2 -- the same sum, an addition of two floats does
// not depend on their order -- z runs (d1 + d2) + t, minimum and maximum of the three terms, sum - min - max.
// tab: the level's X table, followed by Y, Z and ONES.
__device__ __forceinline__ float lattice_scan_level(const uint32_t *tab, int n, V3 p, float h, float inv, float tau, float run, uint32_t &n_evals) {
    const uint32_t *ONES = tab + 3 * n;
    int lo0, hi0, lo1, hi1, lo2, hi2;
    axis_range(tab, n, p.x, h, inv, tau, lo0, hi0);
    axis_range(tab + n, n, p.y, h, inv, tau, lo1, hi1);
    axis_range(tab + 2 * n, n, p.z, h, inv, tau, lo2, hi2);
    _Pragma("nounroll") for (int along = 0; along < 3; ++along) {
        // the row's fixed axes a < b, in x, y, z order
        const uint32_t *A = tab + (along == 0 ? n : 0), *B = tab + (along == 2 ? n : 2 * n), *T = tab + along * n;
        const float pa = along == 0 ? p.y : p.x, pb = along == 2 ? p.y : p.z, pt = along == 0 ? p.x : (along == 1 ? p.y : p.z);
        const int loa = along == 0 ? lo1 : lo0, hia = along == 0 ? hi1 : hi0, lob = along == 2 ? lo1 : lo2, hib = along == 2 ? hi1 : hi2;
        _Pragma("nounroll") for (int i = loa; i <= hia; ++i) {
            _Pragma("nounroll") for (int j = lob; j <= hib; ++j) {
                const uint32_t fixed_ones = ONES[i] | ONES[j];
                if (ONES[i] & ONES[j]) continue;   // two zero offsets at one subdivision step: no such sub-cube
                const float da = axis_term(A, i, pa, h), db = axis_term(B, j, pb, h), both = da + db;
                _Pragma("nounroll") for (int k = 0; k < n; ++k) {
                    const float t = axis_term(T, k, pt, h);
                    const float sum = along == 2 ? both + t : (t + da) + db;
                    const float dmin = __builtin_fminf(__builtin_fminf(t, da), db), dmax = __builtin_fmaxf(__builtin_fmaxf(t, da), db);
                    run = __builtin_fmaxf(run, (ONES[k] & fixed_ones) ? -__builtin_inff() : -(sum - dmin - dmax));
                }
                n_evals += (uint32_t)n;
            }
        }
    }
    return run;
}

// The sum of 3^position over the bits of a digit mask: what takes an index with digit 1 at those positions to the index
// with digit 0 there.
// n: centres per axis of the level, 3^(digit positions).
__device__ __forceinline__ int digits_to_zero(uint32_t mask, int n) {
    int sum = 0;
    for (int power = 1; power < n; power *= 3, mask >>= 1) sum += (mask & 1u) ? power : 0;
    return sum;
}

// One cross of level l (n = 3^l centres per axis) on each of the three rows through the centres nearest to p: the row's
// two fixed indices are the nearest ones (if they name a row: no digit position where both have digit 1), the running index
// is the nearest one with the digits that clash with the fixed indices set to 0, which names a cross of the lattice.
// Returns the largest -cross(p) of the three, each with the reference's operands in their places: values of real crosses,
// never above the lattice's maximum -- and where p lies in a tunnel of this level, or next to one, a bar of one of these
// three crosses is that tunnel (the bars are endless: every cross of a row carries the row's bar), so the value is the
// lattice's own there.
__device__ __forceinline__ float lattice_nearest_rows(const uint32_t *w, ConstIndex g, uint32_t l, int n, V3 p) {
    const ConstFloats gf = (ConstFloats)g;
    const float h = gf[kLatHalf + l], inv = gf[kLatInvSpacing + l];
    const uint32_t *X = w + g[kLatOffset + l], *Y = X + n, *Z = Y + n, *ONES = Z + n;
    const int ix = nearest_index(p.x, gf[kLatOrigin + 3 * l], inv, n - 1);
    const int iy = nearest_index(p.y, gf[kLatOrigin + 3 * l + 1], inv, n - 1);
    const int iz = nearest_index(p.z, gf[kLatOrigin + 3 * l + 2], inv, n - 1);
    const float dx = axis_term(X, ix, p.x, h), dy = axis_term(Y, iy, p.y, h), dz = axis_term(Z, iz, p.z, h);
    const uint32_t ox = ONES[ix], oy = ONES[iy], oz = ONES[iz];
    float best = -__builtin_inff();
    if (!(ox & oy)) best = __builtin_fmaxf(best, -cross_value(dx, dy, axis_term(Z, iz - digits_to_zero(oz & (ox | oy), n), p.z, h)));
    if (!(ox & oz)) best = __builtin_fmaxf(best, -cross_value(dx, axis_term(Y, iy - digits_to_zero(oy & (ox | oz), n), p.y, h), dz));
    if (!(oy & oz)) best = __builtin_fmaxf(best, -cross_value(axis_term(X, ix - digits_to_zero(ox & (oy | oz), n), p.x, h), dy, dz));
    return best;
}

template <bool TRACK, bool COUNT>
__device__ __forceinline__ void lattice_fold(const uint32_t *w, ConstIndex g, uint32_t levels, uint32_t first_row, V3 p,
                                             float &acc, uint32_t &acc_prim, uint32_t &n_evals) {
    // w: the lattice block in LDS (lookups at per-lane indices); g: the same block in global memory,
    // read through the scalar unit (header fields, whole rows)
    const ConstFloats gf = (ConstFloats)g;
    const float bound = __builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + gf[2];
    const float eps = 2e-6f * bound;   // > 8x the worst rounding of dx + dy + dz - min - max at this magnitude
    float run = acc;
    int n = 1;
    bool raised = false;
    for (uint32_t l = 0; l < levels; ++l, n *= 3) {
        const float h = gf[kLatHalf + l];
        if (!(run < h * 1.00001f + eps)) break;   // no cross this fine can beat the running value
        float tau = eps - run;
        const float inv = gf[kLatInvSpacing + l];
        if (!((h + tau) * inv <= 0.45f) && !raised) {
            // More than one centre per axis could be in range at this level: the running value is far below what the
            // lattice is worth at p -- the usual cause is a point in one of the fine tunnels, a long way from the coarse
            // ones, where the list's value so far is minus the depth inside the sponge's box.  Before scanning ranges of
            // centres, one cross of each row through the nearest centres of this and of every finer level is folded
            // (lattice_nearest_rows): each is a cross of the lattice, so the maximum cannot overshoot, and in a tunnel one
            // of them carries the tunnel's own (small) value, after which the nearest-centre argument holds again on
            // every level.  Once per walk; a point that is really deep inside material still takes the scans below.
            raised = true;
            int m = n;
            for (uint32_t fine = l; fine < levels; ++fine, m *= 3) run = __builtin_fmaxf(run, lattice_nearest_rows(w, g, fine, m, p));
            if (COUNT) n_evals += 3u * (levels - l);
            if (!(run < h * 1.00001f + eps)) break;
            tau = eps - run;
        }
        const uint32_t offset = g[kLatOffset + l];
        const uint32_t *X = w + offset;
        const uint32_t *Y = X + n;
        const uint32_t *Z = Y + n;
        const uint32_t *ONES = Z + n;
        if (COUNT) n_evals += 3;
        if ((h + tau) * inv <= 0.45f) {
            // The usual case (p is not deep inside solid material): an axis term below tau needs
            // |p - centre| < h + tau <= 0.45 spacings, so on each axis only the centre nearest to p can
            // be in range.  (Tables are evenly spaced to 1 % of a step, and the three axes' spacings
            // agree to 1 %: whenever some centre is in range, it is the one the estimate picks; when
            // none is, the pick fails the test.)  One round of independent loads, no scans.
            const int ix = nearest_index(p.x, gf[kLatOrigin + 3 * l], inv, n - 1);
            const int iy = nearest_index(p.y, gf[kLatOrigin + 3 * l + 1], inv, n - 1);
            const int iz = nearest_index(p.z, gf[kLatOrigin + 3 * l + 2], inv, n - 1);
            const float dx = axis_term(X, ix, p.x, h), dy = axis_term(Y, iy, p.y, h), dz = axis_term(Z, iz, p.z, h);
            const uint32_t ox = ONES[ix], oy = ONES[iy], oz = ONES[iz];
            const bool bx = dx < tau, by = dy < tau, bz = dz < tau;
            // two zero offsets at one subdivision step: no such sub-cube
            if (bx && by && !(ox & oy)) {
                run = lattice_row<2>((const float *)Z, ONES, n, p.z, h, dx, dy, ox | oy, run);
                if (COUNT) n_evals += n;
            }
            if (bx && bz && !(ox & oz)) {
                run = lattice_row<1>((const float *)Y, ONES, n, p.y, h, dx, dz, ox | oz, run);
                if (COUNT) n_evals += n;
            }
            if (by && bz && !(oy & oz)) {
                run = lattice_row<0>((const float *)X, ONES, n, p.x, h, dy, dz, oy | oz, run);
                if (COUNT) n_evals += n;
            }
            continue;
        }
        run = lattice_scan_level(X, n, p, h, inv, tau, run, n_evals);
    }
    if (TRACK) acc_prim = (acc < run) ? first_row : acc_prim;   // all crosses of a lattice share one colour
    acc = run;
}

// The lattice's exact maximum by the shortest code there is for it: every level's nearest rows first (values of real
// crosses, which bring `run` close to the lattice's value), then every level by ranges.  What the kernels generated for
// a scene fall back to from the rare level their nearest-centre walk cannot decide (a point deep inside material): it
// sits in every copy of the scene's walk, so its size counts and its speed does not.
template <bool COUNT>
__device__ __forceinline__ void lattice_fold_scans(const uint32_t *w, ConstIndex g, uint32_t levels, V3 p, float &run, uint32_t &n_evals) {
    const ConstFloats gf = (ConstFloats)g;
    const float eps = 2e-6f * (__builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + gf[2]);
    int n = 1;
    _Pragma("nounroll") for (uint32_t l = 0; l < levels; ++l, n *= 3) run = __builtin_fmaxf(run, lattice_nearest_rows(w, g, l, n, p));
    uint32_t evals = 3u * levels;
    n = 1;
    _Pragma("nounroll") for (uint32_t l = 0; l < levels; ++l, n *= 3) {
        const float h = gf[kLatHalf + l];
        if (!(run < h * 1.00001f + eps)) break;   // no cross this fine can beat the running value
        run = lattice_scan_level(w + g[kLatOffset + l], n, p, h, gf[kLatInvSpacing + l], eps - run, run, evals);
    }
    if (COUNT) n_evals += evals;
}

// One LDS word per workgroup: set by an optimistic scene walk that met a point it could not decide (trace_kernel.h books it
// into FrameView::tile_flags).  Why the exact walk is not simply part of every kernel: its mere presence inside the scene
// walk cost a tenth of the frame -- the compiler keeps the walk's state live for it and schedules around it.  Default
// frame, frames in flight, same GPU: 0.490 ms with lattice_fold_scans in line at the lattice, 0.447 called there as a
// function (and 60 bytes of register spills per sample around the call, 0.45 GB per frame), 0.436 without it; the point's
// whole walk repeated behind the scene's walk: 0.504 in line, 0.571 as a function that takes the scene's description.
__device__ __forceinline__ uint32_t *exact_wanted_word() {
    __shared__ uint32_t word;
    return &word;
}

// One level of lattice_fold with everything about the level known when the kernel is compiled (the
// kernels generated for one scene pass the header fields as literals): N = 3^level centres per axis,
// half-width h, 1 / spacing, the first centre on each axis; `tab` = the level's X table in LDS.
// Returns kLatticeNext to go on with the next level, kLatticeDone when the walk ends here, and
// kLatticeGeneral in the rare case that needs the general scan (p deep inside material): the caller
// then runs lattice_fold_scans, which repeats the earlier levels to no effect (max is idempotent, and they
// pass its early-out because coarser levels have larger half-widths).
constexpr int kLatticeDone = 0, kLatticeNext = 1, kLatticeGeneral = 2;

// rows of at least this many cells count as long (below: the tests that spare a long row its walk)
#ifndef RM_LONG_ROW
#define RM_LONG_ROW 27
#endif

template <int N, bool COUNT>
__device__ __forceinline__ int lattice_level(const uint32_t *tab, float h, float inv, float x0, float y0, float z0, V3 p, float eps,
                                             float &run, uint32_t &n_evals) {
    if (!(run < h * 1.00001f + eps)) return kLatticeDone;
    float tau = eps - run;
    if (N == 1 && !((h + tau) * inv <= 0.45f)) return kLatticeGeneral;
    const float *X = (const float *)tab, *Y = X + N, *Z = Y + N;
    const uint32_t *ONES = tab + 3 * N;
    if (COUNT) n_evals += 3;
    probe<COUNT>(N == 1 ? kProbeLevel0 : (N == 3 ? kProbeLevel1 : (N == 9 ? kProbeLevel2 : kProbeLevelDeep)), 1u);
    if (N == 1) {
        // The one cross of the coarsest level: its centre is the level's origin (a literal here), its digit mask is empty,
        // and the three rows through it are all this one cross, evaluated with the same operands in the same places.
        const float dx = __builtin_fabsf(p.x - x0) - h, dy = __builtin_fabsf(p.y - y0) - h, dz = __builtin_fabsf(p.z - z0) - h;
        const bool bx = dx < tau, by = dy < tau, bz = dz < tau;
        if ((bx && by) || (bx && bz) || (by && bz)) run = __builtin_fmaxf(run, -cross_value(dx, dy, dz));
        if (COUNT) n_evals += (bx && by ? 1u : 0u) + (bx && bz ? 1u : 0u) + (by && bz ? 1u : 0u);
        return kLatticeNext;
    }
    const int ix = nearest_index(p.x, x0, inv, N - 1), iy = nearest_index(p.y, y0, inv, N - 1), iz = nearest_index(p.z, z0, inv, N - 1);
    const float dx = __builtin_fabsf(p.x - X[ix]) - h, dy = __builtin_fabsf(p.y - Y[iy]) - h, dz = __builtin_fabsf(p.z - Z[iz]) - h;
    const uint32_t ox = ONES[ix], oy = ONES[iy], oz = ONES[iz];
    // Long rows (27 cells and more): where two or three of them are wanted, the same single crosses are folded first --
    // the row through the two smallest axis terms then sets a bar that the other rows, as a rule, do not pass, and they
    // are never walked.
    bool several_rows = false;
    if (N >= RM_LONG_ROW) {
        const bool bx = dx < tau, by = dy < tau, bz = dz < tau;
        several_rows = (bx && by && bz) && ((!(ox & oy) ? 1 : 0) + (!(ox & oz) ? 1 : 0) + (!(oy & oz) ? 1 : 0)) >= 2;
    }
    if (several_rows || !((h + tau) * inv <= 0.45f)) {
        // More than one centre per axis could be in range: the running value is far below what this level is worth at p
        // (in a fine tunnel, away from the coarse ones, it is still minus the depth inside the sponge's box).  One cross
        // of each row through the nearest centres is folded first -- see lattice_nearest_rows: values of real crosses, so
        // the maximum cannot overshoot -- which as a rule leaves the nearest-centre argument valid for this level
        // after all; where it does not, the caller takes the general walk.
        float best = -__builtin_inff();
        if (!(ox & oy)) {
            const uint32_t clash = oz & (ox | oy);
            best = __builtin_fmaxf(best, -cross_value(dx, dy, clash ? __builtin_fabsf(p.z - Z[iz - digits_to_zero(clash, N)]) - h : dz));
        }
        if (!(ox & oz)) {
            const uint32_t clash = oy & (ox | oz);
            best = __builtin_fmaxf(best, -cross_value(dx, clash ? __builtin_fabsf(p.y - Y[iy - digits_to_zero(clash, N)]) - h : dy, dz));
        }
        if (!(oy & oz)) {
            const uint32_t clash = ox & (oy | oz);
            best = __builtin_fmaxf(best, -cross_value(clash ? __builtin_fabsf(p.x - X[ix - digits_to_zero(clash, N)]) - h : dx, dy, dz));
        }
        if (COUNT) n_evals += 3;
        run = __builtin_fmaxf(run, best);
        if (!(run < h * 1.00001f + eps)) return kLatticeDone;
        tau = eps - run;
        if (!((h + tau) * inv <= 0.45f)) return kLatticeGeneral;
    }
    const bool bx = dx < tau, by = dy < tau, bz = dz < tau;
    const bool rz = bx && by && !(ox & oy);
    bool ry = bx && bz && !(ox & oz), rx = by && bz && !(oy & oz);
    run = lattice_rows<2, COUNT>(rz, Z, ONES, N, p.z, h, dx, dy, ox | oy, run);
    // a row that has been walked raises the running value, and with it the bar for the rows still to come: they are
    // tested again (the same test, against the value as it is now) -- where a row is long enough for that to pay (Menger-6
    // close-up 9.69 -> 8.40 ms per frame; with the rows of 3 and 9 cells tested again too the default frame lost 2 %)
    if (N >= RM_LONG_ROW && rz && (ry || rx)) {
        tau = eps - run;
        ry = ry && dx < tau && dz < tau;
        rx = rx && dy < tau && dz < tau;
    }
    run = lattice_rows<1, COUNT>(ry, Y, ONES, N, p.y, h, dx, dz, ox | oz, run);
    if (N >= RM_LONG_ROW && ry && rx) {
        tau = eps - run;
        rx = dy < tau && dz < tau;
    }
    run = lattice_rows<0, COUNT>(rx, X, ONES, N, p.x, h, dy, dz, oy | oz, run);
    if (COUNT) n_evals += (uint32_t)N * ((rz ? 1u : 0u) + (ry ? 1u : 0u) + (rx ? 1u : 0u));
    return kLatticeNext;
}

// lattice_fold at NP points with the work booked (distance-only walk)
template <bool COUNT, int NP>
__device__ __forceinline__ void lattice_sd(const uint32_t *w, ConstIndex g, uint32_t levels, uint32_t first_row,
                                           const V3 (&p)[NP], float (&acc)[NP], Work (&work)[NP]) {
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        uint32_t evals = 0, no_prim = kNoPrim;
        lattice_fold<false, COUNT>(w, g, levels, first_row, p[n], acc[n], no_prim, evals);
        if (COUNT) {
            work[n].evals += evals;
            work[n].flops += kFlopsCross * evals;
        }
    }
}


// distance of one SPHERE / BOX / CROSS record's primitive: centre c, half sizes (or radius) h
template <uint32_t CODE>
__device__ __forceinline__ float single_distance(V3 p, float cx, float cy, float cz, float h0, float h1, float h2) {
    if (CODE == REC_SPHERE) {
        const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
        return __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - h0;
    }
    const float dx = __builtin_fabsf(p.x - cx) - h0;
    const float dy = __builtin_fabsf(p.y - cy) - h1;
    const float dz = __builtin_fabsf(p.z - cz) - h2;
    return CODE == REC_CROSS ? cross_value(dx, dy, dz) : __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
}

template <uint32_t CODE, bool NEG, bool MAX, int NP>
__device__ __forceinline__ void single_sd(const V3 (&p)[NP], float (&acc)[NP], float cx, float cy, float cz, float h0,
                                          float h1, float h2) {
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        if (CODE == REC_SPHERE) acc[n] = fold_sphere<NEG, MAX>(acc[n], p[n], cx, cy, cz, h0);
        else acc[n] = fold_sd<NEG, MAX>(acc[n], single_distance<CODE>(p[n], cx, cy, cz, h0, h1, h2));
    }
}

template <uint32_t CODE, bool NEG, bool MAX, int NP>
__device__ __forceinline__ void single_sd(const RecWords &r, const V3 (&p)[NP], float (&acc)[NP]) {
    single_sd<CODE, NEG, MAX, NP>(p, acc, __uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]),
                                  __uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]));
}

// the same with colour tracking and the fold given by flags known at compile time
template <uint32_t CODE, uint32_t FLAGS, int NP>
__device__ __forceinline__ void single_surface(const V3 (&p)[NP], float (&acc)[NP], uint32_t (&accp)[NP], float cx, float cy,
                                               float cz, float h0, float h1, float h2, uint32_t prim) {
    float v[NP];
    uint32_t vp[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        v[n] = single_distance<CODE>(p[n], cx, cy, cz, h0, h1, h2);
        vp[n] = prim;
    }
    fold_value<NP, true>(FLAGS, v, vp, acc, accp);
}

template <bool NEG, bool MAX, int NP>
__device__ __forceinline__ void pop_sd(const uint32_t *slot, float (&acc)[NP]) {
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[n] = fold_sd<NEG, MAX>(__uint_as_float(slot[n * 256]), acc[n]);
}

#define RM_FOLD_CASES(CODE, CALL)                                   \
    case CODE: { CALL(false, false); } break;                       \
    case CODE | FOLD_NEG: { CALL(true, false); } break;             \
    case CODE | FOLD_MAX: { CALL(false, true); } break;             \
    case CODE | FOLD_NEG | FOLD_MAX: { CALL(true, true); } break;

__device__ __forceinline__ V3 camera_view(ConstFloats m, V3 v, float w) {
    return v3(v.x * m[0] + v.y * m[4] + v.z * m[8] + w * m[12], v.x * m[1] + v.y * m[5] + v.z * m[9] + w * m[13],
              v.x * m[2] + v.y * m[6] + v.z * m[10] + w * m[14]);
}

__device__ __forceinline__ float lerp1(float u, float v, float t) { return u + t * (v - u); }

struct LaneCounts {
    uint32_t sdf, steps, rays;
    unsigned long long evals;   // the exhaustive evaluator on a Menger-6 scene makes 3.4 M evaluations, 50 MFLOP, per call
    uint32_t slots;
    unsigned long long flops;
    unsigned long long shared_first_flops;   // of `flops`: the camera rays' first evaluation where its value is known beforehand
};

}  // namespace rmdev
