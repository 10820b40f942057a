// kernels.hip -- gfx950 kernels of the sphere-tracing render path.
//
// Two SSAA samples per lane (two horizontally adjacent sub-samples), 16 x 8 samples per
// wave.  A lane does for each of its samples what scene::raymarch does
// (/root/reference/source/scene.cpp:26-62): primary march, colour of the hit,
// central-difference normal, one shadow march per light, clamped Lambert sum.  A second
// kernel averages each pixel's kernel x kernel samples in the reference's accumulation
// order (render.cpp:103-119) and quantises.
//
// Arithmetic contract: every fp32 operation below is a single correctly rounded IEEE
// operation in the reference's order -- compiled with -ffp-contract=off (no fused
// multiply-add) and hipcc's default correctly rounded sqrt / divide -- so the float
// framebuffer is bit-identical to the CPU path's (signs of zeros and NaN payloads aside).
//
// The scene is a flat record stream (device_scene.h).  All lanes of a wave walk the same
// stream at the same time, so records and table rows are fetched through wave-uniform
// addresses (scalar loads, operands in SGPRs); only the sample positions live in vector
// registers.  The scalar unit, one per CU, pays for every record; evaluating two points
// per lane per walk halves that cost per point.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace rmdev {

namespace {

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

// ---- LDS: lattice tables, then the parking area of nested lists -------------

extern __shared__ uint32_t lds_words[];

struct Lds {
    const uint32_t *lattice;
    uint32_t *park;   // [slot][sd | prim][point][thread]
};

__device__ __forceinline__ Lds stage_lds(const SceneView &sc) {
    for (uint32_t i = threadIdx.x; i < sc.lattice_words; i += blockDim.x) lds_words[i] = sc.lattice[i];
    __syncthreads();
    return Lds{ lds_words, lds_words + sc.lattice_words };
}

inline uint32_t lds_bytes(const SceneView &sc) {
    return (sc.lattice_words + sc.max_depth * 2u * (uint32_t)kMaxPts * 256u) * 4u;
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

template <bool TRACK, bool COUNT>
__device__ __forceinline__ void lattice_fold(const uint32_t *w, uint32_t levels, uint32_t first_row, V3 p, float &acc,
                                             uint32_t &acc_prim, uint32_t &n_evals) {
    const float bound = __builtin_fabsf(p.x) + __builtin_fabsf(p.y) + __builtin_fabsf(p.z) + __uint_as_float(w[2]);
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
        axis_range(X, n, p.x, h, inv, tau, lox, hix);
        axis_range(Y, n, p.y, h, inv, tau, loy, hiy);
        axis_range(Z, n, p.z, h, inv, tau, loz, hiz);
        if (COUNT) n_evals += 3;
        const bool bx = lox <= hix, by = loy <= hiy, bz = loz <= hiz;
        if (bx && by) {
            for (int i = lox; i <= hix; ++i)
                for (int j = loy; j <= hiy; ++j) {
                    const uint32_t oi = ONES[i], oj = ONES[j];
                    if (oi & oj) continue;   // two zero offsets at one subdivision step: no such sub-cube
                    const float dx = axis_term(X, i, p.x, h), dy = axis_term(Y, j, p.y, h);
                    for (int k = 0; k < n; ++k) {
                        const float v = -cross_value(dx, dy, axis_term(Z, k, p.z, h));
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
                    const float dx = axis_term(X, i, p.x, h), dz = axis_term(Z, k, p.z, h);
                    for (int j = 0; j < n; ++j) {
                        const float v = -cross_value(dx, axis_term(Y, j, p.y, h), dz);
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
                    const float dy = axis_term(Y, j, p.y, h), dz = axis_term(Z, k, p.z, h);
                    for (int i = 0; i < n; ++i) {
                        const float v = -cross_value(axis_term(X, i, p.x, h), dy, dz);
                        run = (ONES[i] & (oj | ok)) ? run : __builtin_fmaxf(run, v);
                    }
                    if (COUNT) n_evals += n;
                }
        }
    }
    if (TRACK) acc_prim = (acc < run) ? first_row : acc_prim;   // all crosses of a lattice share one colour
    acc = run;
}

// ---- Body::List::SDF for the whole tree (body.cpp:66-111) at NP points per lane ----------------
//
// Two walkers over the same record stream:
//   scene_sd        distance only (march steps, gradient taps, shadow rays): the hot one.  Every
//                   (record kind, fold) pair is a case of one switch with the fold compiled in, so
//                   a record costs a handful of scalar instructions and no selects.
//   scene_surface   distance and the depth-first index of the primitive whose colour the resulting
//                   Surface carries (once per sample, and the rm_eval_sdf test hook).
// A list's accumulator starts from its fold's identity (+inf for min, -inf for max; the host puts
// it into the PUSH record and SceneView::root_identity), so "the first child replaces the
// accumulator" needs no case of its own; an empty list keeps +inf / no colour.

template <bool NEG, bool MAX>
__device__ __forceinline__ float fold_sd(float acc, float v) {
    return MAX ? __builtin_fmaxf(acc, NEG ? -v : v) : __builtin_fminf(acc, NEG ? -v : v);
}

template <uint32_t CODE, bool NEG, bool MAX, int NP>
__device__ __forceinline__ void single_sd(const RecWords &r, const V3 (&p)[NP], float (&acc)[NP]) {
    const float cx = __uint_as_float(r[0]), cy = __uint_as_float(r[1]), cz = __uint_as_float(r[2]);
    const float h0 = __uint_as_float(r[3]), h1 = __uint_as_float(r[4]), h2 = __uint_as_float(r[5]);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        float v;
        if (CODE == REC_SPHERE) {
            const float dx = cx - p[n].x, dy = cy - p[n].y, dz = cz - p[n].z;
            v = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - h0;
        } else {
            const float dx = __builtin_fabsf(p[n].x - cx) - h0;
            const float dy = __builtin_fabsf(p[n].y - cy) - h1;
            const float dz = __builtin_fabsf(p[n].z - cz) - h2;
            v = CODE == REC_CROSS ? cross_value(dx, dy, dz) : __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
        }
        acc[n] = fold_sd<NEG, MAX>(acc[n], v);
    }
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

template <int NP, bool COUNT>
__device__ __forceinline__ void scene_sd(const SceneView &sc, const Lds &lds, const V3 (&p)[NP], float (&out_sd)[NP],
                                         uint32_t (&n_evals)[NP]) {
    float acc[NP];
    uint32_t no_prim[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        acc[n] = sc.root_identity;
        no_prim[n] = kNoPrim;
    }
    // Records are fetched one ahead: the scalar load of record pc + 1 is in flight while record pc
    // is evaluated (the stream ends with an END record, so pc + 1 is always readable).
    const ConstRecs recs = (ConstRecs)(sc.recs);
    RecWords next = recs[0];
    for (uint32_t pc = 0; pc < sc.n_recs; ++pc) {
        const RecWords cur = next;
        // scalar loads return out of order, so a wait cannot single out the older load: retire
        // record pc first (lgkmcnt(0)), then put record pc + 1 in flight behind it
        __builtin_amdgcn_s_waitcnt(0xc07f);
        next = recs[pc + 1];
        switch (cur[6] & 63u) {
#define RM_SPHERE(NEG, MAX) single_sd<REC_SPHERE, NEG, MAX, NP>(cur, p, acc)
#define RM_BOX(NEG, MAX) single_sd<REC_BOX, NEG, MAX, NP>(cur, p, acc)
#define RM_CROSS(NEG, MAX) single_sd<REC_CROSS, NEG, MAX, NP>(cur, p, acc)
#define RM_POP(NEG, MAX) pop_sd<NEG, MAX, NP>(lds.park + cur[0] * (2u * kMaxPts * 256u) + threadIdx.x, acc)
            RM_FOLD_CASES(REC_SPHERE, RM_SPHERE)
            RM_FOLD_CASES(REC_BOX, RM_BOX)
            RM_FOLD_CASES(REC_CROSS, RM_CROSS)
            RM_FOLD_CASES(REC_POP, RM_POP)
#undef RM_SPHERE
#undef RM_BOX
#undef RM_CROSS
#undef RM_POP
        case REC_PUSH: {
            uint32_t *slot = lds.park + cur[0] * (2u * kMaxPts * 256u) + threadIdx.x;
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                slot[n * 256] = __float_as_uint(acc[n]);
                acc[n] = __uint_as_float(cur[1]);   // the new list's identity
            }
        } break;
        case REC_LATTICE | FOLD_NEG | FOLD_MAX: {
#pragma unroll
            for (int n = 0; n < NP; ++n)
                lattice_fold<false, COUNT>(lds.lattice + cur[1], cur[0], cur[7], p[n], acc[n], no_prim[n], n_evals[n]);
        } break;
        default: {   // REC_RUN with any fold
            const uint32_t first = cur[0], count = cur[1], kind = cur[2];
            if (COUNT) {
#pragma unroll
                for (int n = 0; n < NP; ++n) n_evals[n] += count;
            }
            switch (kind) {
            case K_SPHERE: fold_run_kind<K_SPHERE, NP, false>(sc, cur[6], first, count, p, acc, no_prim); break;
            case K_BOX: fold_run_kind<K_BOX, NP, false>(sc, cur[6], first, count, p, acc, no_prim); break;
            case K_CROSS: fold_run_kind<K_CROSS, NP, false>(sc, cur[6], first, count, p, acc, no_prim); break;
            case K_BOX_ISO: fold_run_kind<K_BOX_ISO, NP, false>(sc, cur[6], first, count, p, acc, no_prim); break;
            default: fold_run_kind<K_CROSS_ISO, NP, false>(sc, cur[6], first, count, p, acc, no_prim); break;
            }
        } break;
        }
        if (COUNT && (cur[6] & 15u) <= REC_CROSS) {
#pragma unroll
            for (int n = 0; n < NP; ++n) n_evals[n] += 1;
        }
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) out_sd[n] = acc[n];
}

template <int NP>
__device__ __forceinline__ void scene_surface(const SceneView &sc, const Lds &lds, const V3 (&p)[NP], float (&out_sd)[NP],
                                              uint32_t (&out_prim)[NP]) {
    float acc[NP];
    uint32_t accp[NP], evals_unused[NP] = {};
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        acc[n] = sc.root_identity;
        accp[n] = kNoPrim;
    }
    const ConstRecs recs = (ConstRecs)(sc.recs);
    for (uint32_t pc = 0; pc < sc.n_recs; ++pc) {
        const RecWords r = recs[pc];
        const uint32_t code = r[6] & 15u;
        if (code <= REC_CROSS) {
            float v[NP];
            uint32_t vp[NP];
            const float cx = __uint_as_float(r[0]), cy = __uint_as_float(r[1]), cz = __uint_as_float(r[2]);
            const float h0 = __uint_as_float(r[3]), h1 = __uint_as_float(r[4]), h2 = __uint_as_float(r[5]);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                if (code == REC_SPHERE) {
                    const float dx = cx - p[n].x, dy = cy - p[n].y, dz = cz - p[n].z;
                    v[n] = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) - h0;
                } else {
                    const float dx = __builtin_fabsf(p[n].x - cx) - h0;
                    const float dy = __builtin_fabsf(p[n].y - cy) - h1;
                    const float dz = __builtin_fabsf(p[n].z - cz) - h2;
                    v[n] = code == REC_CROSS ? cross_value(dx, dy, dz) : __builtin_fmaxf(__builtin_fmaxf(dx, dy), dz);
                }
                vp[n] = r[7];
            }
            fold_value<NP, true>(r[6], v, vp, acc, accp);
        } else if (code == REC_RUN) {
            const uint32_t first = r[0], count = r[1], kind = r[2];
            switch (kind) {
            case K_SPHERE: fold_run_kind<K_SPHERE, NP, true>(sc, r[6], first, count, p, acc, accp); break;
            case K_BOX: fold_run_kind<K_BOX, NP, true>(sc, r[6], first, count, p, acc, accp); break;
            case K_CROSS: fold_run_kind<K_CROSS, NP, true>(sc, r[6], first, count, p, acc, accp); break;
            case K_BOX_ISO: fold_run_kind<K_BOX_ISO, NP, true>(sc, r[6], first, count, p, acc, accp); break;
            default: fold_run_kind<K_CROSS_ISO, NP, true>(sc, r[6], first, count, p, acc, accp); break;
            }
        } else if (code == REC_LATTICE) {
#pragma unroll
            for (int n = 0; n < NP; ++n)
                lattice_fold<true, false>(lds.lattice + r[1], r[0], r[7], p[n], acc[n], accp[n], evals_unused[n]);
        } else if (code == REC_PUSH) {
            uint32_t *slot = lds.park + r[0] * (2u * kMaxPts * 256u) + threadIdx.x;
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                slot[n * 256] = __float_as_uint(acc[n]);
                slot[(kMaxPts + n) * 256] = accp[n];
                acc[n] = __uint_as_float(r[1]);
                accp[n] = kNoPrim;
            }
        } else {   // REC_POP: the finished list is one child of the parked accumulator
            const uint32_t *slot = lds.park + r[0] * (2u * kMaxPts * 256u) + threadIdx.x;
            float v[NP];
            uint32_t vp[NP];
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                v[n] = acc[n];
                vp[n] = accp[n];
                acc[n] = __uint_as_float(slot[n * 256]);
                accp[n] = slot[(kMaxPts + n) * 256];
            }
            fold_value<NP, true>(r[6], v, vp, acc, accp);
        }
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        out_sd[n] = acc[n];
        out_prim[n] = accp[n];
    }
}

__device__ __forceinline__ V3 camera_view(ConstFloats m, V3 v, float w) {
    return v3(v.x * m[0] + v.y * m[4] + v.z * m[8] + w * m[12], v.x * m[1] + v.y * m[5] + v.z * m[9] + w * m[13],
              v.x * m[2] + v.y * m[6] + v.z * m[10] + w * m[14]);
}

__device__ __forceinline__ float lerp1(float u, float v, float t) { return u + t * (v - u); }

struct LaneCounts {
    uint32_t sdf, steps, rays, evals;
};

// scene::surface (scene.cpp:34-42) for the lane's two samples in lockstep: advances p[s]
// along ray[s] while sample s is live; at[s] is the point of its last SDF evaluation (whose
// Surface the reference returns).
template <int kPts, bool COUNT>
__device__ __forceinline__ void march(const SceneView &sc, const Lds &lds, const FrameView &fr, V3 (&p)[kPts],
                                      const V3 (&ray)[kPts], V3 (&at)[kPts], const bool (&live)[kPts], LaneCounts &cnt) {
    bool done[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        done[s] = !live[s];
        if (COUNT && live[s]) cnt.rays++;
    }
    for (int it = 0; it < fr.iterations; ++it) {
        bool all_done = true;
#pragma unroll
        for (int s = 0; s < kPts; ++s) all_done = all_done && done[s];
        if (all_done) break;
        float sd[kPts];
        uint32_t evals[kPts] = {};
        scene_sd<kPts, COUNT>(sc, lds, p, sd, evals);
#pragma unroll
        for (int s = 0; s < kPts; ++s) {
            if (!done[s]) {
                at[s] = p[s];
                p[s] = p[s] + ray[s] * sd[s];
                done[s] = sd[s] < fr.surface_eps;
                if (COUNT) {
                    cnt.sdf++;
                    cnt.steps++;
                    cnt.evals += evals[s];
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
template <int kPts, int MIN_WAVES, bool COUNT>
__global__ __launch_bounds__(256, MIN_WAVES) void trace_kernel(const SceneView sc, const FrameView fr, Float4 *__restrict__ samples,
                                                     WorkCounters *__restrict__ counters) {
    __shared__ unsigned long long block_counts[4];
    if (COUNT) {
        if (threadIdx.x < 4) block_counts[threadIdx.x] = 0;
    }
    const Lds lds = stage_lds(sc);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = fr.kernel;
    const uint32_t sw = fr.width * k;
    const uint32_t sh = fr.band_count * fr.band_rows * k;
    const uint32_t sx0 = blockIdx.x * (16u * kPts) + (wave & 1u) * (8u * kPts) + (lane & 7u) * (uint32_t)kPts;
    const uint32_t sy = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    LaneCounts cnt = { 0, 0, 0, 0 };

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
    const uint32_t pi = (fr.first_band + band * fr.band_stride) * fr.band_rows + (ly - band * fr.band_rows);
    const float uv1y = (float)pi * psy, uv2y = (float)(pi + 1u) * psy;
    const float p1y = lerp1(s1y, s2y, uv1y), p2y = lerp1(s1y, s2y, uv2y);
    const float uvy = (float)(j + 1u) / (float)k;
    const float y = lerp1(p1y, p2y, uvy);

    bool stored[kPts], live[kPts];
#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        stored[s] = sx0 + (uint32_t)s < sw && sy < sh;
        live[s] = stored[s] && pi < fr.height;
    }

    // ---- scene::raymarch, scene.cpp:26-32.  Ray 0 is the camera ray; ray 1 + l is the shadow
    // ray to light l (scene::lighting / scene::shadow, scene.cpp:45-62).  One march instance.
    V3 hit[kPts], normal[kPts], p[kPts], ray[kPts], at[kPts];
    float lighting[kPts];
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
            }
        } else {
            const RowWords lp = ((ConstRows)sc.lights)[r - 1];
            light = v3(lp[0], lp[1], lp[2]);
#pragma unroll
            for (int s = 0; s < kPts; ++s) {
                ray[s] = normalize3(light - hit[s]);
                p[s] = hit[s] + normal[s] * lift;
            }
        }
#pragma unroll
        for (int s = 0; s < kPts; ++s) at[s] = p[s];
        march<kPts, COUNT>(sc, lds, fr, p, ray, at, live, cnt);
        if (r == 0) {
#pragma unroll
            for (int s = 0; s < kPts; ++s) hit[s] = p[s];
            // the Surface scene::surface returned: colour of the last evaluation
            if (fr.iterations > 0) {
                float sd_unused[kPts];
                scene_surface<kPts>(sc, lds, at, sd_unused, prim);
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
                    uint32_t evals[kPts] = {};
                    scene_sd<kPts, COUNT>(sc, lds, q, sd, evals);
#pragma unroll
                    for (int s = 0; s < kPts; ++s) {
                        if (COUNT && live[s]) {
                            cnt.sdf++;
                            cnt.evals += evals[s];
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

#pragma unroll
    for (int s = 0; s < kPts; ++s) {
        if (!stored[s]) continue;
        V3 color = v3(0.0f, 0.0f, 0.0f);
        if (live[s]) {
            const float light = __builtin_fminf(__builtin_fmaxf(fr.saturation, lighting[s]), 1.0f);
            V3 base = v3(0.0f, 0.0f, 0.0f);
            if (prim[s] != kNoPrim) {
                const Float4 c = sc.colors[prim[s]];
                base = v3(c.x, c.y, c.z);
            }
            color = v3(light * base.x, light * base.y, light * base.z);
        }
        samples[(size_t)sy * sw + sx0 + (uint32_t)s] = Float4{ color.x, color.y, color.z, 0.0f };
    }

    if (COUNT) {
        atomicAdd(&block_counts[0], (unsigned long long)cnt.sdf);
        atomicAdd(&block_counts[1], (unsigned long long)cnt.steps);
        atomicAdd(&block_counts[2], (unsigned long long)cnt.rays);
        atomicAdd(&block_counts[3], (unsigned long long)cnt.evals);
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
    const Lds lds = stage_lds(sc);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t tt = t < n ? t : n - 1;   // every lane walks the scene (nested lists park through LDS)
    const V3 p[1] = { v3(xyz[3 * tt], xyz[3 * tt + 1], xyz[3 * tt + 2]) };
    float d[1];
    uint32_t pr[1];
    scene_surface<1>(sc, lds, p, d, pr);
    if (t < n) {
        sd[t] = d[0];
        prim[t] = pr[0];
    }
}

}  // namespace

size_t sample_texels(const FrameView &fr) {
    return (size_t)fr.band_count * fr.band_rows * fr.kernel * (size_t)fr.width * fr.kernel;
}

hipError_t launch_trace(const SceneView &sc, const FrameView &fr, bool wide_runs, Float4 *samples, WorkCounters *counters,
                        hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0) return hipSuccess;
    const uint32_t pts = wide_runs ? 1u : 2u;
    const dim3 grid((sw + 16u * pts - 1u) / (16u * pts), (sh + 15u) / 16u, 1u);
    const uint32_t lds = lds_bytes(sc);
    if (wide_runs) {
        if (counters) hipLaunchKernelGGL((trace_kernel<1, 4, true>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
        else hipLaunchKernelGGL((trace_kernel<1, 4, false>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
    } else {
        if (counters) hipLaunchKernelGGL((trace_kernel<2, 8, true>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
        else hipLaunchKernelGGL((trace_kernel<2, 8, false>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
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
    hipLaunchKernelGGL(eval_sdf_kernel, dim3((n + 255u) / 256u), dim3(256), lds_bytes(sc), stream, sc, xyz, n, sd, prim);
    return hipGetLastError();
}

}  // namespace rmdev
