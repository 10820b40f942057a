// interp_scene.h -- the scene evaluator that walks the record stream (device_scene.h) at run
// time.  It serves every scene; kernels specialised for one scene (jit.cpp) replace it with
// straight-line code generated from the same records.
#pragma once

#include "device_core.h"

namespace rmdev {

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

struct InterpretedScene {
    static constexpr bool kOptimistic = false;   // every walk of the record walker is exact
    template <int NP, bool COUNT>
    static __device__ __forceinline__ void sd(const SceneView &sc, const Lds &lds, const V3 (&p)[NP], float (&out_sd)[NP],
                                             Work (&work)[NP], const float (&cap)[NP], bool (&colourless)[NP]) {
        (void)cap;   // the record walker does not use the caller's upper bound
    #pragma unroll
        for (int n = 0; n < NP; ++n) colourless[n] = false;   // nor does it know a surface's colour before the tracking walk
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
                lattice_sd<COUNT, NP>(lds.lattice + cur[1], (ConstIndex)sc.lattice + cur[1], cur[0], cur[7], p, acc, work);
            } break;
            default: {   // REC_RUN with any fold
                const uint32_t first = cur[0], count = cur[1], kind = cur[2];
                if (cur[6] & RUN_CLUSTERED) {
                    switch (kind) {
                    case K_SPHERE: fold_run_clustered_kind<K_SPHERE, NP, COUNT>(sc, cur, p, acc, work); break;
                    case K_BOX: fold_run_clustered_kind<K_BOX, NP, COUNT>(sc, cur, p, acc, work); break;
                    case K_CROSS: fold_run_clustered_kind<K_CROSS, NP, COUNT>(sc, cur, p, acc, work); break;
                    case K_BOX_ISO: fold_run_clustered_kind<K_BOX_ISO, NP, COUNT>(sc, cur, p, acc, work); break;
                    default: fold_run_clustered_kind<K_CROSS_ISO, NP, COUNT>(sc, cur, p, acc, work); break;
                    }
                    break;
                }
                if (COUNT) {
    #pragma unroll
                    for (int n = 0; n < NP; ++n) {
                        work[n].evals += count;
                        work[n].flops += count * flops_of_kind(kind);
                    }
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
                for (int n = 0; n < NP; ++n) {
                    work[n].evals += 1;
                    work[n].flops += flops_of_code(cur[6] & 15u);
                }
            }
        }
    #pragma unroll
        for (int n = 0; n < NP; ++n) out_sd[n] = acc[n];
    }

    template <int NP>
    static __device__ __forceinline__ void surface(const SceneView &sc, const Lds &lds, const V3 (&p)[NP], float (&out_sd)[NP],
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
                    lattice_fold<true, false>(lds.lattice + r[1], (ConstIndex)sc.lattice + r[1], r[0], r[7], p[n], acc[n], accp[n], evals_unused[n]);
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

};

}  // namespace rmdev
