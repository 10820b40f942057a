// device_scene.h -- the scene as the kernels see it: a flat "program" plus
// structure-of-arrays primitive tables.  Shared by the host compiler
// (program.cpp) and the kernels (kernels.hip).
//
// Body::List::SDF (/root/reference/source/body.cpp:66-111) folds its children
// left to right into one running surface.  The tree is therefore compiled, in
// depth-first order, into three kinds of operations whose list mode is already
// resolved on the host:
//
//   RUN  kind, fold, first, count   fold `count` consecutive primitives of one
//                                   kind (table rows first..first+count-1)
//                                   into the accumulator
//   PUSH slot                       a nested list starts: park the accumulator
//                                   in stack slot `slot`, start a new one
//   POP  slot, fold                 the nested list ends: its accumulator is
//                                   folded, as one child, into the parked one
//   LATTICE levels, fold, first, count, table   a run of `count` cubic crosses that form a
//                                   complete Menger lattice (see below), folded with max(s, -c)
//   END
//
// fold = {negate the child?, keep the larger (max) or the smaller (min)?,
// is this the list's first child (then the child replaces the accumulator)?}:
//   UNION         first: s = c     then s = min(s,  c)
//   COMPLEMENT    first: s = -c    then s = min(s, -c)
//   INTERSECTION  first: s = c     then s = max(s,  c)
//   DIFFERENCE    first: s = c     then s = max(s, -c)
// min/max keep the accumulator on ties (std::min/std::max semantics).
#pragma once

#include <cstdint>

namespace rmdev {

enum OpCode : uint32_t { OP_END = 0, OP_RUN = 1, OP_PUSH = 2, OP_POP = 3, OP_LATTICE = 4 };

// Primitive kinds of a RUN.  The *_ISO kinds are boxes/crosses whose three edge
// lengths are equal (everything the generators emit): one table row suffices.
enum Kind : uint32_t { K_SPHERE = 1, K_BOX = 2, K_CROSS = 3, K_BOX_ISO = 4, K_CROSS_ISO = 5 };

enum FoldBits : uint32_t { FOLD_NEG = 1u, FOLD_MAX = 2u, FOLD_FIRST = 4u };

// 16 bytes, fetched with one scalar load.
struct Op {
    uint32_t code;   // OpCode
    uint32_t arg;    // RUN: kind | fold << 8     PUSH: slot     POP: slot | fold << 8
                     // LATTICE: levels | fold << 8 | (table offset in words) << 12
    uint32_t first;  // RUN: first table row (= depth-first primitive index)
    uint32_t count;  // RUN: number of primitives
};

// Table row i describes the i-th primitive in depth-first order:
//   pa[i] = (center.x, center.y, center.z, half.x or radius)
//   pb[i] = (half.y, half.z, 0, 0)            (unused by spheres and *_ISO kinds)
// half = size / 2, exact in fp32, so precomputing it does not change a bit of
// "abs(p - c) - size / 2" (body.cpp:42, :52).
struct Float4 { float x, y, z, w; };

// ---- Menger lattice tables -------------------------------------------------
// A run of cubic crosses is a "complete Menger lattice of n levels" when cross m (in the
// generator's depth-first order, body.cpp:114-147) sits at (X_l[ix], Y_l[iy], Z_l[iz]) with
// half-width half[l], where l is its level (0 = the one big cross), the per-level, per-axis
// coordinate tables X_l, Y_l, Z_l have 3^l sorted, evenly spaced entries, and (ix,iy,iz)
// runs over exactly the index triples whose base-3 digits have, at every digit position, at
// most one digit equal to 1 (the 20 sub-cubes kept at every subdivision step).  The host
// verifies this bit for bit on the node list itself (program.cpp) before emitting LATTICE.
// A block of 32-bit words per lattice:
//   [0] levels  [2] magnitude bound (float)  [4..11] half[l]  [12..19] 1/spacing[l]
//   [20..27] word offset of level l's tables: X[3^l], Y[3^l], Z[3^l], ONES[3^l]
//   ONES[i] = bit mask of the digit positions where index i has digit 1.
constexpr uint32_t kLatticeMaxLevels = 8;
constexpr uint32_t kLatHalf = 4, kLatInvSpacing = 12, kLatOffset = 20, kLatHeaderWords = 28;
constexpr uint32_t kLatticeMaxWords = 12288;   // 48 KiB of LDS per workgroup at most

constexpr uint32_t kMaxDepth = 8;           // nested lists below the root
constexpr uint32_t kNoPrim = 0xffffffffu;   // "colour of an empty list": black

struct SceneView {
    const Op *ops;
    const Float4 *pa;
    const Float4 *pb;
    const Float4 *colors;   // rgb per primitive
    const Float4 *lights;   // xyz per light
    const uint32_t *lattice;   // lattice blocks (staged into LDS by every workgroup)
    uint32_t lattice_words;
    uint32_t n_lights;
    uint32_t n_prims;
    float cam[16];          // column-major camera matrix
    float focal;
};

struct FrameView {
    uint32_t width, height, kernel;
    int32_t iterations;
    float saturation, surface_eps, offset_eps;
    // row bands rendered by this launch (see rm_rows)
    uint32_t band_rows, first_band, band_stride, band_count;
};

struct WorkCounters {
    unsigned long long sdf_calls;
    unsigned long long march_steps;
    unsigned long long rays;
    unsigned long long prim_evals;   // primitive distance evaluations (incl. lattice axis scans, 3 per level)
};

}  // namespace rmdev
