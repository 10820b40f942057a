// device_scene.h -- the scene as the kernels see it: a flat stream of 32-byte
// records plus structure-of-arrays tables for long homogeneous runs.  Shared by the
// host compiler (program.cpp) and the kernels (kernels.hip).
//
// Body::List::SDF (/root/reference/source/body.cpp:66-111) folds its children left to
// right into one running surface.  The tree is compiled, in depth-first order, into
// records whose list mode is already resolved on the host:
//
//   SPHERE / BOX / CROSS   one primitive (centre, half sizes or radius) and how it folds
//   RUN                    `count` consecutive primitives of one kind and fold, rows
//                          first..first+count-1 of the tables (long homogeneous runs only)
//   LATTICE                a run of cubic crosses that form a complete Menger lattice
//                          (see below), folded with max(s, -c)
//   PUSH slot              a nested list starts: park the accumulator, start a new one
//   POP  slot              the nested list ends: its accumulator is folded, as one child,
//                          into the parked one
//   END
//
// fold = {negate the child?, keep the larger (max) or the smaller (min)?}:
//   UNION         first: s = c     then s = min(s,  c)
//   COMPLEMENT    first: s = -c    then s = min(s, -c)
//   INTERSECTION  first: s = c     then s = max(s,  c)
//   DIFFERENCE    first: s = c     then s = max(s, -c)
// A list's accumulator starts at its fold's identity (+inf for min, -inf for max), so the first
// child needs no special case: it is folded like the others (DIFFERENCE's first child without the
// negation).  An empty list yields +inf.  min/max keep the earlier child on ties
// (std::min/std::max semantics).
#pragma once

#include <cstdint>

namespace rmdev {

enum RecCode : uint32_t {
    REC_END = 0, REC_SPHERE = 1, REC_BOX = 2, REC_CROSS = 3, REC_PUSH = 4, REC_POP = 5, REC_RUN = 6, REC_LATTICE = 7
};

// Primitive kinds of a RUN.  The *_ISO kinds are boxes/crosses whose three edge
// lengths are equal (everything the generators emit): one table row suffices.
enum Kind : uint32_t { K_SPHERE = 1, K_BOX = 2, K_CROSS = 3, K_BOX_ISO = 4, K_CROSS_ISO = 5 };

// fold bits as they sit in Rec::flags (above the 4-bit record code)
enum FoldBits : uint32_t { FOLD_NEG = 16u, FOLD_MAX = 32u };
// RUN only: the run's rows are sorted in space and come with bounding hulls (see below)
constexpr uint32_t RUN_CLUSTERED = 64u;
constexpr uint32_t kClusterSize = 8;        // rows per hull of a clustered run
constexpr uint32_t kNoRow = 0xffffffffu;

// 32 bytes, wave-uniform, fetched with one scalar load (s_load_dwordx8).
//   primitive: a = (centre.x, centre.y, centre.z), h = (half.x | radius, half.y, half.z),
//              prim = depth-first index of the primitive (its row in `colors`)
//   RUN:       a = (first row, count, kind) as integers; the primitives of one list and kind,
//              gathered in depth-first order (a fold may run in any order: ties between equal
//              distances are broken by depth-first index, which keeps the earlier child)
//              With RUN_CLUSTERED (folds that keep small values of v: min(s, c) and max(s, -c)):
//              h = (unused, unused, magnitude bound as float bits).  The
//              run's rows are sorted along a space-filling curve and followed, in the tables, by
//              a hull row per 8 consecutive rows -- a primitive of the same kind (sphere, box)
//              that contains them, so that its distance is a lower bound of theirs.  A hull whose
//              distance cannot beat the accumulator rules out its rows.  Runs of crosses hold three such
//              listings back to back, one per bar direction (device_core.h: bar_hull_needs_rows).
//   LATTICE:   a = (levels, table offset in words, count) as integers, prim = first row
//   PUSH:      a[0] = slot, a[1] = identity of the list that starts (float bits)
//   POP:       a[0] = slot
// half = size / 2 is exact in fp32, so precomputing it does not change a bit of
// "abs(p - c) - size / 2" (body.cpp:42, :52).
struct Rec {
    uint32_t a[3];
    uint32_t h[3];
    uint32_t flags;   // RecCode | fold bits
    uint32_t prim;
};

// Table row i (rows exist for the primitives of RUN records only):
//   pa[i] = (center.x, center.y, center.z, half.x or radius)
//   pb[i] = (half.y, half.z, 0, 0)            (unused by spheres and *_ISO kinds)
//   row_prim[i] = depth-first index of the primitive (colour row, tie-break rank)
struct Float4 { float x, y, z, w; };

// One SSAA sample as the trace kernel leaves it for the resolve kernel: scene::raymarch's result is
// light * colour of the surface's primitive (scene.cpp:31); the product is formed where the samples are added up,
// so a sample is 8 bytes instead of 16.  prim = kNoPrim: no surface (colour 0, 0, 0).
struct Sample { float light; uint32_t prim; };

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
//   [28..51] X_l[0], Y_l[0], Z_l[0] per level (saves a dependent table read in the index estimate)
//   ONES[i] = bit mask of the digit positions where index i has digit 1.
constexpr uint32_t kLatticeMaxLevels = 8;
constexpr uint32_t kLatHalf = 4, kLatInvSpacing = 12, kLatOffset = 20, kLatOrigin = 28, kLatHeaderWords = 52;
constexpr uint32_t kLatticeMaxWords = 12288;   // 48 KiB of LDS per workgroup at most

constexpr uint32_t kMaxDepth = 8;           // nested lists below the root
constexpr uint32_t kNoPrim = 0xffffffffu;   // "colour of an empty list": black
constexpr uint32_t kRunMin = 4;             // shorter homogeneous runs are emitted as single records

// Work of one scene-SDF call at one point, filled by the counting kernel variants only:
// primitive distance evaluations executed and their algorithmic FLOPs (11 per sphere, 9 per box,
// 15 per cross, the fold's compare included; a sphere rejected by its square-root-free test counts
// as evaluated; lattice axis scans count 3 evaluations per level visited).
struct Work {
    uint32_t evals, flops;
};
constexpr uint32_t kFlopsSphere = 11, kFlopsBox = 9, kFlopsCross = 15;
constexpr uint32_t flops_of_code(uint32_t code) {
    return code == REC_SPHERE ? kFlopsSphere : (code == REC_BOX ? kFlopsBox : kFlopsCross);
}
constexpr uint32_t flops_of_kind(uint32_t kind) {
    return kind == K_SPHERE ? kFlopsSphere : ((kind == K_BOX || kind == K_BOX_ISO) ? kFlopsBox : kFlopsCross);
}

// ---- cell grid over the root's long runs ------------------------------------------------------
// Where the scene's root is a union, the rows of its clustered RUNs (hundreds of loose primitives) are also listed per
// cell of a uniform grid: cell C lists every row j whose distance can be the smallest of all rows' anywhere in C, i.e.
// lo_j(C) <= H(C), with lo_j(C) a lower bound of row j's distance over C and H(C) = min_j hi_j(C) the smallest upper
// bound (interval bounds of the distance formulas over the cell's box, body.cpp:32-57; built on the device when the
// scene is uploaded, kernels.hip).  At a point of C the runs' fold min(acc, v_j) over all rows equals the fold over C's
// list: a row not listed is larger than the smallest row value there, so it neither changes the minimum nor ties it.
// A lane then evaluates two or three rows (its own cell's) where the hull walk visits hundreds for the wave.  The entries of
// a list carry lo_j(C) and come by rising lo_j, the three classes mixed: a row whose bound is not below the running
// minimum cannot lower it, nor can the rows after it, so the walk ends there -- the nearest row comes first as a rule, and
// most of the listed rows are never loaded.  What bounds such a walk on the GPU is the address unit of the vector memory
// path (every lane gathers from its own cell): a cell's header is 8 bytes and holds the first entry, an entry is 4 bytes, and
// nothing is read twice.
// Points outside the grid's box take the hull walk.
constexpr uint32_t kGridClassSphere = 0, kGridClassBox = 1, kGridClassCross = 2;
constexpr uint32_t kGridMaxRuns = 8;
constexpr uint32_t kGridListEnd = 0x7f80ffffu;   // (bound +infinity: ends the walk of every running minimum that is a number)
struct GridRun {
    uint32_t first, count, cls;   // table rows [first, first + count) (crosses: their first listing), kGridClass*
};
struct GridView {
    const uint32_t *cells;     // null: no grid.  Per cell 2 words: where the list's other entries start, the list's first entry
                               // (kGridListEnd: the cell lists nothing)
    const uint32_t *entries;   // per entry 1 word: the upper 16 bits of a lower bound of the row's distance over the cell (a
                               // float's sign, exponent and 7 bits: rounded down), class (kGridClass*, 2 bits), table row
                               // (14 bits).  By rising bound; kGridListEnd ends a list.
    float origin[3];
    float inv_cell, cell;
    uint32_t dims[3];
};

struct SceneView {
    const Rec *recs;
    const Float4 *pa;
    const Float4 *pb;
    const uint32_t *row_prim;
    const Float4 *colors;   // rgb per primitive, by depth-first index
    const Float4 *lights;   // xyz per light
    const uint32_t *lattice;   // lattice blocks (staged into LDS by every workgroup)
    uint32_t lattice_words;
    uint32_t n_recs;        // records before the END record
    float root_identity;    // the root list's starting accumulator
    uint32_t max_depth;     // deepest PUSH slot + 1 (sizes the LDS parking area)
    uint32_t n_lights;
    uint32_t n_prims;
    const float *camera;    // 16 floats of the column-major camera matrix, then the focal length
    GridView grid;
    // Every camera ray starts at the camera's position with the same evaluation: its value, found once when the scene
    // is uploaded (camera_sd_known = 0: not found, every ray evaluates it)
    float camera_sd;
    uint32_t camera_sd_known;
    // 4e-6 of the largest |centre| + size among the scene's primitives: the part of the march loop's `cap` margin that
    // covers the rounding of a value decided by a primitive far from the point or much larger than the step
    float cap_slack;
};

constexpr uint32_t kTileBits = 27u, kTileMask = (1u << kTileBits) - 1u;

struct FrameView {
    uint32_t width, height, kernel;
    int32_t iterations;
    float saturation, surface_eps, offset_eps;
    // row bands rendered by this launch (see rm_rows)
    uint32_t band_rows, first_band, band_stride, band_count;
    // Tile scheduling of the trace kernel (either may be null).  Workgroup b renders tile
    // tile_order[b] (tiles numbered row-major over the launch's sample image) instead of tile b, and
    // every workgroup stores how long it ran, in 10 ns ticks, into tile_cost[its tile].  The host
    // sorts one frame's costs to start the next frame's expensive tiles first.
    // A tile_order entry is a tile number in its low kTileBits bits and, in the bits above, 0 for the whole tile,
    // 1 + q for quarter q of it (4 x 4 samples on the wave's first 16 lanes) or 5 + s for sixteenth s (2 x 2 samples
    // on the first 4 lanes): the host lists the few tiles that run longest -- a frame cannot end before they do --
    // as 4 or 16 workgroups instead of one, which ends them sooner (each part waits only for the slowest of its own
    // samples at every step).  extra_slots = the number of entries beyond one per tile; the grid has as many more
    // rows as they take.
    const uint32_t *tile_order;
    uint32_t extra_slots;
    uint32_t *tile_cost;
    // Which image band each band of the launch is (null: band b is first_band + b * band_stride): lets
    // a caller deal bands to GPUs by measured cost instead of cyclically.
    const uint32_t *band_list;
    // Zero: pixels go to a compact image of the launch's bands (row b * band_rows + i of the output is row i of the
    // launch's band b).  Nonzero: the outputs are whole frames of `height` rows and every rendered row is stored at its own
    // place in them; rows of bands that lie below the image are not stored at all.  Several launches -- on several GPUs --
    // then fill one frame between them without a copy.
    uint32_t rows_in_frame;
    // Pixels straight from the trace kernel (either pointer may be null; both null: the kernel writes `samples` for
    // the resolve kernel instead).  Set only for kernels of one wave per workgroup and one sample per lane, and only
    // when the SSAA kernel divides the tile's 8 samples per side, so that a workgroup holds whole pixels.
    float *pixels_f32;
    unsigned char *pixels_rgba8;
    // Nonzero: samples whose surface has no colour skip the normal and the shadow rays (their product with
    // the lighting is zero either way) and shadow rays stop once past their light; zero: every sample and
    // ray does the reference's full work, which is what work counts that are compared with the reference's need.
    uint32_t skip_unlit;
    // One word per tile of the launch's sample image, zero between launches (null: none).  The kernels generated for a scene
    // are optimistic: where a point's lattice walk cannot be decided from the nearest centres (deep inside material,
    // where rays as good as never go) they go on with a lower bound and set the tile's word; the scene's exact kernel,
    // launched behind them, renders the set tiles again -- every walk exact -- and clears their words.
    uint32_t *tile_flags;
    // Null, or a word the exact kernel adds one to for every tile it renders again.
    uint32_t *exact_count;
    // Diagnostic (null: off): four words per workgroup slot of the launch -- its tile_order entry, its duration in 10 ns
    // ticks, the iterations its wave spent in the camera march (low 16 bits) and in its shadow marches (high 16 bits; the
    // loops' trip counts: the wave's longest ray each time), and the clock at its start.  Tells which workgroups a frame
    // waits for, and what for.
    uint32_t *slot_stats;
    // Tile of the kernel whose lanes take the tile's next sample when their own is finished (trace_refill.h):
    // 2^refill_w_log2 x 2^refill_h_log2 samples, both at least 8.
    uint32_t refill_w_log2, refill_h_log2;
};

struct WorkCounters {
    unsigned long long sdf_calls;
    unsigned long long march_steps;
    unsigned long long rays;
    unsigned long long prim_evals;   // primitive distance evaluations (incl. lattice axis scans, 3 per level)
    unsigned long long flops;        // algorithmic FLOPs of the primitive evaluations executed (see Work)
    unsigned long long march_slots;  // march-loop iterations x samples of the lanes that sat through them (live or idle)
    // Lane occupancy of code regions, as pairs (lanes that executed the region x its weight, 64 x weight per wave that
    // went through it): see kProbe* in device_core.h.
    unsigned long long probe[16];
    // the part of `flops` spent on the camera rays' first evaluation where SceneView::camera_sd is known: renders
    // that do not count skip that evaluation (it is the same for every ray)
    unsigned long long flops_shared_first;
};

}  // namespace rmdev
