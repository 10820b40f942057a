// program.h -- host compiler from the pre-order node list (rm_node[]) to the
// flat device program and primitive tables of device_scene.h.
#pragma once

#include <string>
#include <vector>

#include "../../include/raymarch_hip.h"
#include "device_scene.h"

namespace rmdev {

struct HostProgram {
    std::vector<Rec> recs;
    std::vector<Float4> pa, pb;      // table rows of the RUN records
    std::vector<uint32_t> row_prim;  // depth-first primitive index of every table row
    std::vector<Float4> colors;      // rgb per primitive, by depth-first index
    std::vector<uint32_t> lattice;   // Menger lattice blocks (device_scene.h)
    uint32_t n_lattices = 0;
    uint32_t lattice_prims = 0;      // primitives covered by LATTICE records
    uint32_t n_prims = 0;
    uint32_t run_prims = 0;          // primitives evaluated through RUN records (table rows)
    uint32_t n_lists = 0;
    uint32_t max_depth = 0;
    float root_identity = 0.0f;      // starting accumulator of the root list
    // algorithmic cost of one scene SDF evaluation: 11 per sphere, 9 per box,
    // 15 per cross (primitive arithmetic + the one compare/select of the fold)
    uint64_t flops_per_sdf = 0;
    // the root's clustered runs folded with min(acc, v), if the root is a union: what the cell grid covers (device_scene.h)
    std::vector<GridRun> grid_runs;
    float grid_lo[3] = { 0, 0, 0 }, grid_hi[3] = { 0, 0, 0 };   // box around those runs' primitives (crosses: around their centres)
};

// Returns false and sets `error` if the node list is not a well-formed tree.
// use_lattice: recognise complete Menger lattices among the cross runs and emit LATTICE
// operations for them (exact, sub-linear evaluation); false = every primitive is visited.
bool compile_scene(const rm_node *nodes, uint32_t n_nodes, bool use_lattice, HostProgram &out, std::string &error);

}  // namespace rmdev
