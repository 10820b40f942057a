// codegen.h -- turns a compiled scene (record stream) into HIP source: a `CompiledSceneT`
// evaluator whose distance functions are straight-line code with the scene's numbers as
// literals, plus the extern "C" kernels that use it.
#pragma once

#include <string>

#include "program.h"

namespace rmdev {

// Kernels emitted: rm_trace (points_per_lane samples per lane, register budget for min_waves waves
// per SIMD, launched with 1, 4 or 9 waves per workgroup, with or without work counting) and
// rm_eval_sdf; signatures as the kernels of kernels.hip.  Without work counting also rm_trace_exact: rm_trace walks the scene
// optimistically and flags the tiles where that was not enough (FrameView::tile_flags), rm_trace_exact renders those again.
// refill: also rm_trace_refill, the kernel whose lanes take their tile's next sample when their own is finished.
std::string generate_scene_source(const HostProgram &prog, int points_per_lane, int min_waves, bool counting, bool refill);

}  // namespace rmdev
