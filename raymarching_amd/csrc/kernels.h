// kernels.h -- launch wrappers of the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "device_scene.h"

namespace rmdev {

// Sub-sample image: (rows * kernel) x (width * kernel) texels, one per SSAA
// sample, rgb in xyz.  rows = band_count * band_rows.
size_t sample_texels(const FrameView &fr);

// Traces every SSAA sample of the launch's row bands (primary march, colour,
// normal, one shadow ray per light, shading) into `samples`.
// counters may be null (no counting).  wide_runs selects the one-sample-per-lane variant, for
// scenes whose cost lies in long primitive runs.
hipError_t launch_trace(const SceneView &sc, const FrameView &fr, bool wide_runs, Sample *samples,
                        WorkCounters *counters, hipStream_t stream);

// Averages the kernel x kernel samples of each pixel in the reference's
// accumulation order and writes float RGBA and/or RGBA8 (either may be null).
hipError_t launch_resolve(const FrameView &fr, const Float4 *colors, const Sample *samples, float *out_f32,
                          unsigned char *out_rgba8, hipStream_t stream);

// Scene SDF with colour tracking at n points.
hipError_t launch_eval_sdf(const SceneView &sc, const float *xyz, uint32_t n, float *sd,
                           uint32_t *prim, float *sd_march, hipStream_t stream);

// The cell grid over the root's long runs (device_scene.h: GridView), built in two passes: count_grid sizes every cell's
// list (cells[2c + 1]), reserves its place (cells[2c], through the counter *total) and keeps the cell's bound H;
// fill_grid writes the lists.  `total` must be zero before count_grid.
struct GridBuild {
    GridRun runs[kGridMaxRuns];   // grouped by class: spheres, boxes, crosses
    uint32_t n_runs;
    const Float4 *pa, *pb;
    float origin[3], cell;
    uint32_t dims[3];
    uint32_t *cells;
    float *bound;                 // H per cell
    unsigned long long *total;   // 64 bits: per-cell list sizes of heavily overlapping rows can sum beyond 2^32
    uint32_t *entries;            // fill pass: 2 words per entry (table row | class << 30, the row's lower bound over the cell), by rising bound
};
hipError_t launch_count_grid(const GridBuild &b, hipStream_t stream);
hipError_t launch_fill_grid(const GridBuild &b, hipStream_t stream);

// Loads this library's ahead-of-time code object on the current device (the runtime would do it at the first launch,
// inside the first frame).
hipError_t preload_kernels();

}  // namespace rmdev
