// kernels.hip -- the ahead-of-time gfx950 kernels of the sphere-tracing render path: the trace
// kernel that walks the scene's record stream (interp_scene.h), the SDF test hook and the resolve
// kernel.  Kernels specialised for one scene are generated and compiled at upload time instead
// (codegen.cpp, jit.cpp); both share the kernel body (trace_kernel.h).
//
// A lane does for each of its SSAA samples what scene::raymarch does
// (/root/reference/source/scene.cpp:26-62): primary march, colour of the hit,
// central-difference normal, one shadow march per light, clamped Lambert sum.  The resolve
// kernel averages each pixel's kernel x kernel samples in the reference's accumulation
// order (render.cpp:103-119) and quantises.
//
// Arithmetic contract: every fp32 operation that can reach the frame is a single correctly
// rounded IEEE operation in the reference's order -- compiled with -ffp-contract=off (no fused
// multiply-add) and hipcc's correctly rounded sqrt / divide -- so the float framebuffer is
// bit-identical to the CPU path's (signs of zeros and NaN payloads aside).
//
// The record walker: all lanes of a wave walk the same stream at the same time, so records and
// table rows are fetched through wave-uniform addresses (scalar loads, operands in SGPRs); only
// the sample positions live in vector registers.  The scalar unit, one per CU, pays for every
// record; evaluating two points per lane per walk halves that cost per point.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include "interp_scene.h"
#include "trace_kernel.h"

namespace rmdev {

namespace {

// kPts / MIN_WAVES variants: see trace_body.  MIN_WAVES = waves per SIMD the register
// allocation must allow.
// kWaves: waves per workgroup (4: 16 x 16 lanes; 1: 8 x 8, finer scheduling for scenes whose tiles run long).
template <int kPts, int MIN_WAVES, bool COUNT, int kWaves>
__global__ __launch_bounds__(64 * kWaves, MIN_WAVES) void trace_kernel(const SceneView sc, const FrameView fr,
                                                                       Sample *__restrict__ samples,
                                                                       WorkCounters *__restrict__ counters) {
    trace_body<InterpretedScene, kPts, COUNT>(sc, fr, samples, counters);
}

__global__ __launch_bounds__(256) void eval_sdf_kernel(const SceneView sc, const float *__restrict__ xyz, uint32_t n,
                                                        float *__restrict__ sd, uint32_t *__restrict__ prim,
                                                        float *__restrict__ sd_march) {
    eval_sdf_body<InterpretedScene>(sc, xyz, n, sd, prim, sd_march);
}

// render.cpp:103-119: total += colour over i (outer, x) then j (inner, y);
// pixel = total / (k*k), alpha 1.  RGBA8 = round(clamp(c, 0, 1) * 255).
__global__ __launch_bounds__(256) void resolve_kernel(const FrameView fr, const Float4 *__restrict__ colors, const Sample *__restrict__ samples,
                                                       float *__restrict__ out_f32, unsigned char *__restrict__ out_rgba8) {
    const uint32_t rows = fr.band_count * fr.band_rows;
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= fr.width || y >= rows) return;
    const uint32_t out_y = placed_row(fr, y);
    if (out_y == kNoOutputRow) return;
    const uint32_t k = fr.kernel;
    const size_t sw = (size_t)fr.width * k;
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
    for (uint32_t i = 0; i < k; ++i) {
        for (uint32_t j = 0; j < k; ++j) {
            const V3 c = sample_colour(colors, samples[((size_t)y * k + j) * sw + (size_t)x * k + i]);
            tx += c.x;
            ty += c.y;
            tz += c.z;
        }
    }
    store_pixel(tx, ty, tz, k, fr.width, x, out_y, out_f32, out_rgba8);
}

// ---- cell grid build -------------------------------------------------------------------------
// Interval bounds of a row's distance (body.cpp:32-57) over a cell's box [lo, hi], widened on the safe side by more than
// the roundings of these bounds and of the kernels' evaluation of the row at any point of the cell (below 1e-6 of the
// magnitudes involved): lower bound L, upper bound U.
__device__ __forceinline__ float med3(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fminf(a, b), __builtin_fminf(__builtin_fmaxf(a, b), c));
}

__device__ __forceinline__ void row_bounds(uint32_t cls, const Float4 a, const Float4 b, const float (&lo)[3], const float (&hi)[3],
                                           float &L, float &U) {
    const float c[3] = { a.x, a.y, a.z };
    const float h[3] = { a.w, cls == kGridClassSphere ? a.w : b.x, cls == kGridClassSphere ? a.w : b.y };
    float dmin[3], dmax[3], magnitude = 0.0f;
    for (int k = 0; k < 3; ++k) {
        dmin[k] = __builtin_fmaxf(__builtin_fmaxf(lo[k] - c[k], c[k] - hi[k]), 0.0f);
        dmax[k] = __builtin_fmaxf(__builtin_fabsf(lo[k] - c[k]), __builtin_fabsf(hi[k] - c[k]));
        magnitude += __builtin_fabsf(c[k]) + __builtin_fabsf(h[k]) + __builtin_fabsf(lo[k]) + __builtin_fabsf(hi[k]);
    }
    if (cls == kGridClassSphere) {
        L = __builtin_sqrtf(dmin[0] * dmin[0] + dmin[1] * dmin[1] + dmin[2] * dmin[2]) - a.w;
        U = __builtin_sqrtf(dmax[0] * dmax[0] + dmax[1] * dmax[1] + dmax[2] * dmax[2]) - a.w;
    } else if (cls == kGridClassBox) {
        L = __builtin_fmaxf(__builtin_fmaxf(dmin[0] - h[0], dmin[1] - h[1]), dmin[2] - h[2]);
        U = __builtin_fmaxf(__builtin_fmaxf(dmax[0] - h[0], dmax[1] - h[1]), dmax[2] - h[2]);
    } else {   // the cross's distance is the median of its three axis terms, monotone in each of them
        L = med3(dmin[0] - h[0], dmin[1] - h[1], dmin[2] - h[2]);
        U = med3(dmax[0] - h[0], dmax[1] - h[1], dmax[2] - h[2]);
    }
    const float slack = 4e-6f * magnitude;
    L = L - slack - 1e-5f * __builtin_fabsf(L);
    U = U + slack + 1e-5f * __builtin_fabsf(U);
}

// the cell's box, widened so that a point the kernels' index arithmetic puts into the cell is inside it
__device__ __forceinline__ void cell_box(const GridBuild &b, uint32_t cell, float (&lo)[3], float (&hi)[3]) {
    const uint32_t ix = cell % b.dims[0], iy = (cell / b.dims[0]) % b.dims[1], iz = cell / (b.dims[0] * b.dims[1]);
    const uint32_t i[3] = { ix, iy, iz };
    for (int k = 0; k < 3; ++k) {
        const float a = b.origin[k] + (float)i[k] * b.cell, z = b.origin[k] + (float)(i[k] + 1u) * b.cell;
        const float widen = 1e-3f * b.cell + 1e-5f * (__builtin_fabsf(a) + __builtin_fabsf(z));
        lo[k] = a - widen;
        hi[k] = z + widen;
    }
}

// The upper 16 bits of the largest float of 8 significant bits that is not above L (a lower bound stays one when it is
// lowered); minus infinity for a bound that is not a number (such a row is always evaluated).
__device__ __forceinline__ uint32_t grid_bound_bits(float L) {
    if (!(L == L)) return 0xff800000u;
    const uint32_t bits = __float_as_uint(L);
    uint32_t high = bits & 0xffff0000u;
    if ((bits & 0x80000000u) && (bits & 0xffffu)) high += 0x10000u;   // negative: the next one down (larger magnitude; at worst minus infinity)
    return high;
}

template <bool FILL>
__global__ __launch_bounds__(256) void grid_kernel(const GridBuild b) {
    const uint32_t n_cells = b.dims[0] * b.dims[1] * b.dims[2];
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    if (cell >= n_cells) return;
    float lo[3], hi[3];
    cell_box(b, cell, lo, hi);
    float H = __builtin_inff();
    if (FILL) {
        H = b.bound[cell];
    } else {
        for (uint32_t r = 0; r < b.n_runs; ++r)
            for (uint32_t i = b.runs[r].first; i < b.runs[r].first + b.runs[r].count; ++i) {
                float L, U;
                row_bounds(b.runs[r].cls, b.pa[i], b.pb[i], lo, hi, L, U);
                H = U < H ? U : H;
            }
        b.bound[cell] = H;
    }
    uint32_t n = 0u;
    const uint32_t first = FILL ? b.cells[2u * cell] : 0u;
    uint32_t at = first;
    for (uint32_t r = 0; r < b.n_runs; ++r) {
        for (uint32_t i = b.runs[r].first; i < b.runs[r].first + b.runs[r].count; ++i) {
            float L, U;
            row_bounds(b.runs[r].cls, b.pa[i], b.pb[i], lo, hi, L, U);
            if (!(L > H)) {   // NaNs are listed
                if (FILL) {
                    // One word per entry -- the bound's upper 16 bits, the class, the table row (below 2^14: the host
                    // builds no grid otherwise) -- by rising bound: the walk stops at the first entry whose bound is
                    // not below its running minimum.  Lists are a few entries long.
                    const uint32_t key = grid_bound_bits(L);
                    uint32_t k = at++;
                    for (; k > first && __uint_as_float(b.entries[k - 1u] & 0xffff0000u) > __uint_as_float(key); --k) b.entries[k] = b.entries[k - 1u];
                    b.entries[k] = key | b.runs[r].cls << 14 | i;
                } else {
                    n++;
                }
            }
        }
    }
    if (!FILL) {
        // the host uses the grid only if the 64-bit total stays below 2^28, so the 32 bits kept here are the whole offset then
        b.cells[2u * cell] = (uint32_t)atomicAdd(b.total, (unsigned long long)n);
        b.cells[2u * cell + 1u] = kGridListEnd;
    } else if (at > first) {
        // the list's first entry moves into the cell's header, the others move up, and the end mark takes the last place
        b.cells[2u * cell + 1u] = b.entries[first];
        for (uint32_t k = first; k + 1u < at; ++k) b.entries[k] = b.entries[k + 1u];
        b.entries[at - 1u] = kGridListEnd;
    }
}

}  // namespace

hipError_t launch_count_grid(const GridBuild &b, hipStream_t stream) {
    (void)hipGetLastError();   // a launch reports its own error, not one an earlier failed call left behind
    const uint32_t n_cells = b.dims[0] * b.dims[1] * b.dims[2];
    hipLaunchKernelGGL(grid_kernel<false>, dim3((n_cells + 255u) / 256u), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t launch_fill_grid(const GridBuild &b, hipStream_t stream) {
    (void)hipGetLastError();
    const uint32_t n_cells = b.dims[0] * b.dims[1] * b.dims[2];
    hipLaunchKernelGGL(grid_kernel<true>, dim3((n_cells + 255u) / 256u), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t preload_kernels() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&resolve_kernel));
}

size_t sample_texels(const FrameView &fr) {
    return (size_t)fr.band_count * fr.band_rows * fr.kernel * (size_t)fr.width * fr.kernel;
}

hipError_t launch_trace(const SceneView &sc, const FrameView &fr, bool wide_runs, Sample *samples, WorkCounters *counters,
                        hipStream_t stream) {
    const uint32_t sw = fr.width * fr.kernel;
    const uint32_t sh = fr.band_count * fr.band_rows * fr.kernel;
    if (sw == 0 || sh == 0) return hipSuccess;
    (void)hipGetLastError();
    const uint32_t lds = lds_bytes(sc);
    if (wide_runs) {   // one sample per lane, one wave per workgroup
        const dim3 grid((sw + 7u) / 8u, (sh + 7u) / 8u, 1u);
        if (counters) hipLaunchKernelGGL((trace_kernel<1, 4, true, 1>), grid, dim3(64), lds, stream, sc, fr, samples, counters);
        else hipLaunchKernelGGL((trace_kernel<1, 4, false, 1>), grid, dim3(64), lds, stream, sc, fr, samples, counters);
    } else {           // two samples per lane, four waves per workgroup
        const dim3 grid((sw + 31u) / 32u, (sh + 15u) / 16u, 1u);
        if (counters) hipLaunchKernelGGL((trace_kernel<2, 8, true, 4>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
        else hipLaunchKernelGGL((trace_kernel<2, 8, false, 4>), grid, dim3(256), lds, stream, sc, fr, samples, counters);
    }
    return hipGetLastError();
}

hipError_t launch_resolve(const FrameView &fr, const Float4 *colors, const Sample *samples, float *out_f32, unsigned char *out_rgba8,
                          hipStream_t stream) {
    const uint32_t rows = fr.band_count * fr.band_rows;
    if (fr.width == 0 || rows == 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 grid((fr.width + 63u) / 64u, (rows + 3u) / 4u, 1u);
    hipLaunchKernelGGL(resolve_kernel, grid, dim3(256), 0, stream, fr, colors, samples, out_f32, out_rgba8);
    return hipGetLastError();
}

hipError_t launch_eval_sdf(const SceneView &sc, const float *xyz, uint32_t n, float *sd, uint32_t *prim, float *sd_march,
                           hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(eval_sdf_kernel, dim3((n + 255u) / 256u), dim3(256), lds_bytes(sc), stream, sc, xyz, n, sd, prim, sd_march);
    return hipGetLastError();
}

}  // namespace rmdev
