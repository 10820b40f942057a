// valu_bench.hip -- measures what the gfx950 vector ALU sustains for the instruction kinds the
// sphere-tracing kernels are made of (fp32 add/sub with abs and SGPR operands, min3/max3, packed
// add, plain max), at 1/2/4/8 waves per SIMD.  The results define the roofline the render kernels
// are priced against: lane-operations per second for non-fused fp32 VALU work.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int kUnroll = 8;      // instruction groups per loop trip
constexpr int kChains = 8;      // independent registers per group
constexpr int kTrips = 2048;

// Each body issues kChains independent instructions on r0..r7; s is an SGPR operand.
#define BODY_ADD(OP)                                                                        \
    asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" \
                 OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n" \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                 : "s"(s))

template <int VARIANT>
__global__ __launch_bounds__(256) void valu_kernel(float *out, float seed, float sval, unsigned long long *cycles) {
    float r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = r0 * 0.5f, q1 = r1 * 0.5f, q2 = r2 * 0.5f, q3 = r3 * 0.5f, q4 = r4 * 0.5f, q5 = r5 * 0.5f, q6 = r6 * 0.5f, q7 = r7 * 0.5f;
    const float s = __builtin_amdgcn_readfirstlane(sval);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < kTrips; ++t) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (VARIANT == 0) {  // v_add_f32 VOP2 with an SGPR source
                BODY_ADD("v_add_f32");
            } else if (VARIANT == 1) {  // v_sub_f32 VOP3 with |abs| and an SGPR source
                asm volatile("v_sub_f32 %0, |%0|, %8\nv_sub_f32 %1, |%1|, %8\nv_sub_f32 %2, |%2|, %8\nv_sub_f32 %3, |%3|, %8\n"
                             "v_sub_f32 %4, |%4|, %8\nv_sub_f32 %5, |%5|, %8\nv_sub_f32 %6, |%6|, %8\nv_sub_f32 %7, |%7|, %8\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "s"(s));
            } else if (VARIANT == 2) {  // v_min3_f32
                asm volatile("v_min3_f32 %0, %0, %1, %2\nv_min3_f32 %1, %1, %2, %3\nv_min3_f32 %2, %2, %3, %4\nv_min3_f32 %3, %3, %4, %5\n"
                             "v_min3_f32 %4, %4, %5, %6\nv_min3_f32 %5, %5, %6, %7\nv_min3_f32 %6, %6, %7, %0\nv_min3_f32 %7, %7, %0, %1\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
            } else if (VARIANT == 3) {  // v_pk_add_f32 (two fp32 adds per lane)
                asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %5\nv_pk_add_f32 %2, %2, %6\nv_pk_add_f32 %3, %3, %7\n"
                             "v_pk_add_f32 %0, %0, %5\nv_pk_add_f32 %1, %1, %6\nv_pk_add_f32 %2, %2, %7\nv_pk_add_f32 %3, %3, %4\n"
                             : "+v"(*(double *)&r0), "+v"(*(double *)&r2), "+v"(*(double *)&r4), "+v"(*(double *)&r6)
                             : "v"(*(double *)&q0), "v"(*(double *)&q2), "v"(*(double *)&q4), "v"(*(double *)&q6));
            } else if (VARIANT == 4) {  // v_max_f32 VOP3 with a negated source
                asm volatile("v_max_f32 %0, %0, -%1\nv_max_f32 %1, %1, -%2\nv_max_f32 %2, %2, -%3\nv_max_f32 %3, %3, -%4\n"
                             "v_max_f32 %4, %4, -%5\nv_max_f32 %5, %5, -%6\nv_max_f32 %6, %6, -%7\nv_max_f32 %7, %7, -%0\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
            } else if (VARIANT == 5) {  // v_fma_f32 (datasheet-peak instruction, for reference)
                asm volatile("v_fma_f32 %0, %0, %8, %1\nv_fma_f32 %1, %1, %8, %2\nv_fma_f32 %2, %2, %8, %3\nv_fma_f32 %3, %3, %8, %4\n"
                             "v_fma_f32 %4, %4, %8, %5\nv_fma_f32 %5, %5, %8, %6\nv_fma_f32 %6, %6, %8, %7\nv_fma_f32 %7, %7, %8, %0\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                             : "s"(s));
            } else if (VARIANT == 6) {  // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %4, %1\nv_pk_fma_f32 %1, %1, %5, %2\nv_pk_fma_f32 %2, %2, %6, %3\nv_pk_fma_f32 %3, %3, %7, %0\n"
                             "v_pk_fma_f32 %0, %0, %5, %1\nv_pk_fma_f32 %1, %1, %6, %2\nv_pk_fma_f32 %2, %2, %7, %3\nv_pk_fma_f32 %3, %3, %4, %0\n"
                             : "+v"(*(double *)&r0), "+v"(*(double *)&r2), "+v"(*(double *)&r4), "+v"(*(double *)&r6)
                             : "v"(*(double *)&q0), "v"(*(double *)&q2), "v"(*(double *)&q4), "v"(*(double *)&q6));
            } else if (VARIANT == 7) {  // the 13-instruction cross body (6 sub, min3, max3, 2 add, 2 sub, max)
                asm volatile(
                    "v_sub_f32 %3, %0, %8\nv_sub_f32 %4, %1, %8\nv_sub_f32 %5, %2, %8\n"
                    "v_sub_f32 %3, |%3|, %8\nv_sub_f32 %4, |%4|, %8\nv_sub_f32 %5, |%5|, %8\n"
                    "v_min3_f32 %6, %3, %4, %5\nv_max3_f32 %7, %3, %4, %5\n"
                    "v_add_f32 %3, %3, %4\nv_add_f32 %3, %3, %5\nv_sub_f32 %3, %3, %6\nv_sub_f32 %3, %3, %7\n"
                    "v_max_f32 %0, %0, -%3\n"
                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                    : "s"(s));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

struct Variant {
    const char *name;
    int instr_per_group;   // wave-instructions per asm group
    int flops_per_instr;   // per lane
    void (*fn)(float *, float, float, unsigned long long *);
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000.0);
    const Variant variants[] = {
        { "v_add_f32 (vop2, sgpr)", 8, 1, valu_kernel<0> }, { "v_sub_f32 |abs| (vop3, sgpr)", 8, 1, valu_kernel<1> },
        { "v_min3_f32", 8, 2, valu_kernel<2> },             { "v_pk_add_f32", 8, 2, valu_kernel<3> },
        { "v_max_f32 neg (vop3)", 8, 1, valu_kernel<4> },   { "v_fma_f32", 8, 2, valu_kernel<5> },
        { "v_pk_fma_f32", 8, 4, valu_kernel<6> },           { "cross body (13 instr)", 13, 1, valu_kernel<7> },
    };
    float *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-30s %6s %12s %14s %12s %10s\n", "instruction", "w/SIMD", "ms", "Tlane-op/s", "TFLOP/s", "cyc/instr");
    for (const Variant &v : variants) {
        for (int waves = 1; waves <= 8; waves *= 2) {
            const int blocks = cus * waves;  // 256 threads = 4 waves = one wave per SIMD per block
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.25f, cyc);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int reps = 5;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.25f, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            unsigned long long c;
            CHECK(hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost));
            const double instr_per_wave = (double)kTrips * kUnroll * v.instr_per_group;
            const double lane_ops = instr_per_wave * 64.0 * blocks * 4.0;
            // s_memtime ticks per wave-instruction of ONE wave (includes time lost to the other waves on its SIMD)
            printf("%-30s %6d %12.4f %14.2f %12.2f %10.2f\n", v.name, waves, ms, lane_ops / ms / 1e9,
                   lane_ops * v.flops_per_instr / ms / 1e9, (double)c / instr_per_wave);
        }
    }
    return 0;
}
