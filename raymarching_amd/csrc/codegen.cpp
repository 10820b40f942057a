// codegen.cpp -- scene-specialised kernel source.
//
// The reference compiles its compute shader when the program starts (render::setup::shaders,
// /root/reference/source/render.cpp:428-437); this back-end compiles a kernel per uploaded
// scene instead.  The record stream of program.cpp is unrolled into straight-line HIP source:
// every primitive becomes one call with its centre and half sizes as hexadecimal float literals
// (exact), nested lists become local accumulators, long homogeneous runs stay loops over the
// tables and lattices stay calls into lattice_fold.  Arithmetic and operand order are those of
// the record walker (interp_scene.h), so results are bit-identical to it.
#include "codegen.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace rmdev {

namespace {

float as_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, sizeof f);
    return f;
}

std::string lit(float f) {
    if (std::isnan(f)) return "__builtin_nanf(\"\")";
    if (std::isinf(f)) return f > 0 ? "__builtin_inff()" : "(-__builtin_inff())";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%af", (double)f);
    return buf;
}

const char *code_name(uint32_t code) {
    return code == REC_SPHERE ? "REC_SPHERE" : (code == REC_BOX ? "REC_BOX" : "REC_CROSS");
}

const char *kind_name(uint32_t kind) {
    switch (kind) {
    case K_SPHERE: return "K_SPHERE";
    case K_BOX: return "K_BOX";
    case K_CROSS: return "K_CROSS";
    case K_BOX_ISO: return "K_BOX_ISO";
    default: return "K_CROSS_ISO";
    }
}

std::string tf(bool b) { return b ? "true" : "false"; }

// body of sd() (track = false) or surface() (track = true)
std::string emit_body(const HostProgram &prog, bool track) {
    std::string s;
    char buf[512];
    uint32_t depth = 0;
    unsigned long long counted = 0;   // primitive evaluations per call outside lattices
    auto acc = [&](uint32_t d) { return "a" + std::to_string(d); };
    auto accp = [&](uint32_t d) { return "q" + std::to_string(d); };
    auto declare = [&](uint32_t d, float identity) {
        s += "    float " + acc(d) + "[NP];\n";
        if (track) s += "    uint32_t " + accp(d) + "[NP];\n";
        s += "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { " + acc(d) + "[n] = " + lit(identity) + ";" +
             (track ? " " + accp(d) + "[n] = kNoPrim;" : "") + " }\n";
    };
    declare(0, prog.root_identity);
    if (!track) s += "    uint32_t no_prim[NP];\n    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) no_prim[n] = kNoPrim;\n";
    for (const Rec &r : prog.recs) {
        const uint32_t code = r.flags & 15u;
        const bool neg = (r.flags & FOLD_NEG) != 0, mx = (r.flags & FOLD_MAX) != 0;
        if (code == REC_END) break;
        if (code <= REC_CROSS) {
            std::string args = lit(as_float(r.a[0])) + ", " + lit(as_float(r.a[1])) + ", " + lit(as_float(r.a[2])) + ", " +
                               lit(as_float(r.h[0])) + ", " + lit(as_float(r.h[1])) + ", " + lit(as_float(r.h[2]));
            if (track) {
                std::snprintf(buf, sizeof buf, "    single_surface<%s, %uu, NP>(p, %s, %s, %s, %uu);\n", code_name(code),
                              r.flags & (FOLD_NEG | FOLD_MAX), acc(depth).c_str(), accp(depth).c_str(), args.c_str(), r.prim);
            } else {
                std::snprintf(buf, sizeof buf, "    single_sd<%s, %s, %s, NP>(p, %s, %s);\n", code_name(code), tf(neg).c_str(),
                              tf(mx).c_str(), acc(depth).c_str(), args.c_str());
            }
            s += buf;
            counted++;
        } else if (code == REC_RUN) {
            std::snprintf(buf, sizeof buf,
                          "    fold_run<%s, %s, %s, NP, %s>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, %uu, %uu, p, %s, %s);\n",
                          kind_name(r.a[2]), tf(neg).c_str(), tf(mx).c_str(), tf(track).c_str(), r.a[0], r.a[1], acc(depth).c_str(),
                          track ? accp(depth).c_str() : "no_prim");
            s += buf;
            counted += r.a[1];
        } else if (code == REC_LATTICE) {
            std::snprintf(buf, sizeof buf,
                          "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) lattice_fold<%s, %s>(lds.lattice + %uu, %uu, %uu, p[n], %s[n], %s[n], n_evals[n]);\n",
                          tf(track).c_str(), track ? "false" : "COUNT", r.a[1], r.a[0], r.prim, acc(depth).c_str(),
                          track ? accp(depth).c_str() : "no_prim");
            s += buf;
        } else if (code == REC_PUSH) {
            s += "    {\n";
            depth++;
            declare(depth, as_float(r.a[1]));
        } else {   // REC_POP
            if (track) {
                std::snprintf(buf, sizeof buf, "    fold_value<NP, true>(%uu, %s, %s, %s, %s);\n", r.flags & (FOLD_NEG | FOLD_MAX),
                              acc(depth).c_str(), accp(depth).c_str(), acc(depth - 1).c_str(), accp(depth - 1).c_str());
            } else {
                std::snprintf(buf, sizeof buf, "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) %s[n] = fold_sd<%s, %s>(%s[n], %s[n]);\n",
                              acc(depth - 1).c_str(), tf(neg).c_str(), tf(mx).c_str(), acc(depth - 1).c_str(), acc(depth).c_str());
            }
            s += buf;
            s += "    }\n";
            depth--;
        }
    }
    if (!track) {
        std::snprintf(buf, sizeof buf, "    if (COUNT) { _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) n_evals[n] += %lluu; }\n", counted);
        s += buf;
    }
    s += "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { out_sd[n] = a0[n];" + std::string(track ? " out_prim[n] = q0[n];" : "") + " }\n";
    return s;
}

}  // namespace

std::string generate_scene_source(const HostProgram &prog, int points_per_lane, int min_waves) {
    std::string s;
    s += "#include \"trace_kernel.h\"\n\nnamespace rmdev {\n\nstruct CompiledScene {\n";
    s += "template <int NP, bool COUNT>\nstatic __device__ __forceinline__ void sd(const SceneView &sc, const Lds &lds, const V3 (&p)[NP],\n"
         "                                          float (&out_sd)[NP], uint32_t (&n_evals)[NP]) {\n";
    s += emit_body(prog, false);
    s += "}\n\n";
    s += "template <int NP>\nstatic __device__ __forceinline__ void surface(const SceneView &sc, const Lds &lds, const V3 (&p)[NP],\n"
         "                                               float (&out_sd)[NP], uint32_t (&out_prim)[NP]) {\n";
    s += "    uint32_t n_evals[NP] = {};\n";
    s += emit_body(prog, true);
    s += "}\n};\n\n}  // namespace rmdev\n\nusing namespace rmdev;\n\n";
    char buf[512];
    std::snprintf(buf, sizeof buf,
                  "extern \"C\" __global__ __launch_bounds__(256, %d) void rm_trace(const SceneView sc, const FrameView fr, Float4 *samples,\n"
                  "        WorkCounters *counters) {\n    trace_body<CompiledScene, %d, false>(sc, fr, samples, counters);\n}\n\n",
                  min_waves, points_per_lane);
    s += buf;
    s += "extern \"C\" __global__ __launch_bounds__(256) void rm_eval_sdf(const SceneView sc, const float *xyz, uint32_t n, float *sd,\n"
         "        uint32_t *prim) {\n    eval_sdf_body<CompiledScene>(sc, xyz, n, sd, prim);\n}\n";
    return s;
}

}  // namespace rmdev
