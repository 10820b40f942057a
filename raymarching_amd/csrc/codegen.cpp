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

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// ---- the record stream as a tree (children of a list may be emitted in any order: min and max
// are exact selections, and the colour-tracking walk breaks ties by depth-first index) ------------

struct Item {
    uint32_t rec = 0;              // leaf: its record; list: its PUSH record
    uint32_t pop = 0;              // list: its POP record
    bool is_list = false;
    std::vector<Item> children;    // list only
};

// parses recs[i..] up to the END or the POP that closes the current list; returns the index after it
uint32_t parse_children(const std::vector<Rec> &recs, uint32_t i, std::vector<Item> &out) {
    while (i < recs.size()) {
        const uint32_t code = recs[i].flags & 15u;
        if (code == REC_END || code == REC_POP) return i;
        Item it;
        it.rec = i;
        if (code == REC_PUSH) {
            it.is_list = true;
            it.pop = parse_children(recs, i + 1, it.children);
            i = it.pop + 1;
        } else {
            i++;
        }
        out.push_back(std::move(it));
    }
    return i;
}

// The root union's sphere-bounded children (spheres, lists bounded by a sphere) are evaluated after its other children
// and behind one common box test (Emitter::children).  After, because the other lists are where rays take their many
// small steps (the tunnels of a sponge): there the accumulator is tiny by the time the spheres' turn comes, and one test
// rules them all out; in front, the accumulator is still the distance to the floor and the walls and rules out little.
// Shipped frame, ms per frame in flight, same GPU: 0.492 as before, 0.489 with the common test at the spheres' old place,
// 0.466 with the spheres after the lists, 0.451 with both; 4K 8.83 -> 8.09.  (An outer test around the lists and the
// members together, for points far from all of them, measured slower: 0.460 against 0.449, 4K 8.25 against 8.08.)
// $RM_GROUP_TESTS selects the variant for such A/B runs (the generated source, and with it the cache key, differs):
// 0 neither; 1 the common test at the first member's place; 2 the same with box-bounded lists as members too; 3 (default)
// members last, behind the common test; 4 members last without it.
int group_tests_mode() {
    const char *env = std::getenv("RM_GROUP_TESTS");
    return env ? std::atoi(env) : 3;
}
bool group_tests_enabled() { return group_tests_mode() != 0; }


struct Emitter {
    const HostProgram &prog;
    bool track;
    std::string s;
    unsigned evals = 0, flops = 0;    // work of the leaves emitted into the current block, not yet booked
    char buf[640];

    // books the block's pending work (the counting variant only; `COUNT` is a template parameter)
    void book() {
        if (!track && evals) {
            std::snprintf(buf, sizeof buf, "    if (COUNT) { _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { work[n].evals += %uu; work[n].flops += %uu; } }\n",
                          evals, flops);
            s += buf;
        }
        evals = flops = 0;
    }

    // A root child whose value decides, where it is the scene's, that the surface has no colour: a box or cross of colour
    // (0, 0, 0) folded into a union's minimum, with no coloured primitive before it in depth-first order (of equal
    // distances the earlier primitive's colour wins, so a tie between such a child and anything coloured is the child's).
    bool wall_candidate(const Rec &r) const {
        const uint32_t code = r.flags & 15u;
        if ((code != REC_BOX && code != REC_CROSS) || (r.flags & FOLD_MAX) != 0u || !(prog.root_identity > 0)) return false;
        for (uint32_t i = 0; i <= r.prim && i < prog.colors.size(); i++) {
            const Float4 &c = prog.colors[i];
            if (c.x != 0.0f || c.y != 0.0f || c.z != 0.0f) return false;
        }
        return true;
    }
    bool any_wall = false;

    std::string acc(uint32_t d) const { return "a" + std::to_string(d); }
    std::string accp(uint32_t d) const { return track ? "q" + std::to_string(d) : "no_prim"; }

    void declare(uint32_t d, float identity) {
        s += "    float " + acc(d) + "[NP];\n";
        if (track) s += "    uint32_t " + accp(d) + "[NP];\n";
        s += "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { " + acc(d) + "[n] = " + lit(identity) + ";" +
             (track ? " " + accp(d) + "[n] = kNoPrim;" : "") + " }\n";
    }

    std::string centre_and_half(const Rec &r) const {
        return lit(as_float(r.a[0])) + ", " + lit(as_float(r.a[1])) + ", " + lit(as_float(r.a[2])) + ", " + lit(as_float(r.h[0])) +
               ", " + lit(as_float(r.h[1])) + ", " + lit(as_float(r.h[2]));
    }

    void leaf(const Rec &r, uint32_t depth) {
        const uint32_t code = r.flags & 15u;
        const bool neg = (r.flags & FOLD_NEG) != 0, mx = (r.flags & FOLD_MAX) != 0;
        if (code <= REC_CROSS && !track && depth == 0 && wall_candidate(r)) {
            // min(acc, +-v) as single_sd folds it, with the value kept: see `wall` in emit_body
            std::snprintf(buf, sizeof buf,
                          "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { const float v = %ssingle_distance<%s>(p[n], %s); "
                          "a0[n] = __builtin_fminf(a0[n], v); wall[n] = __builtin_fminf(wall[n], v); }\n",
                          neg ? "-" : "", code_name(code), centre_and_half(r).c_str());
            evals += 1;
            flops += flops_of_code(code);
            any_wall = true;
        } else if (code <= REC_CROSS) {
            if (track) {
                std::snprintf(buf, sizeof buf, "    single_surface<%s, %uu, NP>(p, %s, %s, %s, %uu);\n", code_name(code),
                              r.flags & (FOLD_NEG | FOLD_MAX), acc(depth).c_str(), accp(depth).c_str(), centre_and_half(r).c_str(), r.prim);
            } else {
                std::snprintf(buf, sizeof buf, "    single_sd<%s, %s, %s, NP>(p, %s, %s);\n", code_name(code), tf(neg).c_str(),
                              tf(mx).c_str(), acc(depth).c_str(), centre_and_half(r).c_str());
            }
            evals += 1;
            flops += flops_of_code(code);
        } else if (code == REC_RUN && (r.flags & RUN_CLUSTERED) && !track) {
            std::snprintf(buf, sizeof buf,
                          "    fold_run_clustered<%s, %s, %s, NP, COUNT>((ConstRows)sc.pa, (ConstRows)sc.pb, %uu, %uu, %s, p, %s, %s, work);\n",
                          kind_name(r.a[2]), tf(neg).c_str(), tf(mx).c_str(), r.a[0], r.a[1], lit(as_float(r.h[2])).c_str(),
                          acc(depth).c_str(), (depth == 0 && !neg && !mx && prog.root_identity > 0) ? "cap" : "no_cap");
        } else if (code == REC_RUN) {
            std::snprintf(buf, sizeof buf,
                          "    fold_run<%s, %s, %s, NP, %s>((ConstRows)sc.pa, (ConstRows)sc.pb, (ConstIndex)sc.row_prim, %uu, %uu, p, %s, %s);\n",
                          kind_name(r.a[2]), tf(neg).c_str(), tf(mx).c_str(), tf(track).c_str(), r.a[0], r.a[1], acc(depth).c_str(),
                          accp(depth).c_str());
            evals += r.a[1];
            flops += r.a[1] * flops_of_kind(r.a[2]);
        } else if (track) {   // REC_LATTICE
            std::snprintf(buf, sizeof buf,
                          "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) lattice_fold<true, false>(lds.lattice + %uu, (ConstIndex)sc.lattice + %uu, %uu, %uu, p[n], %s[n], %s[n], evals_unused[n]);\n",
                          r.a[1], r.a[1], r.a[0], r.prim, acc(depth).c_str(), accp(depth).c_str());
        } else {
            // the lattice's header fields go into the code as literals, its levels are unrolled
            const uint32_t off = r.a[1], levels = r.a[0];
            const uint32_t *w = prog.lattice.data() + off;
            std::snprintf(buf, sizeof buf,
                          "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) {\n"
                          "        const uint32_t *w = lds.lattice + %uu;\n        const ConstIndex g = (ConstIndex)sc.lattice + %uu;\n"
                          "        const float eps = 2e-6f * (__builtin_fabsf(p[n].x) + __builtin_fabsf(p[n].y) + __builtin_fabsf(p[n].z) + %s);\n"
                          "        float run = %s[n];\n        uint32_t evals = 0;\n        int state = kLatticeNext;\n        do {\n",
                          off, off, lit(as_float(w[2])).c_str(), acc(depth).c_str());
            s += buf;
            uint32_t n = 1;
            for (uint32_t l = 0; l < levels; l++, n *= 3) {
                std::snprintf(buf, sizeof buf,
                              "            state = lattice_level<%u, COUNT>(w + %uu, %s, %s, %s, %s, %s, p[n], eps, run, evals);\n"
                              "            if (state != kLatticeNext) break;\n",
                              n, w[kLatOffset + l], lit(as_float(w[kLatHalf + l])).c_str(), lit(as_float(w[kLatInvSpacing + l])).c_str(),
                              lit(as_float(w[kLatOrigin + 3 * l])).c_str(), lit(as_float(w[kLatOrigin + 3 * l + 1])).c_str(),
                              lit(as_float(w[kLatOrigin + 3 * l + 2])).c_str());
                s += buf;
            }
            std::snprintf(buf, sizeof buf,
                          "        } while (false);\n"
                          "        if (state == kLatticeGeneral) {\n"
                          "            if (COUNT || EXACT) lattice_fold_scans<COUNT>(w, g, %uu, p[n], run, evals);\n"
                          "            // an optimistic walk goes on with the lower bound it has and asks for the tile to be rendered again\n"
                          "            else *exact_wanted_word() = 1u;\n        }\n"
                          "        %s[n] = run;\n"
                          "        if (COUNT) { work[n].evals += evals; work[n].flops += kFlopsCross * evals; }\n    }\n",
                          levels, acc(depth).c_str());
        }
        s += buf;
    }

    // cost rank of a child in the distance-only walk: cheap bounds first, lists last
    static int rank(const std::vector<Rec> &recs, const Item &it) {
        if (it.is_list) {
            if (it.children.size() == 1 && !it.children[0].is_list && (recs[it.children[0].rec].flags & 15u) <= REC_CROSS) return rank(recs, it.children[0]);
            return 5;
        }
        switch (recs[it.rec].flags & 15u) {
        case REC_BOX: return 0;
        case REC_CROSS: return 1;
        case REC_SPHERE: return 2;
        case REC_RUN: return 3;
        default: return 4;
        }
    }

    // The child of a nested list whose value alone bounds the list's as its parent sees it (see list()); null if none.
    const Item *bound_child(const Item &it) const {
        const Rec &push = prog.recs[it.rec], &pop = prog.recs[it.pop];
        const bool list_max = as_float(push.a[1]) < 0;   // -inf: folded with max
        const bool pop_neg = (pop.flags & FOLD_NEG) != 0, pop_max = (pop.flags & FOLD_MAX) != 0;
        const Item *bound = nullptr;
        if (!track && list_max == (pop_neg == pop_max)) {
            for (const Item &c : it.children) {
                if (c.is_list) continue;
                const uint32_t code = prog.recs[c.rec].flags & 15u;
                if (code > REC_CROSS) continue;
                if (!bound || rank(prog.recs, c) < rank(prog.recs, *bound)) bound = &c;
            }
        }
        return bound;
    }

    // For a child of the root union whose value is at least that of a sphere or box (a sphere folded with min as it is; a
    // nested list folded with min whose bounding child is a sphere or box, not negated): that volume's box, lo..hi per
    // axis, and its largest half size.  Such children can sit behind one common test (group_wanted in device_core.h).
    bool group_volume(const Item &it, double (&lo)[3], double (&hi)[3], double &size) const {
        const Rec *b = nullptr;
        if (!it.is_list) {
            const Rec &r = prog.recs[it.rec];
            if ((r.flags & 15u) != REC_SPHERE || (r.flags & (FOLD_NEG | FOLD_MAX)) != 0u) return false;
            b = &r;
        } else {
            const uint32_t pop = prog.recs[it.pop].flags;
            if ((pop & (FOLD_NEG | FOLD_MAX)) != 0u) return false;
            if (it.children.size() == 1 && !it.children[0].is_list) {
                const Rec &r = prog.recs[it.children[0].rec];
                if ((r.flags & 15u) != REC_SPHERE || (r.flags & FOLD_NEG) != 0u) return false;
                b = &r;
            } else {
                const Item *bound = bound_child(it);
                if (!bound) return false;
                const Rec &r = prog.recs[bound->rec];
                const uint32_t code = r.flags & 15u;
                if ((code != REC_SPHERE && !(code == REC_BOX && group_tests_mode() == 2)) || (r.flags & FOLD_NEG) != 0u) return false;
                b = &r;
            }
        }
        const bool sphere = (b->flags & 15u) == REC_SPHERE;
        size = 0.0;
        for (int k = 0; k < 3; k++) {
            const double c = as_float(b->a[k]), h = as_float(b->h[sphere ? 0 : k]);
            if (!std::isfinite(c) || !std::isfinite(h) || h < 0.0) return false;
            lo[k] = c - h;
            hi[k] = c + h;
            size = std::max(size, h);
        }
        return true;
    }

    // `free_of_charge`: the child whose evaluation the caller has booked already
    void children(const std::vector<Item> &items, uint32_t depth, const Item *free_of_charge = nullptr) {
        std::vector<const Item *> order;
        for (const Item &it : items) order.push_back(&it);
        if (!track) std::stable_sort(order.begin(), order.end(), [&](const Item *x, const Item *y) { return rank(prog.recs, *x) < rank(prog.recs, *y); });
        // The root's clustered runs that the cell grid covers (program.cpp: grid_runs) are evaluated through the grid; the
        // hull walk is what lanes outside the grid's box fall back to.
        std::vector<const Item *> gridded;
        if (depth == 0 && !prog.grid_runs.empty()) {
            std::vector<const Item *> others;
            for (const Item *it : order) {
                const Rec &r = prog.recs[it->rec];
                const bool covered = !it->is_list && (r.flags & 15u) == REC_RUN && (r.flags & RUN_CLUSTERED) != 0u &&
                                     (r.flags & (FOLD_NEG | FOLD_MAX)) == 0u;
                (covered ? gridded : others).push_back(it);
            }
            order = others;
        }
        // Sphere-bounded children of the root union behind one common box test (two or more of them)
        std::vector<const Item *> grouped;
        double glo[3] = { 0, 0, 0 }, ghi[3] = { 0, 0, 0 }, gsize = 0.0;
        if (!track && depth == 0 && prog.root_identity > 0 && group_tests_enabled()) {
            for (const Item *it : order) {
                double lo[3], hi[3], size;
                if (it == free_of_charge || !group_volume(*it, lo, hi, size)) continue;
                for (int k = 0; k < 3; k++) {
                    glo[k] = grouped.empty() ? lo[k] : std::min(glo[k], lo[k]);
                    ghi[k] = grouped.empty() ? hi[k] : std::max(ghi[k], hi[k]);
                }
                gsize = std::max(gsize, size);
                grouped.push_back(it);
            }
            if (grouped.size() < 2) grouped.clear();
        }
        const bool no_common_test = group_tests_mode() == 4;
        bool group_open = false;
        const auto in_group = [&](const Item *it) { return std::find(grouped.begin(), grouped.end(), it) != grouped.end(); };
        // members first-to-last in one block at the first member's place; the others keep their order around it
        if (!grouped.empty()) {
            std::vector<const Item *> arranged;
            bool placed = false;
            for (const Item *it : order) {
                if (!in_group(it)) { arranged.push_back(it); continue; }
                if (!placed && group_tests_mode() < 3) { arranged.insert(arranged.end(), grouped.begin(), grouped.end()); placed = true; }
            }
            if (!placed) arranged.insert(arranged.end(), grouped.begin(), grouped.end());
            order = arranged;
        }
        for (const Item *it : order) {
            if (!grouped.empty() && it == grouped.front() && !no_common_test) {
                float c[3], h[3];
                double hmax = 0.0;
                for (int k = 0; k < 3; k++) hmax = std::max(hmax, 0.5 * (ghi[k] - glo[k]));
                for (int k = 0; k < 3; k++) {
                    c[k] = (float)(0.5 * (glo[k] + ghi[k]));
                    const double half = std::max(ghi[k] - (double)c[k], (double)c[k] - glo[k]) + 3e-6 * (hmax + gsize + std::fabs((double)c[k]));
                    h[k] = std::nextafter((float)half, INFINITY);
                }
                book();
                std::snprintf(buf, sizeof buf, "    {\n    const bool group = group_wanted<NP>(p, %s, %s, %s, %s, %s, %s, %s);\n", acc(depth).c_str(),
                              lit(c[0]).c_str(), lit(c[1]).c_str(), lit(c[2]).c_str(), lit(h[0]).c_str(), lit(h[1]).c_str(), lit(h[2]).c_str());
                s += buf;
                evals += 1;
                flops += flops_of_code(REC_BOX);
                book();
                s += "    if (group) {\n    keep_in_branch();\n";
                group_open = true;
            }
            if (!it->is_list) {
                const unsigned e = evals, f = flops;
                leaf(prog.recs[it->rec], depth);
                if (it == free_of_charge) {
                    evals = e;
                    flops = f;
                }
            } else if (it->children.size() == 1 && !it->children[0].is_list && (prog.recs[it->children[0].rec].flags & 15u) <= REC_CROSS) {
                // a list of one primitive is that primitive, negated if the list negates its first child,
                // folded the way the list is folded into its parent
                Rec r = prog.recs[it->children[0].rec];
                const uint32_t pop = prog.recs[it->pop].flags;
                r.flags = (r.flags & 15u) | (((r.flags ^ pop) & FOLD_NEG)) | (pop & FOLD_MAX);
                leaf(r, depth);
            } else {
                list(*it, depth);
            }
            if (group_open && it == grouped.back()) {
                book();
                s += "    }\n    }\n";
                group_open = false;
            }
        }
        book();
        if (!gridded.empty()) {
            s += "    if (sc.grid.cells) {\n    bool outside[NP];\n    bool any_outside = false;\n";
            s += track ? "    grid_surface<NP>(sc.grid, sc.pa, sc.pb, sc.row_prim, p, a0, q0, outside);\n"
                       : "    grid_fold<NP, COUNT>(sc.grid, sc.pa, sc.pb, p, a0, outside, work);\n";
            s += 
                 "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) any_outside = any_outside || outside[n];\n"
                 "    if (__builtin_amdgcn_ballot_w64(any_outside) != 0) {\n    keep_in_branch();\n    if (any_outside) {\n";
            for (const Item *it : gridded) leaf(prog.recs[it->rec], depth);
            s += "    }\n    }\n    } else {\n";
            for (const Item *it : gridded) leaf(prog.recs[it->rec], depth);
            s += "    }\n";
        }
    }

    // A nested list.  In the distance-only walk the list is skipped when one of its children
    // proves that its value cannot change the parent's accumulator: a list folded with max is at
    // least, one folded with min at most, any single child's contribution c, so if folding c
    // alone into the parent would leave the parent unchanged, so does folding the list (when the
    // directions agree: list-max with a parent that keeps small values, and so on).  The test
    // is made per wave; if any lane needs the list, all lanes evaluate it, exactly.
    void list(const Item &it, uint32_t depth) {
        const Rec &push = prog.recs[it.rec], &pop = prog.recs[it.pop];
        const float identity = as_float(push.a[1]);
        const bool list_max = identity < 0;   // -inf: folded with max
        const bool pop_neg = (pop.flags & FOLD_NEG) != 0, pop_max = (pop.flags & FOLD_MAX) != 0;
        const Item *bound = bound_child(it);
        (void)list_max;
        const uint32_t d = depth + 1;
        s += "    {\n";
        if (bound) {
            const Rec &b = prog.recs[bound->rec];
            const uint32_t code = b.flags & 15u;
            const bool neg = ((b.flags & FOLD_NEG) != 0) != pop_neg;   // sign of the child's value as the parent sees it
            if (code == REC_SPHERE) {
                std::snprintf(buf, sizeof buf, "    const bool wanted = sphere_bound_wanted<%s, %s, NP>(p, %s, %s, %s, %s, %s);\n", tf(neg).c_str(),
                              tf(pop_max).c_str(), acc(depth).c_str(), lit(as_float(b.a[0])).c_str(), lit(as_float(b.a[1])).c_str(),
                              lit(as_float(b.a[2])).c_str(), lit(as_float(b.h[0])).c_str());
            } else {
                std::snprintf(buf, sizeof buf,
                              "    bool need = false;\n    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) need = need || value_can_change<%s, %s>(%s[n], "
                              "single_distance<%s>(p[n], %s));\n    const bool wanted = __builtin_amdgcn_ballot_w64(need) != 0;\n",
                              tf(neg).c_str(), tf(pop_max).c_str(), acc(depth).c_str(), code_name(code), centre_and_half(b).c_str());
            }
            s += buf;
            // the test costs one evaluation of the bounding child; inside the list the compiler reuses it
            evals += 1;
            flops += flops_of_code(code);
            book();
            s += "    if (wanted) {\n    keep_in_branch();\n";
        } else {
            book();
        }
        declare(d, identity);
        children(it.children, d, bound);
        if (track) {
            std::snprintf(buf, sizeof buf, "    fold_value<NP, true>(%uu, %s, %s, %s, %s);\n", pop.flags & (FOLD_NEG | FOLD_MAX), acc(d).c_str(),
                          accp(d).c_str(), acc(depth).c_str(), accp(depth).c_str());
        } else {
            std::snprintf(buf, sizeof buf, "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) %s[n] = fold_sd<%s, %s>(%s[n], %s[n]);\n",
                          acc(depth).c_str(), tf(pop_neg).c_str(), tf(pop_max).c_str(), acc(depth).c_str(), acc(d).c_str());
        }
        s += buf;
        if (bound) s += "    }\n";
        s += "    }\n";
    }
};

// body of sd() (track = false) or surface() (track = true)
std::string emit_body(const HostProgram &prog, bool track) {
    Emitter e{ prog, track, std::string(), 0, 0, { 0 }, false };
    std::vector<Item> top;
    parse_children(prog.recs, 0, top);
    e.declare(0, prog.root_identity);
    // wall: the smallest value among the root's colourless leading boxes (see wall_candidate); where it is the scene's
    // value the surface there is colourless, known without the colour-tracking walk
    if (!track) e.s += "    uint32_t no_prim[NP];\n    float no_cap[NP], wall[NP];\n    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { no_prim[n] = kNoPrim; no_cap[n] = __builtin_inff(); wall[n] = __builtin_inff(); }\n    (void)cap;\n";
    else e.s += "    uint32_t evals_unused[NP] = {};\n";
    e.children(top, 0);
    e.s += "    _Pragma(\"unroll\") for (int n = 0; n < NP; ++n) { out_sd[n] = a0[n];" + std::string(track ? " out_prim[n] = q0[n];" : (e.any_wall ? " colourless[n] = wall[n] == a0[n];" : " colourless[n] = false;")) + " }\n";
    return e.s;
}

}  // namespace

std::string generate_scene_source(const HostProgram &prog, int points_per_lane, int min_waves, bool counting, bool refill) {
    std::string s;
    s += "#include \"trace_refill.h\"\n\nnamespace rmdev {\n\n";
    // EXACT = false: the optimistic walk of the frame's kernel (see FrameView::tile_flags); true: every walk decided on the spot
    s += "template <bool EXACT>\nstruct CompiledSceneT {\nstatic constexpr bool kOptimistic = !EXACT;\n";
    s += "template <int NP, bool COUNT>\nstatic __device__ __forceinline__ void sd(const SceneView &sc, const Lds &lds, const V3 (&p)[NP],\n"
         "                                          float (&out_sd)[NP], Work (&work)[NP], const float (&cap)[NP], bool (&colourless)[NP]) {\n";
    s += emit_body(prog, false);
    s += "}\n\n";
    s += "template <int NP>\nstatic __device__ __forceinline__ void surface(const SceneView &sc, const Lds &lds, const V3 (&p)[NP],\n"
         "                                               float (&out_sd)[NP], uint32_t (&out_prim)[NP]) {\n";
    s += emit_body(prog, true);
    s += "}\n};\n\n}  // namespace rmdev\n\nusing namespace rmdev;\n\n";
    // The kernels come in parts, marked for jit.cpp, which may compile them as programs of their own side by side: the frame's
    // kernel (the optimistic scene), the exact pass behind it, and the evaluation of the scene at given points.
    // rm_trace: launched with 1, 4 or 9 waves per workgroup; rm_trace_refill: one wave (jit.cpp: launch_trace_module)
    char buf[1024];
    s += "//@part frame\n";
    std::snprintf(buf, sizeof buf,
                  "extern \"C\" __global__ __launch_bounds__(576, %d) void %s(const SceneView sc, const FrameView fr, Sample *samples,\n"
                  "        WorkCounters *counters) {\n    trace_body<CompiledSceneT<%s>, %d, %s>(sc, fr, samples, counters);\n}\n\n",
                  min_waves, counting ? "rm_trace_counting" : "rm_trace", counting ? "true" : "false", points_per_lane, counting ? "true" : "false");
    s += buf;
    s += "//@part exact\n";
    if (!counting && s.find("exact_wanted_word") != std::string::npos) {
        // the pass behind rm_trace over the tiles it flagged, which can also resolve the sample image into pixels
        // (trace_kernel.h: trace_exact_body); same workgroup shapes.
        // Only scenes with a lattice can flag a tile: the others get no such kernel, and their renders no such launch.
        std::snprintf(buf, sizeof buf,
                      "extern \"C\" __global__ __launch_bounds__(576, %d) void rm_trace_exact(const SceneView sc, const FrameView fr, Sample *samples,\n"
                      "        float *out_f32, unsigned char *out_rgba8) {\n"
                      "    trace_exact_body<CompiledSceneT<true>, %d>(sc, fr, samples, out_f32, out_rgba8);\n}\n\n", min_waves, points_per_lane);
        s += buf;
    }
    if (refill) {
        s += "//@part frame\n";
        std::snprintf(buf, sizeof buf,
                      "extern \"C\" __global__ __launch_bounds__(64, %d) void %s(const SceneView sc, const FrameView fr, Sample *samples,\n"
                      "        WorkCounters *counters) {\n    trace_refill_body<CompiledSceneT<true>, %s>(sc, fr, samples, counters);\n}\n\n",
                      min_waves, counting ? "rm_trace_refill_counting" : "rm_trace_refill", counting ? "true" : "false");
        s += buf;
        s += "//@part exact\n";
    }
    s += "//@part eval\n";
    std::snprintf(buf, sizeof buf,
                  "extern \"C\" __global__ __launch_bounds__(256, %d) void rm_eval_sdf(const SceneView sc, const float *xyz, uint32_t n, float *sd,\n"
                  "        uint32_t *prim, float *sd_march) {\n    eval_sdf_body<CompiledSceneT<true>>(sc, xyz, n, sd, prim, sd_march);\n}\n", min_waves);
    s += buf;
    return s;
}

}  // namespace rmdev
