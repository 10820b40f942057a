// program.cpp -- compiles the CSG tree into the flat device program.
//
// Replaces the reference's host-side flattening, render::shader::genscene and
// packbody (/root/reference/source/render.cpp:246-274, 294-366), which fill
// fixed 1024-stride tables that the shader walks with a per-invocation stack.
// Here the walk is done once on the host: list modes, "first child" cases and
// nesting slots are resolved into the operations (see device_scene.h), so the
// kernels only stream runs of same-kind primitives.
//
// Runs of cubic crosses that form a complete Menger lattice (what
// Body::MengerSponge generates, body.cpp:114-156) are recognised on the node
// data itself and compiled to a LATTICE operation, which the kernels evaluate
// exactly without visiting every cross.
#include "program.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rmdev {

namespace {

uint32_t bits_of(float f) {
    uint32_t u;
    std::memcpy(&u, &f, sizeof u);
    return u;
}

// The 20 kept sub-cubes of one subdivision step as base-3 digit triples
// (digit = offset + 1), in the generator's order (body.cpp:122-145).
const uint8_t kKept[20][3] = {
    { 2, 0, 0 }, { 1, 0, 0 }, { 0, 0, 0 }, { 2, 2, 0 }, { 1, 2, 0 }, { 0, 2, 0 }, { 0, 1, 0 }, { 2, 1, 0 },
    { 2, 0, 2 }, { 1, 0, 2 }, { 0, 0, 2 }, { 2, 2, 2 }, { 1, 2, 2 }, { 0, 2, 2 }, { 0, 1, 2 }, { 2, 1, 2 },
    { 0, 0, 1 }, { 2, 0, 1 }, { 0, 2, 1 }, { 2, 2, 1 },
};

uint32_t pow3(uint32_t l) {
    uint32_t r = 1;
    while (l--) r *= 3;
    return r;
}

// number of crosses of a complete lattice with n levels: 1 + 20 + ... + 20^(n-1)
uint64_t lattice_size(uint32_t n) {
    uint64_t total = 0, level = 1;
    for (uint32_t l = 0; l < n; l++) {
        total += level;
        level *= 20;
    }
    return total;
}

// Checks that nodes[at .. at + lattice_size(n)) are a complete Menger lattice of n
// levels and, if so, fills its coordinate tables from the nodes.
struct LatticeMatcher {
    const rm_node *nodes;
    uint32_t n;
    uint32_t cursor;
    std::vector<float> X[kLatticeMaxLevels], Y[kLatticeMaxLevels], Z[kLatticeMaxLevels];
    std::vector<uint8_t> setx[kLatticeMaxLevels], sety[kLatticeMaxLevels], setz[kLatticeMaxLevels];
    float half[kLatticeMaxLevels];
    bool half_set[kLatticeMaxLevels];

    static bool put(std::vector<float> &tab, std::vector<uint8_t> &set, uint32_t i, float v) {
        if (set[i]) return bits_of(tab[i]) == bits_of(v);
        tab[i] = v;
        set[i] = 1;
        return true;
    }

    bool visit(uint32_t level, uint32_t ix, uint32_t iy, uint32_t iz) {
        const rm_node &c = nodes[cursor++];
        if (c.type != RM_CROSS || c.size[0] != c.size[1] || c.size[1] != c.size[2]) return false;
        if (std::memcmp(c.color, nodes[0].color, sizeof c.color) != 0) return false;
        const float h = c.size[0] / 2;
        if (!half_set[level]) {
            half[level] = h;
            half_set[level] = true;
        } else if (bits_of(half[level]) != bits_of(h)) {
            return false;
        }
        if (!put(X[level], setx[level], ix, c.center[0]) || !put(Y[level], sety[level], iy, c.center[1]) ||
            !put(Z[level], setz[level], iz, c.center[2]))
            return false;
        if (level + 1 < n) {
            for (const uint8_t *k : kKept)
                if (!visit(level + 1, 3 * ix + k[0], 3 * iy + k[1], 3 * iz + k[2])) return false;
        }
        return true;
    }

    // sorted, and every entry within 1 % of a step of front + i * step
    static bool on_grid(const std::vector<float> &t, double step) {
        if (!(step > 0.0) || !std::isfinite(step)) return false;
        for (size_t i = 0; i < t.size(); i++) {
            if (i && !(t[i - 1] < t[i])) return false;
            if (std::fabs((double)t[i] - ((double)t.front() + step * (double)i)) > 0.01 * step) return false;
        }
        return true;
    }

    // nodes must point at the first cross; returns true on a verified match
    bool match(uint32_t levels) {
        n = levels;
        cursor = 0;
        for (uint32_t l = 0; l < n; l++) {
            const uint32_t m = pow3(l);
            X[l].assign(m, 0.0f); Y[l].assign(m, 0.0f); Z[l].assign(m, 0.0f);
            setx[l].assign(m, 0); sety[l].assign(m, 0); setz[l].assign(m, 0);
            half_set[l] = false;
        }
        if (!visit(0, 0, 0, 0)) return false;
        for (uint32_t l = 0; l < n; l++) {
            if (!(half[l] > 0.0f) || !std::isfinite(half[l])) return false;
            if (l && !(half[l] < half[l - 1])) return false;
            if (l) {
                // The kernels estimate indices on all three axes with ONE spacing per level (the x table's,
                // as emit() stores it): every table must lie on that grid, and bars of one level must not
                // overlap.
                const float s = (X[l].back() - X[l].front()) / (float)(X[l].size() - 1);
                if (!on_grid(X[l], (double)s) || !on_grid(Y[l], (double)s) || !on_grid(Z[l], (double)s)) return false;
                if (!(2.0f * half[l] < s)) return false;
            }
        }
        return true;
    }

    // appends the lattice block (device_scene.h) and returns its word offset
    uint32_t emit(std::vector<uint32_t> &out) const {
        const uint32_t base = (uint32_t)out.size();
        out.resize(base + kLatHeaderWords, 0u);
        auto put_f = [&](uint32_t at, float v) { out[at] = bits_of(v); };
        out[base + 0] = n;
        double bound = 0.0;
        for (uint32_t l = 0; l < n; l++) {
            put_f(base + kLatHalf + l, half[l]);
            float spacing = 0.0f;
            if (X[l].size() > 1) spacing = (X[l].back() - X[l].front()) / (float)(X[l].size() - 1);
            put_f(base + kLatInvSpacing + l, spacing > 0.0f ? 1.0f / spacing : 0.0f);
            const uint32_t m = pow3(l);
            out[base + kLatOffset + l] = (uint32_t)out.size() - base;
            put_f(base + kLatOrigin + 3 * l, X[l].front());
            put_f(base + kLatOrigin + 3 * l + 1, Y[l].front());
            put_f(base + kLatOrigin + 3 * l + 2, Z[l].front());
            for (float v : X[l]) out.push_back(bits_of(v));
            for (float v : Y[l]) out.push_back(bits_of(v));
            for (float v : Z[l]) out.push_back(bits_of(v));
            for (uint32_t i = 0; i < m; i++) {
                uint32_t ones = 0, r = i;
                for (uint32_t d = 0; d < l; d++) {
                    if (r % 3 == 1) ones |= 1u << d;
                    r /= 3;
                }
                out.push_back(ones);
            }
            for (const std::vector<float> *t : { &X[l], &Y[l], &Z[l] })
                for (float v : *t) bound = std::fmax(bound, std::fabs((double)v));
        }
        put_f(base + 2, (float)(3.0 * bound + 3.0 * half[0]));
        return base;
    }
};

struct Compiler {
    const rm_node *nodes;
    uint32_t n_nodes;
    bool use_lattice;
    HostProgram &out;
    std::string &error;
    uint32_t cursor = 0;

    bool fail(const std::string &what) {
        error = what;
        return false;
    }

    static uint32_t fold_of(uint32_t mode, bool first) {
        uint32_t f = 0u;
        switch (mode) {
        case RM_UNION: break;
        case RM_COMPLEMENT: f |= FOLD_NEG; break;
        case RM_INTERSECTION: f |= FOLD_MAX; break;
        case RM_DIFFERENCE: f |= FOLD_MAX; if (!first) f |= FOLD_NEG; break;
        }
        return f;
    }

    // starting accumulator of a list: the identity of its fold; an empty list is +inf
    static float identity_of(const rm_node &list) {
        const bool is_max = list.mode == RM_INTERSECTION || list.mode == RM_DIFFERENCE;
        return (is_max && list.child_count > 0) ? -INFINITY : INFINITY;
    }

    static bool iso(const rm_node &n) { return n.size[0] == n.size[1] && n.size[1] == n.size[2]; }

    static uint32_t kind_of(const rm_node &n) {
        switch (n.type) {
        case RM_SPHERE: return K_SPHERE;
        case RM_BOX: return iso(n) ? K_BOX_ISO : K_BOX;
        default: return iso(n) ? K_CROSS_ISO : K_CROSS;
        }
    }

    // Every primitive gets its depth-first index (the order Body::List::SDF visits them): it
    // names the primitive's colour and breaks ties between equal distances, so that folds may
    // run in any order and still keep the earlier child, as the sequential fold does.
    uint32_t new_prim(const rm_node &n) {
        out.flops_per_sdf += n.type == RM_SPHERE ? 11 : (n.type == RM_BOX ? 9 : 15);
        out.colors.push_back(Float4{ n.color[0], n.color[1], n.color[2], 0.0f });
        return out.n_prims++;
    }

    static void halves(const rm_node &n, Float4 &a, Float4 &b) {
        a = Float4{ n.center[0], n.center[1], n.center[2], 0.0f };
        b = Float4{ 0.0f, 0.0f, 0.0f, 0.0f };
        if (n.type == RM_SPHERE) {
            a.w = n.size[0];
        } else {
            a.w = n.size[0] / 2;
            b.x = n.size[1] / 2;
            b.y = n.size[2] / 2;
        }
    }

    void emit_single(const rm_node &n, uint32_t prim, uint32_t fold) {
        Float4 a, b;
        halves(n, a, b);
        Rec r = {};
        r.a[0] = bits_of(a.x); r.a[1] = bits_of(a.y); r.a[2] = bits_of(a.z);
        r.h[0] = bits_of(a.w);
        r.h[1] = n.type == RM_SPHERE ? 0u : bits_of(b.x);
        r.h[2] = n.type == RM_SPHERE ? 0u : bits_of(b.y);
        r.flags = (n.type == RM_SPHERE ? REC_SPHERE : n.type == RM_BOX ? REC_BOX : REC_CROSS) | fold;
        r.prim = prim;
        out.recs.push_back(r);
    }

    // If the primitive children starting at `cursor` (at most `left` of them) begin with a
    // complete Menger lattice of >= 2 levels, emits a LATTICE record for it and returns the
    // number of children consumed; otherwise 0.
    uint32_t try_lattice(uint32_t left, uint32_t fold) {
        if (!use_lattice || fold != (FOLD_MAX | FOLD_NEG)) return 0;
        if (nodes[cursor].type != RM_CROSS || !iso(nodes[cursor])) return 0;
        // how many consecutive cubic crosses are there
        uint32_t avail = 0;
        while (avail < left && cursor + avail < n_nodes && nodes[cursor + avail].type == RM_CROSS &&
               iso(nodes[cursor + avail]))
            avail++;
        uint32_t levels = 0;
        while (levels < kLatticeMaxLevels && lattice_size(levels + 1) <= avail) levels++;
        LatticeMatcher m{};
        m.nodes = nodes + cursor;
        for (; levels >= 2; levels--) {
            // table words of this candidate must fit the per-workgroup LDS budget
            uint32_t words = kLatHeaderWords;
            for (uint32_t l = 0; l < levels; l++) words += 4 * pow3(l);
            if (out.lattice.size() + words > kLatticeMaxWords) continue;
            if (!m.match(levels)) continue;
            const uint32_t count = (uint32_t)lattice_size(levels);
            const uint32_t offset = m.emit(out.lattice);
            const uint32_t first_prim = out.n_prims;
            for (uint32_t i = 0; i < count; i++) new_prim(nodes[cursor + i]);
            Rec r = {};
            r.a[0] = levels; r.a[1] = offset; r.a[2] = count;
            r.flags = REC_LATTICE | fold;
            r.prim = first_prim;
            out.recs.push_back(r);
            out.n_lattices++;
            out.lattice_prims += count;
            cursor += count;
            return count;
        }
        return 0;
    }

    struct Bucket {
        std::vector<uint32_t> node, prim;
    };

    // ---- RUN emission, with bounding hulls for long runs --------------------------------------

    struct Row {
        Float4 a, b;
        uint32_t prim;
        uint64_t key;   // position along the space-filling curve
    };

    static uint64_t spread3(uint32_t v) {   // 21 bits -> every third bit
        uint64_t x = v & 0x1fffffu;
        x = (x | x << 32) & 0x1f00000000ffffull;
        x = (x | x << 16) & 0x1f0000ff0000ffull;
        x = (x | x << 8) & 0x100f00f00f00f00full;
        x = (x | x << 4) & 0x10c30c30c30c30c3ull;
        x = (x | x << 2) & 0x1249249249249249ull;
        return x;
    }

    static float round_up(double v) {
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, INFINITY);
        return f;
    }

    // half sizes of a row on the three axes (spheres: the radius on each)
    static void extents(uint32_t kind, const Row &r, double h[3]) {
        const bool one = kind == K_SPHERE || kind == K_BOX_ISO || kind == K_CROSS_ISO;
        h[0] = r.a.w;
        h[1] = one ? r.a.w : r.b.x;
        h[2] = one ? r.a.w : r.b.y;
    }

    // Hull of rows[i0, i1), padded by 1e-6 relative so that the rounding of its own centre and sizes
    // cannot let a row stick out.  dir < 0: a primitive of the run's kind (sphere; box or cross with
    // three half sizes) that contains the rows.  dir = 0, 1, 2: the flat hull of the rows' bars along
    // z, y, x -- a = (centre a, centre b, half a, half b) on the two axes that place the bar.
    static Row hull_of(uint32_t kind, int dir, const std::vector<Row> &rows, size_t i0, size_t i1) {
        double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = i0; i < i1; i++) {
            double h[3];
            extents(kind, rows[i], h);
            const double c[3] = { rows[i].a.x, rows[i].a.y, rows[i].a.z };
            for (int k = 0; k < 3; k++) {
                lo[k] = std::fmin(lo[k], c[k] - h[k]);
                hi[k] = std::fmax(hi[k], c[k] + h[k]);
            }
        }
        float cf[3], hf[3];
        for (int k = 0; k < 3; k++) {
            cf[k] = (float)((lo[k] + hi[k]) / 2);
            const double h = std::fmax(hi[k] - (double)cf[k], (double)cf[k] - lo[k]);
            hf[k] = round_up(h * (1.0 + 1e-6) + 1e-6 * std::fabs((double)cf[k]) + 1e-30);
        }
        Row out{};
        if (dir >= 0) {
            const int ka = dir == 2 ? 1 : 0, kb = dir == 0 ? 1 : 2;
            out.a = Float4{ cf[ka], cf[kb], hf[ka], hf[kb] };
        } else if (kind == K_SPHERE) {
            double radius = 0.0;
            for (size_t i = i0; i < i1; i++) {
                const double dx = (double)rows[i].a.x - cf[0], dy = (double)rows[i].a.y - cf[1], dz = (double)rows[i].a.z - cf[2];
                radius = std::fmax(radius, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)rows[i].a.w);
            }
            out.a = Float4{ cf[0], cf[1], cf[2], round_up(radius * (1.0 + 1e-6) + 1e-30) };
        } else {
            out.a = Float4{ cf[0], cf[1], cf[2], hf[0] };
            out.b = Float4{ hf[1], hf[2], 0.0f, 0.0f };
        }
        return out;
    }

    void push_row(const Row &r) {
        out.pa.push_back(r.a);
        out.pb.push_back(r.b);
        out.row_prim.push_back(r.prim);
    }

    // One listing of a clustered run: the rows sorted along a Morton curve (over all three centre
    // coordinates, or over the two that place the bars of direction `dir`), then a hull per 8 rows.
    bool emit_listing(uint32_t kind, int dir, std::vector<Row> rows) {
        double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (const Row &row : rows) {
            const double c[3] = { row.a.x, row.a.y, row.a.z };
            for (int k = 0; k < 3; k++) {
                lo[k] = std::fmin(lo[k], c[k]);
                hi[k] = std::fmax(hi[k], c[k]);
            }
        }
        const int skip = dir == 0 ? 2 : (dir == 1 ? 1 : (dir == 2 ? 0 : -1));   // the axis the bars run along
        for (Row &row : rows) {
            const double c[3] = { row.a.x, row.a.y, row.a.z };
            uint32_t q[3];
            for (int k = 0; k < 3; k++) {
                const double span = hi[k] - lo[k];
                q[k] = (span > 0.0 && k != skip) ? (uint32_t)std::fmin(2097151.0, (c[k] - lo[k]) / span * 2097151.0) : 0u;
            }
            row.key = spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2;
        }
        std::stable_sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.key < y.key; });
        for (const Row &row : rows) push_row(row);
        const size_t n_clusters = (rows.size() + kClusterSize - 1) / kClusterSize;
        bool sound = true;
        for (size_t c = 0; c < n_clusters; c++) {
            const size_t i0 = c * kClusterSize, i1 = std::min(rows.size(), (c + 1) * kClusterSize);
            const Row hull = hull_of(kind, dir, rows, i0, i1);
            sound = sound && contains(kind, dir, hull, rows, i0, i1);
            push_row(hull);
        }
        return sound;
    }

    // The invariant the kernels' pruning rests on, checked on the float values the device will read:
    // every row's interval on every hull axis lies inside the hull's (spheres: inside the hull sphere).
    static bool contains(uint32_t kind, int dir, const Row &hull, const std::vector<Row> &rows, size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            double h[3];
            extents(kind, rows[i], h);
            const double c[3] = { rows[i].a.x, rows[i].a.y, rows[i].a.z };
            if (dir >= 0) {
                const int ka = dir == 2 ? 1 : 0, kb = dir == 0 ? 1 : 2;
                if (!(std::fabs(c[ka] - (double)hull.a.x) + h[ka] <= (double)hull.a.z)) return false;
                if (!(std::fabs(c[kb] - (double)hull.a.y) + h[kb] <= (double)hull.a.w)) return false;
            } else if (kind == K_SPHERE) {
                const double dx = c[0] - (double)hull.a.x, dy = c[1] - (double)hull.a.y, dz = c[2] - (double)hull.a.z;
                if (!(std::sqrt(dx * dx + dy * dy + dz * dz) + h[0] <= (double)hull.a.w)) return false;
            } else {
                const double hh[3] = { hull.a.w, hull.b.x, hull.b.y }, hc[3] = { hull.a.x, hull.a.y, hull.a.z };
                for (int k = 0; k < 3; k++)
                    if (!(std::fabs(c[k] - hc[k]) + h[k] <= hh[k])) return false;
            }
        }
        return true;
    }

    void emit_run(uint32_t kind, const Bucket &bk, uint32_t fold, uint32_t depth) {
        std::vector<Row> rows(bk.node.size());
        for (size_t i = 0; i < rows.size(); i++) {
            halves(nodes[bk.node[i]], rows[i].a, rows[i].b);
            rows[i].prim = bk.prim[i];
        }
        Rec r = {};
        r.a[0] = (uint32_t)out.pa.size(); r.a[1] = (uint32_t)rows.size(); r.a[2] = kind;
        r.flags = REC_RUN | fold;
        out.run_prims += (uint32_t)rows.size();
        // hulls bound distances from below: of use to the folds that keep small distances
        const bool keeps_small = fold == 0u || fold == (FOLD_MAX | FOLD_NEG);
        bool finite = true;
        double magnitude = 0.0;
        for (const Row &row : rows) {
            double h[3];
            extents(kind, row, h);
            const double c[3] = { row.a.x, row.a.y, row.a.z };
            for (int k = 0; k < 3; k++) {
                finite = finite && std::isfinite(c[k]) && std::isfinite(h[k]) && h[k] >= 0.0;
                magnitude = std::fmax(magnitude, std::fabs(c[k]) + h[k]);
            }
        }
        if (use_lattice && keeps_small && finite && rows.size() >= 2 * kClusterSize) {
            bool sound = true;
            if (kind == K_CROSS || kind == K_CROSS_ISO) {
                for (int dir = 0; dir < 3; dir++) sound = emit_listing(kind, dir, rows) && sound;
            } else {
                sound = emit_listing(kind, -1, rows);
            }
            if (sound) {
                r.h[2] = bits_of((float)(3.0 * magnitude * (1.0 + 1e-5)));
                r.flags |= RUN_CLUSTERED;
                if (depth == 0 && fold == 0u && out.root_identity > 0 && out.grid_runs.size() < kGridMaxRuns) {
                    const uint32_t cls = kind == K_SPHERE ? kGridClassSphere : ((kind == K_BOX || kind == K_BOX_ISO) ? kGridClassBox : kGridClassCross);
                    // the first listing's rows were pushed in sorted order right at r.a[0]
                    out.grid_runs.push_back(GridRun{ r.a[0], (uint32_t)rows.size(), cls });
                    const bool first = out.grid_runs.size() == 1;
                    for (const Row &row : rows) {
                        double h[3];
                        extents(kind, row, h);
                        const double c[3] = { row.a.x, row.a.y, row.a.z };
                        for (int k = 0; k < 3; k++) {
                            const double e = cls == kGridClassCross ? std::fmax(h[0], std::fmax(h[1], h[2])) : h[k];
                            const float lo = (float)(c[k] - e), hi = (float)(c[k] + e);
                            if (first && &row == &rows[0]) { out.grid_lo[k] = lo; out.grid_hi[k] = hi; }
                            out.grid_lo[k] = std::fmin(out.grid_lo[k], lo);
                            out.grid_hi[k] = std::fmax(out.grid_hi[k], hi);
                        }
                    }
                }
            } else {
                // never expected (the hulls are padded well beyond their rounding); a plain run is always right
                out.pa.resize(r.a[0]);
                out.pb.resize(r.a[0]);
                out.row_prim.resize(r.a[0]);
                for (const Row &row : rows) push_row(row);
            }
        } else {
            for (const Row &row : rows) push_row(row);
        }
        out.recs.push_back(r);
    }

    // Compiles the children of the list at nodes[list_at]; cursor is on its first child.
    //
    // min and max are commutative and associative on floats, so the children after the first may
    // be folded in any order as long as ties are broken by depth-first index (the kernels do):
    // primitives are gathered per kind into RUNs over the tables, nested lists and lattices stay
    // where they are.
    bool list_body(uint32_t list_at, uint32_t depth) {
        const uint32_t mode = nodes[list_at].mode;
        const uint32_t n_children = nodes[list_at].child_count;
        if (mode > RM_DIFFERENCE) return fail("list node with an unknown mode");
        out.n_lists++;
        Bucket buckets[K_CROSS_ISO + 1];
        const uint32_t fold_rest = fold_of(mode, false);
        uint32_t child = 0;
        while (child < n_children) {
            if (cursor >= n_nodes) return fail("node list ends inside a list (child_count too large)");
            const rm_node &n = nodes[cursor];
            if (n.type == RM_LIST) {
                if (depth >= kMaxDepth) return fail("lists nested deeper than 8 levels");
                const uint32_t at = cursor++;
                Rec push = {};
                push.a[0] = depth;
                push.a[1] = bits_of(identity_of(nodes[at]));
                push.flags = REC_PUSH;
                out.recs.push_back(push);
                if (depth + 1 > out.max_depth) out.max_depth = depth + 1;
                if (!list_body(at, depth + 1)) return false;
                Rec pop = {};
                pop.a[0] = depth;
                pop.flags = REC_POP | fold_of(mode, child == 0);
                out.recs.push_back(pop);
                child++;
                continue;
            }
            if (n.type > RM_CROSS) return fail("node with an unknown type");
            if (child == 0) {   // the first child replaces the accumulator: always a record of its own
                emit_single(n, new_prim(n), fold_of(mode, true));
                cursor++;
                child++;
                continue;
            }
            if (const uint32_t used = try_lattice(n_children - child, fold_rest)) {
                child += used;
                continue;
            }
            Bucket &bk = buckets[kind_of(n)];
            bk.node.push_back(cursor);
            bk.prim.push_back(new_prim(n));
            cursor++;
            child++;
        }
        for (uint32_t kind = K_SPHERE; kind <= K_CROSS_ISO; kind++) {
            const Bucket &bk = buckets[kind];
            if (bk.node.size() >= kRunMin) {
                emit_run(kind, bk, fold_rest, depth);
            } else {
                for (size_t i = 0; i < bk.node.size(); i++) emit_single(nodes[bk.node[i]], bk.prim[i], fold_rest);
            }
        }
        return true;
    }
};

}  // namespace

bool compile_scene(const rm_node *nodes, uint32_t n_nodes, bool use_lattice, HostProgram &out, std::string &error) {
    out = HostProgram();
    if (!nodes || n_nodes == 0) {
        error = "empty node list (the root list is required)";
        return false;
    }
    if (nodes[0].type != RM_LIST) {
        error = "the first node must be the root list";
        return false;
    }
    Compiler c{ nodes, n_nodes, use_lattice, out, error };
    c.cursor = 1;
    out.root_identity = Compiler::identity_of(nodes[0]);
    if (!c.list_body(0, 0)) return false;
    if (c.cursor != n_nodes) {
        error = "nodes left over after the root list's children";
        return false;
    }
    out.recs.push_back(Rec{});   // REC_END
    // keep the tables non-empty so device pointers are always valid
    if (out.pa.empty()) {
        out.pa.push_back(Float4{ 0, 0, 0, 0 });
        out.pb.push_back(Float4{ 0, 0, 0, 0 });
        out.row_prim.push_back(0u);
    }
    if (out.colors.empty()) out.colors.push_back(Float4{ 0, 0, 0, 0 });
    if (out.lattice.empty()) out.lattice.push_back(0u);
    return true;
}

}  // namespace rmdev
