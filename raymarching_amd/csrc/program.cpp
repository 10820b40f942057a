// program.cpp -- compiles the CSG tree into the flat device program.
//
// Replaces the reference's host-side flattening, render::shader::genscene and
// packbody (/root/reference/source/render.cpp:246-274, 294-366), which fill
// fixed 1024-stride tables that the shader walks with a per-invocation stack.
// Here the walk is done once on the host: list modes, "first child" cases and
// nesting slots are resolved into the operations (see device_scene.h), so the
// kernels only stream runs of same-kind primitives.
#include "program.h"

#include <cmath>

namespace rmdev {

namespace {

struct Compiler {
    const rm_node *nodes;
    uint32_t n_nodes;
    HostProgram &out;
    std::string &error;
    uint32_t cursor = 0;

    bool fail(const std::string &what) {
        error = what;
        return false;
    }

    static uint32_t fold_of(uint32_t mode, bool first) {
        uint32_t f = first ? FOLD_FIRST : 0u;
        switch (mode) {
        case RM_UNION: break;
        case RM_COMPLEMENT: f |= FOLD_NEG; break;
        case RM_INTERSECTION: f |= FOLD_MAX; break;
        case RM_DIFFERENCE: f |= FOLD_MAX; if (!first) f |= FOLD_NEG; break;
        }
        return f;
    }

    static bool iso(const rm_node &n) { return n.size[0] == n.size[1] && n.size[1] == n.size[2]; }

    static uint32_t kind_of(const rm_node &n) {
        switch (n.type) {
        case RM_SPHERE: return K_SPHERE;
        case RM_BOX: return iso(n) ? K_BOX_ISO : K_BOX;
        default: return iso(n) ? K_CROSS_ISO : K_CROSS;
        }
    }

    void add_prim(const rm_node &n) {
        Float4 a = { n.center[0], n.center[1], n.center[2], 0.0f };
        Float4 b = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (n.type == RM_SPHERE) {
            a.w = n.size[0];
            out.flops_per_sdf += 11;
        } else {
            a.w = n.size[0] / 2;
            b.x = n.size[1] / 2;
            b.y = n.size[2] / 2;
            out.flops_per_sdf += (n.type == RM_BOX) ? 9 : 15;
        }
        out.pa.push_back(a);
        out.pb.push_back(b);
        out.colors.push_back(Float4{ n.color[0], n.color[1], n.color[2], 0.0f });
        out.n_prims++;
    }

    // Compiles the children of the list at nodes[list_at]; cursor is on its first child.
    bool list_body(uint32_t list_at, uint32_t depth) {
        const uint32_t mode = nodes[list_at].mode;
        const uint32_t n_children = nodes[list_at].child_count;
        if (mode > RM_DIFFERENCE) return fail("list node with an unknown mode");
        out.n_lists++;
        if (depth > out.max_depth) out.max_depth = depth;
        uint32_t child = 0;
        while (child < n_children) {
            if (cursor >= n_nodes) return fail("node list ends inside a list (child_count too large)");
            const rm_node &n = nodes[cursor];
            if (n.type == RM_LIST) {
                if (depth >= kMaxDepth) return fail("lists nested deeper than 8 levels");
                const uint32_t at = cursor++;
                out.ops.push_back(Op{ OP_PUSH, depth, 0, 0 });
                if (!list_body(at, depth + 1)) return false;
                out.ops.push_back(Op{ OP_POP, depth | (fold_of(mode, child == 0) << 8), 0, 0 });
                child++;
                continue;
            }
            if (n.type > RM_CROSS) return fail("node with an unknown type");
            // a run: consecutive primitive children of the same kind; the first
            // child of the list gets a run of its own (it replaces the accumulator)
            const uint32_t kind = kind_of(n);
            const uint32_t first_row = out.n_prims;
            uint32_t count = 0;
            const bool first_child = (child == 0);
            while (child < n_children && cursor < n_nodes && nodes[cursor].type != RM_LIST &&
                   nodes[cursor].type <= RM_CROSS && kind_of(nodes[cursor]) == kind) {
                add_prim(nodes[cursor]);
                cursor++;
                child++;
                count++;
                if (first_child) break;
            }
            out.ops.push_back(Op{ OP_RUN, kind | (fold_of(mode, first_child) << 8), first_row, count });
        }
        return true;
    }
};

}  // namespace

bool compile_scene(const rm_node *nodes, uint32_t n_nodes, HostProgram &out, std::string &error) {
    out = HostProgram();
    if (!nodes || n_nodes == 0) {
        error = "empty node list (the root list is required)";
        return false;
    }
    if (nodes[0].type != RM_LIST) {
        error = "the first node must be the root list";
        return false;
    }
    Compiler c{ nodes, n_nodes, out, error };
    c.cursor = 1;
    if (!c.list_body(0, 0)) return false;
    if (c.cursor != n_nodes) {
        error = "nodes left over after the root list's children";
        return false;
    }
    out.ops.push_back(Op{ OP_END, 0, 0, 0 });
    // keep the tables non-empty so device pointers are always valid
    if (out.pa.empty()) {
        out.pa.push_back(Float4{ 0, 0, 0, 0 });
        out.pb.push_back(Float4{ 0, 0, 0, 0 });
        out.colors.push_back(Float4{ 0, 0, 0, 0 });
    }
    return true;
}

}  // namespace rmdev
