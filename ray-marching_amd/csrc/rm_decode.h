// rm_decode.h -- validation of the reference wire format and decoding into RmRecord[].
// Wire format: csg/builder.rs:26-62 (opcode word followed by f32::to_bits parameters),
// sphere.rs:15-21 (5 words), box.rs:14-20 (7 words), operations/mod.rs:12-18 (1 word,
// post-order).  Host-only, no HIP.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rm_device.h"
#include "rm_units.h"

struct RmDecoded {
    std::vector<RmRecord> rec;
    uint32_t n_words = 0;      // words consumed by cmd_count commands
    uint32_t max_depth = 0;    // value-stack depth of the reference machine
    uint32_t spill_depth = 0;  // LDS slots the accumulator machine needs
    uint32_t n_sphere = 0, n_box = 0;  // cone / slab entries of the miss-test tables; RmRecord::p[6] = slot
    uint32_t n_plane = 0;              // unbounded primitives: they veto miss-ray culling
    uint32_t n_leaves = 0;             // sphere + box leaves the program EVALUATES (subtracted ones included, which have no table
                                       // slot): what the automatic pruning decision (RM_OPT_PRUNE = 2) counts
    bool has_extensions = false;       // uses node types the reference does not implement
    // How far SmoothUnion operators can pull the tree value below the minimum over its leaves: the miss tests and
    // the pixel-cone pre-pass inflate every bound by it.  Tracked per value-stack entry while decoding (slack[]):
    //   leaf 0;  min / max(a,-b) / max(a,b): max(S_a, S_b);  value * s (ScalePop): S * |s|;
    //   smin_k(a, b) with one operand exact (S = 0):  max(S_other, k)      [*]
    //   smin_k(a, b) with both inexact:               max(S_a, S_b) + k/4
    // [*] a chain a_0 ~ a_1 ~ a_2 ... of smooth unions never falls more than max k below the smallest operand:
    //     with acc >= M - K (K >= k), b >= B exactly and d = |acc - b| < k, the result min(acc,b) - (k-d)^2/(4k)
    //     is >= min(M,B) - K in each of the three cases (b the new minimum: d + (k-d)^2/4k <= k; acc still below
    //     M: the gap g = M - acc < d, so g + (k-d)^2/4k < g + (k-g)^2/4k <= k; b < acc: only k/4 is lost).
    //     The first version charged k/4 per operator, 3.75 for the 15-operator chain of the smooth-min test scene
    //     instead of 0.25, which inflated every bounding sphere past its neighbours and disabled most of the culling.
    double smooth_slack = 0.0;
    // Far-primitive pruning in specialised kernels (rm_kernel_v5.h "Pruning"): allowed when every node is a
    // 1-Lipschitz leaf or a min/max operator; scene_scale = 1 + max over primitives of |centre|_1 + |size|_1
    bool prunable = true;
    float scene_scale = 1.0f;
    // Chain program: record 0 pushes a sphere / box, every later record is a sphere / box fused with a Union / Subtraction
    // (RM_OP_FASTCLASS): the interpreter kernels run such programs through map_scene_chain (rm_interp.h)
    bool is_chain = false;
    // Tree program: every record is one of the eight shapes of RM_OP_FASTCLASS (spheres, boxes, Union, Subtraction in any
    // arrangement): the interpreter kernels run it through map_scene_tree.
    // tree: with unit records (unit_mode RM_UNITS_LATTICE) and at most 128 records, one entry per RECORD from which a wave turns
    // its unit mask into the set of records it has to execute at all (rm_kernel_v5.h tree_keep; executed by
    // map_scene_tree_masked).  L / R: the units (leaves) of the record's left / right operand, as 64-bit masks:
    //   p[0], p[1]  L, low and high word: a leaf fused with its operator -- everything the accumulator holds when it is reached;
    //               an operator record -- the operand it pops; a pushed leaf -- 0
    //   p[2], p[3]  R: a leaf -- its own unit; an operator record -- the operand in the accumulator
    //   p[4]        bits 0-1: 0 pushed leaf, 1 fused leaf, 2 operator record; bit 2: the operator is a Subtraction;
    //               bits 8-13: the unit of L's FIRST leaf (always a pushed leaf)
    bool is_tree = false;
    std::vector<RmRecord> tree;
    // The miss test of a ray can run the program on lower bounds (rm_kernel_v5.h "Miss test on lower bounds"): the program
    // blends with SmoothUnion, holds a Plane the tables cannot clear, or an Intersection (otherwise the plain tests are as
    // sharp), its leaves
    // are in world space (no transforms), and the accumulator machine never holds more than one spilled value
    bool bound_walk = false;
    // Wave-level culling (rm_kernel_v5.h; which records form a unit: rm_units.h): one record per unit, in unit order, behind the
    // program in device memory and in LDS.  A unit stands for a leaf (p[0..2] its centre) bounded from both sides:
    //   p[3]  outer radius R' * 1.000005 rounded up:  value(q) >= |q - c| - p[3]    (+inf for an opaque unit)
    //         R' = (R + 2e-6 (|c|_1 + R)) (1 + 1e-6),  R: sphere max(r, 0), box |max(h, 0)|, cylinder |(max(r,0), max(hh,0))|
    //         -- the slack covers the leaf's own evaluation error near the unit (~4e-7 of |q|_1 + |c|_1 + R) and the rounding of c
    //   p[4]  inner radius, rounded down:             value(q) <= |q - c| - p[4]    (-inf for an opaque unit)
    //         sphere r, box min h, cylinder min(r, hh); with a negative size in it, minus the sum of the absolute sizes
    //   p[5]  the blend radius of the unit's SmoothUnion, max(k, 0); 0 for every other unit
    //   p[6]  the unit's kind (RM_UNIT_*), as an integer
    //   op    first | last << 16: the records [first, last] behind the unit's bit (not an opcode word)
    // unit_mode: RM_UNITS_LATTICE (prunable programs), RM_UNITS_BLEND (programs that blend whose top level is a chain), or
    // RM_UNITS_NONE -- also when there are more than 64 units.  unit_kmax: the largest p[5].
    std::vector<RmRecord> units;
    uint32_t unit_mode = RM_UNITS_NONE;
    float unit_kmax = 0.0f;
    // Space transformations (extension): deepest nesting, and -- because a transformed primitive's parameters no longer
    // say where it is -- one world-space bounding sphere (x, y, z, radius) per bounded primitive for the miss tests.
    // cull_veto: some transform is not a similarity (non-unit quaternion, scale not positive and finite): no culling.
    uint32_t xform_depth = 0;
    bool has_xforms = false, cull_veto = false;
    std::vector<float> bounds;  // 4 floats per cone slot; empty unless has_xforms
    // Materials (extension): `rec` never contains the tags (the march does not depend on them, so a tagged scene
    // keeps its untagged structure, fusion and specialised kernel); `mrec` is the program decoded with the tags in
    // place, for the one evaluation per hit that asks which material the surface carries.
    bool has_materials = false;
    uint32_t max_material = 0;       // largest index a tag names
    std::vector<RmRecord> mrec;
    uint32_t mat_spill_depth = 0;    // value-stack spill slots of mrec (each holds a distance and an index)
    uint32_t mat_xform_depth = 0;
};

// One open transform scope during decoding.
struct RmXformScope {
    uint32_t op;      // the push opcode
    uint32_t depth;   // value-stack depth at the push
    double p[4];      // its parameters
};

// Returns RM_OK or a negative rm_status.  `cap_words` is the number of u32 words that
// exist after the cmd_count word (255 for the reference's 1024-byte buffer).
static inline int rm_decode_core(uint32_t cmd_count, const uint32_t* words, uint32_t cap_words, bool keep_materials,
                                 RmDecoded* out) {
    RmDecoded d;
    if (cmd_count && cap_words && !words) return RM_ERR_NULL;
    // Every command is at least one word, so at most cap_words records can ever be produced.
    d.rec.reserve(cmd_count < cap_words ? cmd_count : cap_words);
    uint32_t ptr = 0, depth = 0, spilled = 0;
    {   // does the program use transforms at all?  (decides where the miss tests take a primitive's position from)
        uint32_t q = 0;
        for (uint32_t i = 0; i < cmd_count && q < cap_words; i++) {
            const uint32_t op = words[q++];
            if (op >= RM_CMD_TRANSLATION_PUSH && op <= RM_CMD_SCALE_POP) { d.has_xforms = true; break; }
            q += op == RM_CMD_SPHERE || op == RM_CMD_PLANE ? 4u : op == RM_CMD_BOX ? 6u : op == RM_CMD_CYLINDER ? 5u
               : op == RM_CMD_SMOOTH_UNION || op == RM_CMD_MATERIAL ? 1u : 0u;
        }
    }
    std::vector<RmXformScope> scopes;
    std::vector<double> slack;  // per value-stack entry, see RmDecoded::smooth_slack
    std::vector<size_t> first;  // per value-stack entry: index of the first record of the sub-tree that produced it
    auto combine = [&](uint32_t op_mode, double k) {  // pops b and a, pushes the operator's slack
        const double sb = slack.back(); slack.pop_back();
        const double sa = slack.back(); slack.pop_back();
        const size_t fb = first.back(); first.pop_back();  // a's first record stays: the result spans both
        if (op_mode == RM_MODE_SUB)  // everything b was built from: out of the miss-test tables (RM_OP_NOCULL)
            for (size_t j = fb; j < d.rec.size(); j++) d.rec[j].op |= RM_OP_NOCULL;
        double sv = sa > sb ? sa : sb;
        if (op_mode == RM_MODE_SMOOTH) {
            if (k > 0.0) sv = (sa == 0.0 || sb == 0.0) ? (sv > k ? sv : k) : sv + 0.25 * k;
            else if (!(k <= 0.0)) sv = 1.0 / 0.0;  // NaN k: nothing can be bounded
        }
        slack.push_back(sv);
    };
    for (uint32_t i = 0; i < cmd_count; i++) {
        if (ptr >= cap_words) return RM_ERR_TRUNCATED;
        uint32_t op = words[ptr++];
        RmRecord r;
        std::memset(&r, 0, sizeof r);
        uint32_t kind = RM_KIND_POP, np = 0;
        if (op == RM_CMD_TRANSLATION_PUSH || op == RM_CMD_ROTATION_PUSH || op == RM_CMD_SCALE_PUSH) {
            const uint32_t n = op == RM_CMD_TRANSLATION_PUSH ? 3u : op == RM_CMD_ROTATION_PUSH ? 4u : 1u;
            if (ptr + n > cap_words) return RM_ERR_TRUNCATED;
            if (scopes.size() == RM_MAX_XFORM_DEPTH) return RM_ERR_TRANSFORM;
            std::memcpy(r.p, words + ptr, n * 4);
            ptr += n;
            RmXformScope sc{op, depth, {r.p[0], r.p[1], r.p[2], r.p[3]}};
            if (op == RM_CMD_ROTATION_PUSH) {
                const double nn = sc.p[0] * sc.p[0] + sc.p[1] * sc.p[1] + sc.p[2] * sc.p[2] + sc.p[3] * sc.p[3];
                if (!(std::fabs(nn - 1.0) < 1.0e-4)) d.cull_veto = true;  // not a rotation (also takes NaN)
            } else if (op == RM_CMD_SCALE_PUSH) {
                if (!(sc.p[0] > 0.0 && sc.p[0] < 1.0e30)) d.cull_veto = true;
            } else if (!(std::fabs(sc.p[0]) + std::fabs(sc.p[1]) + std::fabs(sc.p[2]) < 1.0e30)) {
                d.cull_veto = true;
            }
            const uint32_t level = (uint32_t)scopes.size();
            std::memcpy(&r.p[6], &level, 4);
            scopes.push_back(sc);
            if (scopes.size() > d.xform_depth) d.xform_depth = (uint32_t)scopes.size();
            r.op = RM_OP(RM_KIND_XFORM, op == RM_CMD_TRANSLATION_PUSH ? RM_XF_T_PUSH : op == RM_CMD_ROTATION_PUSH ? RM_XF_R_PUSH : RM_XF_S_PUSH, 0);
            d.has_extensions = true;
            d.prunable = false;
            d.rec.push_back(r);
            continue;
        }
        if (op == RM_CMD_TRANSLATION_POP || op == RM_CMD_ROTATION_POP || op == RM_CMD_SCALE_POP) {
            // closes the innermost scope, which must be of its kind and have produced exactly one value
            if (scopes.empty() || scopes.back().op + 1u != op || depth != scopes.back().depth + 1u) return RM_ERR_TRANSFORM;
            const uint32_t level = (uint32_t)scopes.size() - 1u;
            r.p[0] = (float)scopes.back().p[0];  // ScalePop multiplies the child's value by its push's scale
            if (op == RM_CMD_SCALE_POP && !slack.empty()) slack.back() *= std::fabs(scopes.back().p[0]);
            std::memcpy(&r.p[6], &level, 4);
            scopes.pop_back();
            r.op = RM_OP(RM_KIND_XFORM, op == RM_CMD_TRANSLATION_POP ? RM_XF_T_POP : op == RM_CMD_ROTATION_POP ? RM_XF_R_POP : RM_XF_S_POP, 0);
            d.rec.push_back(r);
            continue;
        }
        if (op == RM_CMD_MATERIAL && keep_materials) {  // tags the value on top of the stack (inside a scope: the scope's own)
            if (ptr + 1 > cap_words) return RM_ERR_TRUNCATED;
            const uint32_t index = words[ptr++];
            if (index >= RM_MAX_MATERIALS) return RM_ERR_MATERIAL;
            if (depth < 1 || (!scopes.empty() && depth <= scopes.back().depth)) return RM_ERR_STACK_UNDERFLOW;
            std::memcpy(&r.p[0], &index, 4);
            r.op = RM_OP(RM_KIND_MATERIAL, 0, 0);
            d.has_materials = true;
            d.has_extensions = true;
            if (index > d.max_material) d.max_material = index;
            d.rec.push_back(r);
            continue;
        }
        switch (op) {
        case RM_CMD_SPHERE: kind = RM_KIND_SPHERE; np = 4; break;
        case RM_CMD_BOX: kind = RM_KIND_BOX; np = 6; break;
        case RM_CMD_CYLINDER: kind = RM_KIND_CYLINDER; np = 5; d.has_extensions = true; break;
        case RM_CMD_PLANE: kind = RM_KIND_PLANE; np = 4; d.has_extensions = true; d.n_plane++; break;
        default: break;
        }
        if (kind != RM_KIND_POP) {
            if (ptr + np > cap_words) return RM_ERR_TRUNCATED;
            std::memcpy(r.p, words + ptr, np * 4);
            ptr += np;
            {
                double ext = 0.0;
                for (uint32_t k = 0; k < np; k++) ext += std::fabs((double)r.p[k]);  // NaN / inf propagate: nothing is pruned then
                const float up = std::nextafterf((float)(1.0 + ext), INFINITY);
                if (!(up <= d.scene_scale)) d.scene_scale = up;  // also takes a NaN
                if (kind == RM_KIND_PLANE) d.prunable = false;   // |n| need not be 1
                // pruning threshold of a sphere, pre-multiplied (see spec_sphere_far): radius * 1.000005, rounded up
                if (kind == RM_KIND_SPHERE) r.p[4] = std::nextafterf((float)((double)r.p[3] * 1.000005), INFINITY);
            }
            // slot in the kernels' per-kind miss-test tables: cones for spheres, slabs for boxes and cylinders; in a
            // program with transforms every bounded primitive is a cone around its world-space bounding sphere
            uint32_t slot = 0u;
            if (d.has_xforms && kind != RM_KIND_PLANE) {
                slot = d.n_sphere++;
                double c[3] = {r.p[0], r.p[1], r.p[2]};
                double rho = kind == RM_KIND_SPHERE ? std::fmax((double)r.p[3], 0.0)
                           : kind == RM_KIND_BOX ? std::sqrt(std::pow(std::fmax((double)r.p[3], 0.0), 2) + std::pow(std::fmax((double)r.p[4], 0.0), 2) +
                                                             std::pow(std::fmax((double)r.p[5], 0.0), 2))
                           : std::sqrt(std::pow(std::fmax((double)r.p[3], 0.0), 2) + std::pow(std::fmax((double)r.p[4], 0.0), 2));
                for (size_t k = scopes.size(); k-- > 0;) {  // innermost scope first: local -> world
                    const RmXformScope& sc = scopes[k];
                    if (sc.op == RM_CMD_TRANSLATION_PUSH) {
                        c[0] += sc.p[0]; c[1] += sc.p[1]; c[2] += sc.p[2];
                    } else if (sc.op == RM_CMD_SCALE_PUSH) {
                        c[0] *= sc.p[0]; c[1] *= sc.p[0]; c[2] *= sc.p[0];
                        rho *= std::fabs(sc.p[0]);
                    } else {  // x_world = q x conj(q)
                        const double w = sc.p[0], a[3] = {sc.p[1], sc.p[2], sc.p[3]};
                        const double cx[3] = {a[1] * c[2] - a[2] * c[1], a[2] * c[0] - a[0] * c[2], a[0] * c[1] - a[1] * c[0]};
                        const double t[3] = {2.0 * cx[0], 2.0 * cx[1], 2.0 * cx[2]};
                        const double u[3] = {a[1] * t[2] - a[2] * t[1], a[2] * t[0] - a[0] * t[2], a[0] * t[1] - a[1] * t[0]};
                        for (int m = 0; m < 3; m++) c[m] = c[m] + w * t[m] + u[m];
                        rho *= 1.0 + 1.0e-3;  // |q| within 5e-5 of 1 (else cull_veto): lengths change by < 1e-4
                    }
                }
                const double mag = std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]) + rho;
                d.bounds.push_back((float)c[0]); d.bounds.push_back((float)c[1]); d.bounds.push_back((float)c[2]);
                d.bounds.push_back(std::nextafterf((float)(rho * (1.0 + 1.0e-5) + 1.0e-6 * mag), INFINITY));
                if (!(mag < 1.0e30)) d.cull_veto = true;
            } else {
                slot = kind == RM_KIND_SPHERE ? d.n_sphere++ : (kind == RM_KIND_PLANE ? 0u : d.n_box++);
            }
            std::memcpy(&r.p[6], &slot, 4);
            // Fuse with a directly following parameter-less binary operator: its rhs is this leaf.
            uint32_t mode = RM_MODE_PUSH;
            if (i + 1 < cmd_count && ptr < cap_words && depth >= 1) {
                if (words[ptr] == RM_CMD_UNION) mode = RM_MODE_UNION;
                else if (words[ptr] == RM_CMD_SUBTRACTION) mode = RM_MODE_SUB;
                else if (words[ptr] == RM_CMD_INTERSECTION) { mode = RM_MODE_INTER; d.has_extensions = true; }
            }
            slack.push_back(0.0);
            first.push_back(d.rec.size());
            if (mode != RM_MODE_PUSH) {
                combine(mode, 0.0);
                ptr++;  // consume the operator: depth is unchanged (push, then pop 2 push 1)
                i++;
                if (depth + 1 > 32) return RM_ERR_STACK_OVERFLOW;  // the reference machine peaks one higher
                if (depth + 1 > d.max_depth) d.max_depth = depth + 1;
                r.op = RM_OP(kind, mode, 0) | (mode == RM_MODE_SUB ? (uint32_t)RM_OP_NOCULL : 0u);  // this leaf IS the right operand
                if ((kind == RM_KIND_SPHERE || kind == RM_KIND_BOX) && (mode == RM_MODE_UNION || mode == RM_MODE_SUB))
                    r.op |= ((kind == RM_KIND_SPHERE ? 1u : 2u) + (mode == RM_MODE_SUB ? 2u : 0u)) << 16;  // RM_OP_FASTCLASS
            } else {
                uint32_t spill = depth >= 1 ? 1u : 0u;  // a live accumulator must be saved
                if (spill) { spilled++; if (spilled > d.spill_depth) d.spill_depth = spilled; }
                depth++;
                if (depth > 32) return RM_ERR_STACK_OVERFLOW;
                if (depth > d.max_depth) d.max_depth = depth;
                r.op = RM_OP(kind, RM_MODE_PUSH, spill);
                if (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX) r.op |= (kind == RM_KIND_SPHERE ? 5u : 6u) << 16;  // RM_OP_FASTCLASS
            }
            d.rec.push_back(r);
        } else if (op == RM_CMD_UNION || op == RM_CMD_SUBTRACTION || op == RM_CMD_INTERSECTION ||
                   op == RM_CMD_SMOOTH_UNION) {
            uint32_t mode = op == RM_CMD_UNION ? RM_MODE_UNION : op == RM_CMD_SUBTRACTION ? RM_MODE_SUB
                          : op == RM_CMD_INTERSECTION ? RM_MODE_INTER : RM_MODE_SMOOTH;
            if (op == RM_CMD_SMOOTH_UNION) {  // one parameter: k
                if (ptr + 1 > cap_words) return RM_ERR_TRUNCATED;
                std::memcpy(&r.p[0], words + ptr, 4);
                ptr += 1;
                // the blend radius is one of the magnitudes the float margins of the skipping rules scale with
                const float up = std::nextafterf((float)(1.0 + std::fabs((double)r.p[0])), INFINITY);
                if (!(up <= d.scene_scale)) d.scene_scale = up;  // also takes a NaN
            }
            if (op == RM_CMD_INTERSECTION || op == RM_CMD_SMOOTH_UNION) d.has_extensions = true;
            if (op == RM_CMD_SMOOTH_UNION) d.prunable = false;  // not a lattice operator
            if (depth < 2) return RM_ERR_STACK_UNDERFLOW;
            combine(mode, (double)r.p[0]);
            depth--;
            spilled--;
            r.op = RM_OP(RM_KIND_POP, mode, 0);
            if (mode == RM_MODE_UNION || mode == RM_MODE_SUB) r.op |= (mode == RM_MODE_UNION ? 7u : 8u) << 16;  // RM_OP_FASTCLASS
            d.rec.push_back(r);
        } else {
            return RM_ERR_OPCODE;
        }
    }
    if (!scopes.empty()) return RM_ERR_TRANSFORM;
    if (cmd_count && depth < 1) return RM_ERR_EMPTY_RESULT;
    // Table slots of the primitives the miss tests have to clear.  A program with transforms keeps one cone per bounded
    // primitive (their slots index `bounds`, filled above); otherwise subtracted primitives get none (RM_OP_NOCULL).
    static const bool keep_subtracted = std::getenv("RM_CULL_SUBTRACTED") && std::atoi(std::getenv("RM_CULL_SUBTRACTED")) == 0;  // A/B
    if (keep_subtracted) {
        for (RmRecord& r : d.rec) r.op &= ~(uint32_t)RM_OP_NOCULL;
    }
    if (d.has_xforms) {  // every bounded primitive keeps its cone slot (it indexes `bounds`); a subtracted one gets a cone no ray meets
        for (RmRecord& r : d.rec) {
            const uint32_t kind = RM_OP_KIND(r.op);
            if (kind != RM_KIND_SPHERE && kind != RM_KIND_BOX && kind != RM_KIND_CYLINDER && kind != RM_KIND_PLANE) r.op &= ~(uint32_t)RM_OP_NOCULL;
        }
    } else {
        d.n_sphere = d.n_box = 0u;
        for (RmRecord& r : d.rec) {
            const uint32_t kind = RM_OP_KIND(r.op);
            if (kind == RM_KIND_PLANE) continue;  // no table entry either way; a subtracted Plane (RM_OP_NOCULL) does not veto the tables
            if (kind != RM_KIND_SPHERE && kind != RM_KIND_BOX && kind != RM_KIND_CYLINDER) { r.op &= ~(uint32_t)RM_OP_NOCULL; continue; }
            uint32_t slot = 0u;
            if (!(r.op & RM_OP_NOCULL)) slot = kind == RM_KIND_SPHERE ? d.n_sphere++ : d.n_box++;
            std::memcpy(&r.p[6], &slot, 4);
        }
    }
    for (const RmRecord& r : d.rec) d.n_leaves += RM_OP_KIND(r.op) == RM_KIND_SPHERE || RM_OP_KIND(r.op) == RM_KIND_BOX;
    {   // unit records, see RmDecoded::units
        std::vector<RmUnit> us;
        if (d.prunable && rm_lattice_units(d.rec, &us)) d.unit_mode = RM_UNITS_LATTICE;
        else if (!d.prunable && !d.has_xforms && rm_has_blend(d.rec) && rm_blend_units(d.rec, &us)) d.unit_mode = RM_UNITS_BLEND;
        const float inf = INFINITY;
        for (size_t ui = 0; ui < us.size(); ui++)  // RM_OP_UNIT: which unit guards a record
            for (int k = us[ui].first; k <= us[ui].last; k++) d.rec[(size_t)k].op |= (uint32_t)(ui + 1u) << 8;
        for (const RmUnit& u : us) {
            RmRecord g;
            std::memset(&g, 0, sizeof g);
            g.op = (uint32_t)u.first | ((uint32_t)u.last << 16);  // the records the unit's bit guards (the interpreter's walk over a blending chain's units)
            std::memcpy(&g.p[6], &u.kind, 4);
            if (u.leaf < 0) {
                g.p[3] = inf;
                g.p[4] = -inf;
            } else {
                const RmRecord& q = d.rec[(size_t)u.leaf];
                const uint32_t kind = RM_OP_KIND(q.op);
                const double c1 = std::fabs((double)q.p[0]) + std::fabs((double)q.p[1]) + std::fabs((double)q.p[2]);
                double R, rin;
                if (kind == RM_KIND_SPHERE) {
                    R = std::fmax((double)q.p[3], 0.0);
                    rin = (double)q.p[3];
                } else if (kind == RM_KIND_BOX) {
                    const double h[3] = {q.p[3], q.p[4], q.p[5]};
                    R = std::sqrt(std::pow(std::fmax(h[0], 0.0), 2) + std::pow(std::fmax(h[1], 0.0), 2) + std::pow(std::fmax(h[2], 0.0), 2));
                    rin = std::fmin(h[0], std::fmin(h[1], h[2]));
                    if (!(h[0] >= 0.0 && h[1] >= 0.0 && h[2] >= 0.0)) rin = -(std::fabs(h[0]) + std::fabs(h[1]) + std::fabs(h[2]));  // (also takes a NaN)
                } else {  // cylinder: radius, half height
                    const double r = q.p[3], hh = q.p[4];
                    R = std::sqrt(std::pow(std::fmax(r, 0.0), 2) + std::pow(std::fmax(hh, 0.0), 2));
                    rin = std::fmin(r, hh);
                    if (!(r >= 0.0 && hh >= 0.0)) rin = -(std::fabs(r) + std::fabs(hh));
                }
                g.p[0] = q.p[0]; g.p[1] = q.p[1]; g.p[2] = q.p[2];
                // a parameter that is not a finite number: the formulas above say nothing (max(NaN, 0) = 0 makes a box with a NaN
                // extent an infinite slab, not a small box): no bound at all
                bool finite = std::isfinite(c1);
                for (int k = 3; k < (kind == RM_KIND_SPHERE ? 4 : kind == RM_KIND_BOX ? 6 : 5); k++) finite = finite && std::isfinite((double)q.p[k]);
                if (!finite) { R = 1.0 / 0.0; rin = -1.0 / 0.0; }
                const double Rp = (R + 2.0e-6 * (c1 + R)) * (1.0 + 1.0e-6);
                g.p[3] = std::nextafterf((float)((double)std::nextafterf((float)Rp, INFINITY) * 1.000005), INFINITY);  // a NaN stays one: never far
                const double rd = rin - 2.0e-6 * (c1 + std::fabs(rin)) - 1.0e-30;
                g.p[4] = std::nextafterf((float)rd, -INFINITY);
                if (!finite) { g.p[0] = g.p[1] = g.p[2] = 0.0f; g.p[3] = inf; g.p[4] = -inf; }
                if (u.k_rec >= 0) {
                    const float k = d.rec[(size_t)u.k_rec].p[0];
                    g.p[5] = k > 0.0f ? k : 0.0f;
                    if (g.p[5] > d.unit_kmax) d.unit_kmax = g.p[5];
                }
            }
            d.units.push_back(g);
        }
    }
    if (!d.rec.empty()) {
        const RmRecord& r0 = d.rec[0];
        d.is_chain = (RM_OP_KIND(r0.op) == RM_KIND_SPHERE || RM_OP_KIND(r0.op) == RM_KIND_BOX) && RM_OP_MODE(r0.op) == RM_MODE_PUSH &&
                     (r0.op & RM_OP_SPILL) == 0u;
        for (size_t i = 1; d.is_chain && i < d.rec.size(); i++) d.is_chain = RM_OP_FASTCLASS(d.rec[i].op) >= 1u && RM_OP_FASTCLASS(d.rec[i].op) <= 4u;
        d.is_tree = true;
        for (const RmRecord& r : d.rec) d.is_tree = d.is_tree && RM_OP_FASTCLASS(r.op) != 0u;
        if (d.is_tree && !d.is_chain && d.unit_mode == RM_UNITS_LATTICE && d.rec.size() <= 128u) {  // RmDecoded::tree
            struct Operand { unsigned long long leaves; uint32_t first; };
            std::vector<Operand> stack;
            Operand acc{0ull, 0u};
            bool ok = true;
            for (const RmRecord& r : d.rec) {
                const uint32_t cls = RM_OP_FASTCLASS(r.op), un = RM_OP_UNIT(r.op);
                unsigned long long Lm = 0ull, Rm = 0ull;
                uint32_t info = 0u;
                if (cls <= 6u) {
                    if (un == 0u) { ok = false; break; }  // (cannot be: every sphere and box of a lattice program is a unit)
                    Rm = 1ull << (un - 1u);
                    if (cls >= 5u) {  // pushed
                        if (r.op & RM_OP_SPILL) stack.push_back(acc);
                        acc = Operand{Rm, un - 1u};
                    } else {
                        Lm = acc.leaves;
                        info = 1u | (cls >= 3u ? 4u : 0u) | (acc.first << 8);
                        acc.leaves |= Rm;
                    }
                } else {
                    if (stack.empty()) { ok = false; break; }
                    const Operand a = stack.back();
                    stack.pop_back();
                    Lm = a.leaves; Rm = acc.leaves;
                    info = 2u | (cls == 8u ? 4u : 0u) | (a.first << 8);
                    acc = Operand{a.leaves | acc.leaves, a.first};
                }
                RmRecord t;
                std::memset(&t, 0, sizeof t);
                const uint32_t w[5] = {(uint32_t)Lm, (uint32_t)(Lm >> 32), (uint32_t)Rm, (uint32_t)(Rm >> 32), info};
                std::memcpy(&t.p[0], w, sizeof w);
                d.tree.push_back(t);
            }
            if (!ok || !stack.empty()) d.tree.clear();
        }
    }
    for (double sv : slack) d.smooth_slack = sv > d.smooth_slack || sv != sv ? sv : d.smooth_slack;  // map_scene returns the top; be generous
    // (a Plane the tables would have to clear: they cannot; an Intersection: they ask a ray to clear BOTH operands where
    // clearing one is enough)
    bool sharper = false;
    for (const RmRecord& r : d.rec)
        sharper = sharper || (RM_OP_KIND(r.op) == RM_KIND_PLANE && !(r.op & RM_OP_NOCULL)) || RM_OP_MODE(r.op) == RM_MODE_INTER;
    d.bound_walk = ((d.smooth_slack > 0.0 && d.smooth_slack < 1.0e30) || (sharper && d.smooth_slack == 0.0)) && !d.has_xforms &&
                   d.spill_depth <= 1u && d.scene_scale < 1.0e12f;
    d.n_words = ptr;
    *out = std::move(d);
    return RM_OK;
}

// Decodes a program of the reference wire format (plus the extension opcodes).  A program with Material tags is
// decoded twice: without them (`rec`, what every distance evaluation runs) and with them (`mrec`).
static inline int rm_decode_program(uint32_t cmd_count, const uint32_t* words, uint32_t cap_words,
                                    RmDecoded* out) {
    bool tagged = false;
    if (words)
        for (uint32_t i = 0, q = 0; i < cmd_count && q < cap_words; i++) {
            const uint32_t op = words[q++];
            if (op == RM_CMD_MATERIAL) { tagged = true; break; }
            q += op == RM_CMD_SPHERE || op == RM_CMD_PLANE || op == RM_CMD_ROTATION_PUSH ? 4u : op == RM_CMD_BOX ? 6u
               : op == RM_CMD_CYLINDER ? 5u : op == RM_CMD_TRANSLATION_PUSH ? 3u
               : op == RM_CMD_SMOOTH_UNION || op == RM_CMD_SCALE_PUSH ? 1u : 0u;
        }
    if (!tagged) return rm_decode_core(cmd_count, words, cap_words, false, out);
    RmDecoded with;
    int rc = rm_decode_core(cmd_count, words, cap_words, true, &with);  // validates everything, tags included
    if (rc != RM_OK) return rc;
    std::vector<uint32_t> plain;
    plain.reserve(with.n_words);
    uint32_t n_plain = 0;
    for (uint32_t i = 0, q = 0; i < cmd_count; i++) {
        const uint32_t op = words[q];
        const uint32_t n = 1u + (op == RM_CMD_SPHERE || op == RM_CMD_PLANE || op == RM_CMD_ROTATION_PUSH ? 4u : op == RM_CMD_BOX ? 6u
                               : op == RM_CMD_CYLINDER ? 5u : op == RM_CMD_TRANSLATION_PUSH ? 3u
                               : op == RM_CMD_SMOOTH_UNION || op == RM_CMD_SCALE_PUSH || op == RM_CMD_MATERIAL ? 1u : 0u);
        if (op != RM_CMD_MATERIAL) {
            plain.insert(plain.end(), words + q, words + q + n);
            n_plain++;
        }
        q += n;
    }
    RmDecoded d;
    rc = rm_decode_core(n_plain, plain.data(), (uint32_t)plain.size(), false, &d);
    if (rc != RM_OK) return rc;
    d.n_words = with.n_words;
    d.has_materials = true;
    d.has_extensions = true;
    d.max_material = with.max_material;
    d.mat_spill_depth = with.spill_depth;
    d.mat_xform_depth = with.xform_depth;
    d.mrec = std::move(with.rec);
    *out = std::move(d);
    return RM_OK;
}
