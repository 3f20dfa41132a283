// rm_units.h -- the UNITS of a decoded program that wave-level culling decides about (rm_kernel_v5.h "Wave-level culling"),
// one bit each in the 64-bit mask a wave computes before every evaluation of the scene.  Host only; shared by the decoder
// (rm_decode.h: the unit records -- bounding data -- that follow the program in device memory and LDS) and the code generator
// (rm_jit.h: which records a bit guards), so that both see the same units for the same structure.
//
// Two kinds of programs have units:
//  * LATTICE programs (min / max / subtraction over spheres, boxes, cylinders: RmDecoded::prunable): every such leaf is a unit
//    (RM_UNIT_LEAF), skipped when its value exceeds the threshold `thr` for every live ray of the wave -- the leaf is then
//    replaced by +inf, which the operators ignore (rm_kernel_v5.h "Pruning").
//  * Programs that BLEND (SmoothUnion) whose top level is a CHAIN: one value -- the accumulator -- flows from record 0 to the
//    end, and every unit takes it to its next value:
//        RM_UNIT_START   record 0, the leaf that starts the chain
//        RM_UNIT_UM      a leaf and the Union (fused) or SmoothUnion (the record behind it) that blends it in:
//                        smin_k(acc, v) = acc, bit for bit, when v >= acc + k  (h = 0: min(acc, v) - 0);  = v when v <= acc - k
//        RM_UNIT_SUB     a leaf fused with a Subtraction:   max(acc, -v) = acc when v + acc >= 0
//        RM_UNIT_INTER   a leaf fused with an Intersection: max(acc, v) = acc when v <= acc
//        RM_UNIT_OPAQUE  anything else that takes the accumulator to its next value (a Plane, a sub-tree popped into the chain
//                        by its operator): never skipped on its own account, and nothing is known about the accumulator behind it
//    Units in front of a RESTART -- a Union / SmoothUnion whose leaf is at least k below the accumulator for every live ray --
//    are dead: the operator returns the leaf's value whatever the accumulator was.
#pragma once
#include <cmath>
#include <vector>

#include "rm_device.h"

// (the kinds RM_UNIT_* and the modes RM_UNITS_* are in rm_device.h: the kernels read them)
constexpr size_t kMaxUnits = 64;

struct RmUnit {
    uint32_t kind;
    int first, last;  // the records the unit's bit guards: [first, last]
    int leaf;         // the leaf record whose bounding sphere stands for the unit, -1 for an opaque unit
    int k_rec;        // RM_UNIT_UM: the SmoothUnion record (its p[0] is k), -1 for a fused Union
};

inline bool rm_bounded_leaf(uint32_t kind) { return kind == RM_KIND_SPHERE || kind == RM_KIND_BOX || kind == RM_KIND_CYLINDER; }

// Lattice programs: every bounded leaf, in program order.  False when there is none or more than the mask has bits for.
inline bool rm_lattice_units(const std::vector<RmRecord>& rec, std::vector<RmUnit>* out) {
    std::vector<RmUnit> u;
    for (size_t i = 0; i < rec.size(); i++)
        if (rm_bounded_leaf(RM_OP_KIND(rec[i].op))) u.push_back({RM_UNIT_LEAF, (int)i, (int)i, (int)i, -1});
    if (u.empty() || u.size() > kMaxUnits) return false;
    *out = std::move(u);
    return true;
}

inline bool rm_has_blend(const std::vector<RmRecord>& rec) {
    for (const RmRecord& r : rec)
        if (RM_OP_KIND(r.op) == RM_KIND_POP && RM_OP_MODE(r.op) == RM_MODE_SMOOTH) return true;
    return false;
}
inline bool rm_finite_k(float k) { return k == k && std::fabs(k) < 1.0e30f; }

// Programs that blend: the top-level chain as units.  False when the top level is not a chain (record 0 must be a bounded
// leaf that starts it; the program must end with exactly the accumulator), when the program has transforms or materials in
// its records, or when there are fewer than two or more than 64 units.
inline bool rm_blend_units(const std::vector<RmRecord>& rec, std::vector<RmUnit>* out) {
    if (rec.empty() || RM_OP_MODE(rec[0].op) != RM_MODE_PUSH || (rec[0].op & RM_OP_SPILL) || !rm_bounded_leaf(RM_OP_KIND(rec[0].op))) return false;
    for (const RmRecord& r : rec)
        if (RM_OP_KIND(r.op) == RM_KIND_XFORM || RM_OP_KIND(r.op) == RM_KIND_MATERIAL) return false;
    std::vector<RmUnit> u;
    u.push_back({RM_UNIT_START, 0, 0, 0, -1});
    for (size_t i = 1; i < rec.size();) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        if (kind == RM_KIND_POP) return false;  // (cannot be: every unit below consumes its own operators)
        if (mode != RM_MODE_PUSH) {  // a leaf fused with its operator
            const uint32_t uk = !rm_bounded_leaf(kind) ? RM_UNIT_OPAQUE : mode == RM_MODE_UNION ? RM_UNIT_UM : mode == RM_MODE_SUB ? RM_UNIT_SUB
                              : mode == RM_MODE_INTER ? RM_UNIT_INTER : RM_UNIT_OPAQUE;
            u.push_back({uk, (int)i, (int)i, uk == RM_UNIT_OPAQUE ? -1 : (int)i, -1});
            i++;
            continue;
        }
        // a pushed leaf (the accumulator spills); "leaf; SmoothUnion" is the unit UM, anything longer a sub-tree
        if (rm_bounded_leaf(kind) && i + 1 < rec.size() && RM_OP_KIND(rec[i + 1].op) == RM_KIND_POP && RM_OP_MODE(rec[i + 1].op) == RM_MODE_SMOOTH) {
            // a NaN or infinite k blends everything with everything: no rule applies to this unit
            if (rm_finite_k(rec[i + 1].p[0])) u.push_back({RM_UNIT_UM, (int)i, (int)i + 1, (int)i, (int)i + 1});
            else u.push_back({RM_UNIT_OPAQUE, (int)i, (int)i + 1, -1, -1});
            i += 2;
            continue;
        }
        int depth = 1;  // stack depth counted from the accumulator
        size_t j = i;
        for (; j < rec.size(); j++) {
            const uint32_t kj = RM_OP_KIND(rec[j].op), mj = RM_OP_MODE(rec[j].op);
            if (kj == RM_KIND_POP) depth--;
            else if (mj == RM_MODE_PUSH) depth++;
            if (depth == 1) break;
        }
        if (j == rec.size()) return false;  // the program ends with more than one value on the stack
        u.push_back({RM_UNIT_OPAQUE, (int)i, (int)j, -1, -1});
        i = j + 1;
    }
    if (u.size() < 2 || u.size() > kMaxUnits) return false;
    *out = std::move(u);
    return true;
}
