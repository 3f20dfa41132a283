// rm_groups.h -- which leaves of a decoded program the LOCAL skipping rule applies to, and how they pair up.
// Host only; shared by the decoder (rm_decode.h: bounding spheres of the pairs) and the code generator (rm_jit.h: the
// tests), so that both see the same pairs for the same structure.
//
// The local rule (programs that blend with SmoothUnion; exact without any bound on the scene value):
//     smin_k(acc, v) = min(acc, v) - h^2 k / 4,  h = max(k - |acc - v|, 0) / k      (oracle/rm_oracle.c, opcode 110)
// returns acc, bit for bit, whenever v >= acc + k: h is 0 and min(acc, v) - 0 = acc.  Likewise min(acc, v) = acc whenever
// v > acc.  So a leaf whose value is provably at least k above the accumulator it is about to be blended into, for every
// live lane of the wave, need not be evaluated -- and neither does the operator.  "Provably": from a lower bound of the leaf
// value (the distance to a bounding sphere of the leaf, or of a PAIR of leaves that are blended into the same accumulator one
// after the other: if the pair's sphere is far enough, the first member leaves the accumulator alone, so the second meets
// the same accumulator and is skipped by the same test).
#pragma once
#include <cmath>
#include <vector>

#include "rm_device.h"

// How the value of leaf record i is consumed.
struct RmLeafUse {
    bool local = false;  // the local rule applies: the leaf is the right operand of a Union (fused) or of the SmoothUnion
                         // record that directly follows it
    int k_rec = -1;      // index of that SmoothUnion record (its p[0] is k), -1 for a fused Union (k = 0)
    int next = 0;        // index of the first record after the leaf and its operator
};

inline RmLeafUse rm_leaf_use(const std::vector<RmRecord>& rec, size_t i) {
    RmLeafUse u;
    const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
    u.next = (int)i + 1;
    if (kind != RM_KIND_SPHERE && kind != RM_KIND_BOX) return u;
    if (mode == RM_MODE_UNION) {
        u.local = true;
        return u;
    }
    // "leaf; SmoothUnion": the decoder does not fuse an operator that carries a parameter, so the leaf is pushed (the live
    // accumulator spills) and the very next record pops it back as the operator's left operand
    if (mode == RM_MODE_PUSH && (rec[i].op & RM_OP_SPILL) && i + 1 < rec.size() && RM_OP_KIND(rec[i + 1].op) == RM_KIND_POP &&
        RM_OP_MODE(rec[i + 1].op) == RM_MODE_SMOOTH) {
        const float k = rec[i + 1].p[0];
        u.next = (int)i + 2;
        if (k == k && std::fabs(k) < 1.0e30f) {  // a NaN or infinite k blends everything with everything: no rule
            u.local = true;
            u.k_rec = (int)i + 1;
        }
    }
    return u;
}

// Pairs (first, second) of leaf records the local rule tests together: both local, the second directly behind the first's
// operator (so its left operand IS the first's result).  Greedy in program order; a local leaf without a partner keeps a
// test of its own.  Depends on the structure only -- and on whether a k is finite, which is part of the structure key.
inline std::vector<std::pair<int, int>> rm_blend_pairs(const std::vector<RmRecord>& rec) {
    std::vector<std::pair<int, int>> pairs;
    for (size_t i = 0; i < rec.size();) {
        const RmLeafUse a = rm_leaf_use(rec, i);
        if (a.local && (size_t)a.next < rec.size()) {
            const RmLeafUse b = rm_leaf_use(rec, (size_t)a.next);
            if (b.local) {
                pairs.emplace_back((int)i, a.next);
                i = (size_t)b.next;
                continue;
            }
        }
        i = (size_t)a.next;
    }
    return pairs;
}

// Does the program blend at all (a SmoothUnion with k > 0 somewhere)?  Only then does the local rule replace the lattice
// rule (rm_kernel_v5.h "Pruning"), which needs min / max operators throughout.
inline bool rm_has_blend(const std::vector<RmRecord>& rec) {
    for (const RmRecord& r : rec)
        if (RM_OP_KIND(r.op) == RM_KIND_POP && RM_OP_MODE(r.op) == RM_MODE_SMOOTH) return true;
    return false;
}
