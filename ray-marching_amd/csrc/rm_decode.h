// rm_decode.h -- validation of the reference wire format and decoding into RmRecord[].
// Wire format: csg/builder.rs:26-62 (opcode word followed by f32::to_bits parameters),
// sphere.rs:15-21 (5 words), box.rs:14-20 (7 words), operations/mod.rs:12-18 (1 word,
// post-order).  Host-only, no HIP.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "rm_device.h"

struct RmDecoded {
    std::vector<RmRecord> rec;
    uint32_t n_words = 0;      // words consumed by cmd_count commands
    uint32_t max_depth = 0;    // value-stack depth of the reference machine
    uint32_t spill_depth = 0;  // LDS slots the accumulator machine needs
    uint32_t n_sphere = 0, n_box = 0;  // cone / slab entries of the miss-test tables; RmRecord::p[6] = slot
    uint32_t n_plane = 0;              // unbounded primitives: they veto miss-ray culling
    bool has_extensions = false;       // uses node types the reference does not implement
    double smooth_slack = 0.0;         // sum of k/4 over SmoothUnion operators
    // Far-primitive pruning in specialised kernels (rm_kernel_v5.h "Pruning"): allowed when every node is a
    // 1-Lipschitz leaf or a min/max operator; scene_scale = 1 + max over primitives of |centre|_1 + |size|_1
    bool prunable = true;
    float scene_scale = 1.0f;
};

// Returns RM_OK or a negative rm_status.  `cap_words` is the number of u32 words that
// exist after the cmd_count word (255 for the reference's 1024-byte buffer).
static inline int rm_decode_program(uint32_t cmd_count, const uint32_t* words, uint32_t cap_words,
                                    RmDecoded* out) {
    RmDecoded d;
    if (cmd_count && cap_words && !words) return RM_ERR_NULL;
    // Every command is at least one word, so at most cap_words records can ever be produced.
    d.rec.reserve(cmd_count < cap_words ? cmd_count : cap_words);
    uint32_t ptr = 0, depth = 0, spilled = 0;
    for (uint32_t i = 0; i < cmd_count; i++) {
        if (ptr >= cap_words) return RM_ERR_TRUNCATED;
        uint32_t op = words[ptr++];
        RmRecord r;
        std::memset(&r, 0, sizeof r);
        uint32_t kind = RM_KIND_POP, np = 0;
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
            // slot in the kernels' per-kind miss-test tables: cones for spheres, slabs for boxes and cylinders
            const uint32_t slot = kind == RM_KIND_SPHERE ? d.n_sphere++ : (kind == RM_KIND_PLANE ? 0u : d.n_box++);
            std::memcpy(&r.p[6], &slot, 4);
            // Fuse with a directly following parameter-less binary operator: its rhs is this leaf.
            uint32_t mode = RM_MODE_PUSH;
            if (i + 1 < cmd_count && ptr < cap_words && depth >= 1) {
                if (words[ptr] == RM_CMD_UNION) mode = RM_MODE_UNION;
                else if (words[ptr] == RM_CMD_SUBTRACTION) mode = RM_MODE_SUB;
                else if (words[ptr] == RM_CMD_INTERSECTION) { mode = RM_MODE_INTER; d.has_extensions = true; }
            }
            if (mode != RM_MODE_PUSH) {
                ptr++;  // consume the operator: depth is unchanged (push, then pop 2 push 1)
                i++;
                if (depth + 1 > 32) return RM_ERR_STACK_OVERFLOW;  // the reference machine peaks one higher
                if (depth + 1 > d.max_depth) d.max_depth = depth + 1;
                r.op = RM_OP(kind, mode, 0);
            } else {
                uint32_t spill = depth >= 1 ? 1u : 0u;  // a live accumulator must be saved
                if (spill) { spilled++; if (spilled > d.spill_depth) d.spill_depth = spilled; }
                depth++;
                if (depth > 32) return RM_ERR_STACK_OVERFLOW;
                if (depth > d.max_depth) d.max_depth = depth;
                r.op = RM_OP(kind, RM_MODE_PUSH, spill);
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
                if (r.p[0] > 0.0f) d.smooth_slack += (double)r.p[0] * 0.25;
                else if (!(r.p[0] <= 0.0f)) d.smooth_slack = 1.0 / 0.0;  // NaN k: nothing can be bounded
            }
            if (op == RM_CMD_INTERSECTION || op == RM_CMD_SMOOTH_UNION) d.has_extensions = true;
            if (op == RM_CMD_SMOOTH_UNION) d.prunable = false;  // not a lattice operator
            if (depth < 2) return RM_ERR_STACK_UNDERFLOW;
            depth--;
            spilled--;
            r.op = RM_OP(RM_KIND_POP, mode, 0);
            d.rec.push_back(r);
        } else {
            return RM_ERR_OPCODE;
        }
    }
    if (cmd_count && depth < 1) return RM_ERR_EMPTY_RESULT;
    d.n_words = ptr;
    *out = std::move(d);
    return RM_OK;
}
