// rm_decode.h -- validation of the reference wire format and decoding into RmRecord[].
// Wire format: csg/builder.rs:26-62 (opcode word followed by f32::to_bits parameters),
// sphere.rs:15-21 (5 words), box.rs:14-20 (7 words), operations/mod.rs:12-18 (1 word,
// post-order).  Host-only, no HIP.
#pragma once
#include <cstring>
#include <vector>

#include "rm_device.h"

struct RmDecoded {
    std::vector<RmRecord> rec;
    uint32_t n_words = 0;      // words consumed by cmd_count commands
    uint32_t max_depth = 0;    // value-stack depth of the reference machine
    uint32_t spill_depth = 0;  // LDS slots the accumulator machine needs
    uint32_t n_sphere = 0, n_box = 0;  // primitives per kind; RmRecord::p[6] holds each one's slot within its kind
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
        if (op == RM_CMD_SPHERE || op == RM_CMD_BOX) {
            uint32_t np = op == RM_CMD_SPHERE ? 4u : 6u;
            if (ptr + np > cap_words) return RM_ERR_TRUNCATED;
            std::memcpy(r.p, words + ptr, np * 4);
            ptr += np;
            uint32_t kind = op == RM_CMD_SPHERE ? RM_KIND_SPHERE : RM_KIND_BOX;
            const uint32_t slot = op == RM_CMD_SPHERE ? d.n_sphere++ : d.n_box++;
            std::memcpy(&r.p[6], &slot, 4);  // slot in the kernels' per-kind miss-test tables
            // Fuse with a directly following binary operator: its rhs is this leaf.
            uint32_t mode = RM_MODE_PUSH;
            if (i + 1 < cmd_count && ptr < cap_words && depth >= 1) {
                if (words[ptr] == RM_CMD_UNION) mode = RM_MODE_UNION;
                else if (words[ptr] == RM_CMD_SUBTRACTION) mode = RM_MODE_SUB;
            }
            if (mode != RM_MODE_PUSH) {
                ptr++;  // consume the operator: depth is unchanged (push then pop 2 push 1)
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
            // a fused record stands for two reference commands; keep n_rec == records
        } else if (op == RM_CMD_UNION || op == RM_CMD_SUBTRACTION) {
            if (depth < 2) return RM_ERR_STACK_UNDERFLOW;
            depth--;
            spilled--;
            r.op = RM_OP(RM_KIND_POP, op == RM_CMD_UNION ? RM_MODE_UNION : RM_MODE_SUB, 0);
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
