// rm_device.h -- structures shared by the host side of librm_hip.so and its kernels.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif

#include "rm_abi.h"

// One decoded command = 32 bytes, so that a wave fetches it with one s_load_dwordx8
// (scalar-cache variant) or two ds_read_b128 (LDS variant).
//
// The reference's stack machine (ray_marching.wgsl:187-203) pushes every command's value
// on a private array.  The program is straight-line postfix, so the stack slot of every
// operand is known when the program is uploaded: the decoder (rm_decode.h) rewrites it
// for an accumulator machine that keeps the top of stack in a VGPR:
//   prim + PUSH  : [spill acc to LDS if one is live]  acc = prim(pos)
//   prim + UNION : acc = min(acc, prim(pos))        <- "prim; Union" fused (rhs is a leaf)
//   prim + SUB   : acc = max(acc, -prim(pos))       <- "prim; Subtraction" fused
//   POP  + UNION : acc = min(pop(), acc)            <- operator whose rhs is a sub-tree
//   POP  + SUB   : acc = max(pop(), -acc)
// The arithmetic performed on each value is exactly the reference's, in the same order;
// only the bookkeeping (where a value lives) changes.
struct RmRecord {
    uint32_t op;  // kind | mode<<3 | spill<<6 | nocull<<7 | (unit + 1)<<8 | fast class<<16
    float p[7];   // sphere: cx cy cz r [r * 1.000005, for specialised kernels]   box: cx cy cz rx ry rz   cylinder: cx cy cz r half_h
                  // plane: nx ny nz h    POP+SMOOTH: k            p[6] (primitives): slot in the miss-test tables
};
static_assert(sizeof(RmRecord) == 32, "record must be 32 bytes");

enum : uint32_t { RM_KIND_POP = 0, RM_KIND_SPHERE = 1, RM_KIND_BOX = 2, RM_KIND_CYLINDER = 3, RM_KIND_PLANE = 4,
                  RM_KIND_XFORM = 5,    // space transformation: mode = RM_XF_*, p[0..3] = parameters (ScalePop: the scale of
                                        // its push), p[6] = level of the position stack
                  RM_KIND_MATERIAL = 6 };  // material program only: the value on top takes the index in p[0] (as bits)
enum : uint32_t { RM_MODE_PUSH = 0, RM_MODE_UNION = 1, RM_MODE_SUB = 2, RM_MODE_INTER = 3, RM_MODE_SMOOTH = 4 };
enum : uint32_t { RM_XF_T_PUSH = 0, RM_XF_T_POP = 1, RM_XF_R_PUSH = 2, RM_XF_R_POP = 3, RM_XF_S_PUSH = 4, RM_XF_S_POP = 5 };
enum : uint32_t { RM_MAX_XFORM_DEPTH = 8 };
#define RM_OP(kind, mode, spill) ((uint32_t)(kind) | ((uint32_t)(mode) << 3) | ((uint32_t)(spill) << 6))
#define RM_OP_KIND(op) ((op) & 7u)
#define RM_OP_MODE(op) (((op) >> 3) & 7u)
enum : uint32_t { RM_OP_SPILL = 1u << 6 };
// A leaf inside the right operand of a Subtraction has no entry in the miss-test tables: max(a, -b) >= a, so a march position
// registers a hit (wgsl:97) only within the margin of a leaf of the LEFT operand, however close the ray comes to b
// (rm_decode.h, rm_kernel_v5.h cull_build_v5)
enum : uint32_t { RM_OP_NOCULL = 1u << 7 };
// Interpreter fast class (bits 16-19; generated code and the v1 kernel ignore it).  The eight record shapes a reference-only
// program (spheres, boxes, Union, Subtraction) decodes into are dispatched with two or three decisions instead of the generic
// kind / spill / mode ladder (rm_interp.h):
//   1 sphere + union   2 box + union   3 sphere + subtraction   4 box + subtraction      leaf fused with its operator, on the accumulator
//   5 sphere pushed    6 box pushed                                                      (the accumulator spills first if RM_OP_SPILL)
//   7 union            8 subtraction                                                     of the popped value and the accumulator
//   0 generic (extension node types)
// A left-deep chain is classes 5 / 6 for record 0 and 1..4 behind it (RmDecoded::is_chain); a program whose records all have a
// class is a TREE (RmDecoded::is_tree).
#define RM_OP_FASTCLASS(op) (((op) >> 16) & 15u)
// Unit of wave-level culling the record belongs to, plus one (bits 8-15; 0: none, the record is always executed): a lattice
// program's bounded leaves, every record of a blending chain's units (rm_units.h).  The interpreter's general loop skips a
// record whose unit's bit is clear in the wave's mask; generated code has the bits as constants and ignores the field.
#define RM_OP_UNIT(op) (((op) >> 8) & 0xFFu)

// reference opcodes (csg/builder.rs:1-24)
enum : uint32_t { RM_CMD_SPHERE = 0, RM_CMD_BOX = 1, RM_CMD_UNION = 100, RM_CMD_SUBTRACTION = 101 };
// Extension opcodes (not implemented by the reference; DESIGN.md "Extension node types").  Plane and
// Intersection use the slots the reference reserves by comment (builder.rs:8,14); Cylinder and
// SmoothUnion (BASELINE.json configs 2-3) stay clear of every reserved slot (2, 102, 200-205).
enum : uint32_t { RM_CMD_PLANE = 2, RM_CMD_CYLINDER = 10, RM_CMD_INTERSECTION = 102, RM_CMD_SMOOTH_UNION = 110 };
// Space transformations: the slots the reference reserves by comment (builder.rs:16-23).  Push(params), one child, Pop.
enum : uint32_t { RM_CMD_TRANSLATION_PUSH = 200, RM_CMD_TRANSLATION_POP = 201, RM_CMD_ROTATION_PUSH = 202,
                  RM_CMD_ROTATION_POP = 203, RM_CMD_SCALE_PUSH = 204, RM_CMD_SCALE_POP = 205 };

// Material tag (extension; the reference lists a material system as future work, README.md:11): unary postfix,
// one u32 parameter (the index, not f32 bits).  Semantics: oracle/rm_oracle.c map_scene_impl.
enum : uint32_t { RM_CMD_MATERIAL = 300, RM_MAX_MATERIALS = 256 };

// Units of wave-level culling (rm_units.h, rm_kernel_v5.h): kind of a unit record (its p[6], as an integer), and what
// RmLaunch::unit_mode says about the program
enum : uint32_t { RM_UNIT_START = 0, RM_UNIT_UM = 1, RM_UNIT_SUB = 2, RM_UNIT_INTER = 3, RM_UNIT_OPAQUE = 4, RM_UNIT_LEAF = 5 };
enum : uint32_t { RM_UNITS_NONE = 0, RM_UNITS_LATTICE = 1, RM_UNITS_BLEND = 2 };

// Tree programs under the interpreter's masked loop (RmDecoded::tree, rm_kernel_v5.h tree_keep): what ONE record does given the
// wave's unit mask `need` -- L, R: the units of its operands, info: RmDecoded::tree p[4].  Shared with the CPU model of the loop
// (tests/cpp/tree_keep_model.cpp).
#if defined(__HIPCC__)
#define RM_HD __device__ __forceinline__
#else
#define RM_HD static inline
#endif
// a Subtraction whose left operand has no needed leaf while its right operand has one: forces rm_tree_forced_unit back into the mask
RM_HD bool rm_tree_forces(unsigned long long L, unsigned long long R, uint32_t info, unsigned long long need) {
    return (info & 4u) != 0u && (L & need) == 0ull && (R & need) != 0ull;
}
RM_HD uint32_t rm_tree_forced_unit(uint32_t info) { return (info >> 8) & 63u; }
// with the forced units in `need`: is the record executed ...
RM_HD bool rm_tree_keeps(unsigned long long L, unsigned long long R, uint32_t info, unsigned long long need) {
    return (R & need) != 0ull && ((info & 3u) != 2u || (L & need) != 0ull);
}
// ... and does a fused leaf push its value because there is nothing to combine it with
RM_HD bool rm_tree_pushes(unsigned long long L, uint32_t info, unsigned long long need) { return (info & 3u) == 1u && (L & need) == 0ull; }

struct RmLaunch {
    const RmRecord* prog;      // decoded program, device memory
    uint32_t n_rec;            // == cmd_count of the reference program
    uint32_t n_grp;            // unit records that follow the n_rec program records in `prog` (wave-level culling: one bounded
                               // stand-in per unit, RmDecoded::units); staged in LDS with the program
    uint32_t n_tree;           // 0, or n_rec: the per-record operand masks (RmDecoded::tree) follow the unit records; interpreter, masked tree loop
    uint32_t unit_mode;        // RM_UNITS_* (rm_units.h): 0 none, 1 lattice program (threshold rule), 2 blending chain
    float unit_kmax;           // the largest blend radius of a unit (the chain of blends never falls further below its smallest leaf)
    uint32_t spill_depth;      // LDS slots per lane this program needs: value stack, then 3 per transform level
    uint32_t wave_dwords;      // LDS dwords of a march wave's private buffers (rm_kernel_v5.h V5_WAVE_DWORDS, or less: a generated kernel whose
                               // taps run in one pass and that carries no materials has no use for the partial normals)
    uint32_t value_spill_depth; // the value-stack part of spill_depth (saved positions start at this slot)
    const float4* bounds;      // nullptr, or one world-space bounding sphere (centre, radius) per bounded primitive:
                               // programs with transforms (their miss tests use these instead of the parameters)
    uint32_t n_cull;           // entries of the miss-ray culling table (== n_rec when culling is on)
    uint32_t flags;            // bit 0: miss-ray culling enabled; bit 2: chain program (interpreter kernels: map_scene_chain); bit 4: tree program (map_scene_tree); bit 3: ... with the wave's unit mask (masked chain loop, map_scene_tree_masked); bit 5: miss test on lower bounds (RmDecoded::bound_walk); bits 8-14: diagnostics (RM_PRE_NEED_MAX)
    uint32_t n_cone, n_slab;   // v5 miss-test tables: spheres / (boxes + cylinders) of the program
    float smooth_slack;        // sum of k/4 over SmoothUnion operators: how far they can lower the tree value
    float scene_scale;         // 1 + max |centre|_1 + |size|_1 over the primitives (RmDecoded::scene_scale)
    float min_dist, max_dist;  // RayMarchLimits (wgsl:78-82)
    uint32_t max_iter;
    uint32_t W, H, row0, rows;
    // Interleaved strips (multi-GPU tiling): when strip_rows != 0 the `rows` output rows are the
    // concatenation of strips strip_first, strip_first + strip_stride, ... of strip_rows rows each.
    uint32_t strip_rows, strip_first, strip_stride;
    float* out;                // rows*W pixels per frame: 16 B each (RM_FORMAT_RGBA32F) or 4 B each (8-bit formats)
    uint32_t out_format;       // enum rm_format
    unsigned long long* stats; // diagnostics (RM_OPT_WAVE_STATS): 4 x u64 per wave, or nullptr
    const uint32_t* order;     // nullptr: tiles in raster order; else dispatch slot -> tile id, per frame
    // Materials (extension): n_mrec != 0 when the program carries Material tags.  mprog is the SAME program decoded
    // with the tags kept (prog drops them: distances do not depend on them); a hit evaluates it once, at the hit
    // position, on a (distance, index) stack of mat_value_depth spill slots each; materials[index].xyz = albedo.
    const RmRecord* mprog;
    uint32_t n_mrec, mat_value_depth;
    const float4* materials;
    const rm_uniforms* frames; // nullptr: use `u`; else frames[blockIdx.z]
    rm_uniforms u;
};

// Framebuffer row (0 = top) of output row `ry` of this launch.
#if defined(__HIPCC__)
__device__ __forceinline__
#else
static inline
#endif
uint32_t rm_global_row(const RmLaunch& L, uint32_t ry) {
    if (L.strip_rows == 0u) return L.row0 + ry;
    return ((ry / L.strip_rows) * L.strip_stride + L.strip_first) * L.strip_rows + ry % L.strip_rows;
}
