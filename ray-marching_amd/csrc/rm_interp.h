// rm_interp.h -- the interpreter half of the march kernels: exact SDF leaf functions, the accumulator
// machine that executes a decoded program record by record (map_scene, wgsl:187-203), the material walk,
// and the arithmetic self-tests.  Device code only (gfx950, wave64).
//
// Bit-exactness: every value goes through exactly the operation sequence of the arithmetic
// contract; only the correctly-rounded sqrt is computed by a shorter (still exact) sequence:
// see sqrt_rn_fast.
#pragma once
#include "rm_kernels.h"

namespace rmk {

// v_min_f32 / v_max_f32 issued directly: the builtins make the compiler insert a canonicalising
// v_max_f32 x,x in front of every operand it cannot prove quiet (values that went through a
// phi or LDS), which costs two extra VALU per CSG operator.  Semantics are identical for every
// non-signalling input (IEEE mode is on; -0 < +0; a quiet NaN operand loses); a SIGNALLING NaN
// operand would yield a quiet NaN instead of the other operand, but these are only ever applied to
// results of arithmetic instructions (SDF values, accumulators), which are never signalling
// (tests/test_gpu_arithmetic.py checks both facts).
RM_DEV float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RM_DEV float vmax_negb(float a, float b) {  // max(a, -b)
    float r;
    asm("v_max_f32 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Correctly rounded sqrt in v_rsq_f32 + 4 VALU:  y = rsq(x); g = x y; s = g + (x - g g) (y / 2), the residual and
// the final sum each one FMA.  v_rsq_f32 is a 1-ulp approximation, g is within ~1.5 ulp of sqrt(x), the residual
// x - g g is exact, and the rounding of the final FMA lands on the correctly rounded root for EVERY binary32 x in
// [2^-102, FLT_MAX]: established by exhaustion on MI355X (tools/probe_sqrt_range.hip; rm_selftest_sqrt repeats it
// on whatever GPU the library runs on), not by a proof.  It replaces v_sqrt_f32 + the 8-instruction neighbour test
// of LLVM's expansion (which this file used before: 9 VALU + v_sqrt_f32).  Outside that range -- 0, inf, NaN,
// denormals -- the sequence returns NaN or garbage: the caller tracks the range of all arguments of one map_scene
// evaluation in a SqrtGuard and re-evaluates with __builtin_sqrtf when any lane saw an argument outside
// [2^-96, FLT_MAX].  Zero is a legitimate and frequent argument for boxes (inside the slab), so ZERO_OK clamps the
// v_rsq input (g = 0 * y = 0, residual 0, result +0) and keeps zero out of the guard; spheres hit zero only at
// their exact centre and take the slow path there.
struct SqrtGuard {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;  // min / max over the (biased) bit patterns of all sqrt arguments
    RM_DEV bool bad() const { return lo < kLoBits || hi > 0x7F7FFFFFu; }
    // any lane of the wave: two compares whose wave masks are combined on the scalar side (a ballot of the OR would go
    // through a VGPR and back, see spec_any_near in rm_kernel_v5.h)
    RM_DEV bool any_bad() const {
        return (__builtin_amdgcn_ballot_w64(lo < kLoBits) | __builtin_amdgcn_ballot_w64(hi > 0x7F7FFFFFu)) != 0ull;
    }
    static constexpr uint32_t kLoBits = 0x0F800000u - 1u;  // bits(2^-96) - 1
};
RM_DEV float vmax(float a, float b) {  // direct v_max_f32, see vmin
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool ZERO_OK>
RM_DEV float sqrt_rn_fast(float x) {
    const float y = __builtin_amdgcn_rsqf(ZERO_OK ? vmax(x, __uint_as_float(0x0F800000u)) : x);
    const float g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}

template <bool FAST, bool ZERO_OK = false>
RM_DEV float sqrt_sel(float x, SqrtGuard& guard) {
    if constexpr (FAST) {
        // x >= +0 or NaN here (sums of squares).  ZERO_OK: (bits - 1) wraps 0 to the top, so only 0 < x < 2^-96 is low.
        guard.lo = min(guard.lo, ZERO_OK ? __float_as_uint(x) - 1u : __float_as_uint(x));
        guard.hi = max(guard.hi, __float_as_uint(x));
        return sqrt_rn_fast<ZERO_OK>(x);
    } else {
        return __builtin_sqrtf(x);
    }
}

template <bool FAST>
RM_DEV float sdf_sphere_t(float px, float py, float pz, const float (&p)[7], SqrtGuard& tiny) {
    const float dx = px - p[0], dy = py - p[1], dz = pz - p[2];
    return sqrt_sel<FAST>((dx * dx + dy * dy) + dz * dz, tiny) - p[3];
}
template <bool FAST>
RM_DEV float sdf_box_t(float px, float py, float pz, const float (&p)[7], SqrtGuard& tiny) {
    const float qx = __builtin_fabsf(px - p[0]) - p[3];
    const float qy = __builtin_fabsf(py - p[1]) - p[4];
    const float qz = __builtin_fabsf(pz - p[2]) - p[5];
    const float mx = fmax_(qx, 0.0f), my = fmax_(qy, 0.0f), mz = fmax_(qz, 0.0f);
    return sqrt_sel<FAST, true>((mx * mx + my * my) + mz * mz, tiny) + fmin_(fmax_(qx, fmax_(qy, qz)), 0.0f);
}

template <bool FAST>
RM_DEV float sdf_cylinder_t(float px, float py, float pz, const float (&p)[7], SqrtGuard& tiny) {
    // extension: capped cylinder along y.  p = cx cy cz radius half_height
    const float dx = px - p[0], dz = pz - p[2];
    const float qx = sqrt_sel<FAST, true>(dx * dx + dz * dz, tiny) - p[3];
    const float qy = __builtin_fabsf(py - p[1]) - p[4];
    const float mx = fmax_(qx, 0.0f), my = fmax_(qy, 0.0f);
    return fmin_(fmax_(qx, qy), 0.0f) + sqrt_sel<FAST, true>(mx * mx + my * my, tiny);
}

// Space transformations (extension; semantics: oracle/rm_oracle.c map_scene, opcodes 200-205).  Every product and
// difference is one binary32 operation, in the oracle's order.
RM_DEV void xf_rotate_conj(float w, float ax, float ay, float az, float& x, float& y, float& z) {
    const float cx = y * az - z * ay, cy = z * ax - x * az, cz = x * ay - y * ax;  // p x a = (-a) x p
    const float tx = 2.0f * cx, ty = 2.0f * cy, tz = 2.0f * cz;
    const float ux = ty * az - tz * ay, uy = tz * ax - tx * az, uz = tx * ay - ty * ax;  // t x a
    x = (x + w * tx) + ux;
    y = (y + w * ty) + uy;
    z = (z + w * tz) + uz;
}

// One decoded command applied to the R positions of a lane.
// EXT = false compiles the reference's four node types only (the lean, measured path); EXT = true
// adds the extension node types.  Which one runs is decided per program on the host.
template <int R, bool FAST, bool EXT = false>
RM_DEV void exec_command(uint32_t op, const float (&p)[7], float (&qx)[R], float (&qy)[R], float (&qz)[R], float (&acc)[R],
                         float* spill, uint32_t& sp, SqrtGuard& tiny, uint32_t xf_base = 0u) {
    // The opcode is wave-uniform; for the LDS policy it arrives in a VGPR and is made scalar HERE,
    // at its first use, not where the (prefetched) record was loaded: otherwise the wave would
    // wait for the NEXT record's LDS read before starting the current record's arithmetic.
    op = __builtin_amdgcn_readfirstlane(op);
    // The records of a left-deep chain (RM_OP_FASTCLASS): leaf, then min / max(., -leaf) into the accumulator; no stack slot
    // is read or written, nothing merges with the generic path below (whose value copies at the joins of its kind / spill /
    // mode ladder cost ~6 v_mov, ~8 branches and ~20 scalar instructions per record).
    const uint32_t cls = RM_OP_FASTCLASS(op);
    if (cls != 0u && cls <= 4u) {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const float b = (cls & 1u) ? sdf_sphere_t<FAST>(qx[k], qy[k], qz[k], p, tiny) : sdf_box_t<FAST>(qx[k], qy[k], qz[k], p, tiny);
            acc[k] = cls >= 3u ? vmax_negb(acc[k], b) : vmin(acc[k], b);  // wgsl:248-252 / :242-246
        }
        return;
    }
    const uint32_t kind = RM_OP_KIND(op), mode = RM_OP_MODE(op);
    if constexpr (EXT) {
        if (kind == RM_KIND_XFORM) {  // extension: the evaluation position changes; saved positions live in LDS
            float* save = spill + (size_t)(xf_base + 3u * __float_as_uint(p[6])) * R * 64u;
#pragma unroll
            for (int k = 0; k < R; k++) {
                if ((mode & 1u) == 0u) {  // push
                    save[(0 * R + k) * 64] = qx[k]; save[(1 * R + k) * 64] = qy[k]; save[(2 * R + k) * 64] = qz[k];
                    if (mode == RM_XF_T_PUSH) { qx[k] = qx[k] - p[0]; qy[k] = qy[k] - p[1]; qz[k] = qz[k] - p[2]; }
                    else if (mode == RM_XF_R_PUSH) xf_rotate_conj(p[0], p[1], p[2], p[3], qx[k], qy[k], qz[k]);
                    else { qx[k] = qx[k] / p[0]; qy[k] = qy[k] / p[0]; qz[k] = qz[k] / p[0]; }
                } else {  // pop
                    qx[k] = save[(0 * R + k) * 64]; qy[k] = save[(1 * R + k) * 64]; qz[k] = save[(2 * R + k) * 64];
                    if (mode == RM_XF_S_POP) acc[k] = acc[k] * p[0];
                }
            }
            return;
        }
    }
    float a[R], b[R];
    if (kind == RM_KIND_POP) {
        --sp;
#pragma unroll
        for (int k = 0; k < R; k++) {
            b[k] = acc[k];
            a[k] = spill[(sp * R + k) * 64u];
        }
    } else {
        if (kind == RM_KIND_SPHERE) {
#pragma unroll
            for (int k = 0; k < R; k++) b[k] = sdf_sphere_t<FAST>(qx[k], qy[k], qz[k], p, tiny);
        } else if (!EXT || kind == RM_KIND_BOX) {
#pragma unroll
            for (int k = 0; k < R; k++) b[k] = sdf_box_t<FAST>(qx[k], qy[k], qz[k], p, tiny);
        } else if (kind == RM_KIND_CYLINDER) {  // extension
#pragma unroll
            for (int k = 0; k < R; k++) b[k] = sdf_cylinder_t<FAST>(qx[k], qy[k], qz[k], p, tiny);
        } else {  // RM_KIND_PLANE, extension: dot(pos, n) + h
#pragma unroll
            for (int k = 0; k < R; k++) b[k] = ((qx[k] * p[0] + qy[k] * p[1]) + qz[k] * p[2]) + p[3];
        }
        if (op & RM_OP_SPILL) {
#pragma unroll
            for (int k = 0; k < R; k++) spill[(sp * R + k) * 64u] = acc[k];
            ++sp;
        }
#pragma unroll
        for (int k = 0; k < R; k++) a[k] = acc[k];
    }
    if (mode == RM_MODE_PUSH) {
#pragma unroll
        for (int k = 0; k < R; k++) acc[k] = b[k];
    } else if (mode == RM_MODE_UNION) {
#pragma unroll
        for (int k = 0; k < R; k++) acc[k] = vmin(a[k], b[k]);  // wgsl:242-246
    } else if (!EXT || mode == RM_MODE_SUB) {
#pragma unroll
        for (int k = 0; k < R; k++) acc[k] = vmax_negb(a[k], b[k]);  // wgsl:248-252
    } else if (mode == RM_MODE_INTER) {  // extension: max(a, b)
#pragma unroll
        for (int k = 0; k < R; k++) acc[k] = fmax_(a[k], b[k]);
    } else {  // RM_MODE_SMOOTH, extension: min(a,b) - h*h*k/4, h = max(k - |a-b|, 0)/k; k <= 0: plain min
        // (k as a scalar: every lane fetched the same record, and a branch on a vector register is divergent control flow)
        const float kk = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p[0])));
#pragma unroll
        for (int k = 0; k < R; k++) {
            float v = fmin_(a[k], b[k]);
            if (kk > 0.0f) {
                const float h = fmax_(kk - __builtin_fabsf(a[k] - b[k]), 0.0f) / kk;
                v = v - ((h * h) * kk) * 0.25f;
            }
            acc[k] = v;
        }
    }
}

// map_scene (wgsl:187-203) for R positions per lane.  Commands are fetched one ahead of their
// use (two buffers, loop unrolled by two) so that the fetch latency hides behind the VALU work
// of the previous command.
template <int R, bool FAST, class Prog, bool EXT = false>
RM_DEV void map_scene_multi(const Prog& prog, uint32_t n_rec, float* spill, float max_dist, const float (&qx_in)[R],
                            const float (&qy_in)[R], const float (&qz_in)[R], float (&out)[R], SqrtGuard& tiny,
                            uint32_t xf_base = 0u) {
    float qx[R], qy[R], qz[R];  // transform commands (EXT) change the evaluation position
#pragma unroll
    for (int k = 0; k < R; k++) { qx[k] = qx_in[k]; qy[k] = qy_in[k]; qz[k] = qz_in[k]; }
    if (n_rec == 0u) {  // wgsl:189-191
#pragma unroll
        for (int k = 0; k < R; k++) out[k] = max_dist;
        return;
    }
    float acc[R];
#pragma unroll
    for (int k = 0; k < R; k++) acc[k] = 0.0f;
    uint32_t sp = 0, c = 0;
#ifdef RM_SIMPLE_LOOP
    for (; c < n_rec; c++) {
        uint32_t op0;
        float p0[7];
        prog.load(c, op0, p0);
        exec_command<R, FAST, EXT>(op0, p0, qx, qy, qz, acc, spill, sp, tiny, xf_base);
    }
#else
    uint32_t op0, op1;
    float p0[7], p1[7];
    prog.load(0u, op0, p0);
    for (;;) {
        prog.load(c + 1u < n_rec ? c + 1u : c, op1, p1);
        exec_command<R, FAST, EXT>(op0, p0, qx, qy, qz, acc, spill, sp, tiny, xf_base);
        if (++c == n_rec) break;
        prog.load(c + 1u < n_rec ? c + 1u : c, op0, p0);
        exec_command<R, FAST, EXT>(op1, p1, qx, qy, qz, acc, spill, sp, tiny, xf_base);
        if (++c == n_rec) break;
    }
#endif
#pragma unroll
    for (int k = 0; k < R; k++) out[k] = acc[k];
}

// map_scene for a program that BLENDS along a top-level chain (rm_units.h RM_UNITS_BLEND), under the wave's unit mask (rm_kernel_v5.h
// "Wave-level culling": skip / restart / poison rules).  A unit takes the accumulator to its next value through one or more records
// -- "leaf fused with its operator", "leaf; SmoothUnion", an opaque stretch --; a unit whose bit is clear leaves the accumulator alone,
// and the chain starts from +inf (min(+inf, v) = v and smin_k(+inf, v) = v - 0, bit for bit: a restart).  The loop walks the set bits
// and runs each unit's records [first, last] (row 7 of the unit table) through the general record machine: same functions, same order.
template <bool FAST, bool EXT, class Prog>
RM_DEV float map_scene_units(const Prog& prog, const uint32_t* ranges, float* spill, float x, float y, float z, unsigned long long need,
                             SqrtGuard& tiny, uint32_t xf_base) {
    float qx[1] = {x}, qy[1] = {y}, qz[1] = {z}, acc[1] = {__uint_as_float(0x7F800000u)};
    uint32_t sp = 0u;
    unsigned long long m = need;
    while (m != 0ull) {
        const uint32_t u = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t range = __builtin_amdgcn_readfirstlane(ranges[u]);  // (every lane reads the same word)
        for (uint32_t c = range & 0xFFFFu, last = range >> 16; c <= last; c++) {
            uint32_t op;
            float p[7];
            prog.load(c, op, p);
            exec_command<1, FAST, EXT>(op, p, qx, qy, qz, acc, spill, sp, tiny, xf_base);
        }
    }
    return acc[0];
}

// map_scene for a CHAIN program (RmDecoded::is_chain): record 0 pushes a sphere / box, every later record is a sphere / box
// fused with the Union / Subtraction that consumes it (RM_OP_FASTCLASS != 0) -- what the reference's editor produces for
// "a op b op c ..." and what both metric scenes are.  No value stack, no opcode ladder: two decisions per record (which
// leaf, which operator).  Same leaf functions, same operators, same order as exec_command: the same bits.
// masked: `need` is the wave's unit mask (rm_kernel_v5.h "Wave-level culling"; in a chain of at most 64 records unit u IS record
// u): a record whose bit is clear is not even fetched -- its leaf is +inf, which min and max(., -.) ignore -- and the loop
// walks the set bits.  Records are fetched one ahead of their use either way.
template <bool FAST, class Prog>
RM_DEV float map_scene_chain(const Prog& prog, uint32_t n_rec, float x, float y, float z, unsigned long long need, bool masked, SqrtGuard& tiny) {
    uint32_t opa, opb;
    float pa[7], pb[7];
    float acc = __uint_as_float(0x7F800000u);
    auto apply = [&](uint32_t c, uint32_t op, const float (&p)[7]) {
        op = __builtin_amdgcn_readfirstlane(op);
        if (c == 0u) {  // record 0 pushes its value
            acc = RM_OP_KIND(op) == RM_KIND_SPHERE ? sdf_sphere_t<FAST>(x, y, z, p, tiny) : sdf_box_t<FAST>(x, y, z, p, tiny);
            return;
        }
        const uint32_t cls = RM_OP_FASTCLASS(op);
        const float b = (cls & 1u) ? sdf_sphere_t<FAST>(x, y, z, p, tiny) : sdf_box_t<FAST>(x, y, z, p, tiny);
        acc = cls >= 3u ? vmax_negb(acc, b) : vmin(acc, b);  // wgsl:248-252 / :242-246
    };
    if (masked) {
        // Scalar instructions are the dear ones here (4.3 cycles each per SIMD, rm_kernel_v5.h "The scalar unit"): record 0 -- the one
        // record that pushes -- is handled in front of the loop, so the loop body asks no "is this record 0"; the bit of a record is
        // cleared with one s_bitset0_b64 instead of the three instructions of m & (m - 1).
        unsigned long long m = need;  // (bits at and above n_rec are clear)
        if (m & 1ull) {
            prog.load(0u, opa, pa);
            opa = __builtin_amdgcn_readfirstlane(opa);
            acc = RM_OP_KIND(opa) == RM_KIND_SPHERE ? sdf_sphere_t<FAST>(x, y, z, pa, tiny) : sdf_box_t<FAST>(x, y, z, pa, tiny);
        }
        m &= ~1ull;
        if (m == 0ull) return acc;
        auto fused = [&](uint32_t op, const float (&p)[7]) {
            op = __builtin_amdgcn_readfirstlane(op);
            const float b = (op & (1u << 16)) ? sdf_sphere_t<FAST>(x, y, z, p, tiny) : sdf_box_t<FAST>(x, y, z, p, tiny);  // RM_OP_FASTCLASS 1, 3: sphere
            acc = RM_OP_FASTCLASS(op) >= 3u ? vmax_negb(acc, b) : vmin(acc, b);  // wgsl:248-252 / :242-246
        };
        auto take = [&]() -> uint32_t {  // the lowest record still to do (the caller knows there is one)
            const uint32_t c = (uint32_t)__builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(c));
            return c;
        };
        prog.load(take(), opa, pa);
        for (;;) {  // a record waits in (opa, pa)
            const bool more = m != 0ull;
            if (more) prog.load(take(), opb, pb);
            fused(opa, pa);
            if (!more) break;
            const bool more2 = m != 0ull;
            if (more2) prog.load(take(), opa, pa);
            fused(opb, pb);
            if (!more2) break;
        }
        return acc;
    }
    prog.load(0u, opa, pa);
    prog.load(n_rec > 1u ? 1u : 0u, opb, pb);
    apply(0u, opa, pa);
    for (uint32_t c = 1u; c < n_rec; c += 2u) {  // record c waits in (opb, pb)
        prog.load(c + 1u < n_rec ? c + 1u : c, opa, pa);
        apply(c, opb, pb);
        if (c + 1u >= n_rec) break;
        prog.load(c + 2u < n_rec ? c + 2u : c + 1u, opb, pb);
        apply(c + 1u, opa, pa);
    }
    return acc;
}

// map_scene for a TREE program (RmDecoded::is_tree): spheres and boxes under Union / Subtraction in any arrangement -- what the
// reference's node-graph editor can produce at all (csg/mod.rs:28-45) -- decoded into the eight record shapes of
// RM_OP_FASTCLASS.  One dispatch per record, the popped operands in the wave's LDS spill column; same leaf functions, same
// operators, same order as exec_command: the same bits.
template <bool FAST, class Prog>
RM_DEV float map_scene_tree(const Prog& prog, uint32_t n_rec, float* spill, float x, float y, float z, SqrtGuard& tiny) {
    const float inf = __uint_as_float(0x7F800000u);
    float acc = inf;
    uint32_t sp = 0u, c = 0u;
    uint32_t op0, op1;
    float p0[7], p1[7];
    // the record at c waits in (opc, pc); the next one is fetched into (opn, pn) before c is executed.  true: the program is done
    auto step = [&](uint32_t& opc, float (&pc)[7], uint32_t& opn, float (&pn)[7]) -> bool {
        const uint32_t op = __builtin_amdgcn_readfirstlane(opc);
        const uint32_t cls = RM_OP_FASTCLASS(op);
        prog.load(c + 1u < n_rec ? c + 1u : c, opn, pn);
        if (cls <= 4u) {  // (class 0 does not occur in a tree program)
            const float b = (cls & 1u) ? sdf_sphere_t<FAST>(x, y, z, pc, tiny) : sdf_box_t<FAST>(x, y, z, pc, tiny);
            acc = cls >= 3u ? vmax_negb(acc, b) : vmin(acc, b);  // wgsl:248-252 / :242-246
        } else if (cls <= 6u) {
            if (op & RM_OP_SPILL) { spill[sp * 64u] = acc; ++sp; }
            acc = cls == 5u ? sdf_sphere_t<FAST>(x, y, z, pc, tiny) : sdf_box_t<FAST>(x, y, z, pc, tiny);
        } else {
            --sp;
            const float a = spill[sp * 64u];
            acc = cls == 7u ? vmin(a, acc) : vmax_negb(a, acc);
        }
        return ++c == n_rec;
    };
    prog.load(0u, op0, p0);
    for (;;) {
        if (step(op0, p0, op1, p1)) break;
        if (step(op1, p1, op0, p0)) break;
    }
    return acc;
}

// The same with the wave's unit mask (rm_kernel_v5.h "Wave-level culling", lattice rule: a leaf whose bit is clear may be replaced by
// +inf).  An operand none of whose leaves is needed is +inf as a whole -- min(a, +inf) = a, max(a, -(+inf)) = a: the operator that
// would consume it is the identity on its other operand, and neither the operand's records nor the operator's need executing.
// tree_keep (rm_kernel_v5.h) has turned the unit mask into the RECORDS that are left -- keep -- and this loop walks its set bits,
// at most 128, fetching one record ahead: a handful of records of a tree of dozens.  What is left is again a well-formed postfix
// program: every operand with a needed leaf leaves one value, every other none.  Every push spills the accumulator (the first
// one an unused +inf: one slot more than map_scene_tree needs); a fused leaf whose left operand has gone (as_push) pushes.
template <bool FAST, class Prog>
RM_DEV float map_scene_tree_masked(const Prog& prog, float* spill, float x, float y, float z, unsigned long long keep0, unsigned long long keep1,
                                   unsigned long long push0, unsigned long long push1, SqrtGuard& tiny) {
    float acc = __uint_as_float(0x7F800000u);
    uint32_t sp = 0u;
    uint32_t opa, opb;
    float pa[7], pb[7];
    auto apply = [&](uint32_t c, uint32_t op, const float (&p)[7]) {
        op = __builtin_amdgcn_readfirstlane(op);
        const uint32_t cls = RM_OP_FASTCLASS(op);
        if (cls <= 6u) {
            const float b = (cls & 1u) ? sdf_sphere_t<FAST>(x, y, z, p, tiny) : sdf_box_t<FAST>(x, y, z, p, tiny);
            const bool as_push = (((c < 64u ? push0 : push1) >> (c & 63u)) & 1ull) != 0ull;
            if (cls >= 5u || as_push) {
                spill[sp * 64u] = acc;
                ++sp;
                acc = b;
            } else {
                acc = cls >= 3u ? vmax_negb(acc, b) : vmin(acc, b);  // wgsl:248-252 / :242-246
            }
        } else {
            --sp;
            const float a = spill[sp * 64u];
            acc = cls == 7u ? vmin(a, acc) : vmax_negb(a, acc);
        }
    };
    unsigned long long m0 = keep0, m1 = keep1;
    auto next = [&]() -> uint32_t {  // (the caller knows a bit is left)
        uint32_t c;
        if (m0 != 0ull) { c = (uint32_t)__builtin_ctzll(m0); m0 &= m0 - 1ull; }
        else { c = 64u + (uint32_t)__builtin_ctzll(m1); m1 &= m1 - 1ull; }
        return c;
    };
    if ((m0 | m1) == 0ull) return acc;
    uint32_t ca = next(), cb = 0u;
    prog.load(ca, opa, pa);
    for (;;) {  // record ca waits in (opa, pa)
        const bool more = (m0 | m1) != 0ull;
        if (more) { cb = next(); prog.load(cb, opb, pb); }
        apply(ca, opa, pa);
        if (!more) break;
        const bool more2 = (m0 | m1) != 0ull;
        if (more2) { ca = next(); prog.load(ca, opa, pa); }
        apply(cb, opb, pb);
        if (!more2) break;
    }
    return acc;
}

// SmoothUnion as the material walks apply it (exec_command, RM_MODE_SMOOTH): every lane takes the full form.
RM_DEV float material_smooth_union(float kk, float a, float b) {
    float v = fmin_(a, b);
    if (kk > 0.0f) {
        const float h = fmax_(kk - __builtin_fabsf(a - b), 0.0f) / kk;
        v = v - ((h * h) * kk) * 0.25f;
    }
    return v;
}

// Materials (extension; semantics: oracle/rm_oracle.c map_scene_impl with mat_out).  One evaluation of the material
// program (RmLaunch::mprog: the program decoded with its Material tags in place) at the position of a hit, on a stack
// of (distance, index) pairs: the accumulator pair lives in registers, deeper pairs in the wave's LDS spill area
// ([depth] distances, then [depth] indices, then 3 floats per transform level; `spill` already points at this lane).
// Runs once per hit ray against tens of march steps, so it is a plain loop with the generic (correctly rounded)
// square root; the distances are the ones map_scene computes at this point, operation for operation.
RM_DEV uint32_t map_scene_material(const RmRecord* __restrict__ mprog, uint32_t n_mrec, float* spill, uint32_t value_depth,
                                   float x, float y, float z) {
    float acc = 0.0f;
    uint32_t accm = 0u, sp = 0u;
    uint32_t* mspill = reinterpret_cast<uint32_t*>(spill) + (size_t)value_depth * 64u;
    float* saved = spill + (size_t)2u * value_depth * 64u;
    SqrtGuard unused;
    for (uint32_t c = 0; c < n_mrec; c++) {
        const RmRecord& r = mprog[c];  // wave-uniform address: scalar loads
        const uint32_t op = r.op, kind = RM_OP_KIND(op), mode = RM_OP_MODE(op);
        float p[7];
#pragma unroll
        for (int k = 0; k < 7; k++) p[k] = r.p[k];
        if (kind == RM_KIND_MATERIAL) {
            accm = __float_as_uint(p[0]);
            continue;
        }
        if (kind == RM_KIND_XFORM) {
            float* save = saved + (size_t)3u * __float_as_uint(p[6]) * 64u;
            if ((mode & 1u) == 0u) {
                save[0] = x; save[64] = y; save[128] = z;
                if (mode == RM_XF_T_PUSH) { x = x - p[0]; y = y - p[1]; z = z - p[2]; }
                else if (mode == RM_XF_R_PUSH) xf_rotate_conj(p[0], p[1], p[2], p[3], x, y, z);
                else { x = x / p[0]; y = y / p[0]; z = z / p[0]; }
            } else {
                x = save[0]; y = save[64]; z = save[128];
                if (mode == RM_XF_S_POP) acc = acc * p[0];
            }
            continue;
        }
        float a, b;
        uint32_t am, bm;
        if (kind == RM_KIND_POP) {
            --sp;
            b = acc; bm = accm;
            a = spill[sp * 64u]; am = mspill[sp * 64u];
        } else {
            if (kind == RM_KIND_SPHERE) b = sdf_sphere_t<false>(x, y, z, p, unused);
            else if (kind == RM_KIND_BOX) b = sdf_box_t<false>(x, y, z, p, unused);
            else if (kind == RM_KIND_CYLINDER) b = sdf_cylinder_t<false>(x, y, z, p, unused);
            else b = ((x * p[0] + y * p[1]) + z * p[2]) + p[3];
            bm = 0u;
            if (op & RM_OP_SPILL) {
                spill[sp * 64u] = acc; mspill[sp * 64u] = accm;
                ++sp;
            }
            a = acc; am = accm;
        }
        if (mode == RM_MODE_PUSH) {
            acc = b; accm = bm;
        } else if (mode == RM_MODE_UNION) {
            acc = vmin(a, b); accm = b < a ? bm : am;
        } else if (mode == RM_MODE_SUB) {
            acc = vmax_negb(a, b); accm = -b > a ? bm : am;
        } else if (mode == RM_MODE_INTER) {
            acc = fmax_(a, b); accm = b > a ? bm : am;
        } else {  // RM_MODE_SMOOTH
            const float kk = p[0];
            float v = fmin_(a, b);
            if (kk > 0.0f) {
                const float h = fmax_(kk - __builtin_fabsf(a - b), 0.0f) / kk;
                v = v - ((h * h) * kk) * 0.25f;
            }
            acc = v; accm = b < a ? bm : am;
        }
    }
    return accm;
}

// The shader's ray direction is the .xyz of a normalised vec4 (wgsl:62): whenever the w components of
// pt_world and ro differ (any projection with znear != 1) it is SHORTER than 1, and the march walks the
// half-line along d / |d|.  The miss tests are statements about that half-line, so they use the unit
// direction; a zero or non-finite d gives NaN here and every comparison below then says "not clear".
RM_DEV void unit_dir(float& dx, float& dy, float& dz) {
    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    dx *= inv; dy *= inv; dz *= inv;  // |error| ~1e-7, far inside the 1e-5 slack of the table entries
}
// One-float result code of a ray that did not hit (see the resolve step).
RM_DEV float miss_code(const V4& ro, float dx, float dy, float dz) {
    const int c = shade_floor(ro.y, ro.x, ro.z, dx, dy, dz);
    return c < 0 ? -3.0f : -1.0f - (float)c;
}

// ---------------------------------------------------------------------------------------------
// Self-tests of the arithmetic building blocks (diagnostics, rm_selftest_* in the ABI).
// ---------------------------------------------------------------------------------------------
// Exhaustive: for EVERY binary32 bit pattern x >= +0 (what a sum of squares can be) and both forms of the guarded
// fast sqrt, either the guard sends the evaluation to the generic path or the result is the correctly rounded root.
// Also counts the in-range inputs the guard rejects: must be none besides 0 for the ZERO_OK = false form.
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
__global__ __launch_bounds__(256) void rm_selftest_sqrt_kernel(uint32_t first, uint64_t count, unsigned long long* mismatches,
                                                               uint32_t* first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = first + (uint32_t)i;
        if (bits > 0x7FFFFFFFu && bits != 0xFFC00000u) continue;  // negative: never an argument (one NaN with the sign set stays in)
        const float x = __uint_as_float(bits);
        const float want = __builtin_sqrtf(x);
        const bool in_range = bits >= 0x0F800000u && bits <= 0x7F7FFFFFu;
        bool ok = true;
        {
            SqrtGuard g;
            const float a = sqrt_sel<true, false>(x, g);
            ok = ok && (g.bad() ? !in_range : __float_as_uint(a) == __float_as_uint(want));
        }
        {
            SqrtGuard g;
            const float a = sqrt_sel<true, true>(x, g);
            ok = ok && (g.bad() ? !(in_range || bits == 0u) : __float_as_uint(a) == __float_as_uint(want));
        }
        if (!ok) {
            bad++;
            atomicMin(first_bad, bits);
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}
#endif

// Element-wise results of the primitives the kernels rely on, for comparison with the oracle's
// definitions on the host: out[0..7][i] = min, max, direct v_min, direct v_max(a,-b), fast sqrt(a),
// generic sqrt(a), a / b, (float) i32(round(a)).
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
__global__ __launch_bounds__(256) void rm_selftest_ops_kernel(const float* a, const float* b, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out[0u * n + i] = fmin_(x, y);
    out[1u * n + i] = fmax_(x, y);
    out[2u * n + i] = vmin(x, y);
    out[3u * n + i] = vmax_negb(x, y);
    SqrtGuard guard;
    const float fast = sqrt_sel<true, true>(x, guard);  // what a kernel does: fast form unless the guard objects
    out[4u * n + i] = guard.bad() ? __builtin_sqrtf(x) : fast;
    out[5u * n + i] = __builtin_sqrtf(x);
    out[6u * n + i] = x / y;
    out[7u * n + i] = (float)__float2int_rz(__builtin_rintf(x));
}
#endif

}  // namespace rmk
