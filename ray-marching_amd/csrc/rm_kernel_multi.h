// rm_kernel_multi.h -- kernel v3 "raypool-R": the ray-pool kernel of rm_kernels.h with R rays
// in flight per lane and a leaner interpreter.  Device code only (gfx950, wave64).
//
// Why R > 1: the CSG program is the same for every lane, so its fetch (s_load / ds_read),
// opcode decode, branches and loop bookkeeping are per-WAVE costs.  With one ray per lane the
// profile (profiles/r01_*) showed 0.55 scalar instructions per vector instruction and the
// CU's scalar unit ~74 % busy; evaluating R positions per fetched command divides all of that
// by R and gives every wave R independent dependency chains (ILP), so fewer resident waves
// are needed to keep the VALU issuing every cycle.
//
// Bit-exactness: each lane-slot performs exactly the operation sequence of the arithmetic
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
        const float kk = p[0];
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

// ---------------------------------------------------------------------------------------------
// Miss-ray culling (exact).  The colour of a ray that never registers a hit depends only on its
// origin and direction (floor test, wgsl:117-130), not on the march.  A march position can only
// register a hit (scene_dist < min_dist, wgsl:97) if it lies within min_dist of some primitive:
// every union/subtraction tree evaluates to >= the minimum over its leaves (min keeps a leaf,
// max(a,-b) >= a), and a leaf's SDF is >= the distance to its bounding sphere.  So a ray whose
// whole half-line stays farther than min_dist (plus a generous float-error margin) from EVERY
// primitive's bounding sphere cannot hit, whatever the step sequence, max_iter or max_dist:
// it is shaded as a miss without marching.  With o the common ray origin and m = c - o, the ray
// direction d (|d| = 1) misses the sphere (c, Rk) iff  m.d < sqrt(|m|^2 - Rk^2)  -- one cone per
// primitive.  All slack is on the safe side (a ray that is not provably clear is marched).
// ---------------------------------------------------------------------------------------------
RM_DEV float4 cull_entry(const RmRecord& rec, const V4& ro, float min_dist) {
    const uint32_t kind = RM_OP_KIND(rec.op);  // v3 culls only reference-only programs (spheres and boxes)
    const float inf = __uint_as_float(0x7F800000u);
    if (kind == RM_KIND_POP) return make_float4(0.0f, 0.0f, 0.0f, inf);  // operators constrain nothing
    float rho;
    if (kind == RM_KIND_SPHERE) {
        rho = fmax_(rec.p[3], 0.0f);
    } else {
        const float hx = fmax_(rec.p[3], 0.0f), hy = fmax_(rec.p[4], 0.0f), hz = fmax_(rec.p[5], 0.0f);
        rho = __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * 1.0001f;
    }
    const float cx = rec.p[0], cy = rec.p[1], cz = rec.p[2];
    // margin: min_dist plus 1 % of the local coordinate scale (float error of the SDF evaluation
    // is ~1e-6 of that scale; the march positions lie on the ray up to the same error)
    const float scale = 1.0f + __builtin_fabsf(cx) + __builtin_fabsf(cy) + __builtin_fabsf(cz) + rho +
                        __builtin_fabsf(ro.x) + __builtin_fabsf(ro.y) + __builtin_fabsf(ro.z);
    const float Rk = rho + fmax_(min_dist, 0.0f) * 1.01f + 0.01f * scale;
    const float mx = cx - ro.x, my = cy - ro.y, mz = cz - ro.z;
    const float mm = mx * mx + my * my + mz * mz;
    const float lim = mm * 0.998f - Rk * Rk * 1.002f;
    // origin inside (or not clearly outside) the inflated sphere, or anything non-finite:
    // this primitive vetoes culling for every ray (s = -inf fails the test  m.d - s < 0).
    float s = -inf;
    if (lim > 0.0f && lim < inf && mm < inf) s = __builtin_sqrtf(lim) * 0.999f;
    return make_float4(mx, my, mz, s);
}
// The shader's ray direction is the .xyz of a normalised vec4 (wgsl:62): whenever the w components of
// pt_world and ro differ (any projection with znear != 1) it is SHORTER than 1, and the march walks the
// half-line along d / |d|.  The miss tests are statements about that half-line, so they use the unit
// direction; a zero or non-finite d gives NaN here and every comparison below then says "not clear".
RM_DEV void unit_dir(float& dx, float& dy, float& dz) {
    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    dx *= inv; dy *= inv; dz *= inv;  // |error| ~1e-7, far inside the 1e-5 slack of the table entries
}
RM_DEV bool ray_misses_scene(const float4* cullt, uint32_t n_cull, float dx, float dy, float dz) {
    bool clear = true;  // n_cull == 0 (empty scene): every ray misses (wgsl:189-191)
    unit_dir(dx, dy, dz);
    for (uint32_t k = 0; k < n_cull; k++) {
        const float4 e = cullt[k];  // wave-uniform address: LDS broadcast
        const float t = __builtin_fmaf(e.z, dz, __builtin_fmaf(e.y, dy, __builtin_fmaf(e.x, dx, -e.w)));
        clear = clear && (t < 0.0f);  // NaN -> not clear
        if (__ballot(clear) == 0ull) break;
    }
    return clear;
}
// One-float result code of a ray that did not hit (see the resolve step).
RM_DEV float miss_code(const V4& ro, float dx, float dy, float dz) {
    const int c = shade_floor(ro.y, ro.x, ro.z, dx, dy, dz);
    return c < 0 ? -3.0f : -1.0f - (float)c;
}

// ---------------------------------------------------------------------------------------------
// Kernel v3.  One wave owns a tile of 64*R pixels (R=1: 8x8, R=2: 16x8, R=4: 16x16) and the pool
// of its 1024*R rays; lane-slot (lane, k) marches one ray at a time exactly like kernel v2.
// LDS per wave: res[16][64R] + pt_screen table [2][64R] + spill [depth][R][64] (+ program).
// ---------------------------------------------------------------------------------------------
template <int R>
struct TileGeom {
    static constexpr uint32_t PIX = 64u * R;
    static constexpr uint32_t TW = R == 1 ? 8u : 16u;
    static constexpr uint32_t TH = PIX / TW;
    static constexpr uint32_t POOL = PIX * 16u;
};

// WPT waves (one workgroup) share a tile: its pool cursor lives in LDS and is advanced with one
// ds_add per refill round, so a tile full of geometry is marched by 64*WPT lanes at once.  That
// divides the duration of the longest waves -- which set the tail of the kernel -- by WPT and
// multiplies the number of (shorter) waves the dispatcher has to balance with.
template <class Prog, bool PROG_IN_LDS, int R, int WPT>
__global__ __launch_bounds__(64 * WPT) void rm_render_raypool_multi(RmLaunch L, uint32_t refill_min) {
    using G = TileGeom<R>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    float* res = reinterpret_cast<float*>(smem);  // [16][PIX]
    float* sxy = res + 16u * G::PIX;              // [2][PIX]
    float* spill = sxy + 2u * G::PIX + wave * (L.spill_depth * R * 64u) + lane;  // [WPT][spill_depth][R][64]
    float4* cullt = reinterpret_cast<float4*>(sxy + 2u * G::PIX + L.spill_depth * R * 64u * WPT);  // [n_cull]
    uint32_t* lprog = reinterpret_cast<uint32_t*>(cullt + L.n_cull);
    uint32_t* s_next = lprog + (PROG_IN_LDS ? L.n_rec * 8u : 0u);  // shared pool cursor (WPT > 1)

    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];  // wave-uniform
    float* out = L.out + (size_t)blockIdx.z * L.rows * L.W * 4;
    // Tile handled by this wave: dispatch slot blockIdx.x -> tile id, heaviest tiles first when the
    // host ran the cost pre-pass (rm_tile_cost + rm_tile_sort), identity otherwise.
    const uint32_t tiles_x = (L.W + G::TW - 1u) / G::TW;
    const uint32_t tile = L.order ? L.order[(size_t)blockIdx.z * gridDim.x + blockIdx.x] : blockIdx.x;
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;

    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);  // wgsl:39-40
    const float eps = 0.0001f;                                    // wgsl:136
    for (uint32_t p = tid; p < G::PIX; p += 64u * WPT) {  // pt_screen of the tile's pixels (edge tiles clamp)
        const uint32_t tx = tile_x * G::TW + p % G::TW, ty = tile_y * G::TH + p / G::TW;
        const uint32_t px = tx < L.W ? tx : L.W - 1u;
        const uint32_t ry = ty < L.rows ? ty : L.rows - 1u;
        sxy[p] = screen_x(px, L.W);
        sxy[G::PIX + p] = screen_y(rm_global_row(L, ry), L.H);
    }
    if (PROG_IN_LDS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.prog);
        for (uint32_t k = tid; k < L.n_rec * 8u; k += 64u * WPT) lprog[k] = src[k];
    }
    // Miss-ray culling table (see cull_entry): one cone per command, built once per workgroup.
    for (uint32_t k = tid; k < L.n_cull; k += 64u * WPT) cullt[k] = cull_entry(L.prog[k], ro, L.min_dist);
    if (WPT > 1 && tid == 0u) *s_next = 0u;
    __syncthreads();

    Prog prog;
    if constexpr (PROG_IN_LDS) prog.base = lprog;
    else prog.base = L.prog;

    // lane-slot state: evaluation point = b + d * sc
    float bx[R], by[R], bz[R], dx[R], dy[R], dz[R], sc[R], nx[R], ny[R], nz[R];
    uint32_t it[R], mode[R], rid[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        bx[k] = by[k] = bz[k] = dx[k] = dy[k] = dz[k] = sc[k] = nx[k] = ny[k] = nz[k] = 0.0f;
        it[k] = 0u; rid[k] = 0u; mode[k] = M_EMPTY;
    }
    uint32_t next = 0;  // wave-uniform: first unassigned ray of the pool (WPT > 1: last value seen)
    // diagnostics only (L.stats != nullptr): wall-clock stamps and loop statistics of this wave
    const unsigned long long t_start = L.stats ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t n_iter = 0u, n_live = 0u, n_refill = 0u;

    for (;;) {
        // ---- refill: shade parked rays, hand out new ones (per slot, ballot-driven) ----
        unsigned long long live_any = 0ull;
#pragma unroll
        for (int k = 0; k < R; k++) live_any |= __ballot(mode[k] < M_DONE_HIT);
        if constexpr (WPT > 1) next = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t*>(s_next));
        bool refilled = false;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const bool parked = mode[k] >= M_DONE_HIT && mode[k] <= M_EMPTY;
            const unsigned long long parked_mask = __ballot(parked);
            const uint32_t n_parked = (uint32_t)__popcll(parked_mask);
            if (n_parked != 0u && (live_any == 0ull || (n_parked >= refill_min && next < G::POOL))) {
                refilled = true;
                if (mode[k] == M_DONE_HIT) {
                    res[rid[k]] = shade_hit(nx[k], ny[k], nz[k], bx[k], by[k], bz[k]);  // wgsl:98-103
                } else if (mode[k] == M_DONE_MISS) {
                    res[rid[k]] = miss_code(ro, dx[k], dy[k], dz[k]);  // wgsl:117-130
                }
                // Hand out rays in rounds: a lane whose new ray provably misses the scene (cull
                // test) is shaded on the spot and asks again in the next round.
                bool want = parked;
                for (uint32_t round = 0;; round++) {
                    const unsigned long long want_mask = __ballot(want);
                    const uint32_t n_want = (uint32_t)__popcll(want_mask);
                    if (n_want == 0u) break;
                    if (round != 0u && n_want < refill_min && (live_any | __ballot(mode[k] < M_DONE_HIT)) != 0ull) {
                        if (want) mode[k] = M_EMPTY;  // few stragglers: let them wait for the next refill
                        break;
                    }
                    if constexpr (WPT > 1) {  // claim n_want consecutive rays of the shared pool
                        uint32_t base = 0u;
                        if (next < G::POOL && lane_rank(want_mask) == 0u && want) base = atomicAdd(s_next, n_want);
                        else if (next >= G::POOL) base = next;
                        next = __builtin_amdgcn_readfirstlane(__shfl(base, (int)__builtin_ctzll(want_mask)));
                    }
                    const uint32_t r = next + lane_rank(want_mask);
                    next = next + n_want < 2u * G::POOL ? next + n_want : 2u * G::POOL;
                    if (want) {
                        if (r < G::POOL) {
                            rid[k] = r;
                            const uint32_t p = r % G::PIX, s = r / G::PIX;  // sample-major: res[s][p] == res[r]
                            gen_ray(u, ro, sxy[p], sxy[G::PIX + p], s >> 2, s & 3u, dx[k], dy[k], dz[k]);
                            if (L.max_iter == 0u || ((L.flags & 1u) && ray_misses_scene(cullt, L.n_cull, dx[k], dy[k], dz[k]))) {
                                res[r] = miss_code(ro, dx[k], dy[k], dz[k]);  // never marched: wgsl:117-130 only
                            } else {
                                bx[k] = ro.x; by[k] = ro.y; bz[k] = ro.z;
                                sc[k] = 0.0f;  // dist (wgsl:88)
                                it[k] = 0u;
                                mode[k] = M_MARCH;
                                want = false;
                            }
                        } else {
                            mode[k] = M_RETIRED;
                            want = false;
                        }
                    }
                }
            }
        }
        if (refilled) { n_refill++; continue; }  // re-evaluate the masks
        if (live_any == 0ull) break;    // nothing live and nothing to refill: every slot retired

        // ---- one map_scene evaluation per live lane-slot ----
        float qx[R], qy[R], qz[R], v[R];
#pragma unroll
        for (int k = 0; k < R; k++) {  // wgsl:91 / :138-141
            qx[k] = bx[k] + dx[k] * sc[k];
            qy[k] = by[k] + dy[k] * sc[k];
            qz[k] = bz[k] + dz[k] * sc[k];
        }
        n_iter++;
        n_live += (uint32_t)__popcll(live_any);
        SqrtGuard tiny;
        map_scene_multi<R, true>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny);
        if (__ballot(tiny.bad()) != 0ull)  // a sqrt argument outside the fast range (SqrtGuard): redo with the generic sqrt
            map_scene_multi<R, false>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny);

#pragma unroll
        for (int k = 0; k < R; k++) {
            if (mode[k] == M_MARCH) {
                if (v[k] < L.min_dist) {  // wgsl:97: hit -> normal taps around pos = q
                    bx[k] = qx[k]; by[k] = qy[k]; bz[k] = qz[k];
                    sc[k] = eps;
                    uint32_t sx, sy, sz;
                    tap_signs(0u, sx, sy, sz);
                    dx[k] = __uint_as_float(0x3F800000u ^ sx);
                    dy[k] = __uint_as_float(0x3F800000u ^ sy);
                    dz[k] = __uint_as_float(0x3F800000u ^ sz);
                    mode[k] = M_TAP0;
                } else if (v[k] > L.max_dist) {  // wgsl:109-111
                    mode[k] = M_DONE_MISS;
                } else {
                    sc[k] += v[k];  // wgsl:114
                    it[k] += 1u;
                    if (it[k] >= L.max_iter) mode[k] = M_DONE_MISS;  // loop bound, wgsl:90
                }
            } else if (mode[k] < M_DONE_HIT) {
                const uint32_t t = mode[k] - M_TAP0;  // tap t: n (+)= k_t * f, products with +-1 are exact
                uint32_t sx, sy, sz;
                tap_signs(t, sx, sy, sz);
                const float vx = __uint_as_float(__float_as_uint(v[k]) ^ sx);
                const float vy = __uint_as_float(__float_as_uint(v[k]) ^ sy);
                const float vz = __uint_as_float(__float_as_uint(v[k]) ^ sz);
                nx[k] = t == 0u ? vx : nx[k] + vx;
                ny[k] = t == 0u ? vy : ny[k] + vy;
                nz[k] = t == 0u ? vz : nz[k] + vz;
                tap_signs(t + 1u, sx, sy, sz);
                dx[k] = __uint_as_float(0x3F800000u ^ sx);
                dy[k] = __uint_as_float(0x3F800000u ^ sy);
                dz[k] = __uint_as_float(0x3F800000u ^ sz);
                mode[k] += 1u;  // M_TAP3 + 1 == M_DONE_HIT
            }
        }
    }
    __syncthreads();
    if (L.stats && lane == 0u) {
        unsigned long long* st = L.stats + 4ull * (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * WPT + wave);
        st[0] = t_start;
        st[1] = __builtin_amdgcn_s_memrealtime();
        st[2] = ((unsigned long long)tile << 32) | n_iter;
        st[3] = ((unsigned long long)n_refill << 32) | n_live;
    }

    // ---- resolve: one pixel per thread, samples in the reference order (wgsl:44-45, 68-69) ----
    for (uint32_t p = tid; p < G::PIX; p += 64u * WPT) {
        const uint32_t tx = tile_x * G::TW + p % G::TW, ty = tile_y * G::TH + p / G::TW;
        if (tx < L.W && ty < L.rows) {
            float tr = 0.0f, tg = 0.0f, tb = 0.0f;
#pragma unroll 4
            for (uint32_t s = 0; s < 16u; s++) {
                const float code = res[s * G::PIX + p];
                float cr, cg, cb;
                if (code >= 0.0f) {  // hit: (0.4,0.7,0.1) * k  (wgsl:105)
                    cr = 0.4f * code; cg = 0.7f * code; cb = 0.1f * code;
                } else if (code > -2.5f) {  // floor (wgsl:127)
                    const float g = 0.2f * (-1.0f - code);
                    cr = 0.1f + g; cg = 0.1f + g; cb = 0.2f + g;
                } else {
                    cr = 0.0f; cg = 0.0f; cb = 0.0f;  // wgsl:130
                }
                tr += __builtin_sqrtf(cr);
                tg += __builtin_sqrtf(cg);
                tb += __builtin_sqrtf(cb);
            }
            float4 o;
            o.x = tr / 16.0f; o.y = tg / 16.0f; o.z = tb / 16.0f; o.w = 1.0f;  // wgsl:73-75
            reinterpret_cast<float4*>(out)[(size_t)ty * L.W + tx] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Load-balance pre-pass.  Tile cost varies by >100x (a sky tile culls all 1024 rays in 16 cheap
// rounds, a tile full of geometry marches ~16k evaluations) and the frame has only ~1.6x more
// expensive tiles than the chip has wave slots, so dispatching tiles in raster order leaves the
// SIMDs half empty while the last heavy waves finish (measured: 3.7 of 8 wave slots occupied on
// average).  rm_tile_cost estimates every tile's cost as the number of its 64 centre-most-sample
// rays that survive the cull test; rm_tile_sort orders tile ids by that estimate, heaviest first
// (longest-processing-time-first scheduling: the tail of the kernel is made of cheap tiles).
// The order only affects WHEN a tile is rendered, never its pixels.
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64) void rm_tile_cost(RmLaunch L, uint32_t* cost) {
    using G = TileGeom<R>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    float4* cullt = reinterpret_cast<float4*>(smem);
    const uint32_t lane = threadIdx.x;
    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];
    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);
    for (uint32_t k = lane; k < L.n_cull; k += 64u) cullt[k] = cull_entry(L.prog[k], ro, L.min_dist);
    __syncthreads();
    const uint32_t tiles_x = (L.W + G::TW - 1u) / G::TW;
    const uint32_t tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t p = (uint32_t)k * 64u + lane;
        const uint32_t tx = tile_x * G::TW + p % G::TW, ty = tile_y * G::TH + p / G::TW;
        const uint32_t px = tx < L.W ? tx : L.W - 1u, ry = ty < L.rows ? ty : L.rows - 1u;
        float dx, dy, dz;
        gen_ray(u, ro, screen_x(px, L.W), screen_y(rm_global_row(L, ry), L.H), 1u, 2u, dx, dy, dz);
        const bool culled = (L.flags & 1u) && ray_misses_scene(cullt, L.n_cull, dx, dy, dz);
        n += (uint32_t)__popcll(__ballot(!culled));
    }
    if (lane == 0u) cost[(size_t)blockIdx.z * gridDim.x + blockIdx.x] = n / R;  // 0..64
}

// Counting sort of the tile ids of one frame (blockIdx.x = frame) by descending cost.
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
__global__ __launch_bounds__(1024) void rm_tile_sort(const uint32_t* cost, uint32_t* order, uint32_t n_tiles) {
    __shared__ uint32_t hist[65], base[65];
    const uint32_t tid = threadIdx.x;
    const uint32_t* c = cost + (size_t)blockIdx.x * n_tiles;
    uint32_t* o = order + (size_t)blockIdx.x * n_tiles;
    if (tid < 65u) hist[tid] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n_tiles; i += 1024u) atomicAdd(&hist[64u - (c[i] < 64u ? c[i] : 64u)], 1u);
    __syncthreads();
    if (tid == 0u) {
        uint32_t acc = 0u;
        for (uint32_t b = 0; b < 65u; b++) { base[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_tiles; i += 1024u) {
        const uint32_t pos = atomicAdd(&base[64u - (c[i] < 64u ? c[i] : 64u)], 1u);
        o[pos] = i;
    }
}
#endif

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
