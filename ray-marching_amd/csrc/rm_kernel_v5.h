// rm_kernel_v5.h -- kernel v5 "streamed ray pool" (gfx950, wave64).  Device code only.
//
// One workgroup of WPT waves owns an 8x8-pixel tile and the pool of its 1024 rays: a lane marches ONE
// ray at a time; when its ray ends, the lane is refilled (ballot + prefix count) with the next ray, so
// every trip through map_scene runs with (nearly) all 64 lanes live whatever the per-ray step counts
// are.  WHERE the expensive, branchy per-ray work runs:
//
//   produce  Ray generation (2 mat-vec + vec4 normalize: wgsl:52-62) and the exact miss test run
//            for 64 consecutive rays of the pool at a time with ALL lanes active, using scratch
//            registers only (the marching state of the lanes is not touched).  Rays that provably
//            miss the scene are shaded on the spot (res[r] = floor code); survivors are compacted
//            with ballot + prefix count into a per-wave READY ring in LDS (ray id + direction).
//   consume  A lane whose ray ended pops the next ready ray: four LDS reads.
//   shade    A finished ray pushes its shading inputs (hit: normal sum + position, miss:
//            direction) into a per-wave SHADE ring; it is flushed (all waiting entries shaded at
//            once, 2 normalizes = 6 correctly rounded divides + 2 sqrt per hit) when the next push
//            would not fit.
//
// In the retired v2-v4 kernels (DESIGN.md 5 keeps their numbers) these pieces ran inside the march loop
// under a sparse exec mask (typically 8-14 of 64 lanes) and cost ~25 % of all vector instructions; here
// they run at (nearly) full occupancy and the march loop is: evaluation point, map_scene, ~20 VALU of
// state update, three ballots.
// Both rings are private to a wave (no barrier; LDS executes a wave's accesses in order), the
// pool cursor and the result array are shared by the tile's waves.
//
// Miss test: spheres get a cone test, boxes a ray/slab test against the box inflated by the margin
// (min_dist + 1e-4 of the local coordinate scale; float error of positions and SDF values is ~1e-6 of
// that scale).
#pragma once
#include "rm_interp.h"

namespace rmk {

// Structure-specialised evaluation (rm_jit.h): when a program's command sequence has been compiled
// into straight-line code by hipRTC, the generated translation unit defines this function and
// instantiates the kernel body with SPEC = true; the library's own build only declares it.
//   lp: the decoded program in LDS (RmRecord[n], 8 dwords each: parameters are read at fixed offsets)
//   thr, live: far-primitive pruning, see below
// LdsF: an explicit LDS (address space 3) pointer.  The body hands map_scene_spec a base that went through a
// v_mov_b32 in inline asm, so the compiler sees ONE vector register plus compile-time offsets and folds those into
// the ds_read offset fields.  With a scalar base, ROCm 7.2's compiler materialised every record address in an SGPR
// of its own (40 live scalars, spilled and re-read with v_readlane inside the march loop: +33 % kernel time); ROCm
// 7.0's did not.  The laundering makes both produce the same loop.
typedef const __attribute__((address_space(3))) float* LdsF;
RM_DEV LdsF lds_vector_base(const void* generic_lds_ptr) {
    const uint32_t s = (uint32_t)(__SIZE_TYPE__)(const __attribute__((address_space(3))) void*)generic_lds_ptr;
    uint32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return (LdsF)(__SIZE_TYPE__)v;  // LDS pointers are 32 bits wide; the widening only silences the host pass
}
// `need`: the units of the program (rm_units.h) some live lane may depend on, one bit each, from wave_cull_* below (all ones
// when the kernel does not cull); `live`: wave mask of the lanes whose value will be used
template <bool FAST>
RM_DEV float map_scene_spec(LdsF lp, float qx, float qy, float qz, unsigned long long need, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval);
#ifdef RM_JIT_MATERIAL_WALK
// The material walk of a tagged program as straight-line code (rm_jit.h generate_material_walk): which material does the
// surface at (x, y, z) carry.  mp: the tagged records in device memory (uniform addresses: scalar loads).
template <bool FAST>
RM_DEV uint32_t map_scene_material_spec(const RmRecord* __restrict__ mp, float qx, float qy, float qz, SqrtGuard& tiny);
#endif
#ifdef RM_JIT_TAPS4
// The four normal taps of a hit at c in one pass (rm_jit.h generate_map_scene_taps): f[t] = map_scene(c + k_t eps).
template <bool FAST>
RM_DEV void map_scene_taps(LdsF lp, float cx, float cy, float cz, unsigned long long need, unsigned long long live, SqrtGuard& tiny, float (&f)[4]);
#endif

// ---- Pruning of far primitives (exact) ----------------------------------------------------------------
// A tree of min / max / negation over leaf values is monotone in every leaf: as a function of one
// leaf's signed value t (t = v for a leaf the tree value F grows with, t = -v below an odd number of
// subtraction right-hand sides) it is  F(t) = max(lo, min(hi, t))  for some lo <= hi, or constant.
// Hence, if F is known to lie in [-thr, thr] and a leaf has v > thr, then t > thr >= F(t) forces
// hi < t (resp. t < -thr <= F(t) forces lo > t): F does not depend on that leaf any more and is
// unchanged -- bit for bit, min and max only select -- if the leaf value is replaced by +inf.  The
// generated code then skips the leaf: "acc = min(acc, +inf)" and "acc = max(acc, -inf)"
// leave acc alone; a leaf that is pushed becomes the constant +inf.
//
// thr comes from the previous evaluation of the same ray.  Every node type admitted here (sphere,
// box, capped cylinder; union, subtraction, intersection) is 1-Lipschitz in real arithmetic, so the
// value F' at the next point q' satisfies |F' - F| <= |q' - q|; with the evaluation error E of the
// float computation on both sides,  |F'_computed| <= |F_computed| + |q' - q| + 2E.
//   march step (wgsl:114): |q' - q| <= |sd| (|rd| <= 1)      -> thr = 2 |sd| (1 + 1e-5) + m
//   normal taps (wgsl:138-141): |q' - q_hit| = eps sqrt(3)    -> thr = |sd_hit| (1 + 1e-5) + 1.75e-4 + m
//   first evaluation of a ray                                 -> thr = +inf (nothing is skipped)
// m = 4e-6 (scene_scale + |ro|_1 + |q'|_1) covers 3E (E <= 4e-7 of that scale: a handful of binary32
// roundings of quantities no larger than it) plus the rounding of q' itself, with 3x to spare.
// Programs with a Plane (|n| arbitrary) or a SmoothUnion (not a lattice operator) are not pruned this way; programs that
// blend have rules of their own (below).
constexpr float kPruneAbs = 4.0e-6f;
// The thresholds are kept (a register per ray, |sd| of a hit in its hit-buffer entry) by the generated kernels compiled
// with pruning and by the library's interpreter kernels
#if defined(RM_JIT_PRUNE_ON) || !defined(RM_JIT_TU)
#define RM_PRUNE_PLUMBING 1
#endif

// ---- Wave-level culling: WHICH units does this wave have to evaluate ----------------------------------------------------
// Rounds 1 and 2 tested every pair of leaves (and every box) against every lane's own threshold, 64 lanes wide, at every
// evaluation: ~120 of the ~330 vector instructions of a march step of the metric scene.  But the 64 rays of a wave come
// from one 8x8-pixel tile and march in step: their positions lie in a small ball.  So the test is TRANSPOSED -- lane u
// looks at unit u (rm_units.h; its bounded stand-in sits in LDS behind the program, RmDecoded::units) and decides for the
// whole wave:
//   p*   the position of the first live lane,  rho >= max over live lanes |p_l - p*|   (one wave reduction)
//   unit u with centre c, outer radius R, inner radius r_in:  for every live lane
//        L_u = |p* - c| - R - rho  <=  value_u(p_l)  <=  |p* - c| + rho - r_in = H_u      (triangle inequality)
// LATTICE programs (threshold rule above): unit u is far for lane l when |p* - c| - R - |p_l - p*| > thr_l; for the whole wave
//   when |p* - c| - R > S = max over live lanes (|p_l - p*| + thr_l) -- one reduction (the first form took rho and T = max thr_l
//   separately: L_u > T).  A CPU simulation of the metric frame (tools/sim/wave_cull_sim.py) has the rho + T form evaluate 4.5 of
//   16 leaves per step where the per-lane pair tests evaluate 5.5 and a per-lane, per-leaf test 4.2.
// Programs that BLEND, top-level chain (rm_units.h).  With a_hi(u) = min over the Union / SmoothUnion units j < u of H_j
//   (the accumulator a unit meets is at most the smallest leaf blended in so far) and a_lo(u) = min_j L_j - kmax (a chain
//   of blends never falls more than the largest k below its smallest leaf: rm_decode.h [*]; Subtraction and Intersection
//   only raise it) -- two prefix minima over the lanes --:
//        Union / SmoothUnion(k)   L_u >= a_hi + k + m   -> the operator returns the accumulator (h = 0): skipped
//                                 H_u <= a_lo - k - m   -> it returns the leaf: RESTART, every unit in front of u is dead
//        Subtraction              L_u + a_lo >= m       -> max(acc, -v) = acc: skipped
//        Intersection             H_u <= a_lo - m       -> max(acc, v) = acc: skipped
//   a_hi is only an upper bound while no Subtraction / Intersection has raised the accumulator: the first one that is not
//   skipped (and any opaque unit, behind which nothing is known) POISONS the rules -- every unit behind it is evaluated.
//   Same simulation, config 3: 4.4 of 16 units per step against 7.2 for the per-lane rule of the first half of this round.
// All comparisons fail on a NaN (the unit is evaluated); m is the margin of the threshold rule at p*, with rho added.
// Cost: ~50 vector instructions per evaluation for a lattice program, ~100 for a blending one, whatever the number of units.
#define RM_DPP(old, src, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), 0xF, 0xF, false))
// largest value over the wave, wave-uniform (all 64 lanes must be active; lanes without a value pass 0).  Four DPP steps
// leave every row of 16 lanes with its maximum (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), two
// row broadcasts carry it on to lane 63 (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; a lane without a source
// gets 0), one v_readlane fetches it.  (The first form read the four rows' maxima with four v_readlane and combined them with
// three s_max_u32: instructions of the 4.3-cycle kind -- a CU has ONE scalar unit for its four SIMDs,
// profiles/r03_ubench_scalar_issue_cycles.txt -- where a DPP step is an ordinary vector instruction.)
RM_DEV uint32_t wave_max_u32(uint32_t v) {
    uint32_t t;
    t = RM_DPP(0u, v, 0xB1); v = t > v ? t : v;
    t = RM_DPP(0u, v, 0x4E); v = t > v ? t : v;
    t = RM_DPP(0u, v, 0x141); v = t > v ? t : v;
    t = RM_DPP(0u, v, 0x140); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum over the lanes BELOW this one (+inf for lane 0); all 64 lanes active.  row_shr:1,2,3, row_shr:4, row_shr:8 scan a row
// of 16, row_bcast:15 and row_bcast:31 carry the rows' totals on (a lane without a source keeps its value; min is idempotent, so
// it does not matter that a total may be counted twice); then one shift by a lane across the whole wave (wave_shr:1).
// Written as v_min_f32_dpp: through __builtin_fminf and update_dpp every step was four instructions (the identity moved in, the
// DPP move, a canonicalising v_max, the v_min) and the last shift a ds_bpermute -- 65 vector instructions for the two scans of
// a blending chain's mask where 16 do.  (v_min_f32 drops a NaN operand like fminf does; s_nop 1: a DPP source written by the
// previous vector instruction needs two wait states, and the compiler does not look inside an asm.)
RM_DEV float wave_exclusive_min(float x) {
    float s = x;
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(x));
    asm volatile("v_min_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(x));
    asm volatile("v_min_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(x));
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(s));
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(s));
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(s));
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(s));
    float up = __uint_as_float(0x7F800000u);
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(up) : "v"(s));
    return up;
}
struct WaveBall {  // wave-uniform: the live lanes' positions lie within rho of (px, py, pz)
    float px, py, pz, rho;
};
RM_DEV WaveBall wave_ball(float x, float y, float z, bool is_live, unsigned long long live_m) {
    WaveBall b;
    const int first = __builtin_ctzll(live_m);  // (the caller has a live lane)
    b.px = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), first));
    b.py = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), first));
    b.pz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(z), first));
    const float dx = x - b.px, dy = y - b.py, dz = z - b.pz;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    // (non-negative floats order like their bit patterns; a NaN's pattern is above +inf's and wins: rho = NaN, nothing is far)
    const float r2 = __uint_as_float(wave_max_u32(is_live ? __float_as_uint(d2) : 0u));
    b.rho = __builtin_amdgcn_sqrtf(r2) * 1.00001f + 1.0e-30f;
    return b;
}
// The unit table in LDS, one ROW per field so that lane u's reads do not collide with its neighbours' (unit records are 32
// bytes apart in device memory; a workgroup transposes them when it stages the program): row f of n_units floats holds
// RmDecoded::units[u].p[f] -- 0..2 centre, 3 outer radius, 4 inner radius, 5 blend radius, 6 kind; row 7 the record's first word (the
// records the unit stands for, first | last << 16).
constexpr uint32_t kUnitRows = 8u;
RM_DEV void stage_units(uint32_t* lunits, const RmRecord* __restrict__ units, uint32_t n_units, uint32_t tid, uint32_t n_threads) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(units);
    for (uint32_t k = tid; k < kUnitRows * n_units; k += n_threads) {
        const uint32_t f = k / n_units, u = k - f * n_units;
        lunits[k] = src[8u * u + ((f + 1u) & 7u)];
    }
}
struct UnitBounds { float L, H, k; uint32_t kind; };
RM_DEV UnitBounds unit_bounds(const uint32_t* lunits, uint32_t n_units, const WaveBall& b) {
    const uint32_t lane = threadIdx.x & 63u;
    const float* u = reinterpret_cast<const float*>(lunits) + (lane < n_units ? lane : 0u);
    const float dx = u[0] - b.px, dy = u[n_units] - b.py, dz = u[2u * n_units] - b.pz;
    const float d = __builtin_amdgcn_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    UnitBounds r;
    r.L = (d * 0.999998f - u[3u * n_units]) - b.rho;  // v_sqrt_f32 is within an ulp, the fused sum of squares within 3e-7
    r.H = (d * 1.000002f + b.rho) - u[4u * n_units];
    r.k = u[5u * n_units];
    r.kind = __float_as_uint(u[6u * n_units]);
    return r;
}
// (Measured and dropped: a mask good for TWO march steps -- a step moves a ray by at most thr and the next threshold is at most
// 2.0001 thr + m', so "|p* - c| - R - (rho + T) > 2.0001 T + m'" holds at both -- halves the cost of the masks and nearly
// doubles the leaves evaluated, 3.6 -> 6.8 of 16 per step on the metric scene: march kernel 0.544 -> 0.613 ms,
// profiles/r03_wave_level_culling_ab.txt.)
RM_DEV unsigned long long wave_cull_lattice(const uint32_t* lunits, uint32_t n_units, float x, float y, float z, float thr, bool is_live,
                                            unsigned long long live_m) {
    const unsigned long long valid = n_units >= 64u ? ~0ull : ((1ull << n_units) - 1ull);
    // ONE reduction: unit u is far for lane l when value_u(p_l) >= |p* - c| - R - |p_l - p*| > thr_l, i.e. for every live lane when
    // |p* - c| - R > S = max over live lanes (|p_l - p*| + thr_l) -- sharper than rho + T (a maximum of sums, not a sum of maxima)
    // and a wave reduction less.  thr >= 0 or NaN; a NaN's bit pattern is above +inf's and wins the maximum: nothing is far then.
    const int first = __builtin_ctzll(live_m);  // (the caller has a live lane)
    const float px = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), first));
    const float py = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(y), first));
    const float pz = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(z), first));
    const float ex = x - px, ey = y - py, ez = z - pz;
    const float e2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
    const float sl = (__builtin_amdgcn_sqrtf(e2) * 1.00001f + 1.0e-30f) + thr;  // v_sqrt_f32 is within an ulp
    const float S = __uint_as_float(wave_max_u32(is_live ? __float_as_uint(sl) : 0u)) * 1.000005f;
    // far: |p* - c| - R > S, decided on the squares (every term is >= 0; a NaN or an infinity makes it false)
    const uint32_t lane = threadIdx.x & 63u;
    const float* u = reinterpret_cast<const float*>(lunits) + (lane < n_units ? lane : 0u);
    const float dx = u[0] - px, dy = u[n_units] - py, dz = u[2u * n_units] - pz;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float s = S + u[3u * n_units];
    return ~__builtin_amdgcn_ballot_w64(d2 > (s * s) * 1.000004f) & valid;
}
// margin_scale: scene_scale + |ro|_1 (the `prune_scale` of the kernels)
RM_DEV unsigned long long wave_cull_blend(const uint32_t* lunits, uint32_t n_units, float kmax, float x, float y, float z, float extra_margin,
                                          float margin_scale, bool is_live, unsigned long long live_m) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long valid = n_units >= 64u ? ~0ull : ((1ull << n_units) - 1ull);
    const WaveBall b = wave_ball(x, y, z, is_live, live_m);
    const float m = kPruneAbs * (margin_scale + ((__builtin_fabsf(b.px) + __builtin_fabsf(b.py)) + __builtin_fabsf(b.pz)) + 2.0f * b.rho) + extra_margin;
    const UnitBounds u = unit_bounds(lunits, n_units, b);
    const bool in_range = lane < n_units;
    const float inf = __uint_as_float(0x7F800000u);
    // (the lane predicates are combined with & and |, not && and ||: short-circuit evaluation of a lane predicate is divergent control
    // flow -- an exec mask saved, narrowed and restored per operator, a dozen scalar instructions where one s_and_b64 does)
    const bool blends = in_range & ((u.kind == RM_UNIT_UM) | (u.kind == RM_UNIT_START));
    const float a_hi = wave_exclusive_min(blends ? u.H : inf);
    const float a_lo = wave_exclusive_min(blends ? u.L : inf) - kmax;
    const bool is_um = u.kind == RM_UNIT_UM, is_sub = u.kind == RM_UNIT_SUB, is_int = u.kind == RM_UNIT_INTER;
    const bool far_um = u.L >= (a_hi + u.k) + m, far_sub = u.L + a_lo >= m, far_int = u.H <= a_lo - m;  // (false on a NaN)
    const bool skip = (is_um & far_um) | (is_sub & far_sub) | (is_int & far_int);
    const bool restart = is_um & (u.H <= (a_lo - u.k) - m);
    const unsigned long long opaque_m = __builtin_amdgcn_ballot_w64(u.kind == RM_UNIT_OPAQUE) & valid;
    const unsigned long long before_opaque = opaque_m ? ((1ull << __builtin_ctzll(opaque_m)) - 1ull) : ~0ull;
    const unsigned long long restart_m = __builtin_amdgcn_ballot_w64(restart) & valid & before_opaque;
    const unsigned long long skip_m = __builtin_amdgcn_ballot_w64(skip) & valid & before_opaque;
    const uint32_t r = restart_m ? 63u - (uint32_t)__builtin_clzll(restart_m) : 0u;
    const unsigned long long from_r = ~((1ull << r) - 1ull);  // units r .. 63
    // the first unit at or behind the restart that may have raised the accumulator, or behind which nothing is known
    const unsigned long long poison_m = (opaque_m | (__builtin_amdgcn_ballot_w64((is_sub | is_int) & !skip) & valid)) & from_r;
    const unsigned long long behind_poison = poison_m ? ~((2ull << __builtin_ctzll(poison_m)) - 1ull) : 0ull;  // (2 << 63 = 0: none)
    return valid & from_r & (~skip_m | behind_poison | (1ull << r));
}
// Tree programs under the interpreter (rm_interp.h map_scene_tree_masked): from the wave's unit mask to the RECORDS the wave has
// to execute.  Lane r looks at records r and r + 64 (RmDecoded::tree, staged like the unit table: one row per field), L and R
// being the units of the record's operands:
//   a leaf is kept iff its unit is needed; an operator record iff both operands hold a needed leaf -- with one of them all +inf a
//   Union or the right side of a Subtraction is the identity on the other: dropped, with the operand's records;
//   a fused leaf whose left operand has gone (Union only, see below) pushes instead of combining: as_push;
//   a Subtraction whose LEFT operand has gone while the right one has not is max(+inf, .) = +inf only in name -- nothing would
//   push that +inf.  It FORCES the first leaf of its left operand back into the mask (evaluating a far leaf instead of replacing
//   it is always allowed).  One pass suffices: the forced leaf lies in an operand that was entirely empty, so no other
//   record's decision changes -- operators inside that operand still have an empty side, operators above it had a needed leaf on
//   this side already (the right operand's).
struct TreeKeep { unsigned long long keep0, keep1, push0, push1; };
RM_DEV TreeKeep tree_keep(const uint32_t* ltree, uint32_t n_rec, unsigned long long need) {
    const uint32_t lane = threadIdx.x & 63u;
    TreeKeep k{0ull, 0ull, 0ull, 0ull};
    const bool two = n_rec > 64u;  // wave-uniform
    unsigned long long Lm[2], Rm[2];
    uint32_t info[2];
    bool in[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t r = lane + 64u * (uint32_t)h;
        in[h] = r < n_rec;
        const uint32_t* t = ltree + (in[h] ? r : 0u);
        if (h == 0 || two) {
            Lm[h] = (unsigned long long)t[0] | ((unsigned long long)t[n_rec] << 32);
            Rm[h] = (unsigned long long)t[2u * n_rec] | ((unsigned long long)t[3u * n_rec] << 32);
            info[h] = t[4u * n_rec];
        } else {
            Lm[h] = Rm[h] = 0ull;
            info[h] = 0u;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && !two) break;
        unsigned long long fm = __builtin_amdgcn_ballot_w64(in[h] && rm_tree_forces(Lm[h], Rm[h], info[h], need));
        while (fm != 0ull) {  // rare
            const int l = __builtin_ctzll(fm);
            fm &= fm - 1ull;
            need |= 1ull << rm_tree_forced_unit((uint32_t)__builtin_amdgcn_readlane((int)info[h], l));
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && !two) break;
        const bool keep = in[h] && rm_tree_keeps(Lm[h], Rm[h], info[h], need);
        const unsigned long long km = __builtin_amdgcn_ballot_w64(keep), pm = __builtin_amdgcn_ballot_w64(keep && rm_tree_pushes(Lm[h], info[h], need));
        if (h == 0) { k.keep0 = km; k.push0 = pm; } else { k.keep1 = km; k.push1 = pm; }
    }
    return k;
}
RM_DEV void stage_tree(uint32_t* ltree, const RmRecord* __restrict__ tree, uint32_t n_tree, uint32_t tid, uint32_t n_threads) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tree);
    for (uint32_t k = tid; k < 5u * n_tree; k += n_threads) {
        const uint32_t f = k / n_tree, r = k - f * n_tree;
        ltree[k] = src[8u * r + 1u + f];
    }
}
#undef RM_DPP

#if !defined(RM_JIT_TU)  // diagnostics (rm_selftest_wave)
__global__ __launch_bounds__(64) void rm_selftest_wave_kernel(const float* in, float* out, uint32_t n_waves) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const float x = in[i];
    out[i] = __uint_as_float(wave_max_u32(__float_as_uint(x)));
    out[64u * n_waves + i] = wave_exclusive_min(x);
}
#endif

// Is unit u (a compile-time constant in generated code) in the wave's mask.  The mask goes through an empty asm at every use:
// left alone the compiler computes all the tests where the mask is born and keeps sixteen 64-bit lane masks alive across the
// whole evaluation -- 30 more scalar registers spilled, which costs the vector register that decides between 6 and 5 waves
// per SIMD.  With the asm each test is born where it is used: s_bitcmp1_b64 + s_cbranch_scc.
RM_DEV bool unit_needed(unsigned long long need, uint32_t u) {
#if defined(RM_UNIT_TEST_FORM) && RM_UNIT_TEST_FORM == 2  // A/B (RM_JIT_UNIT_TEST): the plain test
    return ((need >> u) & 1ull) != 0ull;
#elif defined(RM_UNIT_TEST_FORM) && RM_UNIT_TEST_FORM == 0  // A/B: the whole mask through the asm (s_mov_b64 + s_and_b32 + s_cmp_eq_u64)
    asm volatile("" : "+s"(need));
    return ((need >> u) & 1ull) != 0ull;
#else  // the half of the mask that holds the bit: s_mov_b32 + s_bitcmp1_b32
    uint32_t w = (uint32_t)(need >> (u & 32u));
    asm volatile("" : "+s"(w));
    return ((w >> (u & 31u)) & 1u) != 0u;
#endif
}

// A group of units behind one test (rm_jit.h): the 32-bit word of the mask that holds unit u, through the asm once for the group;
// the tests inside the group read it as it is (s_bitcmp1_b32)
RM_DEV uint32_t unit_word(unsigned long long need, uint32_t u) {
    uint32_t w = (uint32_t)(need >> (u & 32u));
    asm volatile("" : "+s"(w));
    return w;
}
RM_DEV bool unit_in_word(uint32_t w, uint32_t u) { return ((w >> (u & 31u)) & 1u) != 0u; }

typedef float lds_f4 __attribute__((ext_vector_type(4)));
typedef float lds_f2 __attribute__((ext_vector_type(2)));
RM_DEV lds_f4 lds_load4(LdsF r) { return *reinterpret_cast<const __attribute__((address_space(3))) lds_f4*>(r); }
RM_DEV lds_f2 lds_load2(LdsF r) { return *reinterpret_cast<const __attribute__((address_space(3))) lds_f2*>(r); }
// Parameter reads of the generated code.  A specialised kernel stages every record ROTATED by one dword -- p[0..6] at
// dwords 0..6, the opcode (which generated code never reads) at dword 7 -- so that the parameters of a record start on
// a 32-byte boundary and are fetched with ONE ds_read_b128 (+ a ds_read_b64 for a box) instead of two or three
// ds_read2_b32: the LDS pipe of a CU serves a wave64 b32 / b64 read in 2 cycles and a read2_b32 / b128 in 4
// (MI355X_MICROARCH.md, LDS), it is shared by the four SIMDs, and at ~40 parameter reads per evaluation it was ~60 %
// busy (SQ_ACTIVE_INST_LDS) next to a vector unit at ~80 %.
RM_DEV float spec_sphere_a(LdsF r, float qx, float qy, float qz) {
    const lds_f4 p = lds_load4(r);  // cx cy cz r
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    return (dx * dx + dy * dy) + dz * dz;  // the argument of sdf_sphere_t's sqrt, same operations
}
template <bool FAST>
RM_DEV float spec_sphere_v(LdsF r, float a, SqrtGuard& tiny) { return sqrt_sel<FAST>(a, tiny) - r[3]; }
// Subtracted leaves (mode SUB fused into the leaf): max(acc, -v) = acc whenever -v <= acc, in particular at every position
// OUTSIDE the leaf (v > 0) while acc >= 0.  Sphere: a > (r k)^2 (r k: RmRecord::p[4], k = 1.000005) puts sqrt(a) above r by
// more than its rounding, so the computed v = sqrt(a) - r is > 0; NaN fails the comparison and the leaf is evaluated.
RM_DEV bool spec_sub_sphere_near(unsigned long long live, LdsF r, float a, float acc) {
    const float rk = r[4];
    return ((__builtin_amdgcn_ballot_w64(!(a > rk * rk)) | __builtin_amdgcn_ballot_w64(!(acc >= 0.0f))) & live) != 0ull;
}
// Box: a = |max(q, 0)|^2 > 0 means some q_i > 0: the inside term is +0 and v = sqrt(a) > 0.
RM_DEV bool spec_sub_box_near(unsigned long long live, float a, float acc) {
    return ((__builtin_amdgcn_ballot_w64(!(a > 0.0f)) | __builtin_amdgcn_ballot_w64(!(acc >= 0.0f))) & live) != 0ull;
}
struct SpecBox { float qx, qy, qz, a; };
RM_DEV SpecBox spec_box_a(LdsF r, float px, float py, float pz) {
    const lds_f4 c = lds_load4(r);       // cx cy cz rx
    const lds_f2 h = lds_load2(r + 4);   // ry rz
    SpecBox b;
    b.qx = __builtin_fabsf(px - c.x) - c.w;
    b.qy = __builtin_fabsf(py - c.y) - h.x;
    b.qz = __builtin_fabsf(pz - c.z) - h.y;
    const float mx = fmax_(b.qx, 0.0f), my = fmax_(b.qy, 0.0f), mz = fmax_(b.qz, 0.0f);
    b.a = (mx * mx + my * my) + mz * mz;  // as sdf_box_t
    return b;
}
template <bool FAST>
RM_DEV float spec_box_v(const SpecBox& b, SqrtGuard& tiny) {
    return sqrt_sel<FAST, true>(b.a, tiny) + fmin_(fmax_(b.qx, fmax_(b.qy, b.qz)), 0.0f);
}
// Leaves and the one operator with a parameter, as the generated code calls them: `r` points at the
// record's parameters in LDS (wave-uniform address, constant offset: a broadcast read).
template <bool FAST>
RM_DEV float spec_sphere(LdsF r, float qx, float qy, float qz, SqrtGuard& tiny) {
    const lds_f4 c = lds_load4(r);
    const float p[7] = {c.x, c.y, c.z, c.w, 0.0f, 0.0f, 0.0f};
    return sdf_sphere_t<FAST>(qx, qy, qz, p, tiny);
}
template <bool FAST>
RM_DEV float spec_box(LdsF r, float qx, float qy, float qz, SqrtGuard& tiny) {
    const lds_f4 c = lds_load4(r);
    const lds_f2 h = lds_load2(r + 4);
    const float p[7] = {c.x, c.y, c.z, c.w, h.x, h.y, 0.0f};
    return sdf_box_t<FAST>(qx, qy, qz, p, tiny);
}
template <bool FAST>
RM_DEV float spec_cylinder(LdsF r, float qx, float qy, float qz, SqrtGuard& tiny) {
    const lds_f4 c = lds_load4(r);
    const float p[7] = {c.x, c.y, c.z, c.w, r[4], 0.0f, 0.0f};
    return sdf_cylinder_t<FAST>(qx, qy, qz, p, tiny);
}
RM_DEV float spec_plane(LdsF r, float qx, float qy, float qz) {
    const lds_f4 n = lds_load4(r);
    return ((qx * n.x + qy * n.y) + qz * n.z) + n.w;  // as exec_command
}
// k of a SmoothUnion record as a SCALAR: every lane reads the same LDS word, but the compiler cannot know, and "if (k > 0)" on a vector
// register is divergent control flow (exec saved, narrowed, restored: a dozen scalar instructions per blend, at 4.3 cycles each)
RM_DEV float uniform_k(LdsF r) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r[0]))); }
RM_DEV float spec_smooth_union(LdsF r, float a, float b, unsigned long long live) {  // as exec_command, RM_MODE_SMOOTH
    const float kk = uniform_k(r);
    float v = fmin_(a, b);
    if (kk > 0.0f) {
        const float t = kk - __builtin_fabsf(a - b);
        // Outside the blend zone (t <= 0, or NaN) h is 0 and, k being finite, the result is v - 0 = v exactly.  When
        // that holds for every live lane of the wave -- most evaluations: two operands are within k of each other
        // only near the seams -- the division and the five operations after it are skipped (dead lanes' values are
        // never used).
        if (kk < __uint_as_float(0x7F800000u) && (__builtin_amdgcn_ballot_w64(t > 0.0f) & live) == 0ull) return v;
        const float h = fmax_(t, 0.0f) / kk;
        v = v - ((h * h) * kk) * 0.25f;
    }
    return v;
}

// Pins the running minimum / maximum of a SqrtGuard at this point of the generated code.  The guard folds the bit pattern
// of every sqrt argument of an evaluation into two integers; integer min / max may be reassociated, and the compiler
// rebuilt the sequential updates of the four-tap function (64 arguments for a 16-leaf program) into trees that combine
// the FIRST arguments LAST -- every argument stayed live to the end: +55 VGPRs, the difference between 6 and 3-4 waves
// per SIMD for the smooth-min and material scenes (round 1 blamed the four copies of SmoothUnion's division and kept such
// programs on one-position taps).  Measured on the smooth-min scene at 4K: 15.6 ms without the fence, 11.6 ms with it.
RM_DEV void guard_fence(SqrtGuard& g) { asm("" : "+v"(g.lo), "+v"(g.hi)); }

// The same operator applied to the four tap values of a hit (map_scene_taps): ONE wave-uniform "is any live lane of any
// tap inside the blend zone" test for the four.  A lane outside the zone has h = 0 and gets v - 0 = v: the value the
// skipping form returns.
RM_DEV void spec_smooth_union4(LdsF r, const float (&a)[4], const float (&b)[4], unsigned long long live, float (&out)[4]) {
    const float kk = uniform_k(r);
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[i] = fmin_(a[i], b[i]);
        t[i] = kk - __builtin_fabsf(a[i] - b[i]);
    }
    if (!(kk > 0.0f)) return;
    const unsigned long long in_zone = (__builtin_amdgcn_ballot_w64(t[0] > 0.0f) | __builtin_amdgcn_ballot_w64(t[1] > 0.0f)) |
                                       (__builtin_amdgcn_ballot_w64(t[2] > 0.0f) | __builtin_amdgcn_ballot_w64(t[3] > 0.0f));
    if (kk < __uint_as_float(0x7F800000u) && (in_zone & live) == 0ull) return;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float h = fmax_(t[i], 0.0f) / kk;
        out[i] = out[i] - ((h * h) * kk) * 0.25f;
        __builtin_amdgcn_sched_barrier(0);  // one correctly rounded division in flight: interleaved, the four need ~60 more VGPRs
    }
}

constexpr uint32_t V5_RQ = 64u;   // ready buffer entries per wave (refilled only when empty)
constexpr uint32_t V5_SQ = 64u;   // miss buffer entries per wave
constexpr uint32_t V5_HQ = 128u;  // hit buffer entries per wave (64 are taken whenever 64 are waiting)
// per wave: the three buffers, the screen coordinates of the tile's 8 columns and 8 rows, and LAST the partial normals of a tap phase
// that takes its four taps one at a time or is followed by a material phase -- a generated kernel with the four-tap function and no
// materials leaves those 768 bytes away (RmLaunch::wave_dwords): 25.6 -> 22.5 KB per workgroup for the metric scene, 7 instead of 6
// workgroups per CU
constexpr uint32_t V5_TN_DWORDS = 3u * 64u;
constexpr uint32_t V5_WAVE_DWORDS = 4u * (V5_RQ + V5_SQ + V5_HQ) + 16u + V5_TN_DWORDS;

// Miss-test tables of a program, built per workgroup in LDS from the decoded records (the
// decoder stores each primitive's slot within its kind in RmRecord::p[6]):
//   cone[n_cone]  one float4 per sphere: (m.x, m.y, m.z, s); a ray clears it iff m.d - s < 0
//   slab[n_slab]  three float4 per box / cylinder: lo - o and hi - o of the bounding box inflated by
//                 the margin, then the cone of that box's bounding sphere (a cheap pre-test: only
//                 if some lane's ray enters that cone does the wave run the 6-multiply slab test)
//   veto          bit 0: something is non-finite, nothing may be culled; bit 1: the program has a Plane, the tables cannot
//                 clear a ray (the miss test on lower bounds still can)
struct CullTables {
    const float4* cone;
    const float4* slab;
    const uint32_t* veto;
    uint32_t n_cone, n_slab;
};

// `slack`: how far SmoothUnion operators can pull the tree value below the minimum over its leaves
// (each lowers min(a,b) by at most k/4; nested operators add up).  0 for reference-only programs.
RM_DEV float cull_margin(float cx, float cy, float cz, float rho, const V4& ro, float min_dist, float slack) {
    const float scale = 1.0f + __builtin_fabsf(cx) + __builtin_fabsf(cy) + __builtin_fabsf(cz) + rho +
                        __builtin_fabsf(ro.x) + __builtin_fabsf(ro.y) + __builtin_fabsf(ro.z) + slack;
    return (fmax_(min_dist, 0.0f) + slack) * 1.01f + 1.0e-4f * scale;
}

// Table entry (or entries) of one record; no-op for operators.  Called once per record per workgroup.
RM_DEV void cull_build_v5(const RmRecord& rec, const V4& ro, float min_dist, float slack, float4* cone, float4* slab,
                          uint32_t* veto, const float4* bounds = nullptr) {
    uint32_t kind = RM_OP_KIND(rec.op);
    if (kind == RM_KIND_POP || kind == RM_KIND_XFORM) return;
    if (rec.op & RM_OP_NOCULL) {  // subtracted: a hit needs the left operand's surface, whatever this one does
        // (a program with transforms keeps a cone slot per bounded primitive: this one's clears every ray and every pixel)
        if (bounds && kind != RM_KIND_PLANE) cone[__float_as_uint(rec.p[6])] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        return;
    }
    if (kind == RM_KIND_PLANE) { atomicOr(veto, 2u); return; }  // unbounded primitive: the tables cannot clear anything (bit 1)
    const float inf = __uint_as_float(0x7F800000u);
    const uint32_t slot = __float_as_uint(rec.p[6]);
    float cx = rec.p[0], cy = rec.p[1], cz = rec.p[2], sphere_r = rec.p[3];
    if (bounds) {  // program with transforms: the host computed a world-space bounding sphere for every bounded primitive
        const float4 b = bounds[slot];
        cx = b.x; cy = b.y; cz = b.z; sphere_r = b.w;
        kind = RM_KIND_SPHERE;
    }
    bool finite = __builtin_fabsf(cx) < inf && __builtin_fabsf(cy) < inf && __builtin_fabsf(cz) < inf &&
                  __builtin_fabsf(ro.x) < inf && __builtin_fabsf(ro.y) < inf && __builtin_fabsf(ro.z) < inf &&
                  __builtin_fabsf(min_dist) < inf;
    if (kind == RM_KIND_SPHERE) {
        const float rho = fmax_(sphere_r, 0.0f);
        finite = finite && rho < inf;
        const float Rk = rho + cull_margin(cx, cy, cz, rho, ro, min_dist, slack);
        const float mx = cx - ro.x, my = cy - ro.y, mz = cz - ro.z;
        const float mm = mx * mx + my * my + mz * mz;
        // s < sqrt(|m|^2 - Rk^2), the slack covering the rounding of this computation and of the
        // per-ray dot product (|error| <= ~4e-7 |m|)
        const float lim = mm * (1.0f - 4.0e-6f) - Rk * Rk * (1.0f + 4.0e-6f);
        float sv = -inf;  // origin inside (or not clearly outside) the inflated sphere: never clear
        if (lim > 0.0f && lim < inf) sv = __builtin_sqrtf(lim) * (1.0f - 1.0e-5f) - 1.0e-5f * __builtin_sqrtf(mm);
        cone[slot] = make_float4(mx, my, mz, sv);
    } else {
        // box: half extents; cylinder (extension): its bounding box (radius, half height, radius)
        const bool cyl = kind == RM_KIND_CYLINDER;
        const float hx = fmax_(rec.p[3], 0.0f), hy = fmax_(rec.p[4], 0.0f), hz = cyl ? hx : fmax_(rec.p[5], 0.0f);
        finite = finite && hx < inf && hy < inf && hz < inf;
        const float M = cull_margin(cx, cy, cz, hx + hy + hz, ro, min_dist, slack);
        slab[3u * slot] = make_float4((cx - hx - M) - ro.x, (cy - hy - M) - ro.y, (cz - hz - M) - ro.z, 0.0f);
        slab[3u * slot + 1u] = make_float4((cx + hx + M) - ro.x, (cy + hy + M) - ro.y, (cz + hz + M) - ro.z, 0.0f);
        // bounding-sphere cone of the inflated box (same construction as for spheres)
        const float Rk = __builtin_sqrtf((hx + M) * (hx + M) + (hy + M) * (hy + M) + (hz + M) * (hz + M)) * (1.0f + 1.0e-5f);
        const float mx = cx - ro.x, my = cy - ro.y, mz = cz - ro.z;
        const float mm = mx * mx + my * my + mz * mz;
        const float lim = mm * (1.0f - 4.0e-6f) - Rk * Rk * (1.0f + 4.0e-6f);
        float sv = -inf;  // origin not clearly outside the bounding sphere: always run the slab test
        if (lim > 0.0f && lim < inf) sv = __builtin_sqrtf(lim) * (1.0f - 1.0e-5f) - 1.0e-5f * __builtin_sqrtf(mm);
        slab[3u * slot + 2u] = make_float4(mx, my, mz, sv);
    }
    if (!finite || !(slack < inf)) atomicOr(veto, 1u);  // bit 0: non-finite data, nothing may be culled by any test
}

// true iff the half-line o + t d (t >= 0) provably stays clear of every primitive's margin zone.
RM_DEV bool ray_misses_scene_v5(const CullTables& T, float dx, float dy, float dz) {
    bool clear = *T.veto == 0u;  // no primitives (empty scene): every ray misses (wgsl:189-191)
    unit_dir(dx, dy, dz);        // the tests are about the half-line, whatever the length of d (see unit_dir)
    for (uint32_t k = 0; k < T.n_cone; k++) {
        const float4 a = T.cone[k];  // wave-uniform address: LDS broadcast
        const float t = __builtin_fmaf(a.z, dz, __builtin_fmaf(a.y, dy, __builtin_fmaf(a.x, dx, -a.w)));
        clear = clear && (t < 0.0f);  // NaN -> not clear
        if ((k & 3u) == 3u && __ballot(clear) == 0ull) return false;
    }
    if (T.n_slab == 0u || __ballot(clear) == 0ull) return clear;
    // 1/d with |d_i| clamped away from 0 so that no 0 * inf = NaN can appear in the slab test
    const float tiny = 1.0e-30f;
    const float ix = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(dx), tiny), dx));
    const float iy = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(dy), tiny), dy));
    const float iz = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(dz), tiny), dz));
    for (uint32_t k = 0; k < T.n_slab; k++) {
        const float4 c = T.slab[3u * k + 2u];
        const float t = __builtin_fmaf(c.z, dz, __builtin_fmaf(c.y, dy, __builtin_fmaf(c.x, dx, -c.w)));
        if (__ballot(clear && !(t < 0.0f)) == 0ull) continue;  // every still-clear ray misses the bounding sphere
        const float4 a = T.slab[3u * k], b = T.slab[3u * k + 1u];
        const float x1 = a.x * ix, x2 = b.x * ix, y1 = a.y * iy, y2 = b.y * iy, z1 = a.z * iz, z2 = b.z * iz;
        const float tn = fmax_(fmin_(x1, x2), fmax_(fmin_(y1, y2), fmin_(z1, z2)));
        const float tf = fmin_(fmax_(x1, x2), fmin_(fmax_(y1, y2), fmax_(z1, z2)));
        clear = clear && !(tf >= fmax_(tn, 0.0f));  // a dropped NaN only widens the interval
        if (__ballot(clear) == 0ull) return false;
    }
    return clear;
}

// ---- Miss test on lower bounds (programs with SmoothUnion; RmDecoded::bound_walk) -------------------------------
// The plain tests clear a ray when it stays farther than the margin from EVERY primitive, and for a program that blends
// the margin carries the blend's reach (RmDecoded::smooth_slack, = k for a chain of SmoothUnion(k)): a ray that passes one
// primitive at 0.1 is marched although nothing else is within 0.5 of it and the tree value never drops below 0.1 there.
// Sharper, and still exact: every operator admitted here is monotone non-decreasing in each operand --
//     min(a, b), max(a, b), smin_k(a, b) = min(a, b) - h^2 k / 4 with h = max(k - |a - b|, 0) / k  (d/da, d/db in [0, 1])
// -- except the right operand of a Subtraction, and max(a, -b) >= a.  So with c_i <= v_i(q) for every position q of the
// half-line (c_i: how close the half-line comes to leaf i), running the PROGRAM on the c_i -- Subtraction keeping its left
// operand -- gives L <= F(q) for every q of the half-line: if L exceeds the hit threshold (plus the same float slack as the
// plain tests), no march position of the ray can register a hit (wgsl:97), whatever the steps, and the ray is shaded as
// a miss.  Bounds along the half-line o + t d, |d| = 1, t >= 0:
//     sphere  |q - c| - r >= dist(half-line, c) - r                     (exact: |m x d| if m.d > 0, else |m|)
//     box     sdf_box >= Chebyshev distance to the box [c - h+, c + h+], h+ = max(h, 0): the smallest inflation delta at
//             which the slab intervals [tn_i - delta / |d_i|, tf_i + delta / |d_i|] and t >= 0 have a common point, i.e.
//             max over i != j of (tn_i - tf_j) / (1 / |d_i| + 1 / |d_j|) and over j of -tf_j |d_j|
// All of it may use fused multiply-adds and approximate reciprocals: these are bounds, not values of the arithmetic
// contract; every bound is lowered by 1e-5 of itself plus 2e-6 of the scene's scale, NaN becomes -3e38 (v_max drops a NaN
// operand, a min over bounds must not).  Runs only for the rays the plain tests could not clear.
// CONE (the pre-pass): (dx, dy, dz) is the unit centre direction of a pixel whose sixteen sample directions e all satisfy
// |e - c| <= rho.  The point at distance t on such a ray is within t rho of the centre ray's, leaves are 1-Lipschitz, and
// beyond T = |m| + R + max(c_i, 0) a leaf of bounding radius R around m is farther than max(c_i, 0) anyway, so
//     c_i(pixel) >= c_i(centre ray) - rho (|m| + R + max(c_i, 0)),
// the same inflation the plain pixel test applies to its cones and slabs.
template <bool CONE, class LoadRecord>
RM_DEV bool ray_misses_by_bounds_v5(const LoadRecord& load, uint32_t n_rec, const V4& ro, float dx, float dy, float dz, float min_dist,
                                    float scale, float rho = 0.0f) {
    if (!CONE) unit_dir(dx, dy, dz);
    const float tiny = 1.0e-18f;  // |d_i| is clamped away from 0: 1 / |d_i| times a coordinate (< 1e12, RmDecoded::bound_walk) stays finite
    const float ax = fmax_(__builtin_fabsf(dx), tiny), ay = fmax_(__builtin_fabsf(dy), tiny), az = fmax_(__builtin_fabsf(dz), tiny);
    const float ix = __builtin_amdgcn_rcpf(__builtin_copysignf(ax, dx)), iy = __builtin_amdgcn_rcpf(__builtin_copysignf(ay, dy));
    const float iz = __builtin_amdgcn_rcpf(__builtin_copysignf(az, dz));
    const float wxy = (ax * ay) * __builtin_amdgcn_rcpf(ax + ay), wxz = (ax * az) * __builtin_amdgcn_rcpf(ax + az);
    const float wyz = (ay * az) * __builtin_amdgcn_rcpf(ay + az);
    const float lower = 2.0e-6f * scale;
    const float inf = __uint_as_float(0x7F800000u);
    float acc = inf, spilled = inf;
    for (uint32_t i = 0u; i < n_rec; i++) {
        uint32_t op;
        float p[7];
        load(i, op, p);
        op = __builtin_amdgcn_readfirstlane(op);
        const uint32_t kind = RM_OP_KIND(op), mode = RM_OP_MODE(op);
        float a, b;
        if (kind == RM_KIND_POP) {
            a = spilled; b = acc;
        } else {
            const float mx = p[0] - ro.x, my = p[1] - ro.y, mz = p[2] - ro.z;
            const float mm = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
            float reach = 0.0f;  // CONE: bounding radius of the leaf
            bool bounded = true;
            if (kind == RM_KIND_PLANE) {
                // dot(q, n) + h is linear along a ray: v0 + t (n.d); over t >= 0 its infimum is v0 when n.d >= 0, else -inf.
                // CONE: every sample direction e has n.e >= n.c - |n| rho.  |n| is arbitrary (the value is not a distance): the
                // slack scales with it.
                const float nn = __builtin_amdgcn_sqrtf(__builtin_fmaf(p[2], p[2], __builtin_fmaf(p[1], p[1], p[0] * p[0]))) * 1.00001f;
                const float v0 = __builtin_fmaf(p[2], ro.z, __builtin_fmaf(p[1], ro.y, p[0] * ro.x)) + p[3];
                const float nd = __builtin_fmaf(p[2], dz, __builtin_fmaf(p[1], dy, p[0] * dx));
                const float slope = CONE ? nd - nn * (rho * 1.00001f) : nd;
                const float away = 1.0e-5f * nn;  // covers the rounding of n.d and of the march positions' own n.q
                b = slope >= away ? v0 - (1.0e-5f * (__builtin_fabsf(v0) + __builtin_fabsf(p[3])) + nn * (4.0f * lower)) : -inf;
                bounded = false;
            } else if (kind == RM_KIND_SPHERE) {
                const float cx = __builtin_fmaf(my, dz, -(mz * dy)), cy = __builtin_fmaf(mz, dx, -(mx * dz)), cz = __builtin_fmaf(mx, dy, -(my * dx));
                const float perp2 = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx));
                reach = fmax_(p[3], 0.0f);
                const float along = __builtin_fmaf(mz, dz, __builtin_fmaf(my, dy, mx * dx));
                const float dist = __builtin_amdgcn_sqrtf(along > 0.0f ? perp2 : mm);
                b = dist * (1.0f - 1.0e-5f) - p[3];
            } else {  // RM_KIND_BOX; RM_KIND_CYLINDER through its bounding box (radius, half height, radius): a solid inside
                      // another is at least as far as that one
                const bool cyl = kind == RM_KIND_CYLINDER;
                const float hx = fmax_(p[3], 0.0f), hy = fmax_(p[4], 0.0f), hz = cyl ? hx : fmax_(p[5], 0.0f);
                reach = (hx + hy) + hz;  // >= |h|
                const float x1 = (mx - hx) * ix, x2 = (mx + hx) * ix, y1 = (my - hy) * iy, y2 = (my + hy) * iy;
                const float z1 = (mz - hz) * iz, z2 = (mz + hz) * iz;
                const float nx = fmin_(x1, x2), fx = fmax_(x1, x2), ny = fmin_(y1, y2), fy = fmax_(y1, y2);
                const float nz = fmin_(z1, z2), fz = fmax_(z1, z2);
                const float d0 = fmax_((nx - fy) * wxy, (ny - fx) * wxy);
                const float d1 = fmax_((nx - fz) * wxz, (nz - fx) * wxz);
                const float d2 = fmax_((ny - fz) * wyz, (nz - fy) * wyz);
                const float d3 = fmax_(-fx * ax, fmax_(-fy * ay, -fz * az));
                // (every product is finite: coordinates < 1e12 -- the caller checks the scale -- times 1 / |d_i| <= 1e18; a NaN
                // term dropped by v_max could only lower the maximum)
                const float delta = fmax_(fmax_(d0, d1), fmax_(d2, d3));
                b = delta * (1.0f - 1.0e-5f);
            }
            if (CONE && bounded) b = b - (rho * 1.00001f) * ((__builtin_amdgcn_sqrtf(mm) * 1.00001f + reach) + fmax_(b, 0.0f));
            b = fmax_(b - (lower + 1.0e-5f * __builtin_fabsf(b)), -3.0e38f);  // NaN -> -3e38
            if (op & RM_OP_SPILL) spilled = acc;
            a = acc;
        }
        if (mode == RM_MODE_PUSH) acc = b;
        else if (mode == RM_MODE_UNION) acc = fmin_(a, b);
        else if (mode == RM_MODE_SUB) acc = a;                      // max(a, -b) >= a
        else if (mode == RM_MODE_INTER) acc = fmax_(a, b);
        else {  // RM_MODE_SMOOTH
            const float kk = p[0];
            float v = fmin_(a, b);
            if (kk > 0.0f) {
                const float h = fmax_(kk - __builtin_fabsf(a - b), 0.0f) * __builtin_amdgcn_rcpf(kk);
                v = v - ((h * h) * kk) * 0.25f;
                v = v - (1.0e-6f * (__builtin_fabsf(v) + kk));
            }
            acc = fmax_(v, -3.0e38f);  // NaN (k = inf) -> -3e38
        }
    }
    const float margin = fmax_(min_dist, 0.0f) * 1.01f + 1.0e-4f * scale;
    return acc > margin;  // NaN -> false
}

// Work list produced by the pre-pass (rm_tile_pre_v5 + rm_tile_sort_v5), one per frame.
struct V5Work {
    const uint32_t* order;  // [n_frames][n_tiles] ids of the tiles that need marching, heaviest first
    uint32_t* counters;     // [n_frames][4]: {number of such tiles, cursor of the persistent workgroups,
                            //                 bits of f0 = map_scene(ro) (the shared first march step), unused}
    uint32_t* measured;     // nullptr, or [n_frames][n_tiles]: how long each marched tile took (10 ns ticks), written here
                            // and read by the NEXT draw's rm_tile_sort_v5 (RM_OPT_BALANCE = 3: longest tiles first)
};

// (Round 3 tried to drop the sort kernel -- the pre-pass appending every tile that needs marching to one of four lists by the
// class of its last duration, one atomic per tile, the march kernel walking the classes: the frame's ~7 500 returning atomics
// on four addresses took the pre-pass from 0.06 to 0.21 ms, and four classes order the tiles worse than the sort's 64 buckets
// (8-node scene: march kernel 0.244 -> 0.277 ms).  How much the order is worth: last frame's durations 0.532 ms, pending
// pixels 0.65, order of arrival 0.66 on the metric frame.  profiles/r03_work_list_by_atomics_negative.txt)
// b for the lanes whose bit is set in the wave mask m, a for the others: one v_cndmask_b32 with the mask as its
// scalar operand.
RM_DEV float select_by_mask(float a, float b, unsigned long long m) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
RM_DEV uint32_t select_by_mask(uint32_t a, uint32_t b, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
// this lane's bit of a wave mask as a predicate
RM_DEV bool lane_of(unsigned long long m, uint32_t lane) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
    (void)lane;
    return r != 0u;
}

struct TrueTag { static constexpr bool value = true; };
struct FalseTag { static constexpr bool value = false; };

// Persistent workgroups: the grid holds about as many workgroups as the chip has room for; each
// takes the next tile of the work list with one atomic and leaves when the list is exhausted
// (no spinning, no inter-workgroup dependency).  Tiles whose 1024 rays are all culled never reach
// this kernel: the pre-pass writes their pixels directly.
// MAT (extension, materials): compiled into the kernels that can meet a tagged program -- the extension
// interpreter and the kernels specialised for one; it adds a fifth phase after the four normal taps of a batch of
// hits (one evaluation of the material program at the hit positions) and a byte per ray next to its result code.
// LOOP (interpreter kernels): 0 the record loop is chosen at run time from the launch's flags; 1 chain, 2 chain over the records the unit
// mask names, 3 tree, 4 tree over the records the mask leaves -- a kernel compiled for ONE loop: no dispatch per evaluation, and above
// all none of the other loops' code and registers (the lean kernel with all four kept 176 scalar registers spilled in vector lanes)
template <class Prog, bool PROG_IN_LDS, int WPT, bool EXT, bool SPEC, bool MAT = false, int LOOP = 0>
RM_DEV void rm_render_v5_body(const RmLaunch& L, const V5Work& work, uint32_t n_tiles, uint32_t refill_min) {
    constexpr uint32_t POOL = 1024u;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // ---- LDS carve-up (all offsets multiples of 16 bytes) ----
    float* res = reinterpret_cast<float*>(smem);                       // [1024] result code per ray (tile)
    uint32_t* wbase = smem + POOL + wave * L.wave_dwords;             // this wave's buffers
    uint32_t* rq_rid = wbase;                                          // ready rays (SoA): id, direction
    float* rq_d = reinterpret_cast<float*>(wbase + V5_RQ);             // [3][V5_RQ]
    uint32_t* sq_rid = wbase + 4u * V5_RQ;                             // rays that ended without a hit: id, direction
    float* sq_v = reinterpret_cast<float*>(sq_rid + V5_SQ);            // [3][V5_SQ]
    uint32_t* hq_rid = sq_rid + 4u * V5_SQ;                            // hits waiting for their normal: id, position
    float* hq_v = reinterpret_cast<float*>(hq_rid + V5_HQ);            // [3][V5_HQ]
    float* wxy = hq_v + 3u * V5_HQ;                                    // [16] pt_screen.x of the tile's 8 columns, .y of its 8 rows (wgsl:41-43)
    float* tn = wxy + 16u;                                             // [3][64] partial normals of the running tap phase (absent when L.wave_dwords says so)
    uint32_t* after = smem + POOL + WPT * L.wave_dwords;
    float* spill = reinterpret_cast<float*>(after) + wave * (L.spill_depth * 64u) + lane;  // [WPT][depth][64]
    float4* t_cone = reinterpret_cast<float4*>(after + WPT * L.spill_depth * 64u);         // [n_cone]
    float4* t_slab = t_cone + L.n_cone;                                                     // [2 * n_slab]
    uint32_t* lprog = reinterpret_cast<uint32_t*>(t_slab + 3u * L.n_slab);
    uint32_t* s_next = lprog + (PROG_IN_LDS ? (L.n_rec + L.n_grp + L.n_tree) * 8u : 0u);  // shared pool cursor
    uint32_t* s_tile = s_next + 1;                                     // work-list slot of the current tile
    uint32_t* s_veto = s_next + 2;
    float* s_off = reinterpret_cast<float*>(s_next + 4);               // [16][2] screen offsets of the AA samples (wgsl:47-53)
    uint8_t* rmat = reinterpret_cast<uint8_t*>(s_next + 36);           // MAT, tagged program: [1024] material per ray
    const bool tagged = MAT && L.n_mrec != 0u;                         // wave-uniform
    constexpr uint32_t TAP_IDLE = MAT ? 5u : 4u;                       // tap_t: 0..3 normal taps, 4 (MAT) material, else idle
    const CullTables cullt{t_cone, t_slab, s_veto, L.n_cone, L.n_slab};

    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];  // wave-uniform
    const uint32_t tiles_x = (L.W + 7u) / 8u;
    const uint32_t* order = work.order + (size_t)blockIdx.z * n_tiles;
    uint32_t* counters = work.counters + 4u * blockIdx.z;
    const uint32_t n_active = counters[0];

    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);  // wgsl:39-40
    const float prune_scale = L.scene_scale + ((__builtin_fabsf(ro.x) + __builtin_fabsf(ro.y)) + __builtin_fabsf(ro.z));  // "Pruning"
    const float eps = 0.0001f;                                    // wgsl:136
    if (PROG_IN_LDS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.prog);
        // program (generated code reads the records rotated by one dword: parameters first, lds_load4), then the unit table
        for (uint32_t k = tid; k < L.n_rec * 8u; k += 64u * WPT) lprog[SPEC ? ((k & ~7u) | ((k + 7u) & 7u)) : k] = src[k];
        stage_units(lprog + 8u * L.n_rec, L.prog + L.n_rec, L.n_grp, tid, 64u * WPT);  // (7 of a unit record's 8 dwords: it fits the same space)
        if constexpr (!SPEC) stage_tree(lprog + 8u * (L.n_rec + L.n_grp), L.prog + L.n_rec + L.n_grp, L.n_tree, tid, 64u * WPT);
    }
    if (tid == 0u) *s_veto = 0u;
    if (tid < 16u) sample_offset(u, tid >> 2, tid & 3u, s_off[2u * tid], s_off[2u * tid + 1u]);  // two divisions per sample, once
    __syncthreads();
    if (L.flags & 1u)
        for (uint32_t k = tid; k < L.n_rec; k += 64u * WPT) cull_build_v5(L.prog[k], ro, L.min_dist, L.smooth_slack, t_cone, t_slab, s_veto, L.bounds);

    Prog prog;
    if constexpr (PROG_IN_LDS) prog.base = lprog;
    else prog.base = L.prog;

    // record i of the program as (opcode, parameters), wherever this kernel keeps it (SPEC: the rotated LDS copy)
    auto load_record = [&](uint32_t i, uint32_t& op, float (&p)[7]) {
        if constexpr (PROG_IN_LDS) {
            const uint32_t* r = lprog + 8u * i;
            op = r[SPEC ? 7u : 0u];
#pragma unroll
            for (int k = 0; k < 7; k++) p[k] = __uint_as_float(r[(SPEC ? 0u : 1u) + (uint32_t)k]);
        } else {
            prog.load(i, op, p);
        }
    };
    const float bound_scale = prune_scale + L.smooth_slack;  // scale of the coordinates the bounds are computed from
    const LdsF lprog_v = lds_vector_base(lprog);  // SPEC: the program's LDS copy, base address in a VGPR (see LdsF)
    uint32_t n_eval = 0u;  // diagnostics (pruned kernels compiled with statistics): leaves actually evaluated, per wave
    // Wave-level culling (above): the units of the program this wave has to evaluate at its live lanes' positions.  Generated
    // kernels know at compile time whether and how they cull; the interpreter kernels look at the launch (flags bit 3).
    const uint32_t* lunits = lprog + 8u * L.n_rec;  // the unit records behind the program (PROG_IN_LDS kernels only)
    auto units_needed = [&](float x, float y, float z, float thr, float extra_margin, bool is_live, unsigned long long live_mask) -> unsigned long long {
        if constexpr (SPEC) {
#if defined(RM_JIT_PRUNE_ON)
            return wave_cull_lattice(lunits, L.n_grp, x, y, z, thr, is_live, live_mask);
#elif defined(RM_JIT_BLEND_PRUNE)
            return wave_cull_blend(lunits, L.n_grp, L.unit_kmax, x, y, z, extra_margin, prune_scale, is_live, live_mask);
#else
            return ~0ull;
#endif
        } else if constexpr (LOOP == 2 || LOOP == 4) {  // (the masked loops are for lattice programs)
            return wave_cull_lattice(lunits, L.n_grp, x, y, z, thr, is_live, live_mask);
        } else if constexpr (LOOP == 1 || LOOP == 3) {
            return ~0ull;
        } else if constexpr (PROG_IN_LDS) {
            if ((L.flags & 8u) == 0u) return ~0ull;
            if (L.unit_mode == RM_UNITS_LATTICE) return wave_cull_lattice(lunits, L.n_grp, x, y, z, thr, is_live, live_mask);
            return wave_cull_blend(lunits, L.n_grp, L.unit_kmax, x, y, z, extra_margin, prune_scale, is_live, live_mask);
        } else {
            return ~0ull;
        }
    };
    // map_scene (wgsl:187-203) at one point per lane
    auto eval_scene = [&](float x, float y, float z, float thr, bool is_live, unsigned long long live_mask) -> float {
        float qx[1] = {x}, qy[1] = {y}, qz[1] = {z}, v[1];
        SqrtGuard tiny;
        if constexpr (SPEC) {  // straight-line code compiled for this program's structure (rm_jit.h)
            const unsigned long long need = units_needed(x, y, z, thr, 0.0f, is_live, live_mask);
            v[0] = map_scene_spec<true>(lprog_v, x, y, z, need, live_mask, tiny, n_eval);
            if (tiny.any_bad()) {
                // (an empty volatile asm keeps this a branch: for a program of one or two leaves the compiler otherwise
                // evaluates both forms at every step and selects -- a second square root and 16 more vector instructions)
                asm volatile("");
                uint32_t again = 0u;
                v[0] = map_scene_spec<false>(lprog_v, x, y, z, need, live_mask, tiny, again);
            }
        } else if (LOOP == 1 || LOOP == 2 || (LOOP == 0 && (L.flags & 4u))) {  // chain program (wave-uniform): the stack-free record loop
            const unsigned long long need = units_needed(x, y, z, thr, 0.0f, is_live, live_mask);
            const bool masked = LOOP == 0 ? (PROG_IN_LDS && (L.flags & 8u) != 0u) : LOOP == 2;  // (the unit records are staged in LDS with the program)
#ifdef RM_INTERP_STATS
            n_eval += (uint32_t)__builtin_popcountll(masked ? need : (L.n_rec >= 64u ? ~0ull : (1ull << L.n_rec) - 1ull));
#endif
            v[0] = map_scene_chain<true>(prog, L.n_rec, x, y, z, need, masked, tiny);
            if (tiny.any_bad()) v[0] = map_scene_chain<false>(prog, L.n_rec, x, y, z, need, masked, tiny);
        } else if (LOOP == 3 || LOOP == 4 || (LOOP == 0 && !EXT)) {  // tree program -- reference node types in any arrangement, all a kernel
                                                                      // without the extensions ever gets --: one dispatch per record
            const bool masked = LOOP == 0 ? (PROG_IN_LDS && (L.flags & 8u) != 0u) : LOOP == 4;
            if (masked) {  // ... over the records the wave's unit mask leaves (L.n_tree != 0)
                const unsigned long long need = units_needed(x, y, z, thr, 0.0f, is_live, live_mask);
                const TreeKeep k = tree_keep(lunits + 8u * L.n_grp, L.n_rec, need);
#ifdef RM_INTERP_STATS  // diagnostics build (tools/wave_stats.py --interp-stats): records executed, per wave
                n_eval += (uint32_t)(__builtin_popcountll(k.keep0) + __builtin_popcountll(k.keep1));
#endif
                v[0] = map_scene_tree_masked<true>(prog, spill, x, y, z, k.keep0, k.keep1, k.push0, k.push1, tiny);
                if (tiny.any_bad()) v[0] = map_scene_tree_masked<false>(prog, spill, x, y, z, k.keep0, k.keep1, k.push0, k.push1, tiny);
            } else {
                v[0] = map_scene_tree<true>(prog, L.n_rec, spill, x, y, z, tiny);
                if (tiny.any_bad()) v[0] = map_scene_tree<false>(prog, L.n_rec, spill, x, y, z, tiny);
            }
        } else if (PROG_IN_LDS && (L.flags & 8u)) {  // a blending chain: the general record machine over the units the wave's mask names
            const unsigned long long need = units_needed(x, y, z, thr, 0.0f, is_live, live_mask);
            const uint32_t* ranges = lunits + 7u * L.n_grp;  // row 7 of the unit table
            v[0] = map_scene_units<true, EXT>(prog, ranges, spill, x, y, z, need, tiny, L.value_spill_depth);
            if (tiny.any_bad())  // a sqrt argument outside the fast range (SqrtGuard): redo with the generic sqrt
                v[0] = map_scene_units<false, EXT>(prog, ranges, spill, x, y, z, need, tiny, L.value_spill_depth);
        } else {
            map_scene_multi<1, true, Prog, EXT>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny, L.value_spill_depth);
            if (tiny.any_bad()) map_scene_multi<1, false, Prog, EXT>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny, L.value_spill_depth);
        }
        return v[0];
    };

    // The first march step of EVERY ray of the frame evaluates the same point: pos = ro + rd * 0 (wgsl:88-94).
    // rm_tile_sort_v5 took it once per frame (f0); rays start from its outcome (wgsl:97-114 applied to f0):
    //   START_HIT    f0 < min_dist: the hit is at ro, the ray goes straight to its normal taps
    //   START_DONE   f0 > max_dist, or max_iter == 1: the loop ends without a hit
    //   START_MARCH  dist = 0 + f0, one iteration used
    // (only for finite rd: with a NaN / inf component rd * 0 is NaN and the ray starts the ordinary way)
    enum : uint32_t { START_MARCH = 0u, START_HIT = 1u, START_DONE = 2u };
    const float f0 = __uint_as_float(__builtin_amdgcn_readfirstlane(counters[2]));
    uint32_t start = START_DONE;
    if (L.max_iter != 0u)
        start = f0 < L.min_dist ? START_HIT : (f0 > L.max_dist || L.max_iter <= 1u) ? START_DONE : START_MARCH;
    const float inf_f = __uint_as_float(0x7F800000u);

    const unsigned long long t_start = L.stats ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // (diagnostics, RM_OPT_WAVE_STATS; a generated kernel keeps them only when it was compiled for that: RM_NO_WAVE_STATS, rm_jit.h)
    uint32_t n_iter = 0u, n_live = 0u, n_prod = 0u, n_tiles_done = 0u;

  for (;;) {  // ---- next tile of the work list ----
    if (tid == 0u) {
        *s_tile = atomicAdd(&counters[1], 1u);
        *s_next = 0u;
        s_next[3] = 0u;  // "the per-ray miss tests clear nothing in this tile", see the production of rays below
    }
    __syncthreads();
    const uint32_t slot = *s_tile;
    if (slot >= n_active) break;
    const uint32_t tile = order[slot];
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    n_tiles_done++;
    const unsigned long long tile_t0 = work.measured ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // Ray r of the pool: pixel r & 63, AA sample r >> 6 (the resolve step reads res[] that way).  A produce round makes the 64
    // rays of one BATCH: the sixteen samples of a 2x2 block of pixels (lane: sample lane >> 2, pixel lane & 3 of the block) --
    // rays that stay within two pixels of each other march alike (they finish together: fewer idle lanes) and lie in a small
    // ball (wave-level culling tests the units against that ball).  Round 2's batch was one sample of all 64 pixels.
    if (lane < 16u) {
        const uint32_t k = lane & 7u, tx = tile_x * 8u + k, ty = tile_y * 8u + k;
        wxy[lane] = lane < 8u ? screen_x(tx < L.W ? tx : L.W - 1u, L.W)                                // edge tiles clamp
                              : screen_y(rm_global_row(L, ty < L.rows ? ty : L.rows - 1u), L.H);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // lane state of a marching ray: evaluation point = ro + d * sc
    float dx = 0.f, dy = 0.f, dz = 0.f, sc = 0.f;
    // itr: the ray's id in the pool (bits 0-9) and the march steps it has taken (bits 10 and up; max_iter <= 65536) in ONE
    // register -- a step adds 1024, the loop bound compares with max_iter << 10 -- the one register between 6 and 5 waves per SIMD
    // ... and the lane's state in its top two bits: 0 marching, ITR_EMPTY idle (waiting for a ray), ITR_RETIRED nothing left to take
    constexpr uint32_t ITR_EMPTY = 0x80000000u, ITR_RETIRED = 0x40000000u;  // (a marching ray's itr stays below 2^27)
    uint32_t itr = ITR_EMPTY;
    auto lane_empty = [&]() { return (int32_t)itr < 0; };
    auto lane_marching = [&]() { return itr < ITR_RETIRED; };
    const uint32_t iter_limit = L.max_iter << 10;
    float thr_base = inf_f;  // SPEC: pruning threshold without its position term ("Pruning")
    // (a hit whose normal is being sampled keeps its state -- position, partial normal -- in LDS: tap phase, below)
    uint32_t rq_pos = 0u, rq_cnt = 0u, sq_n = 0u, hq_n = 0u;  // wave-uniform cursors: ready / miss / hit buffers
    uint32_t tap_t = TAP_IDLE, tap_n = 0u;                     // wave-uniform: tap phase step (TAP_IDLE = not in one), its entries
    bool pool_open = true;                                     // wave-uniform: the shared pool may still hold rays

    auto flush_misses = [&]() {  // floor / black for every waiting ray that ended without a hit (<= 64): wgsl:117-130
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < sq_n) res[sq_rid[lane]] = miss_code(ro, sq_v[lane], sq_v[V5_SQ + lane], sq_v[2u * V5_SQ + lane]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        sq_n = 0u;
    };

    for (;;) {
        if (tap_t == TAP_IDLE) {
            // ---- A. idle lanes take ready rays; when the buffer is empty it is refilled with the
            //         survivors of the next 64 candidates of the pool ----
            const unsigned long long want0 = __ballot(lane_empty());
            const unsigned long long live0 = __ballot(lane_marching());
            if (want0 != 0ull && (live0 == 0ull || (uint32_t)__popcll(want0) >= refill_min)) {
                unsigned long long want = want0;
                while (want != 0ull) {
                    if (rq_pos == rq_cnt) {  // buffer empty: produce
                        if (!pool_open) {
                            if (lane_empty()) itr = ITR_RETIRED;  // pool exhausted and buffer drained
                            break;
                        }
                        uint32_t base = 0u;
                        if (lane == 0u) base = atomicAdd(s_next, 64u);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= POOL) { pool_open = false; continue; }
                        n_prod++;
                        // batch -> block of the 4x4 blocks of the tile: the corners first, then the centre (see test_rays below)
                        const uint32_t blk = (uint32_t)(0xEDB87421A965FC30ull >> (4u * (base >> 6))) & 15u;
                        // (the lane's part of these is recomputed per round: hoisted out of the loop it would sit in registers through
                        // the four-tap function, whose pressure decides between 6 and 5 waves per SIMD)
                        uint32_t ln = lane;
                        asm volatile("" : "+v"(ln));
                        const uint32_t bpx = 2u * (blk & 3u) + (ln & 1u), bpy = 2u * (blk >> 2) + ((ln >> 1) & 1u), s = ln >> 2;
                        const uint32_t r = s * 64u + bpy * 8u + bpx;
                        float gx, gy, gz;
                        gen_ray_at(u.inv_proj, u.inv_view, ro, wxy[bpx], wxy[8u + bpy], s_off[2u * s], s_off[2u * s + 1u], gx, gy, gz);
                        const bool finite_d = __builtin_fabsf(gx) < inf_f && __builtin_fabsf(gy) < inf_f && __builtin_fabsf(gz) < inf_f;
                        // The miss tests are optional (a ray they do not clear is marched and ends as the same miss): in a tile
                        // the scene covers completely they clear nothing -- 92 % of the rays that reach this kernel are marched --
                        // so once the batches of the tile's four CORNER blocks (the first four of the pool) went through without
                        // a single ray cleared, the rest of the tile's batches skip them.
                        const bool test_rays = (L.flags & 1u) != 0u && __builtin_amdgcn_readfirstlane(s_next[3]) < 4u;
                        bool culled = L.max_iter == 0u || (start == START_DONE && finite_d) || (test_rays && ray_misses_scene_v5(cullt, gx, gy, gz));
#if !defined(RM_JIT_TU) || defined(RM_JIT_BOUND_WALK)  // a generated kernel carries it only if its program's structure can use it
                        if (test_rays && (L.flags & 32u) && (*s_veto & 1u) == 0u && bound_scale < 1.0e12f && __ballot(!culled) != 0ull)  // "Miss test on lower bounds"
                            culled = culled || ray_misses_by_bounds_v5<false>(load_record, L.n_rec, ro, gx, gy, gz, L.min_dist, bound_scale);
#endif
                        if (test_rays && base < 256u && __ballot(culled) == 0ull && lane == 0u) atomicAdd(&s_next[3], 1u);
                        if (culled) res[r] = miss_code(ro, gx, gy, gz);  // never marched: wgsl:117-130 only
                        const unsigned long long keep = __ballot(!culled);
                        if (!culled) {
                            const uint32_t e = lane_rank(keep);
                            rq_rid[e] = r;
                            rq_d[e] = gx; rq_d[V5_RQ + e] = gy; rq_d[2u * V5_RQ + e] = gz;
                        }
                        rq_pos = 0u;
                        rq_cnt = (uint32_t)__popcll(keep);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        continue;
                    }
                    const uint32_t avail = rq_cnt - rq_pos, n_want = (uint32_t)__popcll(want);
                    if (lane_empty()) {
                        const uint32_t rank = lane_rank(want);
                        if (rank < avail) {
                            const uint32_t e = rq_pos + rank;
                            const uint32_t rid = rq_rid[e];
                            dx = rq_d[e]; dy = rq_d[V5_RQ + e]; dz = rq_d[2u * V5_RQ + e];
                            const bool finite_d = __builtin_fabsf(dx) < inf_f && __builtin_fabsf(dy) < inf_f && __builtin_fabsf(dz) < inf_f;
                            // The shared first step applies to finite rd that neither hit nor ended there (START_MARCH);
                            // everything else takes its own first step: rd * 0 = NaN, or the hit at ro (rare: the camera
                            // inside a solid), which then goes through the ordinary hit path.
                            const bool shared = finite_d && start == START_MARCH;
                            sc = shared ? 0.0f + f0 : 0.0f;  // dist (wgsl:88), dist += scene_dist (wgsl:114)
                            itr = rid | (shared ? 1024u : 0u);
                            thr_base = shared ? __builtin_fabsf(f0) * 2.00002f : inf_f;
                        }
                    }
                    rq_pos += n_want < avail ? n_want : avail;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    want = __ballot(lane_empty());
                }
            }
            // ---- hits waiting for their normal: a tap phase runs when 64 are waiting, or when nothing else is left ----
            const bool any_live = __ballot(lane_marching()) != 0ull;
            if (!any_live && __ballot(lane_empty()) != 0ull) continue;  // idle lanes remain: force a refill round
            if (hq_n >= 64u || (!any_live && hq_n != 0u)) {
                tap_n = hq_n < 64u ? hq_n : 64u;
                hq_n -= tap_n;  // the most recent tap_n entries: [hq_n, hq_n + tap_n); nothing is pushed during the phase
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                tap_t = 0u;
            } else if (!any_live) {
                break;  // every lane retired, nothing waiting
            }
        }
        const bool tapping = tap_t != TAP_IDLE;  // wave-uniform
        if constexpr (MAT) {
            if (tap_t == 4u) {  // the normals of this batch are complete (tn): which material does each hit carry?
                const uint32_t e = hq_n + lane;
                const float hx = hq_v[e], hy = hq_v[V5_HQ + e], hz = hq_v[2u * V5_HQ + e];
#ifdef RM_JIT_MATERIAL_WALK
                SqrtGuard mtiny;
                uint32_t m = map_scene_material_spec<true>(L.mprog, hx, hy, hz, mtiny);
                if (mtiny.any_bad()) m = map_scene_material_spec<false>(L.mprog, hx, hy, hz, mtiny);
#else
                const uint32_t m = map_scene_material(L.mprog, L.n_mrec, spill, L.mat_value_depth, hx, hy, hz);
#endif
                if (lane < tap_n) {
                    const uint32_t r = hq_rid[e] & 1023u;
                    res[r] = shade_hit(tn[lane], tn[64u + lane], tn[128u + lane], hx, hy, hz);  // wgsl:98-103
                    rmat[r] = (uint8_t)m;
                }
                tap_t = TAP_IDLE;
                continue;
            }
        }

#ifdef RM_JIT_TAPS4
        if (tapping) {  // all four normal taps of the batch in one pass over the program (wgsl:135-144)
            const uint32_t e = hq_n + lane;
            const float cx = hq_v[e], cy = hq_v[V5_HQ + e], cz = hq_v[2u * V5_HQ + e];
            const bool live4 = lane < tap_n;
            // |F(tap)| <= |sd_hit| + eps sqrt(3) + error: the threshold of a tap ("Pruning"); the far test runs at c,
            // another eps sqrt(3) and one more leaf-evaluation error away from every tap
            const float m = kPruneAbs * (prune_scale + (((__builtin_fabsf(cx) + __builtin_fabsf(cy)) + __builtin_fabsf(cz)) + 1.0e-3f));
            const float thr_c = __uint_as_float(hq_rid[e] & ~1023u) * 1.00001f + 3.5e-4f + (m + m);
            float f[4];
            SqrtGuard tiny;
            const unsigned long long live4_m = __builtin_amdgcn_ballot_w64(live4);
            // the units are decided at the hit positions c, eps sqrt(3) from every tap: the threshold (lattice programs) or
            // the margin (programs that blend) carries that distance twice over
            const unsigned long long need4 = units_needed(cx, cy, cz, thr_c, 3.5e-4f, live4, live4_m);
            map_scene_taps<true>(lprog_v, cx, cy, cz, need4, live4_m, tiny, f);
            if (tiny.any_bad()) map_scene_taps<false>(lprog_v, cx, cy, cz, need4, live4_m, tiny, f);
#ifndef RM_NO_WAVE_STATS
            n_iter++;
            n_live += (uint32_t)__popcll(live4_m);
#endif
            // n = ((k0 f0 + k1 f1) + k2 f2) + k3 f3, k = (+,-,-), (-,-,+), (-,+,-), (+,+,+); products with +-1 are exact
            const float nx = ((f[0] + -f[1]) + -f[2]) + f[3];
            const float ny = ((-f[0] + -f[1]) + f[2]) + f[3];
            const float nz = ((-f[0] + f[1]) + -f[2]) + f[3];
            if (tagged) {
                tn[lane] = nx; tn[64u + lane] = ny; tn[128u + lane] = nz;
                tap_t = 4u;  // the material phase follows
            } else {
                // (the hit position is read again rather than kept in three registers through the four-tap function, whose
                // pressure is the kernel's peak: the acquire keeps the compiler from forwarding the earlier loads)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (live4) res[hq_rid[e] & 1023u] = shade_hit(nx, ny, nz, hq_v[e], hq_v[V5_HQ + e], hq_v[2u * V5_HQ + e]);  // wgsl:98-103
                tap_t = TAP_IDLE;
            }
            continue;
        }
#endif
        // ---- B. map_scene evaluations: march steps of the live rays, or the normal taps of the waiting hits ----
        // An inner loop of its own: evaluation follows evaluation until something else is due -- a refill, a tap phase, the end
        // of the tile's rays --, and that can only change when a ray finishes.  Section A's questions (who is idle, who is live,
        // how many hits wait) are then asked once per finished ray instead of once per step: a dozen scalar instructions and a
        // handful of branches less per march step, in a kernel whose scalar side is as busy as its vector side (a CU has one
        // scalar unit for four SIMDs: 4.3 cycles per scalar instruction per SIMD, profiles/r03_ubench_scalar_issue_cycles.txt).
        // The sequence of evaluations is the one the flat loop produced.
        uint32_t n_idle = tapping ? 0u : (uint32_t)__popcll(__ballot(lane_empty()));  // wave-uniform: lanes waiting for a ray
        for (;;) {
            // (the interpreter kernels keep ONE copy of the evaluation for march steps and taps: their record loops are large, and the
            // loop unswitched on `tapping` costs the lean kernel its registers)
            bool tapping_i = tapping;
            if constexpr (!SPEC) {
                uint32_t t = tapping ? 1u : 0u;
                asm volatile("" : "+v"(t));
                tapping_i = __builtin_amdgcn_readfirstlane(t) != 0u;
            }
            float ex, ey, ez, thr = inf_f;
            bool is_live;
            uint32_t sgx = 0u, sgy = 0u, sgz = 0u;
            const uint32_t he = hq_n + lane;  // this lane's hit-buffer entry in a tap phase (< V5_HQ for every lane)
            if (tapping_i) {  // pos + k_t * eps (wgsl:138-141); k_t * eps = +-eps exactly
                tap_signs(tap_t, sgx, sgy, sgz);
                ex = hq_v[he] + __uint_as_float(__float_as_uint(eps) ^ sgx);
                ey = hq_v[V5_HQ + he] + __uint_as_float(__float_as_uint(eps) ^ sgy);
                ez = hq_v[2u * V5_HQ + he] + __uint_as_float(__float_as_uint(eps) ^ sgz);
                is_live = lane < tap_n;
#ifdef RM_PRUNE_PLUMBING
                // the tap is eps * sqrt(3) away from the hit position, where the scene value was sd_hit ("Pruning")
                thr = __uint_as_float(hq_rid[he] & ~1023u) * 1.00001f + 1.75e-4f +
                      kPruneAbs * (prune_scale + ((__builtin_fabsf(ex) + __builtin_fabsf(ey)) + __builtin_fabsf(ez)));
#endif
            } else {
                ex = ro.x + dx * sc; ey = ro.y + dy * sc; ez = ro.z + dz * sc;  // wgsl:91
                is_live = lane_marching();
                thr = thr_base + kPruneAbs * (prune_scale + ((__builtin_fabsf(ex) + __builtin_fabsf(ey)) + __builtin_fabsf(ez)));
            }
            const unsigned long long live_m = __builtin_amdgcn_ballot_w64(is_live);
#ifndef RM_NO_WAVE_STATS
            n_iter++;
            n_live += (uint32_t)__popcll(live_m);
#endif
#ifdef RM_PRIO_LONG_RAYS
            // A ray that needs hundreds of steps is a serial chain of that many evaluations; at full load a wave gets a
            // fifth of its SIMD's issue slots, so such a ray started late IS the kernel's tail.  Waves that carry one get
            // issue priority: the chain runs at the lone-wave rate and the short rays fill in behind it.
            if (!tapping_i) {
                const uint32_t old_rays = (uint32_t)__popcll(__ballot(is_live && itr >= (RM_PRIO_LONG_RAYS << 10)));
                if (old_rays != 0u) __builtin_amdgcn_s_setprio(3);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            const float sd = eval_scene(ex, ey, ez, thr, is_live, live_m);

            if (tapping_i) {  // n (+)= k_t * f; products with +-1 are exact (wgsl:138-143)
                const float vx = __uint_as_float(__float_as_uint(sd) ^ sgx);
                const float vy = __uint_as_float(__float_as_uint(sd) ^ sgy);
                const float vz = __uint_as_float(__float_as_uint(sd) ^ sgz);
                // the partial sum waits in LDS between taps, not in three registers that every march iteration would carry
                const float nx = tap_t == 0u ? vx : tn[lane] + vx;
                const float ny = tap_t == 0u ? vy : tn[64u + lane] + vy;
                const float nz = tap_t == 0u ? vz : tn[128u + lane] + vz;
                if (++tap_t == 4u && !tagged) {
                    if (is_live) res[hq_rid[he] & 1023u] = shade_hit(nx, ny, nz, hq_v[he], hq_v[V5_HQ + he], hq_v[2u * V5_HQ + he]);  // wgsl:98-103
                    tap_t = TAP_IDLE;
                    break;
                }
                tn[lane] = nx; tn[64u + lane] = ny; tn[128u + lane] = nz;
                if (tap_t == 4u) break;  // (tagged program: the material phase follows)
                continue;                // the next tap of the same hits
            }

            // ---- C. march bookkeeping (wgsl:97-114); finished rays leave their lane ----
            // Written on wave masks (64-bit scalars) with explicit selects: left to itself the compiler turned the lane
            // predicates into 0/1 integers and back (17 vector instructions for what takes 7).
            const unsigned long long hit_mask = live_m & __ballot(sd < L.min_dist);                       // wgsl:97
            const unsigned long long on_m = live_m & ~hit_mask;                                           // (a NaN is not a hit)
            const unsigned long long esc_m = on_m & __ballot(sd > L.max_dist);                            // wgsl:109-111
            const unsigned long long go_m = on_m & ~esc_m;
            sc = select_by_mask(sc, sc + sd, go_m);                                                       // wgsl:114
            itr = select_by_mask(itr, itr + 1024u, go_m);
#ifdef RM_PRUNE_PLUMBING  // only pruning kernels read it
            thr_base = select_by_mask(thr_base, __builtin_fabsf(sd) * 2.00002f, go_m);  // the next point is |sd| |rd| away
#endif
            const unsigned long long miss_mask = esc_m | (go_m & __ballot(itr >= iter_limit));             // loop bound, wgsl:90
            const unsigned long long done_m = hit_mask | miss_mask;
            if (done_m == 0ull) continue;  // every live ray goes on: nothing section A asks about has changed
            const bool hit = is_live && sd < L.min_dist, miss = lane_of(miss_mask, lane);
            if (hit_mask != 0ull) {  // -> hit buffer (capacity 128: a tap phase takes 64 as soon as 64 are waiting)
                if (hit) {
                    const uint32_t e = hq_n + lane_rank(hit_mask);
#ifdef RM_PRUNE_PLUMBING
                    // |sd| of the hit, rounded up to 22 bits, rides in the upper bits of the entry's ray id (< 1024): the
                    // pruning threshold of its normal taps ("Pruning"); a NaN stays a NaN (nothing is skipped then)
                    hq_rid[e] = (itr & 1023u) | ((__float_as_uint(__builtin_fabsf(sd)) + 1023u) & ~1023u);
#else
                    hq_rid[e] = itr & 1023u;
#endif
                    hq_v[e] = ex; hq_v[V5_HQ + e] = ey; hq_v[2u * V5_HQ + e] = ez;
                    itr = ITR_EMPTY;
                }
                hq_n += (uint32_t)__popcll(hit_mask);
            }
            if (miss_mask != 0ull) {
                const uint32_t n_miss = (uint32_t)__popcll(miss_mask);
                if (sq_n + n_miss > V5_SQ) flush_misses();
                if (miss) {
                    const uint32_t e = sq_n + lane_rank(miss_mask);
                    sq_rid[e] = itr & 1023u;
                    sq_v[e] = dx; sq_v[V5_SQ + e] = dy; sq_v[2u * V5_SQ + e] = dz;
                    itr = ITR_EMPTY;
                }
                sq_n += n_miss;
            }
            // back to section A when it has something to do: no ray left marching, enough idle lanes for a refill (refill_min; when
            // the pool is exhausted A retires them), or 64 hits waiting for their normals
            n_idle += (uint32_t)__popcll(done_m);
            if ((live_m & ~done_m) == 0ull || n_idle >= refill_min || hq_n >= 64u) break;
        }
    }
    if (sq_n != 0u) flush_misses();
    __syncthreads();  // all waves of the tile are done: res[] is complete

    // ---- resolve: one pixel per thread, samples in the reference order (wgsl:44-45, 68-69) ----
    for (uint32_t p = tid; p < 64u; p += 64u * WPT) {
        const uint32_t px = tile_x * 8u + (p & 7u), py = tile_y * 8u + (p >> 3);
        if (px < L.W && py < L.rows) {
            // the 48 gamma square roots of a pixel (wgsl:68) take the short form of the leaves' sqrt (rm_interp.h sqrt_rn_fast:
            // the same bits for every argument its guard lets through); a colour it does not -- negative, NaN, infinite: only a
            // material table can produce one -- sends the pixel through the generic form
            auto sum_samples = [&](auto fast_tag, float& tr, float& tg, float& tb, SqrtGuard& guard) {
                constexpr bool FAST = decltype(fast_tag)::value;
                tr = 0.0f; tg = 0.0f; tb = 0.0f;
#pragma unroll 4
                for (uint32_t s = 0; s < 16u; s++) {
                    const float code = res[s * 64u + p];
                    float cr, cg, cb;
                    if (code >= 0.0f) {  // hit: (0.4,0.7,0.1) * k  (wgsl:105)
                        cr = 0.4f * code; cg = 0.7f * code; cb = 0.1f * code;
                        if constexpr (MAT) {
                            if (tagged) {  // extension: the albedo of the material the surface carries
                                const float4 al = L.materials[rmat[s * 64u + p]];
                                cr = al.x * code; cg = al.y * code; cb = al.z * code;
                            }
                        }
                    } else if (code > -2.5f) {  // floor (wgsl:127)
                        const float g = 0.2f * (-1.0f - code);
                        cr = 0.1f + g; cg = 0.1f + g; cb = 0.2f + g;
                    } else {
                        cr = 0.0f; cg = 0.0f; cb = 0.0f;  // wgsl:130
                    }
                    tr += sqrt_sel<FAST, true>(cr, guard);
                    tg += sqrt_sel<FAST, true>(cg, guard);
                    tb += sqrt_sel<FAST, true>(cb, guard);
                }
            };
            float tr, tg, tb;
            SqrtGuard gamma_guard;
            sum_samples(TrueTag(), tr, tg, tb, gamma_guard);
            if (gamma_guard.bad()) sum_samples(FalseTag(), tr, tg, tb, gamma_guard);  // per lane: the rare pixel, not the wave
            store_pixel(L, blockIdx.z, (size_t)py * L.W + px, tr / 16.0f, tg / 16.0f, tb / 16.0f);  // wgsl:73-75
        }
    }
    if (work.measured && tid == 0u) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - tile_t0;
        work.measured[(size_t)blockIdx.z * n_tiles + tile] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (dt == 0ull ? 1u : (uint32_t)dt);
    }
    __syncthreads();  // res[] / rings / cursor are reused by the next tile
  }
    if (L.stats && lane == 0u) {
        unsigned long long* st = L.stats + 4ull * (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * WPT + wave);
        st[0] = t_start;
        st[1] = __builtin_amdgcn_s_memrealtime();
        st[2] = ((unsigned long long)n_tiles_done << 32) | n_iter;
        st[3] = ((unsigned long long)(n_eval != 0u ? n_eval : n_prod) << 32) | n_live;
    }
}

template <class Prog, bool PROG_IN_LDS, int WPT, bool EXT>
__global__ __launch_bounds__(64 * WPT) void rm_render_v5(RmLaunch L, V5Work work, uint32_t n_tiles, uint32_t refill_min) {
    rm_render_v5_body<Prog, PROG_IN_LDS, WPT, EXT, false, EXT>(L, work, n_tiles, refill_min);
}
// The interpreter for reference-only programs staged in LDS -- north_star's design, and what runs while a structure compiles --
// with its vector registers capped at 80 (6 waves per SIMD, which is also what its LDS footprint allows): left alone the
// allocator takes 81 and a sixth of the occupancy (march kernel of the metric frame 0.83 -> 0.87 ms).
#ifndef RM_LEAN_WAVES
#define RM_LEAN_WAVES 6
#endif
template <int WPT, int LOOP>
__global__ __launch_bounds__(64 * WPT) __attribute__((amdgpu_waves_per_eu(RM_LEAN_WAVES, 8)))
void rm_render_v5_lean(RmLaunch L, V5Work work, uint32_t n_tiles, uint32_t refill_min) {
    rm_render_v5_body<ProgLds, true, WPT, false, false, false, LOOP>(L, work, n_tiles, refill_min);
}

// ---------------------------------------------------------------------------------------------
// Pre-pass: one WAVE per tile, lane = pixel, four tiles per workgroup.
//
// Each lane decides whether ALL 16 AA rays of its pixel provably miss the scene, with one test per
// primitive instead of sixteen.  The 16 sample points are an affine 4x4 grid on the screen and
// pt_world.xyz - ro.xyz is affine in the screen position (two mat-vecs, wgsl:56-61), so the 16 ray
// directions lie in the convex cone spanned by the four extreme samples (i, j in {0, 3}).  With the
// unit directions e_0..e_3 of those four, c = normalize(e_0 + .. + e_3) and rho = max |e_i - c|,
// every sample direction e of the pixel satisfies |e - c| <= rho (the distance to a fixed point is
// maximal at a vertex of a convex spherical polygon smaller than a hemisphere; rho < 0.05 is required,
// which also rejects a pixel whose parallelogram contains the origin).  Then for a primitive with
// bounding cone (m, s) -- a ray clears it iff m.e < s, see cull_build_v5 --
//     m.e = m.c + m.(e - c) <= m.c + |m| rho,
// so  m.c + |m| rho < s  clears the primitive for the whole pixel.  Boxes get a second chance when
// that fails: a point P of any of the pixel's rays inside the (margin-inflated) box has |P| <= D =
// |m| + radius, and the centre ray passes within D rho of it, so if the centre ray misses the box
// inflated by a further D rho, every ray of the pixel misses the box.  rho carries 0.1 % + 2e-6 of
// slack for the rounding of e_i and c.
//
// cost = number of pixels that are not provably clear (0..64).  A tile with cost 0 is finished
// here: its wave sums the 16 gamma-corrected floor colours of every pixel in the reference's order
// (wgsl:44-45, 68-69, 73-75) and writes it.  Tiles with cost > 0 go on the work list; the march
// kernel repeats the (sharper) test per ray.  ~60 % of the tiles of the metric frame end here.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t V5_PRE_TILES = 4u;  // waves per pre-pass workgroup, one tile at a time each
// Tiles each of those waves walks through: one.  (Four -- to spread the workgroup's prologue: records fetched, tables built, three
// barriers -- made the kernel slower, 60 -> 75 us: it is bound by its work, and expensive tiles are neighbours.)
constexpr uint32_t V5_PRE_TILES_PER_WAVE = 1u;

#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
// cone_out: (unit centre direction, rho) of the pixel's sixteen sample directions; rho = NaN when the cone is not usable.
// corner[c]: pt_world.xyz - ro.xyz of corner sample c (samples (0,0), (3,0), (0,3), (3,3)), as gen_ray computes it before its
// normalisation; the caller's sky / floor tests reuse them.  m_proj / m_view: the matrices (in vector registers), off: the
// table of the sixteen sample offsets.
RM_DEV bool pixel_misses_scene_v5(const CullTables& T, const float2* aux, const float* m_proj, const float* m_view, const float* off,
                                  const V4& ro, float sx, float sy, float (&cone_out)[4], float (&corner)[4][3]) {
    float cx = 0.0f, cy = 0.0f, cz = 0.0f, ex[4], ey[4], ez[4];
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        const uint32_t s = ((c & 1u) * 3u) * 4u + (c >> 1) * 3u;  // sample (i, j) = table entry 4 i + j
        gen_ray_unnormalized_at(m_proj, m_view, ro, sx, sy, off[2u * s], off[2u * s + 1u], ex[c], ey[c], ez[c]);  // unit_dir normalises
        corner[c][0] = ex[c]; corner[c][1] = ey[c]; corner[c][2] = ez[c];
        unit_dir(ex[c], ey[c], ez[c]);
        cx += ex[c]; cy += ey[c]; cz += ez[c];
    }
    unit_dir(cx, cy, cz);
    float rho2 = 0.0f;
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        const float ax = ex[c] - cx, ay = ey[c] - cy, az = ez[c] - cz;
        rho2 = fmax_(rho2, __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
    }
    const float rho = __builtin_sqrtf(rho2) * 1.001f + 2.0e-6f;
    // NaN anywhere above (zero / non-finite directions) makes this false; v_max drops a NaN operand, so
    // the corners are checked one by one as well
    bool cone_ok = rho < 0.05f;
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) cone_ok = cone_ok && (ex[c] - cx) * (ex[c] - cx) < 1.0f && (ey[c] - cy) * (ey[c] - cy) < 1.0f && (ez[c] - cz) * (ez[c] - cz) < 1.0f;
    cone_out[0] = cx; cone_out[1] = cy; cone_out[2] = cz;
    cone_out[3] = cone_ok ? rho : __uint_as_float(0x7FC00000u);
    bool clear = cone_ok && *T.veto == 0u;  // a Plane (veto bit 1) leaves the tables unusable, not the cone
    for (uint32_t k = 0; k < T.n_cone; k++) {
        const float4 a = T.cone[k];  // wave-uniform address: LDS broadcast
        const float t = __builtin_fmaf(aux[k].x, rho, __builtin_fmaf(a.z, cz, __builtin_fmaf(a.y, cy, a.x * cx)));
        clear = clear && (t < a.w);  // NaN -> not clear
        if ((k & 3u) == 3u && __ballot(clear) == 0ull) return false;
    }
    if (T.n_slab == 0u || __ballot(clear) == 0ull) return clear;
    const float tiny = 1.0e-30f;
    const float ix = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(cx), tiny), cx));
    const float iy = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(cy), tiny), cy));
    const float iz = __builtin_amdgcn_rcpf(__builtin_copysignf(fmax_(__builtin_fabsf(cz), tiny), cz));
    for (uint32_t k = 0; k < T.n_slab; k++) {
        const float4 c = T.slab[3u * k + 2u];
        const float2 x = aux[T.n_cone + k];  // (|m|, D) both rounded up
        const float t = __builtin_fmaf(x.x, rho, __builtin_fmaf(c.z, cz, __builtin_fmaf(c.y, cy, c.x * cx)));
        if (__ballot(clear && !(t < c.w)) == 0ull) continue;  // every still-clear pixel clears the bounding sphere
        const float infl = x.y * rho;
        const float4 a = T.slab[3u * k], b = T.slab[3u * k + 1u];
        const float x1 = (a.x - infl) * ix, x2 = (b.x + infl) * ix, y1 = (a.y - infl) * iy, y2 = (b.y + infl) * iy;
        const float z1 = (a.z - infl) * iz, z2 = (b.z + infl) * iz;
        const float tn = fmax_(fmin_(x1, x2), fmax_(fmin_(y1, y2), fmin_(z1, z2)));
        const float tf = fmin_(fmax_(x1, x2), fmin_(fmax_(y1, y2), fmax_(z1, z2)));
        // infl must be a number for the comparison to mean anything (a dropped NaN would shrink the box)
        clear = clear && infl < __uint_as_float(0x7F800000u) && !(tf >= fmax_(tn, 0.0f));
        if (__ballot(clear) == 0ull) return false;
    }
    return clear;
}

__global__ __launch_bounds__(64 * V5_PRE_TILES) void rm_tile_pre_v5(RmLaunch L, uint32_t* cost, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* s_veto = smem;
    float4* t_cone = reinterpret_cast<float4*>(smem + 4u);
    float4* t_slab = t_cone + L.n_cone;
    float2* t_aux = reinterpret_cast<float2*>(t_slab + 3u * L.n_slab);  // [n_cone + n_slab] (|m|, |m| + bounding radius)
    const CullTables cullt{t_cone, t_slab, s_veto, L.n_cone, L.n_slab};
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];
    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);
    const bool tables = (L.flags & 1u) != 0u;
    __shared__ float s_off[32];  // screen offsets of the 16 AA samples: the same for every pixel, computed once
    __shared__ float s_sxy[V5_PRE_TILES][128];          // per wave: pixel centres, sample codes and the list of the pixels that
    __shared__ uint8_t s_code[V5_PRE_TILES][64 * 16];   // need them, for the sample-parallel finish of a clear tile (below)
    __shared__ uint8_t s_list[V5_PRE_TILES][64];
    if (tid == 0u) *s_veto = 0u;
    if (tid < 16u) sample_offset(u, tid >> 2, tid & 3u, s_off[2u * tid], s_off[2u * tid + 1u]);
    __syncthreads();
    if (tables)
        for (uint32_t k = tid; k < L.n_rec; k += 64u * V5_PRE_TILES)
            cull_build_v5(L.prog[k], ro, L.min_dist, L.smooth_slack, t_cone, t_slab, s_veto, L.bounds);
    __syncthreads();
    if (tables) {
        const float up = 1.0f + 1.0e-6f;
        for (uint32_t k = tid; k < L.n_cone + L.n_slab; k += 64u * V5_PRE_TILES) {
            float4 m;
            float radius = 0.0f;
            if (k < L.n_cone) {
                m = t_cone[k];
            } else {
                const uint32_t j = k - L.n_cone;
                m = t_slab[3u * j + 2u];
                const float4 a = t_slab[3u * j], b = t_slab[3u * j + 1u];
                const float hx = b.x - a.x, hy = b.y - a.y, hz = b.z - a.z;  // full extents of the inflated box
                radius = 0.5f * __builtin_sqrtf(hx * hx + hy * hy + hz * hz) * up;
            }
            const float len = __builtin_sqrtf(m.x * m.x + m.y * m.y + m.z * m.z) * up;
            t_aux[k] = make_float2(len, (len + radius) * up);
        }
    }
    __syncthreads();
    const uint32_t tiles_x = (L.W + 7u) / 8u;
  auto do_tile = [&](uint32_t tile) {  // (whole wave; no barrier inside)
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const uint32_t tx = tile_x * 8u + (lane & 7u), ty = tile_y * 8u + (lane >> 3);
    const uint32_t px = tx < L.W ? tx : L.W - 1u, ry = ty < L.rows ? ty : L.rows - 1u;
    const float sx = screen_x(px, L.W), sy = screen_y(rm_global_row(L, ry), L.H);
    // the two matrices in vector registers (32 of them): as scalar operands every multiply of the mat-vecs -- the four corner rays
    // here, the sixteen sample rays of the finishing loop -- would issue at half rate
    float mp[16], mv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mp[k] = u.inv_proj[k]; mv[k] = u.inv_view[k];
        asm volatile("" : "+v"(mp[k]), "+v"(mv[k]));
    }
    bool clear;
    float corner[4][3];  // pt_world.xyz - ro.xyz of the pixel's four corner samples
    if (L.max_iter == 0u || !tables) {
        clear = L.max_iter == 0u;  // no march step is ever taken: every ray is a miss (wgsl:92); without tables nothing is known
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) {
            const uint32_t s = ((c & 1u) * 3u) * 4u + (c >> 1) * 3u;
            gen_ray_unnormalized_at(mp, mv, ro, sx, sy, s_off[2u * s], s_off[2u * s + 1u], corner[c][0], corner[c][1], corner[c][2]);
        }
    } else {
        float cone[4];
        clear = pixel_misses_scene_v5(cullt, t_aux, mp, mv, s_off, ro, sx, sy, cone, corner);
        // a program that blends: the pixels the inflated bounds could not clear get the program run on lower bounds of its
        // leaves over the pixel's cone ("Miss test on lower bounds"; the march kernel repeats it per ray for what is left)
        const float bound_scale = (L.scene_scale + L.smooth_slack) + ((__builtin_fabsf(ro.x) + __builtin_fabsf(ro.y)) + __builtin_fabsf(ro.z));
        if ((L.flags & 32u) && (*s_veto & 1u) == 0u && bound_scale < 1.0e12f && __ballot(!clear && cone[3] == cone[3]) != 0ull) {
            auto load_record = [&](uint32_t i, uint32_t& op, float (&p)[7]) {
                const RmRecord& r = L.prog[i];  // wave-uniform address: scalar loads
                op = r.op;
#pragma unroll
                for (int k = 0; k < 7; k++) p[k] = r.p[k];
            };
            const bool by_bounds = ray_misses_by_bounds_v5<true>(load_record, L.n_rec, ro, cone[0], cone[1], cone[2], L.min_dist, bound_scale, cone[3]);
            clear = clear || (cone[3] == cone[3] && by_bounds);
        }
    }
    const uint32_t pending = (uint32_t)__popcll(__ballot(!clear));
    if (lane == 0u) cost[(size_t)blockIdx.z * n_tiles + tile] = pending;  // 0 = finished here
    if (pending != 0u) return;
    bool known = false;   // this pixel's sixteen samples provably share one colour code ...
    int known_code = -1;  // ... -1 black (sky), 0 / 1 the checker bit
    // Sky: a sample whose ray points away from the floor plane (t <= 0 in wgsl:120-121) is black, and the sum of sixteen
    // zeros is zero.  pt_world.y - ro.y is affine in the sample's screen offset, so its sign over the 4 x 4 grid is decided at
    // the four corner samples up to rounding; with the camera above the plane (-1.5 - ro.y < 0), "all four corners point up by
    // more than the rounding can move any of the sixteen" means dy > 0, t < 0 for all of them.  The margin is 1e-5 of `mag`,
    // the sum of the absolute values of every term the two mat-vecs add up for the y component (each of the ~30 roundings
    // on the way is <= 6e-8 of a partial sum, hence of mag; interior and corner values each move by that, the affine
    // argument needs twice it).  A tile whose 64 pixels are all such writes zeros and skips the loop below (about half of
    // the clear tiles of the metric frame).  Anything non-finite fails the comparison and takes the loop.
    {
        float lo = __uint_as_float(0x7F800000u);
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) {
            lo = corner[c][1] < lo ? corner[c][1] : lo;
            if (!(corner[c][1] == corner[c][1])) lo = corner[c][1];  // a NaN sticks and makes the test below false
        }
        const float bx = __builtin_fabsf(sx) + __builtin_fabsf(s_off[0]), by = __builtin_fabsf(sy) + __builtin_fabsf(s_off[1]);  // sample (0, 0) has the largest offsets
        float mag = __builtin_fabsf(ro.y);
#pragma unroll
        for (int k = 0; k < 4; k++)  // |inv_view row y| . (|inv_proj| (|x|, |y|, 1, 1))
            mag += __builtin_fabsf(u.inv_view[1 + 4 * k]) * (((__builtin_fabsf(u.inv_proj[k]) * bx + __builtin_fabsf(u.inv_proj[k + 4]) * by) +
                                                              __builtin_fabsf(u.inv_proj[k + 8])) + __builtin_fabsf(u.inv_proj[k + 12]));
        const bool sky = (-1.5f - ro.y) < 0.0f && lo > 1.0e-5f * mag;
        if (__ballot(!sky) == 0ull) {
            if (tx < L.W && ty < L.rows) store_pixel(L, blockIdx.z, (size_t)ty * L.W + tx, 0.0f, 0.0f, 0.0f);  // 0 / 16 = 0 (wgsl:73-75)
            return;
        }
        // Floor, one checker cell per pixel.  With every sample ray pointing down (t > 0) the floor point of a sample is
        // fx = o.x + C ex / ey, C = -1.5 - o.y (the normalisation cancels), a linear-fractional function of the sample's screen
        // position: monotone along every segment where ey keeps its sign, so over the pixel's rectangle of samples it stays between
        // its values at the four corner samples.  If the interval those span -- widened by everything rounding can do to a
        // sample's own fx: the mat-vec errors E (4e-6 of `mag`, per component) pushed through |C| / |ey| (E_x + |ex| E_y / |ey|),
        // once for the corners and once for the sample, the approximate reciprocal used here and the five roundings of the
        // sample's own chain -- rounds to ONE integer after the +0.5 of wgsl:124 (rint is monotone), and the same holds for z,
        // all sixteen samples carry the same checker bit, and the pixel is the sum of sixteen equal gamma values: a constant
        // per bit, accumulated below exactly as the loop would.  A tile whose 64 pixels are each sky or such a pixel skips the
        // loop: what remains for it are the pixels a cell edge crosses, the horizon and the far floor.
        const float Cf = -1.5f - ro.y;  // as shade_floor computes it
        float ex[4], eyc[4], ez[4], ey_hi = -__uint_as_float(0x7F800000u), ax_hi = 0.0f, az_hi = 0.0f;
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) {
            ex[c] = corner[c][0]; eyc[c] = corner[c][1]; ez[c] = corner[c][2];
            ey_hi = eyc[c] > ey_hi ? eyc[c] : ey_hi;
            if (!(eyc[c] == eyc[c])) ey_hi = eyc[c];
            ax_hi = fmax_(ax_hi, __builtin_fabsf(ex[c]));
            az_hi = fmax_(az_hi, __builtin_fabsf(ez[c]));
        }
        float mag_x = __builtin_fabsf(ro.x), mag_z = __builtin_fabsf(ro.z);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float pvk = ((__builtin_fabsf(u.inv_proj[k]) * bx + __builtin_fabsf(u.inv_proj[k + 4]) * by) + __builtin_fabsf(u.inv_proj[k + 8])) +
                              __builtin_fabsf(u.inv_proj[k + 12]);
            mag_x += __builtin_fabsf(u.inv_view[0 + 4 * k]) * pvk;
            mag_z += __builtin_fabsf(u.inv_view[2 + 4 * k]) * pvk;
        }
        const float Ey = 4.0e-6f * mag, Ex = 4.0e-6f * mag_x, Ez = 4.0e-6f * mag_z;
        const float my = -ey_hi - 2.0f * Ey;  // every sample's |ey| is at least this
        bool uniform = Cf < -1.0e-20f && my > 0.0f && ey_hi < -4.0f * Ey;
        const float inv_my = __builtin_amdgcn_rcpf(my) * 1.00001f, absC = -Cf;
        float gx_lo = __uint_as_float(0x7F800000u), gx_hi = -gx_lo, gz_lo = gx_lo, gz_hi = -gx_lo;
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++) {
            const float r = __builtin_amdgcn_rcpf(eyc[c]);
            const float gx = ex[c] * r, gz = ez[c] * r;
            gx_lo = fmin_(gx_lo, gx); gx_hi = fmax_(gx_hi, gx);
            gz_lo = fmin_(gz_lo, gz); gz_hi = fmax_(gz_hi, gz);
            uniform = uniform && gx == gx && gz == gz;  // (v_min / v_max drop a NaN)
        }
        // C < 0: the interval of fx - o.x is [C gx_hi, C gx_lo]
        const float dx_lo = Cf * gx_hi, dx_hi = Cf * gx_lo, dz_lo = Cf * gz_hi, dz_hi = Cf * gz_lo;
        const float dev_x = fmax_(__builtin_fabsf(dx_lo), __builtin_fabsf(dx_hi)), dev_z = fmax_(__builtin_fabsf(dz_lo), __builtin_fabsf(dz_hi));
        const float del_x = 4.0f * (absC * inv_my) * (Ex + ((ax_hi + Ex) * inv_my) * Ey) + 2.0e-5f * ((1.0f + __builtin_fabsf(ro.x)) + dev_x);
        const float del_z = 4.0f * (absC * inv_my) * (Ez + ((az_hi + Ez) * inv_my) * Ey) + 2.0e-5f * ((1.0f + __builtin_fabsf(ro.z)) + dev_z);
        const float nx = __builtin_rintf(((ro.x + dx_lo) - del_x) + 0.5f), nz = __builtin_rintf(((ro.z + dz_lo) - del_z) + 0.5f);
        uniform = uniform && nx == __builtin_rintf(((ro.x + dx_hi) + del_x) + 0.5f) && nz == __builtin_rintf(((ro.z + dz_hi) + del_z) + 0.5f);
        known = sky || uniform;
        known_code = sky ? -1 : ((__float2int_rz(nx) ^ __float2int_rz(nz)) & 1);  // as shade_floor: saturating, wgsl:124-126
    }
    // The sample loop.  Everything that does not depend on the pixel is taken out of it: the sample offsets (two divisions each)
    // come from the table, the matrices sit in vector registers, and the gamma-corrected colour of a sample is one of three
    // values -- black, or the checker colour for bit 0 / 1 (wgsl:127) -- whose square roots are taken once, by the same
    // expressions.
    float gamma_rg[2], gamma_b[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float g = 0.2f * (float)c;
        gamma_rg[c] = __builtin_sqrtf(0.1f + g);
        gamma_b[c] = __builtin_sqrtf(0.2f + g);
    }
    auto gamma_of = [&](int c, float& rg, float& b) {
        rg = c < 0 ? 0.0f : (c ? gamma_rg[1] : gamma_rg[0]);  // sqrt(0) = 0 (wgsl:130)
        b = c < 0 ? 0.0f : (c ? gamma_b[1] : gamma_b[0]);
    };
    float tr = 0.0f, tg = 0.0f, tb = 0.0f;
    const unsigned long long need = __ballot(!known);  // pixels whose samples have to be looked at one by one
    const uint32_t n_need = (uint32_t)__popcll(need);
    const uint32_t need_max = (L.flags >> 8) & 0x7Fu ? ((L.flags >> 8) & 0x7Fu) - 1u : 24u;  // (diagnostics: RM_PRE_NEED_MAX)
    if (n_need <= need_max) {
        // Few such pixels (a checker edge crossing the tile): their samples -- 16 n_need of them -- are spread over the 64
        // lanes, 64 per pass instead of one sample of all 64 pixels per pass; each sample's colour code goes to LDS and the
        // pixel's lane then adds its sixteen gamma values in the reference order.  Same functions, same sums.
        float* w_sxy = s_sxy[wave];
        uint8_t* w_code = s_code[wave];
        uint8_t* w_list = s_list[wave];
        w_sxy[lane] = sx; w_sxy[64u + lane] = sy;
        if (!known) w_list[lane_rank(need)] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t base = 0u; base < 16u * n_need; base += 64u) {
            const uint32_t item = base + lane, q = item >> 4, s = item & 15u;
            const uint32_t pl = w_list[q < n_need ? q : 0u];
            float dx, dy, dz;
            gen_ray_at(mp, mv, ro, w_sxy[pl], w_sxy[64u + pl], s_off[2u * s], s_off[2u * s + 1u], dx, dy, dz);
            const int c = shade_floor(ro.y, ro.x, ro.z, dx, dy, dz);  // wgsl:117-130
            if (q < n_need) w_code[pl * 16u + s] = (uint8_t)(c + 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t s = 0; s < 16u; s++) {  // reference order: wgsl:44-45, 68-69
            const int c = known ? known_code : (int)w_code[lane * 16u + s] - 1;
            float rg, b;
            gamma_of(c, rg, b);
            tr += rg;
            tg += rg;
            tb += b;
        }
    } else {
        for (uint32_t s = 0; s < 16u; s++) {  // reference order: wgsl:44-45, 68-69
            float dx, dy, dz;
            gen_ray_at(mp, mv, ro, sx, sy, s_off[2u * s], s_off[2u * s + 1u], dx, dy, dz);
            const int c = shade_floor(ro.y, ro.x, ro.z, dx, dy, dz);  // wgsl:117-130
            float rg, b;
            gamma_of(c, rg, b);
            tr += rg;
            tg += rg;
            tb += b;
        }
    }
    if (tx < L.W && ty < L.rows) {
        store_pixel(L, blockIdx.z, (size_t)ty * L.W + tx, tr / 16.0f, tg / 16.0f, tb / 16.0f);  // wgsl:73-75
    }
  };
    const uint32_t first_tile = (blockIdx.x * V5_PRE_TILES + wave) * V5_PRE_TILES_PER_WAVE;
    for (uint32_t t = 0; t < V5_PRE_TILES_PER_WAVE && first_tile + t < n_tiles; t++) {
        do_tile(first_tile + t);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's LDS scratch is reused by its next tile
    }
}
#endif

// Work list of one frame (blockIdx.x = frame): ids of the tiles with cost > 0, by descending cost
// when `balance` is set.  Also resets the persistent kernel's cursor.
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
// With `prev` (what the previous draw of the same shape measured per tile) the key of a tile that was marched then is
// its duration on a 4-steps-per-octave log scale, so that the LONGEST tiles are dispatched first and the kernel ends
// on its shortest ones; a tile without a measurement keeps its pending-pixel count (64 for a fully covered tile, i.e.
// it is treated as heavy).  The order only affects when a tile is rendered, never its pixels.
RM_DEV uint32_t duration_key(uint32_t ticks) {  // 1..64; 64 ticks (0.64 us) -> 1, doubling adds 4
    if (ticks < 64u) return 1u;
    const uint32_t lz = 31u - (uint32_t)__builtin_clz(ticks);
    const uint32_t q = 4u * lz + ((ticks >> (lz - 2u)) & 3u) - 23u;
    return q > 64u ? 64u : q;
}
__global__ __launch_bounds__(1024) void rm_tile_sort_v5(RmLaunch L, const uint32_t* cost, uint32_t* order, uint32_t* counters,
                                                        uint32_t n_tiles, uint32_t balance, const uint32_t* prev) {
    __shared__ uint32_t hist[65], base[65];
    __shared__ float f0_spill[(32 + 3 * RM_MAX_XFORM_DEPTH) * 64];
    const uint32_t tid = threadIdx.x;
    // f0 = map_scene(ro): the first march step every ray of this frame shares (see rm_render_v5_body).  The
    // interpreter with the correctly rounded library sqrt gives the same bits as every kernel's evaluation.
    if (tid >= 960u) {  // the last wave; it has the least to do in the loops below
        rm_uniforms u = L.u;
        if (L.frames) u = L.frames[blockIdx.x];
        const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);
        ProgSmem prog;
        prog.base = L.prog;
        const float qx[1] = {ro.x}, qy[1] = {ro.y}, qz[1] = {ro.z};
        float v[1];
        SqrtGuard tiny;
        map_scene_multi<1, false, ProgSmem, true>(prog, L.n_rec, f0_spill + (tid & 63u), L.max_dist, qx, qy, qz, v, tiny, L.value_spill_depth);
        if (tid == 960u) counters[4u * blockIdx.x + 2u] = __float_as_uint(v[0]);
    }
    const uint32_t* c = cost + (size_t)blockIdx.x * n_tiles;
    const uint32_t* pm = prev ? prev + (size_t)blockIdx.x * n_tiles : nullptr;
    uint32_t* o = order + (size_t)blockIdx.x * n_tiles;
    // keys of tiles i0 + tid + 1024 j, j = 0..7: the loads of a batch are issued together (one workgroup walks the whole list;
    // with one dependent load per trip the two passes below were 32 round trips to L2 each, 35 us for a 1080p frame)
    constexpr uint32_t B = 8u;
    auto keys_of = [&](uint32_t i0, uint32_t (&v)[B]) {
        uint32_t t[B];
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
            const uint32_t i = i0 + j * 1024u + tid;
            v[j] = i < n_tiles ? c[i] : 0u;
            t[j] = (pm && i < n_tiles) ? pm[i] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < B; j++)
            if (v[j] != 0u && t[j] != 0u) v[j] = duration_key(t[j]);
    };
    if (tid < 65u) hist[tid] = 0u;
    __syncthreads();
    // bucket 0 = heaviest (cost 64) ... bucket 63 = cost 1; cost 0 (finished in the pre-pass) is dropped
    auto bucket = [&](uint32_t v) { return balance ? 64u - (v < 64u ? v : 64u) : 0u; };
    for (uint32_t i0 = 0; i0 < n_tiles; i0 += 1024u * B) {
        uint32_t vs[B];
        keys_of(i0, vs);
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
            const uint32_t v = vs[j];
            // most active tiles have the same cost (64): count those once per wave, not once per lane
            const unsigned long long heavy = __ballot(v != 0u && bucket(v) == 0u);
            if (heavy != 0ull && (tid & 63u) == (uint32_t)__builtin_ctzll(heavy)) atomicAdd(&hist[0], (uint32_t)__popcll(heavy));
            if (v != 0u && bucket(v) != 0u) atomicAdd(&hist[bucket(v)], 1u);
        }
    }
    __syncthreads();
    if (tid == 0u) {
        uint32_t acc = 0u;
        if (balance == 2u) {  // partially covered tiles (silhouettes: that is where the rays that never converge are) first
            for (uint32_t b = 1; b < 65u; b++) { base[b] = acc; acc += hist[b]; }
            base[0] = acc;
            acc += hist[0];
        } else {
            for (uint32_t b = 0; b < 65u; b++) { base[b] = acc; acc += hist[b]; }
        }
        counters[4u * blockIdx.x] = acc;      // tiles on the work list
        counters[4u * blockIdx.x + 1u] = 0u;  // cursor
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_tiles; i0 += 1024u * B) {
        uint32_t vs[B];
        keys_of(i0, vs);
#pragma unroll
        for (uint32_t j = 0; j < B; j++) {
            const uint32_t i = i0 + j * 1024u + tid;
            const uint32_t v = vs[j];
            const bool is_heavy = v != 0u && bucket(v) == 0u;
            const unsigned long long heavy = __ballot(is_heavy);
            uint32_t pos0 = 0u;
            if (heavy != 0ull) {
                const int leader = __builtin_ctzll(heavy);
                if ((int)(tid & 63u) == leader) pos0 = atomicAdd(&base[0], (uint32_t)__popcll(heavy));
                pos0 = __shfl(pos0, leader);
            }
            if (is_heavy) o[pos0 + lane_rank(heavy)] = i;
            else if (v != 0u) o[atomicAdd(&base[bucket(v)], 1u)] = i;
        }
    }
}
#endif

}  // namespace rmk
