// rm_kernel_queue.h -- kernel v4 "queued ray pool" (gfx950, wave64).  Device code only.
//
// Same per-wave ray pool as kernels v2/v3 (one wave = one 8x8-pixel tile = 1024 rays, one ray
// per lane in flight), but the two expensive, branchy pieces of per-ray work no longer run
// with a handful of active lanes inside the march loop:
//
//   produce  ray generation (2 mat-vec, vec4 normalize) + the exact miss-ray cull test run for
//            64 candidate rays at a time, ALL lanes active; rays that provably miss are shaded
//            on the spot, survivors are compacted (ballot + prefix count) into a READY queue in
//            LDS (ray id + direction, 16 B).
//   consume  a lane whose ray ended pops the next ready ray: four LDS reads, no arithmetic.
//   shade    a finished ray pushes its shading inputs (hit: normal sum + position; miss:
//            direction) into a SHADE queue; whenever 64 are waiting the whole wave shades 64
//            rays at once (2 normalizes = 6 correctly-rounded divides + 2 sqrt per hit).
//
// The march loop itself is then: evaluation point (6 VALU), map_scene (interpreter), state
// update (~20 VALU), two ballots.  Profiles of v3 showed the refill block (ray-gen + cull +
// shading under a sparse exec mask) costing about as much as map_scene itself once most rays
// of a tile are culled; here it runs at full lane occupancy.
//
// LDS per wave: res[1024] (4 KiB) + ready queue 128 x 16 B (2 KiB) + shade queue 128 x 28 B
// (3.5 KiB) + value-stack spill + cull table (+ program for the LDS policy).
#pragma once
#include "rm_kernel_multi.h"

namespace rmk {

constexpr uint32_t QCAP = 128u;  // both rings; a producer/flush never adds more than 64 to < 64

template <class Prog, bool PROG_IN_LDS>
__global__ __launch_bounds__(64) void rm_render_queue(RmLaunch L) {
    constexpr uint32_t POOL = 1024u;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x;
    float* res = reinterpret_cast<float*>(smem);                 // [1024] one result code per ray
    uint32_t* rq_rid = smem + POOL;                              // ready queue (SoA)
    float* rq_d = reinterpret_cast<float*>(rq_rid + QCAP);       // [3][QCAP]
    uint32_t* sq_rid = reinterpret_cast<uint32_t*>(rq_d + 3u * QCAP);  // shade queue: rid | hit<<31
    float* sq_v = reinterpret_cast<float*>(sq_rid + QCAP);       // [6][QCAP]
    float* spill = sq_v + 6u * QCAP + lane;                      // [spill_depth][64]
    float4* cullt = reinterpret_cast<float4*>(sq_v + 6u * QCAP + L.spill_depth * 64u);
    uint32_t* lprog = reinterpret_cast<uint32_t*>(cullt + L.n_cull);

    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];  // wave-uniform
    float* out = L.out + (size_t)blockIdx.z * L.rows * L.W * 4;

    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);  // wgsl:39-40
    const float eps = 0.0001f;                                    // wgsl:136
    for (uint32_t k = lane; k < L.n_cull; k += 64u) cullt[k] = cull_entry(L.prog[k], ro, L.min_dist);
    if (PROG_IN_LDS) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.prog);
        for (uint32_t k = lane; k < L.n_rec * 8u; k += 64u) lprog[k] = src[k];
    }
    __syncthreads();
    Prog prog;
    if constexpr (PROG_IN_LDS) prog.base = lprog;
    else prog.base = L.prog;

    // this lane's pixel: ray r of the pool belongs to pixel r & 63 (edge tiles clamp)
    const uint32_t tx = blockIdx.x * 8u + (lane & 7u), ty = blockIdx.y * 8u + (lane >> 3);
    const float my_sx = screen_x(tx < L.W ? tx : L.W - 1u, L.W);
    const float my_sy = screen_y(rm_global_row(L, ty < L.rows ? ty : L.rows - 1u), L.H);

    // lane state: evaluation point = b + d * sc
    float bx = 0.f, by = 0.f, bz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, sc = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    uint32_t it = 0u, rid = 0u;
    uint32_t mode = M_EMPTY;
    // wave-uniform cursors
    uint32_t next = 0u;                 // first ungenerated ray of the pool
    uint32_t rq_head = 0u, rq_tail = 0u;  // ready queue ring (monotonic counters)
    uint32_t sq_head = 0u, sq_tail = 0u;  // shade queue ring

    auto shade_batch = [&](uint32_t count) {  // shade `count` (<= 64) entries from the head of the shade queue
        if (lane < count) {
            const uint32_t e = (sq_head + lane) & (QCAP - 1u);
            const uint32_t tag = sq_rid[e];
            const float a0 = sq_v[e], a1 = sq_v[QCAP + e], a2 = sq_v[2u * QCAP + e];
            float code;
            if (tag >> 31) {  // hit: wgsl:98-103
                code = shade_hit(a0, a1, a2, sq_v[3u * QCAP + e], sq_v[4u * QCAP + e], sq_v[5u * QCAP + e]);
            } else {  // marched, did not hit: floor / black, wgsl:117-130
                code = miss_code(ro, a0, a1, a2);
            }
            res[tag & 0x7FFFFFFFu] = code;
        }
        sq_head += count;
    };

    for (;;) {
        // ---- A. produce: keep >= 64 ready rays queued while the pool lasts ----
        while (rq_tail - rq_head < 64u && next < POOL) {
            const uint32_t r = next + lane;  // sample-major: pixel = r & 63 == lane, sample = r >> 6
            const uint32_t s = next >> 6;
            next += 64u;
            float gx, gy, gz;
            gen_ray(u, ro, my_sx, my_sy, s >> 2, s & 3u, gx, gy, gz);
            const bool culled = L.max_iter == 0u || ((L.flags & 1u) && ray_misses_scene(cullt, L.n_cull, gx, gy, gz));
            if (culled) res[r] = miss_code(ro, gx, gy, gz);  // never marched: wgsl:117-130 only
            const unsigned long long keep = __ballot(!culled);
            if (!culled) {
                const uint32_t e = (rq_tail + lane_rank(keep)) & (QCAP - 1u);
                rq_rid[e] = r;
                rq_d[e] = gx; rq_d[QCAP + e] = gy; rq_d[2u * QCAP + e] = gz;
            }
            rq_tail += (uint32_t)__popcll(keep);
        }
        __syncthreads();  // queue writes visible to the whole wave

        // ---- B. consume: idle lanes pop ready rays ----
        {
            const unsigned long long want = __ballot(mode == M_EMPTY);
            const uint32_t avail = rq_tail - rq_head;
            if (want != 0ull && avail != 0u) {
                const uint32_t rank = lane_rank(want);
                if (mode == M_EMPTY && rank < avail) {
                    const uint32_t e = (rq_head + rank) & (QCAP - 1u);
                    rid = rq_rid[e];
                    dx = rq_d[e]; dy = rq_d[QCAP + e]; dz = rq_d[2u * QCAP + e];
                    bx = ro.x; by = ro.y; bz = ro.z;
                    sc = 0.0f;  // dist (wgsl:88)
                    it = 0u;
                    mode = M_MARCH;
                }
                const uint32_t n_want = (uint32_t)__popcll(want);
                rq_head += n_want < avail ? n_want : avail;
            }
        }
        const unsigned long long live = __ballot(mode < M_DONE_HIT);
        if (live == 0ull) break;  // pool generated, queue drained, every ray finished

        // ---- C. one map_scene evaluation per live lane ----
        uint32_t fin = 0u;  // 1: finished as hit, 2: finished without hit
        if (mode < M_DONE_HIT) {
            float qx[1], qy[1], qz[1], v[1];
            qx[0] = bx + dx * sc; qy[0] = by + dy * sc; qz[0] = bz + dz * sc;  // wgsl:91 / :138-141
            SqrtGuard tiny;
            map_scene_multi<1, true>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny);
            if (__ballot(tiny.bad()) != 0ull)  // a sqrt argument outside the fast range (SqrtGuard): redo with the generic sqrt
                map_scene_multi<1, false>(prog, L.n_rec, spill, L.max_dist, qx, qy, qz, v, tiny);
            const float sd = v[0];
            if (mode == M_MARCH) {
                if (sd < L.min_dist) {  // wgsl:97: hit -> normal taps around pos = q
                    bx = qx[0]; by = qy[0]; bz = qz[0];
                    sc = eps;
                    uint32_t sx, sy, sz;
                    tap_signs(0u, sx, sy, sz);
                    dx = __uint_as_float(0x3F800000u ^ sx);
                    dy = __uint_as_float(0x3F800000u ^ sy);
                    dz = __uint_as_float(0x3F800000u ^ sz);
                    mode = M_TAP0;
                } else if (sd > L.max_dist) {  // wgsl:109-111
                    fin = 2u;
                } else {
                    sc += sd;  // wgsl:114
                    it += 1u;
                    if (it >= L.max_iter) fin = 2u;  // loop bound, wgsl:90
                }
            } else {
                const uint32_t t = mode - M_TAP0;  // tap t: n (+)= k_t * f; products with +-1 are exact
                uint32_t sx, sy, sz;
                tap_signs(t, sx, sy, sz);
                const float vx = __uint_as_float(__float_as_uint(sd) ^ sx);
                const float vy = __uint_as_float(__float_as_uint(sd) ^ sy);
                const float vz = __uint_as_float(__float_as_uint(sd) ^ sz);
                nx = t == 0u ? vx : nx + vx;
                ny = t == 0u ? vy : ny + vy;
                nz = t == 0u ? vz : nz + vz;
                tap_signs(t + 1u, sx, sy, sz);
                dx = __uint_as_float(0x3F800000u ^ sx);
                dy = __uint_as_float(0x3F800000u ^ sy);
                dz = __uint_as_float(0x3F800000u ^ sz);
                mode += 1u;
                if (mode == M_DONE_HIT) fin = 1u;
            }
        }

        // ---- D. finished rays -> shade queue; the lane becomes idle ----
        const unsigned long long fin_mask = __ballot(fin != 0u);
        if (fin_mask != 0ull) {
            if (fin != 0u) {
                const uint32_t e = (sq_tail + lane_rank(fin_mask)) & (QCAP - 1u);
                if (fin == 1u) {
                    sq_rid[e] = rid | 0x80000000u;
                    sq_v[e] = nx; sq_v[QCAP + e] = ny; sq_v[2u * QCAP + e] = nz;
                    sq_v[3u * QCAP + e] = bx; sq_v[4u * QCAP + e] = by; sq_v[5u * QCAP + e] = bz;
                } else {
                    sq_rid[e] = rid;
                    sq_v[e] = dx; sq_v[QCAP + e] = dy; sq_v[2u * QCAP + e] = dz;
                }
                mode = M_EMPTY;
            }
            sq_tail += (uint32_t)__popcll(fin_mask);
            __syncthreads();
            if (sq_tail - sq_head >= 64u) shade_batch(64u);
        }
    }
    __syncthreads();
    while (sq_tail != sq_head) {  // flush what is left (< 64 entries unless the loop never shaded)
        const uint32_t n = sq_tail - sq_head;
        shade_batch(n < 64u ? n : 64u);
    }
    __syncthreads();

    // ---- resolve: this lane's pixel, samples in the reference order (wgsl:44-45, 68-69) ----
    if (tx < L.W && ty < L.rows) {
        float tr = 0.0f, tg = 0.0f, tb = 0.0f;
#pragma unroll 4
        for (uint32_t s = 0; s < 16u; s++) {
            const float code = res[s * 64u + lane];
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

}  // namespace rmk
