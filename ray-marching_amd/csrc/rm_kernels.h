// rm_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64).  Device code only.
//
// Replaces ray_marching.wgsl (fs_main :36-76, ray_march :87-131, calculate_normal :135-144,
// map_scene :187-203, eval_cmd* :205-252).  Compile with -ffp-contract=off: every float
// operation below is one IEEE binary32 operation in exactly the order of the arithmetic
// contract (DESIGN.md "Arithmetic contract"), so the result is bit-identical to the oracle.
#pragma once
#if !defined(__HIPCC_RTC__)  // hipRTC pre-includes its own runtime header
#include <hip/hip_runtime.h>
#endif

#include "rm_device.h"

#define RM_DEV __device__ __forceinline__

namespace rmk {

// ---- exact scalar pieces ---------------------------------------------------------------
RM_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }  // v_min_f32: -0 < +0, NaN loses
RM_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }  // v_max_f32

RM_DEV float sdf_sphere(float px, float py, float pz, float cx, float cy, float cz, float r) {
    // wgsl:229-233  length(pos - center) - radius
    float dx = px - cx, dy = py - cy, dz = pz - cz;
    return __builtin_sqrtf((dx * dx + dy * dy) + dz * dz) - r;
}
RM_DEV float sdf_box(float px, float py, float pz, float cx, float cy, float cz, float rx, float ry, float rz) {
    // wgsl:235-240  q = abs(pos-center)-radius; length(max(q,0)) + min(max(q.x,max(q.y,q.z)),0)
    float qx = __builtin_fabsf(px - cx) - rx;
    float qy = __builtin_fabsf(py - cy) - ry;
    float qz = __builtin_fabsf(pz - cz) - rz;
    float mx = fmax_(qx, 0.0f), my = fmax_(qy, 0.0f), mz = fmax_(qz, 0.0f);
    return __builtin_sqrtf((mx * mx + my * my) + mz * mz) + fmin_(fmax_(qx, fmax_(qy, qz)), 0.0f);
}

struct V4 { float x, y, z, w; };
// column-major mat4 * vec4: ((c0*x + c1*y) + c2*z) + c3*w
RM_DEV V4 matvec(const float* m, float x, float y, float z, float w) {
    V4 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

// Pixel centre -> pt_screen (vs_main wgsl:7-20 + rasteriser), framebuffer rows top-down.
RM_DEV float screen_x(uint32_t px, uint32_t W) { return (((float)px + 0.5f) / (float)W) * 2.0f - 1.0f; }
RM_DEV float screen_y(uint32_t py, uint32_t H) { return 1.0f - (((float)py + 0.5f) / (float)H) * 2.0f; }

// Ray direction of AA sample (i,j) through pt_screen (sx,sy): wgsl:52-62.
// AA sample (i, j) of a pixel: its offset on the screen (wgsl:47-53).  The same 16 pairs for every pixel of a frame.
RM_DEV void sample_offset(const rm_uniforms& u, uint32_t i, uint32_t j, float& ox, float& oy) {
    float rx = ((float)i + 0.5f) / 4.0f - 0.5f;
    float ry = ((float)j + 0.5f) / 4.0f - 0.5f;
    ox = rx / u.viewport_extent[0] * 2.0f;
    oy = ry / u.viewport_extent[1] * 2.0f;
}
// The ray through pt_screen + (ox, oy) (wgsl:54-62); m_proj / m_view are the uniform block's matrices (callers may hold
// them in vector registers: an instruction with a scalar-register operand issues at half rate).
RM_DEV void gen_ray_at(const float* m_proj, const float* m_view, const V4& ro, float sx, float sy, float ox, float oy,
                       float& dx, float& dy, float& dz) {
    V4 pv = matvec(m_proj, sx + ox, sy + oy, -1.0f, 1.0f);
    V4 pw = matvec(m_view, pv.x, pv.y, pv.z, pv.w);
    float ex = pw.x - ro.x, ey = pw.y - ro.y, ez = pw.z - ro.z, ew = pw.w - ro.w;
    float len = __builtin_sqrtf(((ex * ex + ey * ey) + ez * ez) + ew * ew);
    dx = ex / len;
    dy = ey / len;
    dz = ez / len;
}
RM_DEV void gen_ray(const rm_uniforms& u, const V4& ro, float sx, float sy, uint32_t i, uint32_t j,
                    float& dx, float& dy, float& dz) {
    float ox, oy;
    sample_offset(u, i, j, ox, oy);
    gen_ray_at(u.inv_proj, u.inv_view, ro, sx, sy, ox, oy, dx, dy, dz);
}

// The same ray up to a positive factor, for the miss tests only (they normalise with unit_dir themselves): pt_world - ro_world
// without wgsl:62's vec4 normalize -- no sqrt, no three correctly rounded divisions.  NOT a value of the arithmetic contract.
RM_DEV void gen_ray_unnormalized_at(const float* m_proj, const float* m_view, const V4& ro, float sx, float sy, float ox, float oy,
                                    float& ex, float& ey, float& ez) {
    V4 pv = matvec(m_proj, sx + ox, sy + oy, -1.0f, 1.0f);
    V4 pw = matvec(m_view, pv.x, pv.y, pv.z, pv.w);
    ex = pw.x - ro.x; ey = pw.y - ro.y; ez = pw.z - ro.z;
}
RM_DEV void gen_ray_unnormalized(const rm_uniforms& u, const V4& ro, float sx, float sy, uint32_t i, uint32_t j,
                                 float& ex, float& ey, float& ez) {
    float rx = ((float)i + 0.5f) / 4.0f - 0.5f;
    float ry = ((float)j + 0.5f) / 4.0f - 0.5f;
    float ox = rx / u.viewport_extent[0] * 2.0f;
    float oy = ry / u.viewport_extent[1] * 2.0f;
    V4 pv = matvec(u.inv_proj, sx + ox, sy + oy, -1.0f, 1.0f);
    V4 pw = matvec(u.inv_view, pv.x, pv.y, pv.z, pv.w);
    ex = pw.x - ro.x; ey = pw.y - ro.y; ez = pw.z - ro.z;
}

// Diffuse intensity of a hit (wgsl:98-103) from the un-normalised tetrahedron sum n.
RM_DEV float shade_hit(float nx, float ny, float nz, float px, float py, float pz) {
    float nl = __builtin_sqrtf((nx * nx + ny * ny) + nz * nz);
    nx = nx / nl; ny = ny / nl; nz = nz / nl;
    float lx = px - 2.0f, ly = py - (-5.0f), lz = pz - 3.0f;  // pos - light_position
    float ll = __builtin_sqrtf((lx * lx + ly * ly) + lz * lz);
    lx = lx / ll; ly = ly / ll; lz = lz / ll;
    return fmax_(0.02f, (nx * lx + ny * ly) + nz * lz);
}

// Floor plane for a ray that did not hit (wgsl:117-130).  Returns the checker bit (0/1)
// or -1 for "black".
RM_DEV int shade_floor(float oy, float ox, float oz, float dx, float dy, float dz) {
    float t = (-1.5f - oy) / dy;
    if (t > 0.0f) {
        float fx = ox + dx * t;
        float fz = oz + dz * t;
        int ix = __float2int_rz(__builtin_rintf(fx + 0.5f));  // v_rndne_f32 + v_cvt_i32_f32 (saturating, NaN->0)
        int iz = __float2int_rz(__builtin_rintf(fz + 0.5f));
        return (ix ^ iz) & 1;
    }
    return -1;
}

// ---- output stage -------------------------------------------------------------------------
// UNORM8 quantisation of one channel as a colour target does it: clamp (NaN -> 0), * 255, round to nearest even.
RM_DEV uint32_t unorm8(float x) { return (uint32_t)__float2int_rn(fmin_(fmax_(x, 0.0f), 1.0f) * 255.0f); }
// Pixel `index` (within the frame that starts `frame` frames into L.out) = (r, g, b, 1).
RM_DEV void store_pixel(const RmLaunch& L, uint32_t frame, size_t index, float r, float g, float b) {
    const size_t at = (size_t)frame * L.rows * L.W + index;
    if (L.out_format == RM_FORMAT_RGBA32F) {
        float4 o;
        o.x = r; o.y = g; o.z = b; o.w = 1.0f;  // wgsl:73-75
        reinterpret_cast<float4*>(L.out)[at] = o;
    } else {
        const uint32_t qr = unorm8(r), qg = unorm8(g), qb = unorm8(b);
        const uint32_t lo = L.out_format == RM_FORMAT_BGRA8_UNORM ? qb : qr, hi = L.out_format == RM_FORMAT_BGRA8_UNORM ? qr : qb;
        reinterpret_cast<uint32_t*>(L.out)[at] = lo | (qg << 8) | (hi << 16) | 0xFF000000u;
    }
}

// ---- program access policies -------------------------------------------------------------
// LDS: the decoded program was staged into shared memory by the workgroup; every lane reads
// the same address (broadcast), the opcode is made scalar with readfirstlane.
struct ProgLds {
    const uint32_t* base;
    RM_DEV void load(uint32_t c, uint32_t& op, float (&p)[7]) const {
        const uint4* q = reinterpret_cast<const uint4*>(base + c * 8);
        uint4 a = q[0], b = q[1];
        op = a.x;  // made scalar (readfirstlane) by the consumer, see exec_command / map_scene
        p[0] = __uint_as_float(a.y); p[1] = __uint_as_float(a.z); p[2] = __uint_as_float(a.w);
        p[3] = __uint_as_float(b.x); p[4] = __uint_as_float(b.y); p[5] = __uint_as_float(b.z);
        p[6] = __uint_as_float(b.w);
    }
};
// SMEM: the program stays in global memory and is fetched through the scalar data cache
// straight into SGPRs (s_load_dwordx8); parameters are then SGPR operands of the VALU ops.
struct ProgSmem {
    const RmRecord* __restrict__ base;
    RM_DEV void load(uint32_t c, uint32_t& op, float (&p)[7]) const {
        const RmRecord& r = base[c];
        op = r.op;
#pragma unroll
        for (int k = 0; k < 7; k++) p[k] = r.p[k];
    }
};

// Value stack below the accumulator: LDS, [slot][thread] so a wave's access is conflict-free.
struct SpillLds {
    float* base;      // &stack[0][tid]
    uint32_t stride;  // threads per workgroup
    RM_DEV void push(uint32_t slot, float v) const { base[slot * stride] = v; }
    RM_DEV float pop(uint32_t slot) const { return base[slot * stride]; }
};

// map_scene (wgsl:187-203): wave-uniform control flow, per-lane position.
template <class Prog>
RM_DEV float map_scene(const Prog& prog, uint32_t n_rec, const SpillLds& st, float max_dist, float px, float py,
                       float pz) {
    if (n_rec == 0u) return max_dist;  // wgsl:189-191
    float acc = 0.0f;
    uint32_t sp = 0;
    for (uint32_t c = 0; c < n_rec; c++) {
        uint32_t op;
        float p[7];
        prog.load(c, op, p);
        op = __builtin_amdgcn_readfirstlane(op);
        const uint32_t kind = RM_OP_KIND(op), mode = RM_OP_MODE(op);  // reference node types only (host rejects extensions here)
        float a, b;
        if (kind == RM_KIND_POP) {
            b = acc;
            a = st.pop(--sp);
        } else {
            float v = kind == RM_KIND_SPHERE ? sdf_sphere(px, py, pz, p[0], p[1], p[2], p[3])
                                             : sdf_box(px, py, pz, p[0], p[1], p[2], p[3], p[4], p[5]);
            if (op & RM_OP_SPILL) st.push(sp++, acc);
            a = acc;
            b = v;
        }
        if (mode == RM_MODE_PUSH) acc = b;
        else if (mode == RM_MODE_UNION) acc = fmin_(a, b);  // wgsl:242-246
        else acc = fmax_(a, -b);                            // wgsl:248-252
    }
    return acc;
}

// =============================================================================================
// Kernel v1 "pixel": one thread per pixel, 16x16-pixel workgroup (4 waves of 8x8), program
// staged once per workgroup into LDS, AA samples and march steps in the reference's loop order.
// =============================================================================================
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
__global__ __launch_bounds__(256) void rm_render_pixel(RmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t prog_dwords = L.n_rec * 8u;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.prog);
        for (uint32_t k = tid; k < prog_dwords; k += 256u) smem[k] = src[k];
    }
    __syncthreads();
    ProgLds prog{smem};
    SpillLds st{reinterpret_cast<float*>(smem + prog_dwords) + tid, 256u};

    rm_uniforms u = L.u;
    if (L.frames) u = L.frames[blockIdx.z];  // wave-uniform: stays in SGPRs
    float* out = L.out + (size_t)blockIdx.z * L.rows * L.W * 4;

    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);
    const uint32_t px = blockIdx.x * 16u + lx;
    const uint32_t ry = blockIdx.y * 16u + ly;  // row inside the band
    if (px >= L.W || ry >= L.rows) return;       // no barrier after this point
    const uint32_t py = rm_global_row(L, ry);

    const V4 ro = matvec(u.inv_view, 0.0f, 0.0f, 0.0f, 1.0f);  // wgsl:39-40
    const float sx = screen_x(px, L.W), sy = screen_y(py, L.H);
    float tr = 0.0f, tg = 0.0f, tb = 0.0f;

#pragma unroll 1
    for (uint32_t s = 0; s < 16u; s++) {  // wgsl:44-45: i outer, j inner
        float dx, dy, dz;
        gen_ray(u, ro, sx, sy, s >> 2, s & 3u, dx, dy, dz);
        float dist = 0.0f;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        bool hit = false;
#pragma unroll 1
        for (uint32_t it = 0; it < L.max_iter; it++) {  // wgsl:90
            float qx = ro.x + dx * dist, qy = ro.y + dy * dist, qz = ro.z + dz * dist;
            float sd = map_scene(prog, L.n_rec, st, L.max_dist, qx, qy, qz);
            if (sd < L.min_dist) {  // wgsl:97
                const float eps = 0.0001f;
                // wgsl:135-144, taps k.xyy, k.yyx, k.yxy, k.xxx with k = (1,-1)
                float f0 = map_scene(prog, L.n_rec, st, L.max_dist, qx + eps, qy + -eps, qz + -eps);
                float f1 = map_scene(prog, L.n_rec, st, L.max_dist, qx + -eps, qy + -eps, qz + eps);
                float f2 = map_scene(prog, L.n_rec, st, L.max_dist, qx + -eps, qy + eps, qz + -eps);
                float f3 = map_scene(prog, L.n_rec, st, L.max_dist, qx + eps, qy + eps, qz + eps);
                float nx = ((f0 + -f1) + -f2) + f3;
                float ny = ((-f0 + -f1) + f2) + f3;
                float nz = ((-f0 + f1) + -f2) + f3;
                float k = shade_hit(nx, ny, nz, qx, qy, qz);
                cr = 0.4f * k; cg = 0.7f * k; cb = 0.1f * k;  // wgsl:105
                hit = true;
                break;
            }
            if (sd > L.max_dist) break;  // wgsl:109-111
            dist += sd;                  // wgsl:114
        }
        if (!hit) {
            int c = shade_floor(ro.y, ro.x, ro.z, dx, dy, dz);
            if (c >= 0) {
                float g = 0.2f * (float)c;  // wgsl:127
                cr = 0.1f + g; cg = 0.1f + g; cb = 0.2f + g;
            }
        }
        tr += __builtin_sqrtf(cr);  // wgsl:68-69
        tg += __builtin_sqrtf(cg);
        tb += __builtin_sqrtf(cb);
    }
    float4 o;
    o.x = tr / 16.0f; o.y = tg / 16.0f; o.z = tb / 16.0f; o.w = 1.0f;  // wgsl:73-75
    reinterpret_cast<float4*>(out)[(size_t)ry * L.W + px] = o;
}
#endif

// ---- pieces shared by the ray-pool kernels (rm_kernel_v5.h) ---------------------------------
// Per-sample results are parked in LDS as ONE float each (hit: diffuse intensity k >= 0.02;
// miss: -1 - checker bit; black: -3) and resolved per pixel in the reference's sample order
// (i outer, j inner: wgsl:44-45, 68-69), so the sum is bit-identical to the nested loops.
enum : uint32_t { M_MARCH = 0, M_TAP0 = 1, M_TAP3 = 4, M_DONE_HIT = 5, M_DONE_MISS = 6, M_EMPTY = 7, M_RETIRED = 8 };

RM_DEV uint32_t lane_rank(unsigned long long mask) {  // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Sign bits of tetrahedron tap t (wgsl:137-143): k.xyy, k.yyx, k.yxy, k.xxx with k = (1,-1).
RM_DEV void tap_signs(uint32_t t, uint32_t& sx, uint32_t& sy, uint32_t& sz) {
    sx = ((t ^ (t >> 1)) & 1u) << 31;  // negative for t = 1, 2
    sy = ((~t >> 1) & 1u) << 31;       // negative for t = 0, 1
    sz = (~t & 1u) << 31;              // negative for t = 0, 2
}

// Stream-write calibration kernel: 16 B per lane, grid-stride.
#if !defined(RM_JIT_TU)  // not part of a specialised translation unit (rm_jit.h)
__global__ __launch_bounds__(256) void rm_fill(float4* dst, size_t n_vec, float v) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256u;
    float4 val = make_float4(v, v, v, v);
    for (; i < n_vec; i += stride) dst[i] = val;
}
#endif

}  // namespace rmk
