/*
 * rm_oracle.c -- CPU restatement of the reference SDF ray-marcher (TEST INFRASTRUCTURE).
 *
 * This file is the parity ORACLE for the HIP hot path.  It is test infrastructure:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (ray-marching_amd/, include/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" against the real wgpu/naga render.  The reference
 * (Rust + WGSL) ships no tests, golden images or fixtures, and neither rustc/cargo nor
 * any WGSL/Vulkan implementation exists in the build container, so the reference can
 * not be executed.  This restatement follows the reference TEXT line by line (cited
 * below as file:line relative to the reference root) and is pinned only by
 *   (a) the hand-derived known answers of SURVEY.md section 8(c) (tests/test_oracle_*.py),
 *   (b) an independently written numpy restatement (oracle/rm_oracle_np.py) that must
 *       agree bit-for-bit, and
 *   (c) closed-form ray/sphere intersection checks.
 *
 * ARITHMETIC CONTRACT (shared with the HIP kernel; every op is IEEE-754 binary32,
 * round-to-nearest-even, NO fused multiply-add, correctly rounded sqrt and divide):
 *   mat*vec   : r = ((c0*x + c1*y) + c2*z) + c3*w     per component, column-major
 *   dot3      : (x*x' + y*y') + z*z'                   dot4 continues "+ w*w'"
 *   length(v) : sqrt(dot(v,v))          normalize(v): v / length(v)   (true divides)
 *   min/max   : IEEE-754-2019 minimumNumber/maximumNumber (-0 < +0, NaN loses)
 *   round     : ties-to-even (rintf)    f32->i32: truncate, saturate, NaN -> 0
 * Build with: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RMO_API __attribute__((visibility("default")))

/* ---- blob layouts (renderer.rs:29-41, ray_marching.wgsl:22-31,78-82) ------------- */
typedef struct {
    float viewport_extent[2]; /* @0  */
    float _pad[2];            /* @8  (std140-style alignment of mat4x4 to 16) */
    float inv_proj[16];       /* @16 column-major */
    float inv_view[16];       /* @80 column-major */
} rmo_uniforms;               /* 144 B */

typedef struct {
    float min_dist;
    float max_dist;
    uint32_t max_iter;
} rmo_limits; /* 12 B */

typedef struct {
    uint64_t march_steps;   /* map_scene calls made from the march loop */
    uint64_t normal_taps;   /* map_scene calls made from calculate_normal */
    uint64_t rays;
    uint64_t hits;
    uint64_t floor_hits;
    uint64_t sky;
} rmo_counters;

enum {
    RMO_OK = 0,
    RMO_ERR_NULL = -1,
    RMO_ERR_TRUNCATED = -2,  /* a command reads past n_words */
    RMO_ERR_UNDERFLOW = -3,  /* binary operator with < 2 values on the stack */
    RMO_ERR_OVERFLOW = -4,   /* value stack deeper than 32 (wgsl:173) */
    RMO_ERR_EMPTY_RESULT = -5,
    RMO_ERR_OPCODE = -6,     /* unknown opcode (only in strict mode) */
    RMO_ERR_TRANSFORM = -12, /* transform push/pop not nested properly, deeper than 8, or not around exactly one value */
    RMO_ERR_MATERIAL = -13   /* a Material command names an index outside the material table (or >= 256) */
};

/* ---- opcode numbering (csg/builder.rs:1-24) ------------------------------------- */
#define RMO_CMD_SPHERE 0u
#define RMO_CMD_BOX 1u
#define RMO_CMD_UNION 100u
#define RMO_CMD_SUBTRACTION 101u
/* ---- extensions (NOT implemented by the reference; semantics defined by this repo, DESIGN.md
 * "Extension node types"; parity with the reference is undefined for them) ------------------
 * Plane = 2 and Intersection = 102 are the slots the reference reserves by comment
 * (builder.rs:8,14; csg/mod.rs:34,39).  Cylinder and SmoothUnion (BASELINE.json configs 2-3)
 * take numbers outside every reserved slot (2, 102, 200-205). */
#define RMO_CMD_PLANE 2u          /* normal vec3, h f32:            dot(p, n) + h                  */
#define RMO_CMD_CYLINDER 10u      /* center vec3, radius, half_h:   capped cylinder along y         */
#define RMO_CMD_INTERSECTION 102u /*                                max(a, b)                       */
#define RMO_CMD_SMOOTH_UNION 110u /* k f32:                         polynomial smooth minimum       */
/* Space transformations: the six slots the reference reserves by comment (builder.rs:16-23: "1 child, transforms
 * space").  A node is  Push(params), <the child's commands>, Pop.  Push saves the evaluation position and replaces
 * it; Pop restores it (ScalePop also multiplies the child's value by the scale).  Both count as commands. */
#define RMO_CMD_TRANSLATION_PUSH 200u /* t vec3:             pos = pos - t                                  */
#define RMO_CMD_TRANSLATION_POP 201u
#define RMO_CMD_ROTATION_PUSH 202u    /* q = (w,i,j,k) unit:  pos = conj(q) pos q  (the CHILD is rotated by q) */
#define RMO_CMD_ROTATION_POP 203u
#define RMO_CMD_SCALE_PUSH 204u       /* s f32 (uniform):     pos = pos / s                                  */
#define RMO_CMD_SCALE_POP 205u        /*                      value = value * s                              */
#define RMO_MAX_XFORM_DEPTH 8u
/* Materials (extension; README.md:11 lists a material system as future work, the reference has none): a unary
 * postfix tag.  Every value on the stack carries a material index next to its distance; primitives push index 0,
 * Material(i) sets the index of the value on top, binary operators keep the index of the operand that decides the
 * result (see map_scene_impl).  The hit colour becomes albedo[index] * diffuse; entry 0 of the default table is the
 * reference's (0.4, 0.7, 0.1) (wgsl:105), so a program without Material commands renders as before. */
#define RMO_CMD_MATERIAL 300u     /* index u32 (a plain integer word, not f32 bits) */
#define RMO_MAX_MATERIALS 256u

/* ---- scalar helpers --------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* minimumNumber / maximumNumber with -0 < +0 (what v_min_f32 / v_max_f32 compute). */
static inline float rmo_min(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) >> 31) ? a : b;
    return a < b ? a : b;
}
static inline float rmo_max(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) >> 31) ? b : a;
    return a > b ? a : b;
}
/* WGSL i32(f32): round toward zero, clamp to range; NaN -> 0 (v_cvt_i32_f32). */
static inline int32_t rmo_f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v4 mat4_mul_vec4(const float* m, v4 v) {
    v4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) {
    float l = length3(a);
    v3 r = { a.x / l, a.y / l, a.z / l };
    return r;
}

/* ---- scene program ------------------------------------------------------------------ */
typedef struct {
    uint32_t cmd_count;
    const uint32_t* words;
    uint32_t n_words;
    rmo_limits limits;
    const float* materials; /* extension: n_materials x (r, g, b); NULL = the one-entry default table */
    uint32_t n_materials;
    int has_materials;      /* the program contains Material commands */
} rmo_scene;

/* Validates what the reference leaves as UB (wgsl:177-185 has no bounds checks). */
RMO_API int rmo_validate_program(uint32_t cmd_count, const uint32_t* words, uint32_t n_words,
                                 int strict_opcodes, uint32_t* out_max_depth) {
    uint32_t ptr = 0, depth = 0, max_depth = 0;
    uint32_t xop[RMO_MAX_XFORM_DEPTH], xdepth[RMO_MAX_XFORM_DEPTH], xsize = 0; /* open transform scopes */
    if (cmd_count && !words) return RMO_ERR_NULL;
    for (uint32_t i = 0; i < cmd_count; i++) {
        if (ptr >= n_words) return RMO_ERR_TRUNCATED;
        uint32_t op = words[ptr++];
        switch (op) {
        case RMO_CMD_TRANSLATION_PUSH:
        case RMO_CMD_ROTATION_PUSH:
        case RMO_CMD_SCALE_PUSH: {
            uint32_t np = op == RMO_CMD_TRANSLATION_PUSH ? 3u : op == RMO_CMD_ROTATION_PUSH ? 4u : 1u;
            if (ptr + np > n_words) return RMO_ERR_TRUNCATED;
            ptr += np;
            if (xsize == RMO_MAX_XFORM_DEPTH) return RMO_ERR_TRANSFORM;
            xop[xsize] = op;
            xdepth[xsize++] = depth;
        } continue;
        case RMO_CMD_TRANSLATION_POP:
        case RMO_CMD_ROTATION_POP:
        case RMO_CMD_SCALE_POP:
            /* closes the innermost scope, which must be of its kind and have produced exactly one value */
            if (xsize == 0 || xop[xsize - 1] + 1u != op || depth != xdepth[xsize - 1] + 1u) return RMO_ERR_TRANSFORM;
            xsize--;
            continue;
        case RMO_CMD_MATERIAL: /* extension: tags the value on top of the stack */
            if (ptr + 1 > n_words) return RMO_ERR_TRUNCATED;
            if (words[ptr] >= RMO_MAX_MATERIALS) return RMO_ERR_MATERIAL;
            ptr += 1;
            if (depth < 1) return RMO_ERR_UNDERFLOW;
            /* inside a transform scope the tagged value must be the scope's own */
            if (xsize != 0 && depth <= xdepth[xsize - 1]) return RMO_ERR_UNDERFLOW;
            continue;
        case RMO_CMD_SPHERE:
            if (ptr + 4 > n_words) return RMO_ERR_TRUNCATED;
            ptr += 4; depth++; break;
        case RMO_CMD_BOX:
            if (ptr + 6 > n_words) return RMO_ERR_TRUNCATED;
            ptr += 6; depth++; break;
        case RMO_CMD_PLANE:
            if (ptr + 4 > n_words) return RMO_ERR_TRUNCATED;
            ptr += 4; depth++; break;
        case RMO_CMD_CYLINDER:
            if (ptr + 5 > n_words) return RMO_ERR_TRUNCATED;
            ptr += 5; depth++; break;
        case RMO_CMD_SMOOTH_UNION:
            if (ptr + 1 > n_words) return RMO_ERR_TRUNCATED;
            ptr += 1;
            if (depth < 2) return RMO_ERR_UNDERFLOW;
            depth--; break;
        case RMO_CMD_UNION:
        case RMO_CMD_SUBTRACTION:
        case RMO_CMD_INTERSECTION:
            if (depth < 2) return RMO_ERR_UNDERFLOW;
            depth--; break;
        default:
            if (strict_opcodes) return RMO_ERR_OPCODE;
            depth++; break; /* wgsl:223-225: evaluates to 0.0, which is pushed (wgsl:199) */
        }
        if (depth > 32) return RMO_ERR_OVERFLOW;
        if (depth > max_depth) max_depth = depth;
    }
    if (xsize != 0) return RMO_ERR_TRANSFORM;
    if (cmd_count && depth < 1) return RMO_ERR_EMPTY_RESULT;
    if (out_max_depth) *out_max_depth = max_depth;
    return RMO_OK;
}

/* Rotation of p by conj(q), q = (w, a) a unit quaternion: p' = p + w t + (-a) x t with t = 2 ((-a) x p), every
 * product and difference a separate binary32 operation in the order written (the kernels do the same). */
static v3 rotate_conj(float w, v3 a, v3 p) {
    v3 c = { p.y * a.z - p.z * a.y, p.z * a.x - p.x * a.z, p.x * a.y - p.y * a.x }; /* p x a = (-a) x p */
    v3 t = { 2.0f * c.x, 2.0f * c.y, 2.0f * c.z };
    v3 u = { t.y * a.z - t.z * a.y, t.z * a.x - t.x * a.z, t.x * a.y - t.y * a.x }; /* t x a = (-a) x t */
    v3 r = { (p.x + w * t.x) + u.x, (p.y + w * t.y) + u.y, (p.z + w * t.z) + u.z };
    return r;
}

/* map_scene (wgsl:187-203) with eval_cmd* (wgsl:205-252), stream reader (wgsl:152-170)
 * and value stack (wgsl:173-185) inlined. */
/* mat_out != NULL (extension): also track the material index of every value; *mat_out = the index of the result.
 * Which operand "decides": Union / SmoothUnion b < a, Subtraction -b > a, Intersection b > a take b's index,
 * otherwise (ties, NaN) a's -- a was pushed first (wgsl:242-252). */
static inline __attribute__((always_inline)) float map_scene_impl(const rmo_scene* sc, v3 pos, uint32_t* mat_out) {
    if (mat_out) *mat_out = 0u;
    if (sc->cmd_count == 0u) return sc->limits.max_dist; /* wgsl:189-191 */
    float stack[32];
    uint32_t mstack[32];
    uint32_t vmat = 0u;
    uint32_t size = 0, ptr = 0; /* wgsl:194-195 */
    v3 pstack[RMO_MAX_XFORM_DEPTH];    /* extension: positions saved by transform pushes */
    float sstack[RMO_MAX_XFORM_DEPTH]; /*            and the scale of ScalePush */
    uint32_t xsize = 0;
    const uint32_t* w = sc->words;
    for (uint32_t idx = 0; idx < sc->cmd_count; idx++) {
        uint32_t cmd = w[ptr++]; /* wgsl:198 */
        float val;
        vmat = 0u;
        switch (cmd) {
        case RMO_CMD_MATERIAL: /* extension */
            if (mat_out) mstack[size - 1] = w[ptr];
            ptr += 1;
            continue;
        case RMO_CMD_TRANSLATION_PUSH: { /* extension */
            v3 t = { u2f(w[ptr]), u2f(w[ptr + 1]), u2f(w[ptr + 2]) };
            ptr += 3;
            pstack[xsize++] = pos;
            pos.x = pos.x - t.x; pos.y = pos.y - t.y; pos.z = pos.z - t.z;
        } continue;
        case RMO_CMD_ROTATION_PUSH: { /* extension */
            float qw = u2f(w[ptr]);
            v3 a = { u2f(w[ptr + 1]), u2f(w[ptr + 2]), u2f(w[ptr + 3]) };
            ptr += 4;
            pstack[xsize++] = pos;
            pos = rotate_conj(qw, a, pos);
        } continue;
        case RMO_CMD_SCALE_PUSH: { /* extension */
            float sfac = u2f(w[ptr]);
            ptr += 1;
            pstack[xsize] = pos;
            sstack[xsize++] = sfac;
            pos.x = pos.x / sfac; pos.y = pos.y / sfac; pos.z = pos.z / sfac;
        } continue;
        case RMO_CMD_TRANSLATION_POP:
        case RMO_CMD_ROTATION_POP:
            pos = pstack[--xsize];
            continue;
        case RMO_CMD_SCALE_POP:
            --xsize;
            pos = pstack[xsize];
            stack[size - 1] = stack[size - 1] * sstack[xsize];
            continue;
        case RMO_CMD_SPHERE: { /* wgsl:229-233 */
            v3 c = { u2f(w[ptr]), u2f(w[ptr + 1]), u2f(w[ptr + 2]) };
            float r = u2f(w[ptr + 3]);
            ptr += 4;
            v3 d = { pos.x - c.x, pos.y - c.y, pos.z - c.z };
            val = length3(d) - r;
        } break;
        case RMO_CMD_BOX: { /* wgsl:235-240 */
            v3 c = { u2f(w[ptr]), u2f(w[ptr + 1]), u2f(w[ptr + 2]) };
            v3 r = { u2f(w[ptr + 3]), u2f(w[ptr + 4]), u2f(w[ptr + 5]) };
            ptr += 6;
            v3 q = { fabsf(pos.x - c.x) - r.x, fabsf(pos.y - c.y) - r.y, fabsf(pos.z - c.z) - r.z };
            v3 qp = { rmo_max(q.x, 0.0f), rmo_max(q.y, 0.0f), rmo_max(q.z, 0.0f) };
            val = length3(qp) + rmo_min(rmo_max(q.x, rmo_max(q.y, q.z)), 0.0f);
        } break;
        case RMO_CMD_PLANE: { /* extension: dot(pos, n) + h */
            v3 n = { u2f(w[ptr]), u2f(w[ptr + 1]), u2f(w[ptr + 2]) };
            float h = u2f(w[ptr + 3]);
            ptr += 4;
            val = dot3(pos, n) + h;
        } break;
        case RMO_CMD_CYLINDER: { /* extension: capped cylinder along y (exact SDF) */
            v3 c = { u2f(w[ptr]), u2f(w[ptr + 1]), u2f(w[ptr + 2]) };
            float r = u2f(w[ptr + 3]), hh = u2f(w[ptr + 4]);
            ptr += 5;
            float dx = pos.x - c.x, dz = pos.z - c.z;
            float qx = sqrtf(dx * dx + dz * dz) - r;
            float qy = fabsf(pos.y - c.y) - hh;
            float mx = rmo_max(qx, 0.0f), my = rmo_max(qy, 0.0f);
            val = rmo_min(rmo_max(qx, qy), 0.0f) + sqrtf(mx * mx + my * my);
        } break;
        case RMO_CMD_INTERSECTION: { /* extension */
            float b = stack[--size];
            float a = stack[--size];
            val = rmo_max(a, b);
            if (mat_out) vmat = b > a ? mstack[size + 1] : mstack[size];
        } break;
        case RMO_CMD_SMOOTH_UNION: { /* extension: min(a,b) - h*h*k/4, h = max(k-|a-b|,0)/k; k <= 0: plain min */
            float k = u2f(w[ptr]);
            ptr += 1;
            float b = stack[--size];
            float a = stack[--size];
            val = rmo_min(a, b);
            if (k > 0.0f) {
                float h = rmo_max(k - fabsf(a - b), 0.0f) / k;
                val = val - ((h * h) * k) * 0.25f;
            }
            if (mat_out) vmat = b < a ? mstack[size + 1] : mstack[size];
        } break;
        case RMO_CMD_UNION: { /* wgsl:242-246 */
            float b = stack[--size];
            float a = stack[--size];
            val = rmo_min(a, b);
            if (mat_out) vmat = b < a ? mstack[size + 1] : mstack[size];
        } break;
        case RMO_CMD_SUBTRACTION: { /* wgsl:248-252 */
            float b = stack[--size];
            float a = stack[--size];
            val = rmo_max(a, -b);
            if (mat_out) vmat = -b > a ? mstack[size + 1] : mstack[size];
        } break;
        default: /* wgsl:223-225 */
            val = 0.0f;
            break;
        }
        if (mat_out) mstack[size] = vmat;
        stack[size++] = val; /* wgsl:199 */
    }
    if (mat_out) *mat_out = mstack[size - 1];
    return stack[--size]; /* wgsl:202 */
}
static float map_scene(const rmo_scene* sc, v3 pos) { return map_scene_impl(sc, pos, NULL); }
static uint32_t map_scene_material(const rmo_scene* sc, v3 pos) {
    uint32_t m = 0u;
    (void)map_scene_impl(sc, pos, &m);
    return m;
}
static int program_has_materials(uint32_t cmd_count, const uint32_t* w, uint32_t* max_index) { /* after validation */
    uint32_t ptr = 0, found = 0, mx = 0;
    for (uint32_t i = 0; i < cmd_count; i++) {
        uint32_t op = w[ptr++];
        switch (op) {
        case RMO_CMD_MATERIAL: found = 1; if (w[ptr] > mx) mx = w[ptr]; ptr += 1; break;
        case RMO_CMD_SPHERE: case RMO_CMD_PLANE: case RMO_CMD_ROTATION_PUSH: ptr += 4; break;
        case RMO_CMD_BOX: ptr += 6; break;
        case RMO_CMD_CYLINDER: ptr += 5; break;
        case RMO_CMD_TRANSLATION_PUSH: ptr += 3; break;
        case RMO_CMD_SMOOTH_UNION: case RMO_CMD_SCALE_PUSH: ptr += 1; break;
        default: break;
        }
    }
    if (max_index) *max_index = mx;
    return (int)found;
}
/* Fills the material fields of a validated scene; RMO_ERR_MATERIAL when the program names an index the table lacks. */
static int scene_set_materials(rmo_scene* sc, const float* rgb, uint32_t n) {
    uint32_t mx = 0;
    sc->has_materials = program_has_materials(sc->cmd_count, sc->words, &mx);
    sc->materials = n ? rgb : NULL;
    sc->n_materials = n ? n : 1u;
    if (n > RMO_MAX_MATERIALS || (n && !rgb)) return RMO_ERR_MATERIAL;
    if (sc->has_materials && mx >= sc->n_materials) return RMO_ERR_MATERIAL;
    return RMO_OK;
}

RMO_API float rmo_map_scene(uint32_t cmd_count, const uint32_t* words, uint32_t n_words,
                            const rmo_limits* lim, const float* pos3) {
    rmo_scene sc = { cmd_count, words, n_words, *lim, NULL, 1u, 0 };
    if (rmo_validate_program(cmd_count, words, n_words, 0, NULL) != RMO_OK) return NAN;
    v3 p = { pos3[0], pos3[1], pos3[2] };
    return map_scene(&sc, p);
}
/* extension: the material index map_scene's result carries at pos (0xFFFFFFFF for an invalid program) */
RMO_API uint32_t rmo_map_scene_material(uint32_t cmd_count, const uint32_t* words, uint32_t n_words,
                                        const rmo_limits* lim, const float* pos3) {
    rmo_scene sc = { cmd_count, words, n_words, *lim, NULL, 1u, 0 };
    if (rmo_validate_program(cmd_count, words, n_words, 0, NULL) != RMO_OK) return 0xFFFFFFFFu;
    v3 p = { pos3[0], pos3[1], pos3[2] };
    return map_scene_material(&sc, p);
}

/* calculate_normal (wgsl:135-144): tetrahedron taps, eps = 0.0001, k = (1,-1). */
static v3 calculate_normal(const rmo_scene* sc, v3 pos, rmo_counters* cnt) {
    const float eps = 0.0001f;
    const float kx = 1.0f, ky = -1.0f;
    /* k.xyy, k.yyx, k.yxy, k.xxx */
    const v3 k[4] = { { kx, ky, ky }, { ky, ky, kx }, { ky, kx, ky }, { kx, kx, kx } };
    v3 acc = { 0, 0, 0 };
    for (int t = 0; t < 4; t++) {
        v3 p = { pos.x + k[t].x * eps, pos.y + k[t].y * eps, pos.z + k[t].z * eps };
        float f = map_scene(sc, p);
        v3 term = { k[t].x * f, k[t].y * f, k[t].z * f };
        if (t == 0) acc = term;
        else { acc.x = acc.x + term.x; acc.y = acc.y + term.y; acc.z = acc.z + term.z; }
    }
    if (cnt) cnt->normal_taps += 4;
    return normalize3(acc);
}

/* ray_march (wgsl:87-131). */
static v3 ray_march(const rmo_scene* sc, v3 o, v3 d, rmo_counters* cnt) {
    float dist = 0.0f;
    const rmo_limits* L = &sc->limits;
    if (cnt) cnt->rays++;
    for (uint32_t i = 0; i < L->max_iter; i++) { /* wgsl:90 */
        v3 pos = { o.x + d.x * dist, o.y + d.y * dist, o.z + d.z * dist }; /* wgsl:91 */
        float scene_dist = map_scene(sc, pos);                               /* wgsl:94 */
        if (cnt) cnt->march_steps++;
        if (scene_dist < L->min_dist) { /* wgsl:97-106 */
            v3 n = calculate_normal(sc, pos, cnt);
            v3 tl = { pos.x - 2.0f, pos.y - (-5.0f), pos.z - 3.0f }; /* pos - light_position */
            v3 dl = normalize3(tl);
            float diffuse = rmo_max(0.02f, dot3(n, dl));
            v3 c = { 0.4f * diffuse, 0.7f * diffuse, 0.1f * diffuse }; /* wgsl:105 */
            if (sc->has_materials) { /* extension: albedo of the material the surface carries at pos */
                uint32_t m = map_scene_material(sc, pos);
                if (sc->materials) {
                    const float* al = sc->materials + 3u * m;
                    c.x = al[0] * diffuse; c.y = al[1] * diffuse; c.z = al[2] * diffuse;
                }
            }
            if (cnt) cnt->hits++;
            return c;
        }
        if (scene_dist > L->max_dist) break; /* wgsl:109-111 */
        dist += scene_dist;                  /* wgsl:114 */
    }
    /* floor plane (wgsl:117-128) */
    const float floor_y = -1.5f;
    float floor_dist = (floor_y - o.y) / d.y;
    if (floor_dist > 0.0f) {
        float px = o.x + d.x * floor_dist;
        float pz = o.z + d.z * floor_dist;
        int32_t ix = rmo_f2i(rintf(px + 0.5f));
        int32_t iz = rmo_f2i(rintf(pz + 0.5f));
        float col = (float)((ix ^ iz) & 1);
        float g = 0.2f * col;
        v3 c = { 0.1f + g, 0.1f + g, 0.2f + g };
        if (cnt) cnt->floor_hits++;
        return c;
    }
    if (cnt) cnt->sky++;
    v3 z = { 0, 0, 0 };
    return z; /* wgsl:130 */
}

RMO_API void rmo_ray_march(uint32_t cmd_count, const uint32_t* words, uint32_t n_words,
                           const rmo_limits* lim, const float* o3, const float* d3, float* rgb3) {
    rmo_scene sc = { cmd_count, words, n_words, *lim, NULL, 1u, 0 };
    v3 o = { o3[0], o3[1], o3[2] }, d = { d3[0], d3[1], d3[2] };
    v3 c = { NAN, NAN, NAN };
    if (rmo_validate_program(cmd_count, words, n_words, 0, NULL) == RMO_OK)
        c = ray_march(&sc, o, d, NULL);
    rgb3[0] = c.x; rgb3[1] = c.y; rgb3[2] = c.z;
}

/* Pixel-centre mapping implied by vs_main (wgsl:7-20) + rasteriser: framebuffer pixel
 * (px,py) counted from the top-left, pt_screen y up. */
static inline float pt_screen_x(uint32_t px, uint32_t W) {
    return (((float)px + 0.5f) / (float)W) * 2.0f - 1.0f;
}
static inline float pt_screen_y(uint32_t py, uint32_t H) {
    return 1.0f - (((float)py + 0.5f) / (float)H) * 2.0f;
}

/* fs_main (wgsl:36-76) for one pixel. */
static void fs_main(const rmo_scene* sc, const rmo_uniforms* u, uint32_t px, uint32_t py,
                    uint32_t W, uint32_t H, float* rgba, rmo_counters* cnt) {
    const v4 ro_view = { 0.0f, 0.0f, 0.0f, 1.0f };
    v4 ro_world = mat4_mul_vec4(u->inv_view, ro_view); /* wgsl:39-40 */
    float sx = pt_screen_x(px, W), sy = pt_screen_y(py, H);
    v3 total = { 0, 0, 0 };
    const float aa = 4.0f; /* wgsl:34 */
    for (uint32_t i = 0; i < 4u; i++) {
        for (uint32_t j = 0; j < 4u; j++) {
            float rx = ((float)i + 0.5f) / aa - 0.5f; /* wgsl:52 */
            float ry = ((float)j + 0.5f) / aa - 0.5f;
            float ox = rx / u->viewport_extent[0] * 2.0f; /* wgsl:53 */
            float oy = ry / u->viewport_extent[1] * 2.0f;
            v4 pt_ndc = { sx + ox, sy + oy, -1.0f, 1.0f }; /* wgsl:56-57 */
            v4 pt_view = mat4_mul_vec4(u->inv_proj, pt_ndc); /* wgsl:58 */
            v4 pt_world = mat4_mul_vec4(u->inv_view, pt_view); /* wgsl:59 */
            v4 df = { pt_world.x - ro_world.x, pt_world.y - ro_world.y,
                      pt_world.z - ro_world.z, pt_world.w - ro_world.w };
            float len = sqrtf(((df.x * df.x + df.y * df.y) + df.z * df.z) + df.w * df.w);
            v3 rd = { df.x / len, df.y / len, df.z / len }; /* wgsl:62 (vec4 normalize, .xyz) */
            v3 ro = { ro_world.x, ro_world.y, ro_world.z };
            v3 c = ray_march(sc, ro, rd, cnt); /* wgsl:65 */
            total.x = total.x + sqrtf(c.x);    /* wgsl:68-69 */
            total.y = total.y + sqrtf(c.y);
            total.z = total.z + sqrtf(c.z);
        }
    }
    rgba[0] = total.x / 16.0f; /* wgsl:73 */
    rgba[1] = total.y / 16.0f;
    rgba[2] = total.z / 16.0f;
    rgba[3] = 1.0f; /* wgsl:75 */
}

/* Render rows [row0,row0+rows) of a WxH image into out (rows*W*4 floats, RGBA32F,
 * row-major, top row first).  Single thread.  counters may be NULL. */
RMO_API int rmo_render(const rmo_uniforms* u, const rmo_limits* lim, uint32_t cmd_count,
                       const uint32_t* words, uint32_t n_words, uint32_t W, uint32_t H,
                       uint32_t row0, uint32_t rows, float* out, rmo_counters* counters) {
    if (!u || !lim || !out) return RMO_ERR_NULL;
    int rc = rmo_validate_program(cmd_count, words, n_words, 0, NULL);
    if (rc != RMO_OK) return rc;
    rmo_scene sc = { cmd_count, words, n_words, *lim, NULL, 1u, 0 };
    rc = scene_set_materials(&sc, NULL, 0);
    if (rc != RMO_OK) return rc;
    rmo_counters c;
    memset(&c, 0, sizeof c);
    for (uint32_t r = 0; r < rows; r++)
        for (uint32_t x = 0; x < W; x++)
            fs_main(&sc, u, x, row0 + r, W, H, out + ((size_t)r * W + x) * 4, counters ? &c : NULL);
    if (counters) *counters = c;
    return RMO_OK;
}

/* ---- multi-threaded variant (the CPU baseline of bench.py) -------------------------- */
typedef struct {
    const rmo_scene* sc;
    const rmo_uniforms* u;
    uint32_t W, H, row0, rows;
    float* out;
    uint32_t* next_row;
    pthread_mutex_t* mu;
    rmo_counters cnt;
} mt_job;

static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    for (;;) {
        pthread_mutex_lock(j->mu);
        uint32_t r = (*j->next_row)++;
        pthread_mutex_unlock(j->mu);
        if (r >= j->rows) break;
        for (uint32_t x = 0; x < j->W; x++)
            fs_main(j->sc, j->u, x, j->row0 + r, j->W, j->H, j->out + ((size_t)r * j->W + x) * 4, &j->cnt);
    }
    return NULL;
}

/* materials: n_materials x (r, g, b) floats, or n_materials = 0 for the default table {(0.4, 0.7, 0.1)} */
RMO_API int rmo_render_mt_materials(const rmo_uniforms* u, const rmo_limits* lim, uint32_t cmd_count,
                                    const uint32_t* words, uint32_t n_words, uint32_t W, uint32_t H,
                                    uint32_t row0, uint32_t rows, float* out, rmo_counters* counters,
                                    uint32_t n_threads, const float* materials, uint32_t n_materials) {
    if (!u || !lim || !out) return RMO_ERR_NULL;
    int rc = rmo_validate_program(cmd_count, words, n_words, 0, NULL);
    if (rc != RMO_OK) return rc;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    rmo_scene sc = { cmd_count, words, n_words, *lim, NULL, 1u, 0 };
    rc = scene_set_materials(&sc, materials, n_materials);
    if (rc != RMO_OK) return rc;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    uint32_t next_row = 0;
    mt_job* jobs = (mt_job*)calloc(n_threads, sizeof(mt_job));
    pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
    for (uint32_t t = 0; t < n_threads; t++) {
        mt_job jb = { &sc, u, W, H, row0, rows, out, &next_row, &mu, { 0, 0, 0, 0, 0, 0 } };
        jobs[t] = jb;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    rmo_counters tot;
    memset(&tot, 0, sizeof tot);
    for (uint32_t t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        tot.march_steps += jobs[t].cnt.march_steps;
        tot.normal_taps += jobs[t].cnt.normal_taps;
        tot.rays += jobs[t].cnt.rays;
        tot.hits += jobs[t].cnt.hits;
        tot.floor_hits += jobs[t].cnt.floor_hits;
        tot.sky += jobs[t].cnt.sky;
    }
    free(jobs);
    free(th);
    if (counters) *counters = tot;
    return RMO_OK;
}
RMO_API int rmo_render_mt(const rmo_uniforms* u, const rmo_limits* lim, uint32_t cmd_count,
                          const uint32_t* words, uint32_t n_words, uint32_t W, uint32_t H,
                          uint32_t row0, uint32_t rows, float* out, rmo_counters* counters,
                          uint32_t n_threads) {
    return rmo_render_mt_materials(u, lim, cmd_count, words, n_words, W, H, row0, rows, out, counters, n_threads, NULL, 0);
}

/* ---- wire format: CSGCommandBufferBuilder (csg/builder.rs:26-62) --------------------- */
typedef struct {
    uint32_t cmd_count;
    uint32_t n_words;
    uint32_t cap;
    uint32_t* buffer;
} rmo_builder;

RMO_API rmo_builder* rmo_builder_new(void) { /* builder.rs:32-37 */
    rmo_builder* b = (rmo_builder*)calloc(1, sizeof *b);
    b->cap = 64;
    b->buffer = (uint32_t*)malloc(b->cap * 4);
    return b;
}
RMO_API void rmo_builder_free(rmo_builder* b) { if (b) { free(b->buffer); free(b); } }
static void bpush(rmo_builder* b, uint32_t w) {
    if (b->n_words == b->cap) { b->cap *= 2; b->buffer = (uint32_t*)realloc(b->buffer, b->cap * 4); }
    b->buffer[b->n_words++] = w;
}
RMO_API void rmo_builder_push_command(rmo_builder* b, uint32_t cmd_type) { /* builder.rs:41-45 */
    b->cmd_count += 1;
    bpush(b, cmd_type);
}
RMO_API void rmo_builder_push_param_vec3(rmo_builder* b, const float* v) { /* builder.rs:49-54 */
    for (int i = 0; i < 3; i++) bpush(b, f2u(v[i]));
}
RMO_API void rmo_builder_push_param_float(rmo_builder* b, float v) { bpush(b, f2u(v)); } /* :58-61 */
RMO_API uint32_t rmo_builder_cmd_count(const rmo_builder* b) { return b->cmd_count; }
RMO_API uint32_t rmo_builder_n_words(const rmo_builder* b) { return b->n_words; }
RMO_API const uint32_t* rmo_builder_words(const rmo_builder* b) { return b->buffer; }

/* BuildCommands impls: sphere.rs:15-21, box.rs:14-20, operations/mod.rs:12-18.
 * A tree is given in a flat node table: kind (0 sphere, 1 box, 100 union, 101 subtraction),
 * params[6] (center xyz, radius or half-extents), lhs/rhs child indices. Post-order. */
typedef struct {
    uint32_t kind;
    float p[6];
    int32_t lhs, rhs;
} rmo_node;

RMO_API void rmo_build_commands(const rmo_node* nodes, int32_t root, rmo_builder* b) {
    const rmo_node* n = &nodes[root];
    switch (n->kind) {
    case RMO_CMD_SPHERE:
        rmo_builder_push_command(b, RMO_CMD_SPHERE);
        rmo_builder_push_param_vec3(b, n->p);
        rmo_builder_push_param_float(b, n->p[3]);
        break;
    case RMO_CMD_BOX:
        rmo_builder_push_command(b, RMO_CMD_BOX);
        rmo_builder_push_param_vec3(b, n->p);
        rmo_builder_push_param_vec3(b, n->p + 3);
        break;
    case RMO_CMD_PLANE: /* extension */
        rmo_builder_push_command(b, RMO_CMD_PLANE);
        rmo_builder_push_param_vec3(b, n->p);
        rmo_builder_push_param_float(b, n->p[3]);
        break;
    case RMO_CMD_CYLINDER: /* extension */
        rmo_builder_push_command(b, RMO_CMD_CYLINDER);
        rmo_builder_push_param_vec3(b, n->p);
        rmo_builder_push_param_float(b, n->p[3]);
        rmo_builder_push_param_float(b, n->p[4]);
        break;
    case RMO_CMD_TRANSLATION_PUSH: /* extension: push(params), the child (lhs), pop */
    case RMO_CMD_ROTATION_PUSH:
    case RMO_CMD_SCALE_PUSH:
        rmo_builder_push_command(b, n->kind);
        for (uint32_t k = 0; k < (n->kind == RMO_CMD_TRANSLATION_PUSH ? 3u : n->kind == RMO_CMD_ROTATION_PUSH ? 4u : 1u); k++)
            rmo_builder_push_param_float(b, n->p[k]);
        rmo_build_commands(nodes, n->lhs, b);
        rmo_builder_push_command(b, n->kind + 1u);
        break;
    case RMO_CMD_MATERIAL: /* extension: the child (lhs), then the tag; p[0] = index as a float-valued integer */
        rmo_build_commands(nodes, n->lhs, b);
        rmo_builder_push_command(b, RMO_CMD_MATERIAL);
        bpush(b, (uint32_t)n->p[0]);
        break;
    case RMO_CMD_SMOOTH_UNION: /* extension: lhs, rhs, operator, k */
        rmo_build_commands(nodes, n->lhs, b);
        rmo_build_commands(nodes, n->rhs, b);
        rmo_builder_push_command(b, RMO_CMD_SMOOTH_UNION);
        rmo_builder_push_param_float(b, n->p[0]);
        break;
    default: /* operations/mod.rs:13-17: lhs, rhs, then the operator */
        rmo_build_commands(nodes, n->lhs, b);
        rmo_build_commands(nodes, n->rhs, b);
        rmo_builder_push_command(b, n->kind);
        break;
    }
}

/* ---- host matrix prep (renderer.rs:205-211) and camera (camera.rs) ------------------
 * nalgebra 0.32.4 is not vendored in the reference tree; the formulas below restate its
 * published algorithms from memory ([dep, from memory] in SURVEY 8(a) a17/a18) and are
 * therefore "parity unpinned" against nalgebra at the ulp level.  The kernel boundary
 * sits AFTER these (it takes finished matrices), so they do not affect kernel parity. */

/* Perspective3::new(aspect, fovy, znear, zfar).inverse() -> column-major 4x4. */
RMO_API void rmo_perspective_inverse(float aspect, float fovy, float znear, float zfar, float* out16) {
    float m11 = 1.0f / tanf(fovy / 2.0f); /* Perspective3::set_fovy: new_m22 = 1/tan(fovy/2) */
    float m00 = m11 / aspect;              /* set_aspect */
    float m22 = (zfar + znear) / (znear - zfar); /* set_znear_and_zfar */
    float m23 = zfar * znear * 2.0f / (znear - zfar);
    const float m32 = -1.0f;
    /* Perspective3::inverse(): res[(r,c)] with column-major storage index c*4+r */
    memset(out16, 0, 64);
    out16[0] = 1.0f / m00;             /* (0,0) */
    out16[5] = 1.0f / m11;             /* (1,1) */
    out16[10] = 0.0f;                  /* (2,2) */
    out16[14] = 1.0f / m32;            /* (2,3) */
    out16[11] = 1.0f / m23;            /* (3,2) */
    out16[15] = -m22 / (m23 * m32);    /* (3,3) */
}

typedef struct { float w, i, j, k; } quat;

/* UnitQuaternion::from_euler_angles(roll, pitch, yaw) */
static quat quat_from_euler(float roll, float pitch, float yaw) {
    float sr = sinf(roll * 0.5f), cr = cosf(roll * 0.5f);
    float sp = sinf(pitch * 0.5f), cp = cosf(pitch * 0.5f);
    float sy = sinf(yaw * 0.5f), cy = cosf(yaw * 0.5f);
    quat q;
    q.w = cr * cp * cy + sr * sp * sy;
    q.i = sr * cp * cy - cr * sp * sy;
    q.j = cr * sp * cy + sr * cp * sy;
    q.k = cr * cp * sy - sr * sp * cy;
    return q;
}
static v3 cross3(v3 a, v3 b) {
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
/* UnitQuaternion * Vector3: t = cross(q.ijk, v)*2; t*w + cross(q.ijk, t) + v */
static v3 quat_rotate(quat q, v3 v) {
    v3 qv = { q.i, q.j, q.k };
    v3 t = cross3(qv, v);
    t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
    v3 c = cross3(qv, t);
    v3 r = { (t.x * q.w + c.x) + v.x, (t.y * q.w + c.y) + v.y, (t.z * q.w + c.z) + v.z };
    return r;
}
/* UnitQuaternion::to_rotation_matrix -> row-major 3x3 */
static void quat_to_mat3(quat q, float* m) {
    float ww = q.w * q.w, ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k;
    float ij = q.i * q.j * 2.0f, wk = q.w * q.k * 2.0f, wj = q.w * q.j * 2.0f;
    float ik = q.i * q.k * 2.0f, jk = q.j * q.k * 2.0f, wi = q.w * q.i * 2.0f;
    m[0] = ww + ii - jj - kk; m[1] = ij - wk;           m[2] = wj + ik;
    m[3] = wk + ij;           m[4] = ww - ii + jj - kk; m[5] = jk - wi;
    m[6] = ik - wj;           m[7] = wi + jk;           m[8] = ww - ii - jj + kk;
}

/* General 4x4 inverse (cofactor expansion, the classic MESA gluInvertMatrix layout that
 * nalgebra's do_inverse4 follows); m, out column-major. Returns 0 if singular. */
static int mat4_inverse(const float* m, float* out) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) return 0;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * inv_det;
    return 1;
}

/* OrbitCameraController (camera.rs:21-85) */
typedef struct {
    float target[3];
    float pitch, yaw, radius;
    float pan_speed, yaw_speed, pitch_speed, dolly_speed;
} rmo_orbit;

RMO_API void rmo_orbit_new(rmo_orbit* c, const float* target3, float radius) { /* camera.rs:38-50 */
    c->target[0] = target3[0]; c->target[1] = target3[1]; c->target[2] = target3[2];
    c->pitch = 0.0f; c->yaw = 0.0f; c->radius = radius;
    c->pan_speed = 0.01f; c->yaw_speed = 0.01f; c->pitch_speed = 0.01f; c->dolly_speed = 0.01f;
}
static quat orbit_rotation(const rmo_orbit* c) { /* camera.rs:52-54 */
    return quat_from_euler(-c->pitch, -c->yaw, 0.0f);
}
/* event: 0 Pan(dx,dy), 1 Orbit(dx,dy), 2 Dolly(dx) -- camera.rs:62-84 */
RMO_API void rmo_orbit_update(rmo_orbit* c, int event, float dx, float dy) {
    if (event == 0) {
        quat q = orbit_rotation(c);
        v3 ex = { 1, 0, 0 }, ey = { 0, 1, 0 };
        v3 right = quat_rotate(q, ex), up = quat_rotate(q, ey);
        float ndx = -dx;
        c->target[0] += (right.x * ndx + up.x * dy) * c->pan_speed;
        c->target[1] += (right.y * ndx + up.y * dy) * c->pan_speed;
        c->target[2] += (right.z * ndx + up.z * dy) * c->pan_speed;
    } else if (event == 1) {
        c->yaw += dx * c->yaw_speed;
        c->pitch += dy * c->pitch_speed;
        if (c->pitch < -1.5f) c->pitch = -1.5f;
        if (c->pitch > 1.5f) c->pitch = 1.5f;
    } else if (event == 2) {
        c->radius += dx * c->dolly_speed * c->radius;
        c->radius = fmaxf(c->radius, 0.1f);
    }
}
/* camera() (camera.rs:56-60): position = target + rotation * z * radius; quaternion out as (w,i,j,k). */
RMO_API void rmo_orbit_camera(const rmo_orbit* c, float* position3, float* quat4) {
    quat q = orbit_rotation(c);
    v3 ez = { 0, 0, 1 };
    v3 rz = quat_rotate(q, ez);
    position3[0] = c->target[0] + rz.x * c->radius;
    position3[1] = c->target[1] + rz.y * c->radius;
    position3[2] = c->target[2] + rz.z * c->radius;
    quat4[0] = q.w; quat4[1] = q.i; quat4[2] = q.j; quat4[3] = q.k;
}
/* Camera::view() (camera.rs:10-12) then .inverse().to_homogeneous() (renderer.rs:211).
 * view = R^-1 * T(-position) as a homogeneous 4x4; inv_view = generic 4x4 inverse. */
RMO_API int rmo_camera_inv_view(const float* position3, const float* quat4, float* out16) {
    quat q = { quat4[0], quat4[1], quat4[2], quat4[3] };
    quat qi = { q.w, -q.i, -q.j, -q.k }; /* unit quaternion inverse = conjugate */
    float r[9];
    quat_to_mat3(qi, r);
    v3 np = { -position3[0], -position3[1], -position3[2] };
    v3 t = quat_rotate(qi, np); /* Isometry: rotation * translation -> translation part = R^-1 * (-p) */
    float view[16] = {
        r[0], r[3], r[6], 0.0f,
        r[1], r[4], r[7], 0.0f,
        r[2], r[5], r[8], 0.0f,
        t.x, t.y, t.z, 1.0f,
    };
    return mat4_inverse(view, out16);
}

/* prepare() (renderer.rs:205-222): fill the 144-byte uniform blob. */
RMO_API int rmo_prepare_uniforms(float vw, float vh, const float* position3, const float* quat4,
                                 rmo_uniforms* u) {
    memset(u, 0, sizeof *u);
    u->viewport_extent[0] = vw;
    u->viewport_extent[1] = vh;
    rmo_perspective_inverse(vw / vh, 0.78539816339744830962f /* FRAC_PI_4 */, 1.0f, 10000.0f, u->inv_proj);
    return rmo_camera_inv_view(position3, quat4, u->inv_view) ? RMO_OK : RMO_ERR_NULL;
}

/* Output stage (extension; the reference's colour target is the 8-bit egui surface, renderer.rs:113).
 * UNORM8 quantisation of RGBA32F pixels as a colour target performs it: clamp to [0,1] (NaN -> 0), times 255, round
 * to nearest even.  out: 4 bytes per pixel, r g b a (bgra = 0) or b g r a (bgra = 1); alpha = 255 for the 1.0 of
 * wgsl:75.  Which rounding a real wgpu backend applies is not pinned by anything in the reference: parity unpinned. */
RMO_API void rmo_quantize_unorm8(const float* rgba, uint64_t n_pixels, int bgra, uint8_t* out) {
    for (uint64_t i = 0; i < n_pixels; i++) {
        uint8_t q[4];
        for (int k = 0; k < 4; k++) {
            float x = rmo_min(rmo_max(rgba[4 * i + k], 0.0f), 1.0f) * 255.0f;
            q[k] = (uint8_t)rmo_f2i(rintf(x));
        }
        out[4 * i + 0] = bgra ? q[2] : q[0];
        out[4 * i + 1] = q[1];
        out[4 * i + 2] = bgra ? q[0] : q[2];
        out[4 * i + 3] = q[3];
    }
}

RMO_API uint32_t rmo_sizeof_uniforms(void) { return (uint32_t)sizeof(rmo_uniforms); }
RMO_API uint32_t rmo_sizeof_limits(void) { return (uint32_t)sizeof(rmo_limits); }
