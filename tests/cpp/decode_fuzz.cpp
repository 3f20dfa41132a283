// Decoder robustness (CPU, built with ASan + UBSan by tests/test_decoder_fuzz_cpu.py): rm_decode_program is the one place
// where bytes from the host application are interpreted (rm_write_buffer / rm_set_program hand it the raw command words).
// Random streams -- well-formed programs from a small generator, the same with words flipped, and pure noise -- must be
// either rejected with a status or decoded into records that satisfy the invariants the kernels rely on.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct float4 { float x, y, z, w; };  // rm_device.h names the HIP vector type in RmLaunch; this is a host-only build
#include "rm_abi.h"
#include "rm_decode.h"

static std::mt19937 rng(12345);
static uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)(rng() >> 8) / 16777216.0f; }

static void leaf(std::vector<uint32_t>& w, uint32_t& cmds) {
    const uint32_t k = rng() % 5u;
    if (k == 0u) { w.push_back(RM_CMD_SPHERE); for (int i = 0; i < 4; i++) w.push_back(fbits(frand(-2, 2))); }
    else if (k == 1u) { w.push_back(RM_CMD_BOX); for (int i = 0; i < 6; i++) w.push_back(fbits(frand(-2, 2))); }
    else if (k == 2u) { w.push_back(RM_CMD_CYLINDER); for (int i = 0; i < 5; i++) w.push_back(fbits(frand(-2, 2))); }
    else if (k == 3u) { w.push_back(RM_CMD_PLANE); for (int i = 0; i < 4; i++) w.push_back(fbits(frand(-2, 2))); }
    else { w.push_back(RM_CMD_SPHERE); w.push_back(fbits(frand(-2, 2))); w.push_back(0x7FC00000u); w.push_back(0x7F800000u); w.push_back(fbits(-0.0f)); }
    cmds++;
}
static void tree(std::vector<uint32_t>& w, uint32_t& cmds, int depth) {
    const uint32_t r = rng() % 10u;
    if (depth == 0 || r < 3u) { leaf(w, cmds); return; }
    if (r < 5u) {  // transform scope
        const uint32_t t = rng() % 3u;
        if (t == 0u) { w.push_back(RM_CMD_TRANSLATION_PUSH); for (int i = 0; i < 3; i++) w.push_back(fbits(frand(-1, 1))); }
        else if (t == 1u) { w.push_back(RM_CMD_ROTATION_PUSH); w.push_back(fbits(1.0f)); for (int i = 0; i < 3; i++) w.push_back(fbits(0.0f)); }
        else { w.push_back(RM_CMD_SCALE_PUSH); w.push_back(fbits(frand(0.5f, 2.0f))); }
        cmds++;
        tree(w, cmds, depth - 1);
        w.push_back(t == 0u ? (uint32_t)RM_CMD_TRANSLATION_POP : t == 1u ? (uint32_t)RM_CMD_ROTATION_POP : (uint32_t)RM_CMD_SCALE_POP);
        cmds++;
        return;
    }
    tree(w, cmds, depth - 1);
    tree(w, cmds, depth - 1);
    const uint32_t o = rng() % 4u;
    if (o == 3u) { w.push_back(RM_CMD_SMOOTH_UNION); w.push_back(fbits(frand(-0.2f, 1.0f))); }
    else w.push_back(o == 0u ? (uint32_t)RM_CMD_UNION : o == 1u ? (uint32_t)RM_CMD_SUBTRACTION : (uint32_t)RM_CMD_INTERSECTION);
    cmds++;
    if (rng() % 8u == 0u) { w.push_back(RM_CMD_MATERIAL); w.push_back(rng() % 300u); cmds++; }
}

static int check(const RmDecoded& d) {
    uint32_t cones = 0, slabs = 0;
    for (const RmRecord& r : d.rec) {
        const uint32_t kind = RM_OP_KIND(r.op), mode = RM_OP_MODE(r.op);
        if (kind > RM_KIND_MATERIAL || mode > (kind == RM_KIND_XFORM ? (uint32_t)RM_XF_S_POP : (uint32_t)RM_MODE_SMOOTH)) return 1;
        const bool bounded = kind == RM_KIND_SPHERE || kind == RM_KIND_BOX || kind == RM_KIND_CYLINDER;
        if ((r.op & RM_OP_NOCULL) && !bounded && kind != RM_KIND_PLANE) return 2;
        uint32_t slot;
        std::memcpy(&slot, &r.p[6], 4);
        if (bounded && !(r.op & RM_OP_NOCULL)) {
            if (d.has_xforms || kind == RM_KIND_SPHERE) { if (slot >= d.n_sphere) return 3; cones++; }
            else { if (slot >= d.n_box) return 4; slabs++; }
        }
        if (bounded && d.has_xforms && slot >= d.n_sphere) return 5;
    }
    if (!d.has_xforms && (cones != d.n_sphere || slabs != d.n_box)) return 6;
    if (d.has_xforms && d.bounds.size() != (size_t)d.n_sphere * 4u) return 7;
    // unit records of wave-level culling: at most 64, none without a mode, one per bounded leaf of a lattice program, kinds in range,
    // an outer radius that is never below the inner one
    if (d.units.size() > 64u || (d.unit_mode == RM_UNITS_NONE) != d.units.empty()) return 8;
    if (d.unit_mode == RM_UNITS_LATTICE && (d.units.size() != d.n_leaves + 0u && d.units.size() < d.n_leaves)) return 12;
    for (const RmRecord& g : d.units) {
        uint32_t k;
        std::memcpy(&k, &g.p[6], 4);
        if (k > RM_UNIT_LEAF || (d.unit_mode == RM_UNITS_LATTICE) != (k == RM_UNIT_LEAF)) return 13;
        if (g.p[3] < g.p[4]) return 14;          // (NaN compares false: a leaf with a NaN size has bounds +inf / -inf)
        if (!(g.p[5] >= 0.0f) || g.p[5] > d.unit_kmax) return 15;
    }
    if (!d.tree.empty() && (d.tree.size() != d.rec.size() || d.rec.size() > 128u || !d.is_tree || d.is_chain || d.unit_mode != RM_UNITS_LATTICE)) return 16;
    if (d.is_chain && d.spill_depth != 0u) return 9;
    if (d.bound_walk && (d.has_xforms || d.spill_depth > 1u)) return 10;
    if (d.max_depth > 32u) return 11;
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    long ok = 0, rejected = 0;
    for (int it = 0; it < rounds; it++) {
        std::vector<uint32_t> w;
        uint32_t cmds = 0;
        const uint32_t flavour = rng() % 4u;
        if (flavour == 3u) {  // noise
            const uint32_t n = rng() % 64u;
            for (uint32_t i = 0; i < n; i++) w.push_back(rng() % 3u ? rng() % 320u : rng());
            cmds = rng() % 40u;
        } else {
            tree(w, cmds, (int)(rng() % 6u));
            if (flavour >= 1u && !w.empty())  // damage: flip words, cut the tail, lie about the count
                for (uint32_t k = 0; k < 1u + rng() % 3u; k++) {
                    const uint32_t what = rng() % 3u;
                    if (w.empty()) break;
                    if (what == 0u) w[rng() % w.size()] = rng() % 320u;
                    else if (what == 1u) w.resize(rng() % (w.size() + 1u));
                    else cmds += (rng() % 5u) - 2u;
                }
        }
        RmDecoded d;
        const int rc = rm_decode_program(cmds, w.empty() ? nullptr : w.data(), (uint32_t)w.size(), &d);
        if (rc != RM_OK) { rejected++; continue; }
        ok++;
        if (const int bad = check(d)) {
            std::printf("invariant %d violated (round %d, %u commands, %zu words)\n", bad, it, cmds, w.size());
            return 1;
        }
    }
    std::printf("decoder fuzz ok: %ld decoded, %ld rejected\n", ok, rejected);
    return ok > 0 && rejected > 0 ? 0 : 1;
}
