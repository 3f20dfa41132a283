// CPU model of the interpreter's masked tree loop (rm_kernel_v5.h tree_keep + rm_interp.h map_scene_tree_masked), run over the
// decoder's real tables (RmDecoded::tree) with the per-record decisions the kernel uses (rm_device.h rm_tree_*).
// For random trees of spheres and boxes under Union / Subtraction, random unit masks and random leaf values it checks the
// identity the loop rests on:
//     the records the mask leaves, executed as a postfix program
//  == the WHOLE program with every leaf outside the (forced) mask replaced by +inf,
// bit for bit, for any leaf values -- and that the value stack never holds more than the program's own depth + 1.
// (That replacing far leaves by +inf does not change a frame is the lattice rule's business: rm_kernel_v5.h "Pruning", GPU parity tests.)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct float4 { float x, y, z, w; };
#include "rm_abi.h"
#include "rm_decode.h"

static std::mt19937_64 rng(777);
static uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)((rng() >> 40) & 0xFFFFFF) / 16777216.0f; }

static void gen(std::vector<uint32_t>& w, uint32_t& cmds, uint32_t leaves, uint32_t sub_pct) {
    if (leaves == 1u) {
        if (rng() & 1u) { w.push_back(RM_CMD_SPHERE); for (int i = 0; i < 4; i++) w.push_back(fbits(frand(-2, 2))); }
        else { w.push_back(RM_CMD_BOX); for (int i = 0; i < 6; i++) w.push_back(fbits(frand(-2, 2))); }
        cmds++;
        return;
    }
    const uint32_t shape = (uint32_t)(rng() % 3u);  // 0 balanced, 1 left-deep, 2 anywhere
    const uint32_t left = shape == 0u ? leaves / 2u : shape == 1u ? leaves - 1u : 1u + (uint32_t)(rng() % (leaves - 1u));
    gen(w, cmds, left, sub_pct);
    gen(w, cmds, leaves - left, sub_pct);
    w.push_back(rng() % 100u < sub_pct ? (uint32_t)RM_CMD_SUBTRACTION : (uint32_t)RM_CMD_UNION);
    cmds++;
}

static float vmin(float a, float b) { return b < a ? b : a; }          // selects, like the kernels' operators: any consistent choice
static float vmaxneg(float a, float b) { return -b > a ? -b : a; }     // serves the identity that is checked here

// the whole program, leaf u having the value v[u]
static float run_full(const RmDecoded& d, const std::vector<float>& v) {
    std::vector<float> st;
    float acc = INFINITY;
    for (const RmRecord& r : d.rec) {
        const uint32_t cls = RM_OP_FASTCLASS(r.op), un = RM_OP_UNIT(r.op);
        if (cls <= 4u) acc = cls >= 3u ? vmaxneg(acc, v[un - 1u]) : vmin(acc, v[un - 1u]);
        else if (cls <= 6u) { if (r.op & RM_OP_SPILL) st.push_back(acc); acc = v[un - 1u]; }
        else { const float a = st.back(); st.pop_back(); acc = cls == 7u ? vmin(a, acc) : vmaxneg(a, acc); }
    }
    return acc;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3000;
    long masks = 0, forced = 0, dropped = 0, with_table = 0;
    for (int it = 0; it < rounds; it++) {
        std::vector<uint32_t> w;
        uint32_t cmds = 0;
        const uint32_t leaves = 2u + (uint32_t)(rng() % 63u);
        gen(w, cmds, leaves, (uint32_t)(rng() % 4u) * 20u);
        RmDecoded d;
        if (rm_decode_program(cmds, w.data(), (uint32_t)w.size(), &d) != RM_OK) { std::printf("generated program rejected\n"); return 1; }
        if (!d.is_tree) { std::printf("not a tree program\n"); return 1; }
        if (d.is_chain || d.rec.size() > 128u) {
            if (!d.tree.empty()) { std::printf("a table where none is used\n"); return 1; }
            continue;
        }
        if (d.unit_mode != RM_UNITS_LATTICE || d.tree.size() != d.rec.size() || d.units.size() != leaves) { std::printf("round %d: no table\n", it); return 1; }
        with_table++;
        const size_t n = d.rec.size();
        std::vector<unsigned long long> Lm(n), Rm(n);
        std::vector<uint32_t> info(n);
        for (size_t r = 0; r < n; r++) {
            uint32_t q[5];
            std::memcpy(q, &d.tree[r].p[0], sizeof q);
            Lm[r] = q[0] | ((unsigned long long)q[1] << 32);
            Rm[r] = q[2] | ((unsigned long long)q[3] << 32);
            info[r] = q[4];
        }
        const unsigned long long valid = leaves >= 64u ? ~0ull : ((1ull << leaves) - 1ull);
        for (int m = 0; m < 40; m++) {
            unsigned long long need = rng() & valid;
            const int thin = (int)(rng() % 4u);  // masks of every density, most of them sparse like the real ones
            for (int k = 0; k < thin; k++) need &= rng();
            if (m == 0) need = valid;
            if (m == 1) need = 0ull;
            std::vector<float> v(leaves);
            for (float& x : v) x = frand(-3, 3);
            // tree_keep
            unsigned long long need2 = need;
            for (size_t r = 0; r < n; r++)
                if (rm_tree_forces(Lm[r], Rm[r], info[r], need)) { need2 |= 1ull << rm_tree_forced_unit(info[r]); forced++; }
            // map_scene_tree_masked
            std::vector<float> st;
            float acc = INFINITY;
            size_t deepest = 0;
            for (size_t r = 0; r < n; r++) {
                if (!rm_tree_keeps(Lm[r], Rm[r], info[r], need2)) { dropped++; continue; }
                const uint32_t cls = RM_OP_FASTCLASS(d.rec[r].op), un = RM_OP_UNIT(d.rec[r].op);
                if (cls <= 6u) {
                    if (cls >= 5u || rm_tree_pushes(Lm[r], info[r], need2)) { st.push_back(acc); acc = v[un - 1u]; }
                    else acc = cls >= 3u ? vmaxneg(acc, v[un - 1u]) : vmin(acc, v[un - 1u]);
                } else {
                    if (st.empty()) { std::printf("round %d: pop from an empty stack\n", it); return 1; }
                    const float a = st.back();
                    st.pop_back();
                    acc = cls == 7u ? vmin(a, acc) : vmaxneg(a, acc);
                }
                deepest = st.size() > deepest ? st.size() : deepest;
            }
            if (deepest > (size_t)d.spill_depth + 1u) { std::printf("round %d: %zu values spilled, the program itself spills %u\n", it, deepest, d.spill_depth); return 1; }
            if (st.size() != (need2 != 0ull ? 1u : 0u)) { std::printf("round %d: %zu values left on the stack\n", it, st.size()); return 1; }
            std::vector<float> v2(v);
            for (uint32_t u = 0; u < leaves; u++)
                if (!((need2 >> u) & 1ull)) v2[u] = INFINITY;
            const float want = run_full(d, v2);
            if (fbits(want) != fbits(acc)) { std::printf("round %d mask %016llx: %g, the whole program gives %g\n", it, need, acc, want); return 1; }
            masks++;
        }
    }
    std::printf("tree keep model ok: %ld programs with a table, %ld masks, %ld forced leaves, %ld records dropped\n", with_table, masks, forced, dropped);
    return with_table > 100 && forced > 0 && dropped > 0 ? 0 : 1;
}
