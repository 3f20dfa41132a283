// scenes.hpp -- the deterministic synthetic scenes bench.py and the tests render:
// G1 / G8 / G32 / G64 plus a balanced-tree variant of G32.  Built only from the reference's
// four node types.  Coordinates are computed in double and rounded once to f32.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

#include "csg.hpp"

namespace ray_marching::scenes {

using csg::CSGNode;

struct Lcg {  // x <- 1664525 x + 1013904223 (mod 2^32); u = (x >> 8) / 2^24
    uint32_t x;
    double u() {
        x = 1664525u * x + 1013904223u;
        return (double)(x >> 8) / 16777216.0;
    }
};

inline std::vector<CSGNode> grid_prims(int nx, int nz, uint32_t seed) {
    Lcg rng{seed};
    std::vector<CSGNode> prims;
    for (int k = 0; k < nx * nz; k++) {
        const int ix = k % nx, iz = k / nx;
        const float x = (float)((ix - (nx - 1) / 2.0) * 1.1);
        const float z = (float)((iz - (nz - 1) / 2.0) * 1.1);
        const double u1 = rng.u(), u2 = rng.u();
        const float y = (float)(0.3 * u1);
        if ((ix + iz) % 2 == 0) {
            prims.emplace_back(csg::Sphere{{x, y, z}, (float)(0.35 + 0.15 * u2)});
        } else {
            const float h = (float)(0.3 + 0.15 * u2);
            prims.emplace_back(csg::Box{{x, y, z}, {h, h, h}});
        }
    }
    return prims;
}

// Left-deep fold: operator k (0-based) is Subtraction when k % 4 == 3, else Union.
inline CSGNode fold_left(std::vector<CSGNode> prims) {
    CSGNode acc = prims[0];
    for (size_t k = 0; k + 1 < prims.size(); k++)
        acc = (k % 4 == 3) ? csg::make_subtraction(std::move(acc), prims[k + 1])
                           : csg::make_union(std::move(acc), prims[k + 1]);
    return acc;
}

inline CSGNode fold_balanced(std::vector<CSGNode> level) {
    size_t k = 0;
    while (level.size() > 1) {
        std::vector<CSGNode> next;
        for (size_t i = 0; i < level.size(); i += 2, k++)
            next.push_back((k % 4 == 3) ? csg::make_subtraction(level[i], level[i + 1])
                                        : csg::make_union(level[i], level[i + 1]));
        level = std::move(next);
    }
    return level[0];
}

inline CSGNode g1() { return CSGNode(csg::Sphere{{0, 0, 0}, 1.0f}); }

inline CSGNode g8() {  // Union(Subtraction(Union(S0,B1),S2),B3)
    CSGNode s0(csg::Sphere{{0, 0, 0}, 1.0f});
    CSGNode b1(csg::Box{{0, 0, 0}, {0.8f, 0.8f, 0.8f}});
    CSGNode s2(csg::Sphere{{0.9f, 0.5f, 0.6f}, 0.6f});
    CSGNode b3(csg::Box{{0, -1.2f, 0}, {1.5f, 0.1f, 1.5f}});
    return csg::make_union(csg::make_subtraction(csg::make_union(s0, b1), s2), b3);
}

inline CSGNode g32() { return fold_left(grid_prims(4, 4, 0x5DF00020u)); }
inline CSGNode g64() { return fold_left(grid_prims(8, 4, 0x5DF00040u)); }
inline CSGNode g32_balanced() { return fold_balanced(grid_prims(4, 4, 0x5DF00020u)); }

// ---- scenes using extension node types (BASELINE.json configs 2-3 as literally worded) ----
inline CSGNode g8x() {  // sphere U box - cylinder (+ floor slab)
    CSGNode s0(csg::Sphere{{0, 0, 0}, 1.0f});
    CSGNode b1(csg::Box{{0, 0, 0}, {0.8f, 0.8f, 0.8f}});
    CSGNode c2(csg::Cylinder{{0.9f, 0.5f, 0.6f}, 0.45f, 0.9f});
    CSGNode b3(csg::Box{{0, -1.2f, 0}, {1.5f, 0.1f, 1.5f}});
    return csg::make_union(csg::make_subtraction(csg::make_union(s0, b1), c2), b3);
}
inline CSGNode g32s() {  // G32 with Union -> SmoothUnion(k = 0.25)
    std::vector<CSGNode> prims = grid_prims(4, 4, 0x5DF00020u);
    CSGNode acc = prims[0];
    for (size_t k = 0; k + 1 < prims.size(); k++)
        acc = (k % 4 == 3) ? csg::make_subtraction(std::move(acc), prims[k + 1])
                           : csg::make_smooth_union(std::move(acc), prims[k + 1], 0.25f);
    return acc;
}
inline CSGNode ext_mix() {  // every extension node type in one tree
    CSGNode a = csg::make_smooth_union(CSGNode(csg::Sphere{{-0.6f, 0, 0}, 0.7f}),
                                       CSGNode(csg::Cylinder{{0.5f, 0.0f, 0.1f}, 0.4f, 0.8f}), 0.3f);
    CSGNode b = csg::make_intersection(CSGNode(csg::Box{{0, 0, 0}, {1.4f, 0.9f, 1.0f}}),
                                       CSGNode(csg::Plane{{0.0f, 1.0f, 0.2f}, 0.35f}));
    CSGNode c = csg::make_smooth_union(std::move(a),
                                       csg::make_subtraction(std::move(b), CSGNode(csg::Sphere{{0.2f, 0.3f, 0.9f}, 0.5f})), 0.15f);
    return csg::make_union(std::move(c), CSGNode(csg::Cylinder{{-1.4f, -0.6f, -0.5f}, 0.25f, 0.5f}));
}

inline CSGNode xform_mix() {  // space transformations, nested, around primitives and around a sub-tree
    const float h = 0.70710678f;  // cos(45 deg) = sin(45 deg): quarter turns about z and x
    CSGNode a = csg::make_translation(csg::make_rotation(CSGNode(csg::Box{{0, 0, 0}, {0.9f, 0.35f, 0.5f}}), {h, 0, 0, h}), {-1.1f, 0.2f, 0.0f});
    CSGNode b = csg::make_scale(csg::make_union(CSGNode(csg::Sphere{{0, 0, 0}, 1.0f}), CSGNode(csg::Box{{0.9f, 0, 0}, {0.5f, 0.3f, 0.3f}})), 0.6f);
    CSGNode c = csg::make_translation(
        csg::make_rotation(csg::make_scale(csg::make_subtraction(CSGNode(csg::Box{{0, 0, 0}, {1, 1, 1}}), CSGNode(csg::Sphere{{0.4f, 0.4f, 0.4f}, 0.9f})), 0.45f),
                           {0.9238795f, 0.2209424f, 0.2209424f, 0.2209424f}),
        {1.2f, -0.3f, 0.4f});
    CSGNode d = csg::make_rotation(CSGNode(csg::Cylinder{{0.0f, -0.9f, -0.9f}, 0.3f, 0.7f}), {h, h, 0, 0});
    return csg::make_union(csg::make_union(csg::make_union(std::move(a), std::move(b)), std::move(c)), std::move(d));
}

inline CSGNode mat_mix() {  // material tags on leaves, on sub-trees, inside transform scopes, under every operator
    using csg::make_material;
    const float h = 0.70710678f;
    CSGNode body = csg::make_union(make_material(CSGNode(csg::Sphere{{0, 0, 0}, 1.0f}), 1), make_material(CSGNode(csg::Box{{0, 0, 0}, {0.8f, 0.8f, 0.8f}}), 2));
    CSGNode carved = csg::make_subtraction(std::move(body), make_material(CSGNode(csg::Sphere{{0.9f, 0.5f, 0.6f}, 0.6f}), 3));
    CSGNode slab(csg::Box{{0, -1.2f, 0}, {1.5f, 0.1f, 1.5f}});
    CSGNode arm = csg::make_translation(csg::make_rotation(make_material(CSGNode(csg::Cylinder{{0, 0, 0}, 0.25f, 0.9f}), 5), {h, 0, 0, h}), {-1.3f, 0.4f, 0.3f});
    CSGNode blob = make_material(csg::make_smooth_union(make_material(CSGNode(csg::Sphere{{1.3f, 0.2f, -0.6f}, 0.45f}), 1),
                                                        CSGNode(csg::Sphere{{1.7f, 0.5f, -0.3f}, 0.35f}), 0.3f), 4);
    CSGNode cut = csg::make_intersection(make_material(CSGNode(csg::Box{{-0.2f, 1.3f, -0.9f}, {0.5f, 0.5f, 0.5f}}), 2),
                                         make_material(CSGNode(csg::Sphere{{-0.2f, 1.3f, -0.9f}, 0.62f}), 3));
    CSGNode twin = csg::make_scale(csg::make_smooth_union(make_material(CSGNode(csg::Sphere{{-2.0f, -0.6f, 1.6f}, 0.5f}), 3),
                                                          make_material(CSGNode(csg::Box{{-1.2f, -0.6f, 1.6f}, {0.4f, 0.4f, 0.4f}}), 5), 0.4f), 0.8f);
    return csg::make_union(csg::make_union(csg::make_union(csg::make_union(csg::make_union(std::move(carved), std::move(slab)), std::move(arm)),
                                                           std::move(blob)), std::move(cut)), std::move(twin));
}

inline std::optional<CSGNode> by_name(const std::string& name) {
    if (name == "mat_mix") return mat_mix();
    if (name == "xform_mix") return xform_mix();
    if (name == "g8x") return g8x();
    if (name == "g32s") return g32s();
    if (name == "ext_mix") return ext_mix();
    if (name == "g1") return g1();
    if (name == "g8") return g8();
    if (name == "g32") return g32();
    if (name == "g64") return g64();
    if (name == "g32_balanced") return g32_balanced();
    return std::nullopt;
}

}  // namespace ray_marching::scenes
