// csg.hpp -- C++ mirror of the reference's scene model and serializer:
//   CSGCommandType / CSGCommandBufferBuilder   src/ray_marching/csg/builder.rs:1-62
//   BuildCommands, CSGNode                     src/ray_marching/csg/mod.rs:16-45
//   Sphere / Box                               csg/primitives/sphere.rs:8-22, box.rs:8-21
//   Union / Subtraction                        csg/operations/mod.rs:7-56
// Same names, same field meaning, same post-order wire format.  The editor-only halves
// (*Template, CSGNodeTemplateTrait) are out of scope.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <utility>
#include <variant>
#include <vector>

namespace ray_marching::csg {

enum class CSGCommandType : uint32_t {  // builder.rs:3-24
    Sphere = 0,
    Box = 1,
    Union = 100,
    Subtraction = 101,
    // ---- extensions: NOT implemented by the reference (DESIGN.md "Extension node types") ----
    Plane = 2,           // slot the reference reserves by comment (builder.rs:8)
    Cylinder = 10,       // BASELINE.json config 2
    Intersection = 102,  // slot the reference reserves by comment (builder.rs:14)
    SmoothUnion = 110,   // BASELINE.json config 3
    // space transformations: the six slots the reference reserves by comment (builder.rs:16-23, "1 child, transforms space")
    TranslationPush = 200,
    TranslationPop = 201,
    RotationPush = 202,
    RotationPop = 203,
    ScalePush = 204,
    ScalePop = 205,
    // material tag (the reference lists a material system as future work, README.md:11): unary postfix, one u32 index
    Material = 300,
};

struct CSGCommandBufferBuilder {  // builder.rs:26-62
    uint32_t cmd_count = 0;
    std::vector<uint32_t> buffer;

    CSGCommandBufferBuilder& push_command(CSGCommandType cmd_type) {
        cmd_count += 1;
        buffer.push_back(static_cast<uint32_t>(cmd_type));
        return *this;
    }
    CSGCommandBufferBuilder& push_param_vec3(const std::array<float, 3>& value) {
        for (float v : value) buffer.push_back(to_bits(v));
        return *this;
    }
    CSGCommandBufferBuilder& push_param_float(float value) {
        buffer.push_back(to_bits(value));
        return *this;
    }
    CSGCommandBufferBuilder& push_param_u32(uint32_t value) {  // extension: integer parameters (Material index)
        buffer.push_back(value);
        return *this;
    }
    static uint32_t to_bits(float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    }
};

class CSGNode;

// Rust's Box<CSGNode> with #[derive(Clone)]: owning, deep-copying.
class NodeBox {
  public:
    NodeBox() = default;
    explicit NodeBox(const CSGNode& n);
    explicit NodeBox(CSGNode&& n);
    NodeBox(const NodeBox& o);
    NodeBox(NodeBox&&) noexcept = default;
    NodeBox& operator=(const NodeBox& o);
    NodeBox& operator=(NodeBox&&) noexcept = default;
    ~NodeBox();
    const CSGNode& operator*() const { return *p_; }
    const CSGNode* operator->() const { return p_.get(); }

  private:
    std::unique_ptr<CSGNode> p_;
};

struct Sphere {  // sphere.rs:8-13
    std::array<float, 3> center{0, 0, 0};
    float radius = 1.0f;
    void build_commands(CSGCommandBufferBuilder& builder) const {  // sphere.rs:15-21
        builder.push_command(CSGCommandType::Sphere).push_param_vec3(center).push_param_float(radius);
    }
};

struct Box {  // box.rs:8-12 ("radius" = half extents)
    std::array<float, 3> center{0, 0, 0};
    std::array<float, 3> radius{1, 1, 1};
    void build_commands(CSGCommandBufferBuilder& builder) const {  // box.rs:14-20
        builder.push_command(CSGCommandType::Box).push_param_vec3(center).push_param_vec3(radius);
    }
};

// ---- extension primitives (no counterpart in the reference) ----
struct Plane {  // dot(p, normal) + h; the normal is used as given
    std::array<float, 3> normal{0, 1, 0};
    float h = 0.0f;
    void build_commands(CSGCommandBufferBuilder& builder) const {
        builder.push_command(CSGCommandType::Plane).push_param_vec3(normal).push_param_float(h);
    }
};
struct Cylinder {  // capped, along y
    std::array<float, 3> center{0, 0, 0};
    float radius = 1.0f;
    float half_height = 1.0f;
    void build_commands(CSGCommandBufferBuilder& builder) const {
        builder.push_command(CSGCommandType::Cylinder).push_param_vec3(center).push_param_float(radius).push_param_float(half_height);
    }
};

struct Union {  // operations/mod.rs:55 via impl_binary_operation!
    NodeBox lhs, rhs;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};
struct Subtraction {  // operations/mod.rs:56
    NodeBox lhs, rhs;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};

struct Intersection {  // extension: max(a, b)
    NodeBox lhs, rhs;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};
struct SmoothUnion {  // extension: polynomial smooth minimum with blend width k; the operator carries k as one parameter
    NodeBox lhs, rhs;
    float k = 0.25f;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};

// Space transformations (extension; the node types csg/mod.rs:41-44 reserves by comment): Push(params), child, Pop.
struct Translation {  // the child moved by `offset`
    NodeBox child;
    std::array<float, 3> offset{0, 0, 0};
    void build_commands(CSGCommandBufferBuilder& builder) const;
};
struct Rotation {  // the child rotated by the unit quaternion (w, i, j, k)
    NodeBox child;
    std::array<float, 4> quaternion{1, 0, 0, 0};
    void build_commands(CSGCommandBufferBuilder& builder) const;
};
struct Scale {  // the child scaled uniformly by `factor` (> 0)
    NodeBox child;
    float factor = 1.0f;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};

struct Material {  // the child with every surface tagged by material `index` (rm_set_materials)
    NodeBox child;
    uint32_t index = 0;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};

class CSGNode {  // csg/mod.rs:28-45 (enum_dispatch over BuildCommands) + the extension node types
  public:
    using Variant = std::variant<Sphere, Box, Union, Subtraction, Plane, Cylinder, Intersection, SmoothUnion, Translation, Rotation, Scale, Material>;
    CSGNode(Sphere s) : v_(std::move(s)) {}
    CSGNode(Box b) : v_(std::move(b)) {}
    CSGNode(Union u) : v_(std::move(u)) {}
    CSGNode(Subtraction s) : v_(std::move(s)) {}
    CSGNode(Plane p) : v_(std::move(p)) {}
    CSGNode(Cylinder c) : v_(std::move(c)) {}
    CSGNode(Intersection i) : v_(std::move(i)) {}
    CSGNode(SmoothUnion s) : v_(std::move(s)) {}
    CSGNode(Translation t) : v_(std::move(t)) {}
    CSGNode(Rotation r) : v_(std::move(r)) {}
    CSGNode(Scale s) : v_(std::move(s)) {}
    CSGNode(Material m) : v_(std::move(m)) {}
    void build_commands(CSGCommandBufferBuilder& builder) const {
        std::visit([&](const auto& n) { n.build_commands(builder); }, v_);
    }
    const Variant& variant() const { return v_; }

  private:
    Variant v_;
};

inline NodeBox::NodeBox(const CSGNode& n) : p_(std::make_unique<CSGNode>(n)) {}
inline NodeBox::NodeBox(CSGNode&& n) : p_(std::make_unique<CSGNode>(std::move(n))) {}
inline NodeBox::NodeBox(const NodeBox& o) : p_(o.p_ ? std::make_unique<CSGNode>(*o.p_) : nullptr) {}
inline NodeBox& NodeBox::operator=(const NodeBox& o) {
    if (this != &o) p_ = o.p_ ? std::make_unique<CSGNode>(*o.p_) : nullptr;
    return *this;
}
inline NodeBox::~NodeBox() = default;

// operations/mod.rs:13-17: lhs, rhs, then the operator -> valid postfix.
inline void Union::build_commands(CSGCommandBufferBuilder& builder) const {
    lhs->build_commands(builder);
    rhs->build_commands(builder);
    builder.push_command(CSGCommandType::Union);
}
inline void Subtraction::build_commands(CSGCommandBufferBuilder& builder) const {
    lhs->build_commands(builder);
    rhs->build_commands(builder);
    builder.push_command(CSGCommandType::Subtraction);
}

inline void Intersection::build_commands(CSGCommandBufferBuilder& builder) const {
    lhs->build_commands(builder);
    rhs->build_commands(builder);
    builder.push_command(CSGCommandType::Intersection);
}
inline void SmoothUnion::build_commands(CSGCommandBufferBuilder& builder) const {
    lhs->build_commands(builder);
    rhs->build_commands(builder);
    builder.push_command(CSGCommandType::SmoothUnion).push_param_float(k);
}
inline void Translation::build_commands(CSGCommandBufferBuilder& builder) const {
    builder.push_command(CSGCommandType::TranslationPush).push_param_vec3(offset);
    child->build_commands(builder);
    builder.push_command(CSGCommandType::TranslationPop);
}
inline void Rotation::build_commands(CSGCommandBufferBuilder& builder) const {
    builder.push_command(CSGCommandType::RotationPush);
    for (float q : quaternion) builder.push_param_float(q);
    child->build_commands(builder);
    builder.push_command(CSGCommandType::RotationPop);
}
inline void Scale::build_commands(CSGCommandBufferBuilder& builder) const {
    builder.push_command(CSGCommandType::ScalePush).push_param_float(factor);
    child->build_commands(builder);
    builder.push_command(CSGCommandType::ScalePop);
}
inline void Material::build_commands(CSGCommandBufferBuilder& builder) const {
    child->build_commands(builder);
    builder.push_command(CSGCommandType::Material).push_param_u32(index);
}
inline CSGNode make_material(CSGNode child, uint32_t index) { return CSGNode(Material{NodeBox(std::move(child)), index}); }
inline CSGNode make_translation(CSGNode child, std::array<float, 3> offset) { return CSGNode(Translation{NodeBox(std::move(child)), offset}); }
inline CSGNode make_rotation(CSGNode child, std::array<float, 4> q) { return CSGNode(Rotation{NodeBox(std::move(child)), q}); }
inline CSGNode make_scale(CSGNode child, float factor) { return CSGNode(Scale{NodeBox(std::move(child)), factor}); }
inline CSGNode make_intersection(CSGNode a, CSGNode b) {
    return CSGNode(Intersection{NodeBox(std::move(a)), NodeBox(std::move(b))});
}
inline CSGNode make_smooth_union(CSGNode a, CSGNode b, float k) {
    return CSGNode(SmoothUnion{NodeBox(std::move(a)), NodeBox(std::move(b)), k});
}

inline CSGNode make_union(CSGNode a, CSGNode b) { return CSGNode(Union{NodeBox(std::move(a)), NodeBox(std::move(b))}); }
inline CSGNode make_subtraction(CSGNode a, CSGNode b) {
    return CSGNode(Subtraction{NodeBox(std::move(a)), NodeBox(std::move(b))});
}

}  // namespace ray_marching::csg
