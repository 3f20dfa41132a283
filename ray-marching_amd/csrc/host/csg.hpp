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
    // Plane (reserved by comment in the reference)
    Union = 100,
    Subtraction = 101,
    // Intersection; 200.. TranslationPush/Pop, RotationPush/Pop, ScalePush/Pop (reserved)
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

struct Union {  // operations/mod.rs:55 via impl_binary_operation!
    NodeBox lhs, rhs;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};
struct Subtraction {  // operations/mod.rs:56
    NodeBox lhs, rhs;
    void build_commands(CSGCommandBufferBuilder& builder) const;
};

class CSGNode {  // csg/mod.rs:28-45 (enum_dispatch over BuildCommands)
  public:
    using Variant = std::variant<Sphere, Box, Union, Subtraction>;
    CSGNode(Sphere s) : v_(std::move(s)) {}
    CSGNode(Box b) : v_(std::move(b)) {}
    CSGNode(Union u) : v_(std::move(u)) {}
    CSGNode(Subtraction s) : v_(std::move(s)) {}
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

inline CSGNode make_union(CSGNode a, CSGNode b) { return CSGNode(Union{NodeBox(std::move(a)), NodeBox(std::move(b))}); }
inline CSGNode make_subtraction(CSGNode a, CSGNode b) {
    return CSGNode(Subtraction{NodeBox(std::move(a)), NodeBox(std::move(b))});
}

}  // namespace ray_marching::csg
