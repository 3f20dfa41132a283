// renderer.hpp -- C++ mirror of src/ray_marching/renderer.rs on top of the C ABI:
//   Uniforms / RayMarchLimits + as_shader_bytes        renderer.rs:17-41
//   RayMarchingResources::new                          renderer.rs:51-175   -> rm_create
//   RayMarchingCallback::new / prepare / paint         renderer.rs:184-256  -> rm_write_buffer x3, rm_draw
// Header-only; link against librm_hip.so (include/rm_abi.h).
#pragma once
#include <array>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "camera.hpp"
#include "csg.hpp"
#include "rm_abi.h"

namespace ray_marching {

// renderer.rs:29-34; as_shader_bytes() (renderer.rs:17-27) yields the 144-byte std140-style blob.
struct Uniforms {
    std::array<float, 2> viewport_extent{0, 0};
    Matrix4 inv_proj{};
    Matrix4 inv_view{};
    std::array<uint8_t, 144> as_shader_bytes() const {
        rm_uniforms u;
        std::memset(&u, 0, sizeof u);
        u.viewport_extent[0] = viewport_extent[0];
        u.viewport_extent[1] = viewport_extent[1];
        std::memcpy(u.inv_proj, inv_proj.m.data(), 64);
        std::memcpy(u.inv_view, inv_view.m.data(), 64);
        std::array<uint8_t, 144> b;
        static_assert(sizeof(rm_uniforms) == 144, "Uniforms blob must be 144 bytes");
        std::memcpy(b.data(), &u, 144);
        return b;
    }
};

struct RayMarchLimits {  // renderer.rs:36-41
    float min_dist = 0.01f;
    float max_dist = 100.0f;
    uint32_t max_iter = 100;
    std::array<uint8_t, 12> as_shader_bytes() const {
        std::array<uint8_t, 12> b;
        std::memcpy(b.data(), &min_dist, 4);
        std::memcpy(b.data() + 4, &max_dist, 4);
        std::memcpy(b.data() + 8, &max_iter, 4);
        return b;
    }
};

// The host half of prepare() (renderer.rs:205-227) with no device involved.
struct PreparedFrame {
    Uniforms uniforms;
    csg::CSGCommandBufferBuilder builder;
};
inline PreparedFrame prepare_frame(const std::optional<csg::CSGNode>& csg_node, std::array<float, 2> viewport,
                                   const Camera& camera) {
    PreparedFrame f;
    const Perspective3 projection(viewport[0] / viewport[1], 0.78539816339744830962f /* FRAC_PI_4 */, 1.0f, 10000.0f);
    f.uniforms.viewport_extent = viewport;
    f.uniforms.inv_proj = projection.inverse();
    f.uniforms.inv_view = camera.view().inverse().to_homogeneous();
    if (csg_node) csg_node->build_commands(f.builder);
    return f;
}

class RmException : public std::runtime_error {
  public:
    RmException(int status, const std::string& what) : std::runtime_error(what), status(status) {}
    int status;
};

// Long-lived GPU state (renderer.rs:43-49).
class RayMarchingResources {
  public:
    static RayMarchingResources new_(int device = 0) { return RayMarchingResources(device); }
    RayMarchingResources(RayMarchingResources&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    RayMarchingResources(const RayMarchingResources&) = delete;
    RayMarchingResources& operator=(const RayMarchingResources&) = delete;
    ~RayMarchingResources() { rm_destroy(ctx_); }
    rm_ctx* ctx() const { return ctx_; }
    // The reference writes the limits once at init and drops the handle (renderer.rs:130-140);
    // BASELINE configs vary max_iter, so the mirror exposes the write.
    void set_limits(const RayMarchLimits& l) { check(rm_write_buffer(ctx_, RM_BUF_LIMITS, 0, l.as_shader_bytes().data(), 12)); }
    void check(int rc) const {
        if (rc != RM_OK) throw RmException(rc, rm_last_error(ctx_));
    }

  private:
    explicit RayMarchingResources(int device) {
        int rc = rm_create(device, &ctx_);
        if (rc != RM_OK) throw RmException(rc, rm_last_error(nullptr));
    }
    rm_ctx* ctx_ = nullptr;
};

// Per-frame value object (renderer.rs:177-193).
class RayMarchingCallback {
  public:
    // RayMarchingCallback::new(time, csg_node, viewport, camera) -- `time` is carried but unused (renderer.rs:178).
    static RayMarchingCallback new_(float time, std::optional<csg::CSGNode> csg_node, std::array<float, 2> viewport,
                                    Camera camera) {
        return RayMarchingCallback{time, std::move(csg_node), viewport, camera};
    }
    // prepare (renderer.rs:196-242): three queue.write_buffer calls.
    void prepare(RayMarchingResources& resources) const {
        const PreparedFrame f = prepare_frame(csg_node, viewport, camera);
        resources.check(rm_write_buffer(resources.ctx(), RM_BUF_UNIFORMS, 0, f.uniforms.as_shader_bytes().data(), 144));
        const uint32_t cmd_count = f.builder.cmd_count;
        resources.check(rm_write_buffer(resources.ctx(), RM_BUF_COMMANDS, 0, &cmd_count, 4));
        resources.check(rm_write_buffer(resources.ctx(), RM_BUF_COMMANDS, 4, f.builder.buffer.data(),
                                        f.builder.buffer.size() * 4));
    }
    // paint (renderer.rs:244-255): draws the full viewport into a host RGBA32F image.
    void paint(RayMarchingResources& resources, uint32_t width, uint32_t height, float* out_rgba) const {
        resources.check(rm_draw(resources.ctx(), width, height, 0, height, out_rgba, 0, nullptr));
    }

    float time;
    std::optional<csg::CSGNode> csg_node;
    std::array<float, 2> viewport;
    Camera camera;
};

}  // namespace ray_marching
