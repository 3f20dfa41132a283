// rm_host.cpp -- C ABI (include/rm_host.h) over the C++ host mirror in csrc/host/.
#include "rm_host.h"

#include <cstring>
#include <new>
#include <optional>

#include "host/camera.hpp"
#include "host/csg.hpp"
#include "host/linalg.hpp"
#include "host/scenes.hpp"

#define RMH_EXPORT extern "C" __attribute__((visibility("default")))

using namespace ray_marching;

struct rmh_node {
    csg::CSGNode node;
};
struct rmh_builder {
    csg::CSGCommandBufferBuilder b;
};

namespace {
std::array<float, 3> a3(const float* p) { return {p[0], p[1], p[2]}; }

Camera to_camera(const rmh_camera* c) {
    Camera cam;
    cam.position = {c->position[0], c->position[1], c->position[2]};
    cam.rotation = {c->rotation[0], c->rotation[1], c->rotation[2], c->rotation[3]};
    return cam;
}
OrbitCameraController to_controller(const rmh_orbit* c) {
    OrbitCameraController o;
    o.target = {c->target[0], c->target[1], c->target[2]};
    o.pitch = c->pitch; o.yaw = c->yaw; o.radius = c->radius;
    o.pan_speed = c->pan_speed; o.yaw_speed = c->yaw_speed;
    o.pitch_speed = c->pitch_speed; o.dolly_speed = c->dolly_speed;
    return o;
}
void from_controller(const OrbitCameraController& o, rmh_orbit* c) {
    c->target[0] = o.target.x; c->target[1] = o.target.y; c->target[2] = o.target.z;
    c->pitch = o.pitch; c->yaw = o.yaw; c->radius = o.radius;
    c->pan_speed = o.pan_speed; c->yaw_speed = o.yaw_speed;
    c->pitch_speed = o.pitch_speed; c->dolly_speed = o.dolly_speed;
}
}  // namespace

RMH_EXPORT rmh_node* rmh_sphere(const float center[3], float radius) {
    if (!center) return nullptr;
    return new (std::nothrow) rmh_node{csg::CSGNode(csg::Sphere{a3(center), radius})};
}
RMH_EXPORT rmh_node* rmh_box(const float center[3], const float radius[3]) {
    if (!center || !radius) return nullptr;
    return new (std::nothrow) rmh_node{csg::CSGNode(csg::Box{a3(center), a3(radius)})};
}
RMH_EXPORT rmh_node* rmh_union(const rmh_node* lhs, const rmh_node* rhs) {
    if (!lhs || !rhs) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_union(lhs->node, rhs->node)};
}
RMH_EXPORT rmh_node* rmh_subtraction(const rmh_node* lhs, const rmh_node* rhs) {
    if (!lhs || !rhs) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_subtraction(lhs->node, rhs->node)};
}
RMH_EXPORT rmh_node* rmh_plane(const float normal[3], float h) {
    if (!normal) return nullptr;
    return new (std::nothrow) rmh_node{csg::CSGNode(csg::Plane{a3(normal), h})};
}
RMH_EXPORT rmh_node* rmh_cylinder(const float center[3], float radius, float half_height) {
    if (!center) return nullptr;
    return new (std::nothrow) rmh_node{csg::CSGNode(csg::Cylinder{a3(center), radius, half_height})};
}
RMH_EXPORT rmh_node* rmh_intersection(const rmh_node* lhs, const rmh_node* rhs) {
    if (!lhs || !rhs) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_intersection(lhs->node, rhs->node)};
}
RMH_EXPORT rmh_node* rmh_smooth_union(const rmh_node* lhs, const rmh_node* rhs, float k) {
    if (!lhs || !rhs) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_smooth_union(lhs->node, rhs->node, k)};
}
RMH_EXPORT rmh_node* rmh_translation(const rmh_node* child, const float offset[3]) {
    if (!child || !offset) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_translation(child->node, a3(offset))};
}
RMH_EXPORT rmh_node* rmh_rotation(const rmh_node* child, const float quaternion_wijk[4]) {
    if (!child || !quaternion_wijk) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_rotation(child->node, {quaternion_wijk[0], quaternion_wijk[1], quaternion_wijk[2], quaternion_wijk[3]})};
}
RMH_EXPORT rmh_node* rmh_scale(const rmh_node* child, float factor) {
    if (!child) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_scale(child->node, factor)};
}
RMH_EXPORT rmh_node* rmh_material(const rmh_node* child, uint32_t index) {
    if (!child) return nullptr;
    return new (std::nothrow) rmh_node{csg::make_material(child->node, index)};
}
RMH_EXPORT rmh_node* rmh_node_clone(const rmh_node* n) { return n ? new (std::nothrow) rmh_node{n->node} : nullptr; }
RMH_EXPORT void rmh_node_free(rmh_node* n) { delete n; }
RMH_EXPORT rmh_node* rmh_scene(const char* name) {
    if (!name) return nullptr;
    std::optional<csg::CSGNode> n = scenes::by_name(name);
    return n ? new (std::nothrow) rmh_node{std::move(*n)} : nullptr;
}

RMH_EXPORT rmh_builder* rmh_builder_new(void) { return new (std::nothrow) rmh_builder(); }
RMH_EXPORT void rmh_builder_free(rmh_builder* b) { delete b; }
RMH_EXPORT void rmh_builder_push_command(rmh_builder* b, uint32_t cmd_type) {
    if (b) b->b.push_command(static_cast<csg::CSGCommandType>(cmd_type));
}
RMH_EXPORT void rmh_builder_push_param_vec3(rmh_builder* b, const float value[3]) {
    if (b && value) b->b.push_param_vec3(a3(value));
}
RMH_EXPORT void rmh_builder_push_param_float(rmh_builder* b, float value) {
    if (b) b->b.push_param_float(value);
}
RMH_EXPORT uint32_t rmh_builder_cmd_count(const rmh_builder* b) { return b ? b->b.cmd_count : 0; }
RMH_EXPORT uint32_t rmh_builder_len(const rmh_builder* b) { return b ? (uint32_t)b->b.buffer.size() : 0; }
RMH_EXPORT const uint32_t* rmh_builder_buffer(const rmh_builder* b) { return b ? b->b.buffer.data() : nullptr; }
RMH_EXPORT void rmh_build_commands(const rmh_node* node, rmh_builder* b) {
    if (node && b) node->node.build_commands(b->b);  // renderer.rs:224-227
}

RMH_EXPORT void rmh_orbit_new(rmh_orbit* c, const float target[3], float radius) {
    if (!c || !target) return;
    from_controller(OrbitCameraController::new_(a3(target), radius), c);
}
RMH_EXPORT void rmh_orbit_update(rmh_orbit* c, int event, float dx, float dy) {
    if (!c) return;
    OrbitCameraController o = to_controller(c);
    switch (event) {
    case RMH_PAN: o.update(OrbitCameraControllerEvent::Pan{{dx, dy}}); break;
    case RMH_ORBIT: o.update(OrbitCameraControllerEvent::Orbit{{dx, dy}}); break;
    case RMH_DOLLY: o.update(OrbitCameraControllerEvent::Dolly{dx}); break;
    default: return;
    }
    from_controller(o, c);
}
RMH_EXPORT void rmh_orbit_camera(const rmh_orbit* c, rmh_camera* out) {
    if (!c || !out) return;
    const Camera cam = to_controller(c).camera();
    out->position[0] = cam.position.x; out->position[1] = cam.position.y; out->position[2] = cam.position.z;
    out->rotation[0] = cam.rotation.w; out->rotation[1] = cam.rotation.i;
    out->rotation[2] = cam.rotation.j; out->rotation[3] = cam.rotation.k;
}
RMH_EXPORT void rmh_camera_view(const rmh_camera* cam, float out16[16]) {
    if (!cam || !out16) return;
    const Matrix4 m = to_camera(cam).view().to_homogeneous();
    std::memcpy(out16, m.m.data(), 64);
}
RMH_EXPORT void rmh_prepare_uniforms(const float viewport[2], const rmh_camera* cam, void* out144) {
    if (!viewport || !cam || !out144) return;
    // renderer.rs:205-222 (same arithmetic as host/renderer.hpp prepare_frame, without the rm_abi dependency)
    const Perspective3 projection(viewport[0] / viewport[1], 0.78539816339744830962f, 1.0f, 10000.0f);
    const Matrix4 inv_proj = projection.inverse();
    const Matrix4 inv_view = to_camera(cam).view().inverse().to_homogeneous();
    float blob[36];
    std::memset(blob, 0, sizeof blob);
    blob[0] = viewport[0];
    blob[1] = viewport[1];
    std::memcpy(blob + 4, inv_proj.m.data(), 64);
    std::memcpy(blob + 20, inv_view.m.data(), 64);
    std::memcpy(out144, blob, 144);
}
