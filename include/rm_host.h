/*
 * rm_host.h -- C ABI of the host-side mirror (librm_host.so, no GPU code): the reference's
 * scene model + serializer and its orbit camera, i.e. everything RayMarchingCallback::prepare
 * computes BEFORE it touches the GPU.  A Rust/C/Python host can use these instead of
 * re-implementing them, or ignore them and feed rm_abi.h with its own blobs.
 *
 *   CSGNode / Sphere / Box / Union / Subtraction     src/ray_marching/csg/mod.rs:28-45,
 *                                                    primitives/sphere.rs:8-13, box.rs:8-12,
 *                                                    operations/mod.rs:7-11
 *   BuildCommands::build_commands                    sphere.rs:15-21, box.rs:14-20, operations/mod.rs:12-18
 *   CSGCommandBufferBuilder                          src/ray_marching/csg/builder.rs:26-62
 *   OrbitCameraController / Camera                   src/camera.rs:3-85
 *   prepare(): inv_proj, inv_view, Uniforms bytes    src/ray_marching/renderer.rs:205-222
 */
#ifndef RM_HOST_H
#define RM_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmh_node rmh_node;       /* a CSGNode (owning tree) */
typedef struct rmh_builder rmh_builder; /* a CSGCommandBufferBuilder */

/* ---- CSGNode constructors; binary operators deep-copy their children (Box::new(x.clone())) */
rmh_node* rmh_sphere(const float center[3], float radius);
rmh_node* rmh_box(const float center[3], const float radius[3]);
rmh_node* rmh_union(const rmh_node* lhs, const rmh_node* rhs);
rmh_node* rmh_subtraction(const rmh_node* lhs, const rmh_node* rhs);
/* extension node types (not implemented by the reference; DESIGN.md "Extension node types") */
rmh_node* rmh_plane(const float normal[3], float h);
rmh_node* rmh_cylinder(const float center[3], float radius, float half_height);
rmh_node* rmh_intersection(const rmh_node* lhs, const rmh_node* rhs);
rmh_node* rmh_smooth_union(const rmh_node* lhs, const rmh_node* rhs, float k);
/* space transformations: the node types csg/mod.rs:41-44 reserves by comment (opcodes 200-205, builder.rs:16-23);
 * the child is deep-copied; quaternion = (w, i, j, k), unit length; factor > 0 */
rmh_node* rmh_translation(const rmh_node* child, const float offset[3]);
rmh_node* rmh_rotation(const rmh_node* child, const float quaternion_wijk[4]);
rmh_node* rmh_scale(const rmh_node* child, float factor);
/* material tag (extension; README.md:11 "Material system" is future work in the reference): the child, deep-copied,
 * with its surfaces tagged by entry `index` (< 256) of the table rm_set_materials uploads; serialises to [child..., 300, index] */
rmh_node* rmh_material(const rmh_node* child, uint32_t index);
rmh_node* rmh_node_clone(const rmh_node* n);
void rmh_node_free(rmh_node* n);
/* Named synthetic scenes (g1, g8, g32, g64, g32_balanced; with extension nodes: g8x, g32s, ext_mix, xform_mix, mat_mix); NULL if unknown. */
rmh_node* rmh_scene(const char* name);

/* ---- CSGCommandBufferBuilder */
rmh_builder* rmh_builder_new(void);
void rmh_builder_free(rmh_builder* b);
void rmh_builder_push_command(rmh_builder* b, uint32_t cmd_type);
void rmh_builder_push_param_vec3(rmh_builder* b, const float value[3]);
void rmh_builder_push_param_float(rmh_builder* b, float value);
uint32_t rmh_builder_cmd_count(const rmh_builder* b);
uint32_t rmh_builder_len(const rmh_builder* b);         /* words in .buffer */
const uint32_t* rmh_builder_buffer(const rmh_builder* b);
/* node.build_commands(&mut builder); node may be NULL (csg_node == None: nothing is pushed) */
void rmh_build_commands(const rmh_node* node, rmh_builder* b);

/* ---- camera */
typedef struct rmh_orbit {
    float target[3];
    float pitch, yaw, radius;
    float pan_speed, yaw_speed, pitch_speed, dolly_speed;
} rmh_orbit;
typedef struct rmh_camera {
    float position[3];
    float rotation[4]; /* unit quaternion (w, i, j, k) */
} rmh_camera;
enum rmh_orbit_event { RMH_PAN = 0, RMH_ORBIT = 1, RMH_DOLLY = 2 };

void rmh_orbit_new(rmh_orbit* c, const float target[3], float radius); /* camera.rs:38-50 */
void rmh_orbit_update(rmh_orbit* c, int event, float dx, float dy);    /* camera.rs:62-84; Dolly uses dx */
void rmh_orbit_camera(const rmh_orbit* c, rmh_camera* out);            /* camera.rs:56-60 */
void rmh_camera_view(const rmh_camera* cam, float out16[16]);          /* camera.rs:10-12, column-major */

/* ---- prepare(): the 144-byte Uniforms blob for (viewport, camera) (renderer.rs:205-222) */
void rmh_prepare_uniforms(const float viewport[2], const rmh_camera* cam, void* out144);

#ifdef __cplusplus
}
#endif
#endif /* RM_HOST_H */
