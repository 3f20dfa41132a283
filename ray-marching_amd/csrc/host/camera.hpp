// camera.hpp -- C++ mirror of src/camera.rs: Camera (:3-13), OrbitCameraControllerEvent
// (:15-19), OrbitCameraController (:21-85).
#pragma once
#include <algorithm>
#include <array>
#include <variant>

#include "linalg.hpp"

namespace ray_marching {

struct Camera {  // camera.rs:3-6
    Point3 position;
    UnitQuaternion rotation;

    // World-to-view transformation (camera.rs:10-12):
    //   convert(self.rotation.inverse() * Translation3::from(-self.position.coords))
    Affine3 view() const {
        const UnitQuaternion ri = rotation.inverse();
        const Vector3 t = ri * (-position);  // Isometry: rotation * translation
        const std::array<float, 9> r = ri.to_rotation_matrix();
        Affine3 a;
        for (int row = 0; row < 3; row++)
            for (int col = 0; col < 3; col++) a.matrix.at(row, col) = r[row * 3 + col];
        a.matrix.at(0, 3) = t.x;
        a.matrix.at(1, 3) = t.y;
        a.matrix.at(2, 3) = t.z;
        return a;
    }
};

namespace OrbitCameraControllerEvent {  // camera.rs:15-19
struct Pan { std::array<float, 2> delta; };
struct Orbit { std::array<float, 2> delta; };
struct Dolly { float delta; };
using Any = std::variant<Pan, Orbit, Dolly>;
}  // namespace OrbitCameraControllerEvent

class OrbitCameraController {  // camera.rs:21-35
  public:
    Point3 target;
    float pitch = 0.0f, yaw = 0.0f, radius = 0.0f;
    float pan_speed = 0.01f, yaw_speed = 0.01f, pitch_speed = 0.01f, dolly_speed = 0.01f;

    // OrbitCameraController::new (camera.rs:38-50); `new` is a C++ keyword.
    static OrbitCameraController new_(std::array<float, 3> target, float radius) {
        OrbitCameraController c;
        c.target = {target[0], target[1], target[2]};
        c.radius = radius;
        return c;
    }
    UnitQuaternion rotation() const {  // camera.rs:52-54
        return UnitQuaternion::from_euler_angles(-pitch, -yaw, 0.0f);
    }
    Camera camera() const {  // camera.rs:56-60
        const UnitQuaternion r = rotation();
        return Camera{target + (r * Vector3::z_axis()) * radius, r};
    }
    void update(const OrbitCameraControllerEvent::Any& event) {  // camera.rs:62-84
        if (const auto* pan = std::get_if<OrbitCameraControllerEvent::Pan>(&event)) {
            const Vector3 right = rotation() * Vector3::x_axis();
            const Vector3 up = rotation() * Vector3::y_axis();
            target = target + (right * -pan->delta[0] + up * pan->delta[1]) * pan_speed;
        } else if (const auto* orbit = std::get_if<OrbitCameraControllerEvent::Orbit>(&event)) {
            yaw += orbit->delta[0] * yaw_speed;
            pitch += orbit->delta[1] * pitch_speed;
            pitch = std::min(std::max(pitch, -1.5f), 1.5f);  // f32::clamp(-1.5, 1.5)
        } else if (const auto* dolly = std::get_if<OrbitCameraControllerEvent::Dolly>(&event)) {
            radius += dolly->delta * dolly_speed * radius;
            radius = std::fmax(radius, 0.1f);
        }
    }
};

}  // namespace ray_marching
