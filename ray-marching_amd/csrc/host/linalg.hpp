// linalg.hpp -- the few nalgebra 0.32.4 operations the reference's host path uses
// (src/camera.rs, src/ray_marching/renderer.rs:205-211), restated in f32.
//
// nalgebra is a crates.io dependency (Cargo.lock:1849-1851) whose source is not in the
// reference tree; the formulas below restate its published algorithms and are "parity
// unpinned" at the ulp level.  They sit BEFORE the kernel boundary (the kernel takes the
// finished matrices), so kernel parity does not depend on them.
#pragma once
#include <array>
#include <cmath>
#include <cstring>

namespace ray_marching {

struct Vector3 {
    float x = 0, y = 0, z = 0;
    static Vector3 x_axis() { return {1, 0, 0}; }
    static Vector3 y_axis() { return {0, 1, 0}; }
    static Vector3 z_axis() { return {0, 0, 1}; }
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3 operator*(Vector3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3 operator-(Vector3 a) { return {-a.x, -a.y, -a.z}; }
inline Vector3 cross(Vector3 a, Vector3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
using Point3 = Vector3;

// Column-major 4x4, element (r,c) at m[c*4+r]: the storage nalgebra and encase agree on.
struct Matrix4 {
    std::array<float, 16> m{};
    float& at(int r, int c) { return m[c * 4 + r]; }
    float at(int r, int c) const { return m[c * 4 + r]; }
    static Matrix4 identity() {
        Matrix4 r;
        r.at(0, 0) = r.at(1, 1) = r.at(2, 2) = r.at(3, 3) = 1.0f;
        return r;
    }
};

// UnitQuaternion<f32>, components (w, i, j, k).
struct UnitQuaternion {
    float w = 1, i = 0, j = 0, k = 0;

    // UnitQuaternion::from_euler_angles(roll, pitch, yaw)
    static UnitQuaternion from_euler_angles(float roll, float pitch, float yaw) {
        const float sr = std::sin(roll * 0.5f), cr = std::cos(roll * 0.5f);
        const float sp = std::sin(pitch * 0.5f), cp = std::cos(pitch * 0.5f);
        const float sy = std::sin(yaw * 0.5f), cy = std::cos(yaw * 0.5f);
        UnitQuaternion q;
        q.w = cr * cp * cy + sr * sp * sy;
        q.i = sr * cp * cy - cr * sp * sy;
        q.j = cr * sp * cy + sr * cp * sy;
        q.k = cr * cp * sy - sr * sp * cy;
        return q;
    }
    UnitQuaternion inverse() const { return {w, -i, -j, -k}; }  // conjugate of a unit quaternion

    // UnitQuaternion * Vector3: t = (q.ijk x v) * 2;  t*w + (q.ijk x t) + v
    Vector3 operator*(Vector3 v) const {
        const Vector3 qv{i, j, k};
        const Vector3 t = cross(qv, v) * 2.0f;
        const Vector3 c = cross(qv, t);
        return (t * w + c) + v;
    }
    // to_rotation_matrix(), row-major 3x3
    std::array<float, 9> to_rotation_matrix() const {
        const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
        const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
        const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
        return {ww + ii - jj - kk, ij - wk,           wj + ik,
                wk + ij,           ww - ii + jj - kk, jk - wi,
                ik - wj,           wi + jk,           ww - ii - jj + kk};
    }
};

// Generic 4x4 inverse by cofactors (the do_inverse4 nalgebra uses for Matrix4::try_inverse).
inline bool try_inverse(const Matrix4& a, Matrix4& out) {
    const std::array<float, 16>& m = a.m;
    std::array<float, 16> c;
    c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    if (det == 0.0f) return false;
    const float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out.m[i] = c[i] * inv_det;
    return true;
}

// Affine3<f32>: a homogeneous 4x4 (what `convert(isometry)` produces, camera.rs:11).
struct Affine3 {
    Matrix4 matrix = Matrix4::identity();
    Affine3 inverse() const {  // Transform::inverse() -> generic matrix inverse
        Affine3 r;
        if (!try_inverse(matrix, r.matrix)) r.matrix = Matrix4::identity();
        return r;
    }
    Matrix4 to_homogeneous() const { return matrix; }
};

// Perspective3<f32> (renderer.rs:206-207).
struct Perspective3 {
    Matrix4 matrix = Matrix4::identity();
    Perspective3(float aspect, float fovy, float znear, float zfar) {
        matrix.at(3, 3) = 0.0f;
        matrix.at(3, 2) = -1.0f;
        const float m11 = 1.0f / std::tan(fovy / 2.0f);  // set_fovy
        matrix.at(1, 1) = m11;
        matrix.at(0, 0) = m11 / aspect;                   // set_aspect
        matrix.at(2, 2) = (zfar + znear) / (znear - zfar);  // set_znear_and_zfar
        matrix.at(2, 3) = zfar * znear * 2.0f / (znear - zfar);
    }
    Matrix4 inverse() const {
        Matrix4 r = matrix;
        r.at(0, 0) = 1.0f / matrix.at(0, 0);
        r.at(1, 1) = 1.0f / matrix.at(1, 1);
        r.at(2, 2) = 0.0f;
        const float m23 = matrix.at(2, 3), m32 = matrix.at(3, 2);
        r.at(2, 3) = 1.0f / m32;
        r.at(3, 2) = 1.0f / m23;
        r.at(3, 3) = -matrix.at(2, 2) / (m23 * m32);
        return r;
    }
};

}  // namespace ray_marching
