// linalg.h -- the few fixed-size fp32 operations the tracking path needs (the reference uses
// Eigen, which is not a dependency here): 3-vectors, row-major 3x3 / 4x4 matrices, unit
// quaternions, and a 9x9 lower-triangular Cholesky (Eigen::LLT in cpp/pnp/lev_marq.h:299-314).
#pragma once

#include <array>
#include <cmath>
#include <cstring>

using Float = float;  // cpp/eigen_typedefs.h
using Vec2f = std::array<float, 2>;
using Vec3f = std::array<float, 3>;
using Vec4f = std::array<float, 4>;
using Mat3f = std::array<float, 9>;   // row-major
using Mat4f = std::array<float, 16>;  // row-major

inline Vec3f operator+(const Vec3f& a, const Vec3f& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3f operator-(const Vec3f& a, const Vec3f& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3f operator*(const Vec3f& a, float s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline Vec3f operator-(const Vec3f& a) { return {-a[0], -a[1], -a[2]}; }
inline float Dot(const Vec3f& a, const Vec3f& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3f Cross(const Vec3f& a, const Vec3f& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline float Norm(const Vec3f& a) { return std::sqrt(Dot(a, a)); }
inline Vec3f MatVec(const Mat3f& m, const Vec3f& v) {
    return {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
            m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
}
inline Mat3f Transpose(const Mat3f& m) { return {m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8]}; }
inline Mat3f Identity3() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
inline Mat4f Identity4() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

inline Mat4f MatMul4(const Mat4f& a, const Mat4f& b) {
    Mat4f c{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
    return c;
}

// general 4x4 inverse (Eigen's .inverse() in cpp/ray_casting.h:55-56), computed in double
inline bool Inverse4(const Mat4f& m, Mat4f* out) {
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = m[4 * i + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 8; j++) std::swap(a[piv][j], a[c][j]);
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] *= inv;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                const double f = a[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) (*out)[4 * i + j] = static_cast<float>(a[i][4 + j]);
    return true;
}

// Unit quaternion, stored like Eigen (x, y, z, w); exposed to Python as WXYZ (polychase_pybind.cc:224-232)
struct Quatf {
    float x = 0, y = 0, z = 0, w = 1;
    static Quatf FromWXYZ(float w, float x, float y, float z) { return Quatf{x, y, z, w}; }
    Quatf Conjugate() const { return {-x, -y, -z, w}; }
    // Eigen::Quaternion::toRotationMatrix
    Mat3f ToRotationMatrix() const {
        const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const float twx = tx * w, twy = ty * w, twz = tz * w;
        const float txx = tx * x, txy = ty * x, txz = tz * x;
        const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
        return {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                txz - twy,       tyz + twx, 1 - (txx + tyy)};
    }
    // Eigen: Quaternion(Matrix3) (Shepperd's method as in Eigen's quaternionbase_assign_impl)
    static Quatf FromRotationMatrix(const Mat3f& m) {
        Quatf q;
        float t = m[0] + m[4] + m[8];
        if (t > 0.f) {
            t = std::sqrt(t + 1.0f);
            q.w = 0.5f * t;
            t = 0.5f / t;
            q.x = (m[7] - m[5]) * t;
            q.y = (m[2] - m[6]) * t;
            q.z = (m[3] - m[1]) * t;
        } else {
            int i = 0;
            if (m[4] > m[0]) i = 1;
            if (m[8] > m[4 * i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
            float v[3];
            v[i] = 0.5f * t;
            t = 0.5f / t;
            q.w = (m[3 * k + j] - m[3 * j + k]) * t;
            v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
            v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
            q.x = v[0];
            q.y = v[1];
            q.z = v[2];
        }
        return q;
    }
};
inline Quatf operator*(const Quatf& a, const Quatf& b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// q * v (Eigen::Quaternion::_transformVector)
inline Vec3f Rotate(const Quatf& q, const Vec3f& v) {
    const Vec3f u{q.x, q.y, q.z};
    Vec3f uv = Cross(u, v);
    uv = uv + uv;
    return v + uv * q.w + Cross(u, uv);
}
// QuatStepPost (cpp/pnp/quaternion.h:11-20): q * AngleAxis(|w|, w/|w|)
inline Quatf QuatStepPost(const Quatf& q, const Vec3f& w_delta) {
    const float angle = Norm(w_delta);
    if (angle > 0) {
        const Vec3f axis = w_delta * (1.0f / angle);
        const float s = std::sin(0.5f * angle), c = std::cos(0.5f * angle);
        return q * Quatf{axis[0] * s, axis[1] * s, axis[2] * s, c};
    }
    return q;
}

// In-place lower Cholesky of an NxN row-major matrix (only the lower triangle is read/written);
// left-looking like Eigen's unblocked llt_inplace.  Returns false if not positive definite.
template <int N>
inline bool CholeskyLower(float* a) {
    for (int k = 0; k < N; k++) {
        float x = a[k * N + k];
        for (int j = 0; j < k; j++) x -= a[k * N + j] * a[k * N + j];
        if (!(x > 0.0f)) return false;
        x = std::sqrt(x);
        a[k * N + k] = x;
        for (int i = k + 1; i < N; i++) {
            float s = a[i * N + k];
            for (int j = 0; j < k; j++) s -= a[i * N + j] * a[k * N + j];
            a[i * N + k] = s / x;
        }
    }
    return true;
}
// Solve L L^T x = b
template <int N>
inline void CholeskySolve(const float* l, const float* b, float* x) {
    float y[N];
    for (int i = 0; i < N; i++) {
        float s = b[i];
        for (int j = 0; j < i; j++) s -= l[i * N + j] * y[j];
        y[i] = s / l[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {
        float s = y[i];
        for (int j = i + 1; j < N; j++) s -= l[j * N + i] * x[j];
        x[i] = s / l[i * N + i];
    }
}
