// glm_min.h -- the handful of GLM types the reference's scene code touches (triangle.h:24-42 UVs,
// mesh.h:22-162 vertex transforms), so that the drop-in headers do not depend on the 47 kLoC GLM tree.
// Single precision, column-major, with GLM 0.9.8's evaluation order where a result can depend on it
// (mat4 * vec4 pairs the products as (c0*x + c1*y) + (c2*z + c3*w); translate/rotate/scale build their
// columns left to right), because mesh vertices pass through these in float before they become the
// double-precision triangles the kernels see.  If the real <glm.hpp> was included first it is used instead.
#ifndef RTK_GLM_MIN_H
#define RTK_GLM_MIN_H

#ifndef GLM_VERSION
#include <cmath>

namespace glm {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
};

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator*(float s, const vec3& v) { return vec3(s * v.x, s * v.y, s * v.z); }
inline vec3 operator*(const vec3& v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};
inline vec4 operator*(const vec4& v, float s) { return vec4(v.x * s, v.y * s, v.z * s, v.w * s); }
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct mat4 {
    vec4 col[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { col[0] = vec4(d, 0, 0, 0); col[1] = vec4(0, d, 0, 0); col[2] = vec4(0, 0, d, 0); col[3] = vec4(0, 0, 0, d); }
    vec4& operator[](int i) { return col[i]; }
    const vec4& operator[](int i) const { return col[i]; }
};
inline vec4 operator*(const mat4& m, const vec4& v) {
    const vec4 a = m[0] * v.x + m[1] * v.y;
    const vec4 b = m[2] * v.z + m[3] * v.w;
    return a + b;
}

inline float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }

inline mat4 translate(const mat4& m, const vec3& v) {
    mat4 r(m);
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}

inline mat4 scale(const mat4& m, const vec3& v) {
    mat4 r;
    r[0] = m[0] * v.x;
    r[1] = m[1] * v.y;
    r[2] = m[2] * v.z;
    r[3] = m[3];
    return r;
}

inline mat4 rotate(const mat4& m, float angle, const vec3& v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const float inv_len = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    const vec3 axis = v * inv_len;
    const vec3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2];
    r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2];
    r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2];
    r[3] = m[3];
    return r;
}

}  // namespace glm
#endif  // GLM_VERSION
#endif  // RTK_GLM_MIN_H
