// vecmath.h -- the few fp32 vector operations the host side needs.
//
// The reference takes these from LiteMath (an absent submodule); the camera
// matrix and the generated primitive centres feed the device kernels, so the
// operation order here is part of the numerical contract: every operation is a
// single fp32 operation, evaluated left to right, never fused (this file is
// compiled with -ffp-contract=off).
#pragma once

#include <cmath>

namespace rmath {

struct float3 {
    float x, y, z;
    float3() : x(0.0f), y(0.0f), z(0.0f) {}
    explicit float3(float v) : x(v), y(v), z(v) {}
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

inline float3 operator+(float3 a, float3 b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(float3 a, float3 b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator-(float3 a) { return float3(-a.x, -a.y, -a.z); }
inline float3 operator*(float3 a, float s) { return float3(a.x * s, a.y * s, a.z * s); }
inline float3 operator*(float s, float3 a) { return float3(s * a.x, s * a.y, s * a.z); }
inline float3 operator/(float3 a, float s) { return float3(a.x / s, a.y / s, a.z / s); }

inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(float3 a) { const float inv = 1.0f / length(a); return a * inv; }
inline float3 cross(float3 a, float3 b) {
    return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Column-major 4x4: m[4*c + r].
struct float4x4 {
    float m[16];
    float4x4() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    void set_col(int c, float x, float y, float z, float w) {
        m[4 * c] = x; m[4 * c + 1] = y; m[4 * c + 2] = z; m[4 * c + 3] = w;
    }
};

// (M * (v, w)).xyz with the column sum taken left to right.
inline float3 transform(const float4x4 &M, float3 v, float w) {
    const float *m = M.m;
    return float3(v.x * m[0] + v.y * m[4] + v.z * m[8] + w * m[12],
                  v.x * m[1] + v.y * m[5] + v.z * m[9] + w * m[13],
                  v.x * m[2] + v.y * m[6] + v.z * m[10] + w * m[14]);
}

constexpr float DEG_TO_RAD = 3.14159265358979323846f / 180.0f;

}  // namespace rmath
