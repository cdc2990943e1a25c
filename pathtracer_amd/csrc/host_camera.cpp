// Camera matrices with glm 0.9.9.9 conventions (right-handed, clip depth -1..1, column-major
// M[col*4 + row]) as used by the reference host: glm::lookAt / glm::perspective in
// Source/Runtime/Camera/Camera.cpp:10-18 and glm::inverse at
// Source/PathTracer/PathTracingApplication.cpp:250.  The 4x4 inverse is also what the kernel
// launch uses for GLSL's inverse(ViewMatrix) (Shaders/pathtracing.comp:159-160), hoisted to
// the host.  Float arithmetic, no FMA contraction.
#include <cmath>

#include "host.h"

namespace ptcore {
namespace {
inline float Dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void Cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - b[1] * a[2];
    o[1] = a[2] * b[0] - b[2] * a[0];
    o[2] = a[0] * b[1] - b[0] * a[1];
}
inline void Normalize3(float v[3])
{
    float inv = 1.0f / std::sqrt(Dot3(v, v));  // glm::inversesqrt
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}
}  // namespace

void LookAt(const float eye[3], const float center[3], const float up[3], float M[16])
{
    float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    Normalize3(f);
    float s[3];
    Cross3(f, up, s);
    Normalize3(s);
    float u[3];
    Cross3(s, f, u);
    for (int i = 0; i < 16; i++) M[i] = 0.0f;
    M[0] = s[0];
    M[4] = s[1];
    M[8] = s[2];
    M[1] = u[0];
    M[5] = u[1];
    M[9] = u[2];
    M[2] = -f[0];
    M[6] = -f[1];
    M[10] = -f[2];
    M[12] = -Dot3(s, eye);
    M[13] = -Dot3(u, eye);
    M[14] = Dot3(f, eye);
    M[15] = 1.0f;
}

void Perspective(float fovy, float aspect, float zNear, float zFar, float M[16])
{
    const float tanHalf = std::tan(fovy / 2.0f);
    for (int i = 0; i < 16; i++) M[i] = 0.0f;
    M[0] = 1.0f / (aspect * tanHalf);
    M[5] = 1.0f / tanHalf;
    M[10] = -(zFar + zNear) / (zFar - zNear);
    M[11] = -1.0f;
    M[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
}

// Cofactor (adjugate) inverse; sub-determinants are shared between the columns.
void Inverse4(const float m[16], float out[16])
{
#define E(c, r) m[(c)*4 + (r)]
    const float c00 = E(2, 2) * E(3, 3) - E(3, 2) * E(2, 3);
    const float c02 = E(1, 2) * E(3, 3) - E(3, 2) * E(1, 3);
    const float c03 = E(1, 2) * E(2, 3) - E(2, 2) * E(1, 3);
    const float c04 = E(2, 1) * E(3, 3) - E(3, 1) * E(2, 3);
    const float c06 = E(1, 1) * E(3, 3) - E(3, 1) * E(1, 3);
    const float c07 = E(1, 1) * E(2, 3) - E(2, 1) * E(1, 3);
    const float c08 = E(2, 1) * E(3, 2) - E(3, 1) * E(2, 2);
    const float c10 = E(1, 1) * E(3, 2) - E(3, 1) * E(1, 2);
    const float c11 = E(1, 1) * E(2, 2) - E(2, 1) * E(1, 2);
    const float c12 = E(2, 0) * E(3, 3) - E(3, 0) * E(2, 3);
    const float c14 = E(1, 0) * E(3, 3) - E(3, 0) * E(1, 3);
    const float c15 = E(1, 0) * E(2, 3) - E(2, 0) * E(1, 3);
    const float c16 = E(2, 0) * E(3, 2) - E(3, 0) * E(2, 2);
    const float c18 = E(1, 0) * E(3, 2) - E(3, 0) * E(1, 2);
    const float c19 = E(1, 0) * E(2, 2) - E(2, 0) * E(1, 2);
    const float c20 = E(2, 0) * E(3, 1) - E(3, 0) * E(2, 1);
    const float c22 = E(1, 0) * E(3, 1) - E(3, 0) * E(1, 1);
    const float c23 = E(1, 0) * E(2, 1) - E(2, 0) * E(1, 1);

    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {E(1, 0), E(0, 0), E(0, 0), E(0, 0)}, v1[4] = {E(1, 1), E(0, 1), E(0, 1), E(0, 1)};
    const float v2[4] = {E(1, 2), E(0, 2), E(0, 2), E(0, 2)}, v3[4] = {E(1, 3), E(0, 3), E(0, 3), E(0, 3)};
    float inv[16];
    for (int k = 0; k < 4; k++) {
        const float sa = (k & 1) ? -1.0f : 1.0f, sb = -sa;
        inv[0 * 4 + k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
        inv[1 * 4 + k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb;
        inv[2 * 4 + k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa;
        inv[3 * 4 + k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
    }
    const float d0 = E(0, 0) * inv[0], d1 = E(0, 1) * inv[4], d2 = E(0, 2) * inv[8], d3 = E(0, 3) * inv[12];
    const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; i++) out[i] = inv[i] * oneOverDet;
#undef E
}

}  // namespace ptcore
