// TEST INFRASTRUCTURE ONLY.
//
// CPU oracle for the path-tracing hot path of Kliaxe/PathTracer: a scalar float32
// restatement of the reference's compute shader and of the host arithmetic that feeds it.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (pathtracer_amd/, libptcore.so) never does.
//
// What pins it.  The reference has no tests, golden vectors or known-answer fixtures for the
// shader (there is no test directory at all), and its GLSL cannot be compiled or run in the
// build environment (no GL context, no GLSL compiler).  SHADING PARITY IS THEREFORE UNPINNED:
// this file follows the shader text line by line and nothing executable confirms it.  The BVH
// builder IS pinned: the reference's own BVH.cpp is compiled unmodified into
// oracle/_ref/libref_shim.so (see Makefile) and the product's builder must match it bit for bit.
// The HDRI-cache builder below IS pinned too: lines 803-901 of the reference's
// PathTracingRenderer.cpp are compiled in place into oracle/_ref/libref_hdri.so and this file's
// restatement (and the product's) must reproduce every float of the cache.
//
// Each function cites the reference lines it restates (paths relative to
// Source/PathTracer/Shaders/ unless noted).  Arithmetic is IEEE binary32, one rounding per
// operation, evaluated left to right in GLSL source order; GLSL built-ins are defined in
// oracle_math.h.  Where GLSL leaves something undefined the choice made here is stated:
//   * argument evaluation order at pathtracing.comp:178 -> left to right;
//   * mat*vec sums -> ((m0*v0 + m1*v1) + m2*v2) + m3*v3;
//   * inverse(mat4) -> cofactor expansion (same scheme as glm::inverse);
//   * traversal-stack overflow (intersection.glsl:122,148,172-179 can write stack[16]) -> the
//     write lands in spare slots, the pixel is flagged in `overflow_mask`, and tests exclude it;
//   * texture(): bilinear, REPEAT, base level, float weights
//     (x = u*W - 0.5, i0 = floor(x), a = x - i0, texel indices wrapped modulo W).
//   * min(x, y) / max(x, y) with a NaN operand (GLSL 4.60 section 8.3 leaves the result undefined;
//     its formula min = y < x ? y : x would return x) -> the other operand, and -0 orders below +0
//     (IEEE 754-2019 minimumNumber / maximumNumber, which is what v_min_f32 / v_max_f32 compute on
//     gfx950).  It matters in HitAABB (intersection.glsl:63-77): a ray that lies in a slab plane with a
//     zero direction component yields 0 * inf = NaN for that slab, and this choice drops the slab from
//     the test instead of turning the whole box into a miss or a hit by operand order;
//   * the dead primary trace at pathtracing.comp:237 has no observable effect and is not run.
// The two DEBUG_ENABLED views of Library/debug.glsl are restated as well (DebugBvh, DebugHdriCache); the
// reference holds no image of either, so they are parity-unpinned like the shading: tests/test_host_misc.py
// checks the values that follow from the shader text by hand.  DebugBvh's seventeenth push lands in spare
// slots like the traversal's.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "oracle_math.h"

namespace oracle {

// ------------------------------------------------------------------------- vectors
struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
};
struct vec4 {
    float x, y, z, w;
};
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }
inline vec3 V3(float s) { return {s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline vec3& operator+=(vec3& a, vec3 b)
{
    a = a + b;
    return a;
}
inline vec3& operator*=(vec3& a, vec3 b)
{
    a = a * b;
    return a;
}
inline vec3& operator*=(vec3& a, float s)
{
    a = a * s;
    return a;
}
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float length(vec3 a) { return gl_sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * gl_inversesqrt(dot(a, a)); }
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 vmin(vec3 a, vec3 b) { return {gl_min(a.x, b.x), gl_min(a.y, b.y), gl_min(a.z, b.z)}; }
inline vec3 vmax(vec3 a, vec3 b) { return {gl_max(a.x, b.x), gl_max(a.y, b.y), gl_max(a.z, b.z)}; }
inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
inline vec3 refract(vec3 I, vec3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f);
    return eta * I - (eta * d + gl_sqrt(k)) * N;
}
inline vec3 pow3(vec3 a, vec3 e) { return {gl_pow(a.x, e.x), gl_pow(a.y, e.y), gl_pow(a.z, e.z)}; }

struct mat4 {
    float m[16];  // column-major: m[c*4 + r]
};
inline vec4 mul(const mat4& M, vec4 v)
{
    vec4 r;
    float* o = &r.x;
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 4; i++) o[i] = ((M.m[0 + i] * vv[0] + M.m[4 + i] * vv[1]) + M.m[8 + i] * vv[2]) + M.m[12 + i] * vv[3];
    return r;
}
// GLSL inverse(mat4), cofactor expansion.
mat4 inverse(const mat4& in)
{
    const float* m = in.m;
#define E(c, r) m[(c)*4 + (r)]
    float c00 = E(2, 2) * E(3, 3) - E(3, 2) * E(2, 3), c02 = E(1, 2) * E(3, 3) - E(3, 2) * E(1, 3);
    float c03 = E(1, 2) * E(2, 3) - E(2, 2) * E(1, 3), c04 = E(2, 1) * E(3, 3) - E(3, 1) * E(2, 3);
    float c06 = E(1, 1) * E(3, 3) - E(3, 1) * E(1, 3), c07 = E(1, 1) * E(2, 3) - E(2, 1) * E(1, 3);
    float c08 = E(2, 1) * E(3, 2) - E(3, 1) * E(2, 2), c10 = E(1, 1) * E(3, 2) - E(3, 1) * E(1, 2);
    float c11 = E(1, 1) * E(2, 2) - E(2, 1) * E(1, 2), c12 = E(2, 0) * E(3, 3) - E(3, 0) * E(2, 3);
    float c14 = E(1, 0) * E(3, 3) - E(3, 0) * E(1, 3), c15 = E(1, 0) * E(2, 3) - E(2, 0) * E(1, 3);
    float c16 = E(2, 0) * E(3, 2) - E(3, 0) * E(2, 2), c18 = E(1, 0) * E(3, 2) - E(3, 0) * E(1, 2);
    float c19 = E(1, 0) * E(2, 2) - E(2, 0) * E(1, 2), c20 = E(2, 0) * E(3, 1) - E(3, 0) * E(2, 1);
    float c22 = E(1, 0) * E(3, 1) - E(3, 0) * E(1, 1), c23 = E(1, 0) * E(2, 1) - E(2, 0) * E(1, 1);
    float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    float v0[4] = {E(1, 0), E(0, 0), E(0, 0), E(0, 0)}, v1[4] = {E(1, 1), E(0, 1), E(0, 1), E(0, 1)};
    float v2[4] = {E(1, 2), E(0, 2), E(0, 2), E(0, 2)}, v3[4] = {E(1, 3), E(0, 3), E(0, 3), E(0, 3)};
    mat4 inv;
    for (int k = 0; k < 4; k++) {
        float sa = (k & 1) ? -1.0f : 1.0f, sb = -sa;
        inv.m[0 + k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
        inv.m[4 + k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb;
        inv.m[8 + k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa;
        inv.m[12 + k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
    }
    float det = (E(0, 0) * inv.m[0] + E(0, 1) * inv.m[4]) + (E(0, 2) * inv.m[8] + E(0, 3) * inv.m[12]);
    float ood = 1.0f / det;
    for (int i = 0; i < 16; i++) inv.m[i] = inv.m[i] * ood;
#undef E
    return inv;
}

// --------------------------------------------------------- constants (Library/common.glsl:6-15)
constexpr float PI = 3.141592653589f;
constexpr float TWO_PI = 6.283185307178f;
constexpr float ONE_OVER_PI = 0.318309886183f;
constexpr float ONE_OVER_TWO_PI = 0.159154943091f;
constexpr float FLT_MAX_ = 3.402823466e+38f;
constexpr float MIN_DIELECTRICS_F0 = 0.04f;
constexpr int BVH_STACKSIZE = 16;
constexpr int STACK_SPARE = 48;  // where out-of-bounds pushes land (see header)

// --------------------------------------------------- buffer records (Library/common.glsl:32-91)
struct BvhNode {  // 48 B, std430
    int32_t left, right, n, index;
    float AA[3], padA;
    float BB[3], padB;
};
struct BvhPrimitive {  // 112 B
    float posAuvX[4], norAuvY[4], posBuvX[4], norBuvY[4], posCuvX[4], norCuvY[4];
    uint32_t meshIndex, pad[3];
};
struct MaterialRecord {  // 176 B
    uint64_t handle[11];
    uint64_t pad0;
    float emission[3], pad1;
    float albedo[3];
    float specular, specularTint, metallic, roughness, subsurface, anisotropy, sheenRoughness, sheenTint, clearcoat,
        clearcoatRoughness, refraction, transmission;
    float pad2;
};
static_assert(sizeof(BvhNode) == 48 && sizeof(BvhPrimitive) == 112 && sizeof(MaterialRecord) == 176, "std430 sizes");

struct Material {  // the shader-side struct after EvaluateMaterial
    uint64_t handle[11];
    vec3 emission, albedo;
    float specular, specularTint, metallic, roughness, subsurface, anisotropy, sheenRoughness, sheenTint, clearcoat,
        clearcoatRoughness, refraction, transmission;
};

// Uniform block (Library/resources.glsl:6-31); same field order as the product's pt_params so the
// tests can hand the same bytes to both.
struct Params {
    float ViewMatrix[16], ProjMatrix[16], InvProjMatrix[16];
    uint32_t FrameCount;
    float FrameDimensions[2];
    uint32_t AntiAliasingEnabled;
    float FocalLength, ApertureSize, ApertureShape[2];
    float SpecularModifier, SpecularTintModifier, MetallicModifier, RoughnessModifier, SubsurfaceModifier,
        AnisotropyModifier, SheenRoughnessModifier, SheenTintModifier, ClearcoatModifier, ClearcoatRoughnessModifier,
        RefractionModifier, TransmissionModifier;
    float DebugValueA, DebugValueB;
    uint32_t MaxBounceCount;  // pathtracing.comp:24 (constant 3 in the reference)
    uint32_t AccumulateMode;  // 0 = running mean (reference), 1 = sum
};

struct Texture {
    int w = 0, h = 0, channels = 0;
    bool srgb = false, isFloat = false;
    std::vector<uint8_t> u8;
    std::vector<float> f32;
};

struct Counters {
    uint64_t samples, raysClosest, raysAny, nodeFetches, pairTests, triangleTests, closestHits, stackOverflows;
    uint32_t maxStack, pad;
};

struct Ray {
    vec3 origin, direction;
};
struct HitInfo {
    bool didHit;
    vec3 hitPosition, hitDirection;
    float dst;
    vec2 uv;
    vec3 shadingNormal, geometryNormal, tangent, bitangent;
    uint32_t primitiveIndex;
};
struct BrdfData {
    float roughness, alpha, alphaSquared, roughnessClearcoat, alphaClearcoat, alphaSquaredClearcoat;
    float alphaAnisotropicX, alphaAnisotropicY;
    vec3 V, N, L, H, X, Y;
    float NdotV, NdotL, VdotH, NdotH, LdotH, VdotX, LdotX, HdotX, VdotY, LdotY, HdotY;
    float eta, f0, f90;
};

struct Scene {
    std::vector<BvhNode> nodes;
    std::vector<BvhPrimitive> prims;
    std::vector<MaterialRecord> materials;
    std::vector<Texture> textures;  // index = slot (1-based; slot 0 unused)
    std::vector<float> hdri, cache;  // RGB32F
    int envW = 0, envH = 0;
    float srgbLut[256];
    Scene()
    {
        for (int i = 0; i < 256; i++) {
            double c = i / 255.0;
            double l = c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
            srgbLut[i] = (float)l;
        }
    }
};

// Per-thread evaluation context (uniforms + scene + counters).
struct Ctx {
    const Scene* s;
    Params u;
    Counters c{};
    bool overflow = false;
};

// ------------------------------------------------------------------ texture() on a sampler2D
static vec4 fetchTexel(const Scene& s, const Texture& t, int x, int y)
{
    vec4 r{0, 0, 0, 1};
    float* o = &r.x;
    size_t base = ((size_t)y * t.w + x) * t.channels;
    for (int c = 0; c < t.channels; c++) {
        if (t.isFloat)
            o[c] = t.f32[base + c];
        else if (t.srgb && c < 3)
            o[c] = s.srgbLut[t.u8[base + c]];
        else
            o[c] = (float)t.u8[base + c] / 255.0f;
    }
    return r;
}
static inline int wrapi(int i, int n)
{
    int m = i % n;
    return m < 0 ? m + n : m;
}
static vec4 bilinear(const Scene& s, const Texture& t, vec2 uv)
{
    float x = uv.x * (float)t.w - 0.5f, y = uv.y * (float)t.h - 0.5f;
    float fx = std::floor(x), fy = std::floor(y);
    float a = x - fx, b = y - fy;
    int x0 = wrapi((int)fx, t.w), y0 = wrapi((int)fy, t.h);
    int x1 = wrapi(x0 + 1, t.w), y1 = wrapi(y0 + 1, t.h);
    vec4 t00 = fetchTexel(s, t, x0, y0), t10 = fetchTexel(s, t, x1, y0);
    vec4 t01 = fetchTexel(s, t, x0, y1), t11 = fetchTexel(s, t, x1, y1);
    vec4 r;
    r.x = gl_mix(gl_mix(t00.x, t10.x, a), gl_mix(t01.x, t11.x, a), b);
    r.y = gl_mix(gl_mix(t00.y, t10.y, a), gl_mix(t01.y, t11.y, a), b);
    r.z = gl_mix(gl_mix(t00.z, t10.z, a), gl_mix(t01.z, t11.z, a), b);
    r.w = gl_mix(gl_mix(t00.w, t10.w, a), gl_mix(t01.w, t11.w, a), b);
    return r;
}
// Library/common.glsl:167-175 -- the 64-bit handle is a 1-based texture slot here.
static vec4 SampleBindlessTexture(const Ctx& c, uint64_t handle, vec2 uv)
{
    if (handle >= c.s->textures.size() || c.s->textures[(size_t)handle].w <= 0) {  // a test's mistake: say so instead of reading past the table
        std::fprintf(stderr, "oracle: material refers to texture slot %llu, which was never set\n", (unsigned long long)handle);
        std::abort();
    }
    return bilinear(*c.s, c.s->textures[(size_t)handle], uv);
}
static vec3 envTexel(const std::vector<float>& img, int w, int x, int y)
{
    const float* p = &img[3 * ((size_t)y * w + x)];
    return {p[0], p[1], p[2]};
}
static vec3 envBilinear(const std::vector<float>& img, int w, int h, vec2 uv)
{
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx = std::floor(x), fy = std::floor(y);
    float a = x - fx, b = y - fy;
    int x0 = wrapi((int)fx, w), y0 = wrapi((int)fy, h);
    int x1 = wrapi(x0 + 1, w), y1 = wrapi(y0 + 1, h);
    vec3 t00 = envTexel(img, w, x0, y0), t10 = envTexel(img, w, x1, y0);
    vec3 t01 = envTexel(img, w, x0, y1), t11 = envTexel(img, w, x1, y1);
    return {gl_mix(gl_mix(t00.x, t10.x, a), gl_mix(t01.x, t11.x, a), b),
            gl_mix(gl_mix(t00.y, t10.y, a), gl_mix(t01.y, t11.y, a), b),
            gl_mix(gl_mix(t00.z, t10.z, a), gl_mix(t01.z, t11.z, a), b)};
}

// ------------------------------------------------------- Library/montecarlo.glsl
static uint32_t NextRandom(uint32_t& state)  // :8-14
{
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
static float RandomValue(uint32_t& state) { return (float)NextRandom(state) / 4294967295.0f; }  // :17-20
static vec2 RandomValueVec2(uint32_t& state)  // :23-28
{
    float R1 = RandomValue(state);
    float R2 = RandomValue(state);
    return {R1, R2};
}
static vec3 RandomValueVec3(uint32_t& state)  // :31-37
{
    float R1 = RandomValue(state);
    float R2 = RandomValue(state);
    float R3 = RandomValue(state);
    return {R1, R2, R3};
}
static vec3 HemispherepointCos(float u, float v)  // :53-59
{
    float phi = v * TWO_PI;
    float cosTheta = gl_sqrt(1.0f - u);
    float sinTheta = gl_sqrt(1.0f - cosTheta * cosTheta);
    return {gl_cos(phi) * sinTheta, gl_sin(phi) * sinTheta, cosTheta};
}
static vec3 ImportanceSampleGGX(float alpha, vec2 Xi)  // :63-76
{
    float a2 = alpha * alpha;
    float cosThetaH = gl_sqrt((1.0f - Xi.y) / (1.0f + (a2 - 1.0f) * Xi.y));
    float sinThetaH = gl_sqrt(gl_max(0.0f, 1.0f - cosThetaH * cosThetaH));
    float phiH = TWO_PI * Xi.x;
    float sinPhiH = gl_sin(phiH);
    float cosPhiH = gl_cos(phiH);
    return {sinThetaH * cosPhiH, sinThetaH * sinPhiH, cosThetaH};
}
static vec3 ImportanceSampleVisibleGGX(vec3 Ve, float ax, float ay, vec2 Xi)  // :83-106
{
    vec3 Vh = normalize(vec3{ax * Ve.x, ay * Ve.y, Ve.z});
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    vec3 T1 = lensq > 0.0f ? vec3{-Vh.y, Vh.x, 0.0f} * gl_inversesqrt(lensq) : vec3{1.0f, 0.0f, 0.0f};
    vec3 T2 = cross(Vh, T1);
    float r = gl_sqrt(Xi.x);
    float phi = TWO_PI * Xi.y;
    float t1 = r * gl_cos(phi);
    float t2 = r * gl_sin(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = gl_mix(gl_sqrt(1.0f - t1 * t1), t2, s);
    vec3 Nh = t1 * T1 + t2 * T2 + gl_sqrt(gl_max(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(vec3{ax * Nh.x, ay * Nh.y, gl_max(0.0f, Nh.z)});
}
static float MisMixWeight(float a, float b)  // :110-114
{
    float t = a * a;
    return t / (b * b + t);
}
static float RescaleRandomNumber(float randomValue, float lowerBound, float upperBound)  // :118-122
{
    const float oneMinusEpsilon = 0.99999994f;
    return gl_min((randomValue - lowerBound) / (upperBound - lowerBound), oneMinusEpsilon);
}

// ------------------------------------------------------------- Library/brdf.glsl
static float D_GGX(float alpha, float NdotH)  // :14-20
{
    float oneMinusNoHSquared = 1.0f - NdotH * NdotH;
    float a = NdotH * alpha;
    float k = alpha / (oneMinusNoHSquared + a * a);
    return k * k * ONE_OVER_PI;
}
static float D_GGX_Anisotropic(float at, float ab, float HdotX, float HdotY, float NdotH)  // :23-30
{
    float a2 = at * ab;
    vec3 d{ab * HdotX, at * HdotY, a2 * NdotH};
    float d2 = dot(d, d);
    float b2 = a2 / d2;
    return a2 * b2 * b2 * ONE_OVER_PI;
}
static float D_Charlie(float roughness, float NdotH)  // :33-39
{
    float invR = 1.0f / roughness;
    float cos2h = NdotH * NdotH;
    float sin2h = 1.0f - cos2h;
    return (2.0f + invR) * gl_pow(sin2h, invR * 0.5f) * ONE_OVER_TWO_PI;
}
static float V_Smith_G1_GGX_Anisotropic(float ax, float ay, float VdotX, float VdotY, float NdotV)  // :50-56
{
    float a = VdotX * ax;
    float b = VdotY * ay;
    float c = NdotV;
    return (2.0f * NdotV) / (NdotV + gl_sqrt(a * a + b * b + c * c));
}
static float V_Smith_G2_Correlated_GGX_Anisotropic(float ax, float ay, float VdotX, float VdotY, float LdotX,
                                                   float LdotY, float NdotV, float NdotL)  // :70-75
{
    float lambdaV = NdotL * length(vec3{ax * VdotX, ay * VdotY, NdotV});
    float lambdaL = NdotV * length(vec3{ax * LdotX, ay * LdotY, NdotL});
    return 0.5f / (lambdaV + lambdaL);
}
static float V_Kelemen(float LdotH) { return 0.25f / (LdotH * LdotH + 1e-5f); }  // :78-82
static float V_Neubelt(float NdotV, float NdotL) { return 1.0f / (4.0f * (NdotL + NdotV - NdotL * NdotV)); }  // :85-88
static float F_SchlickWeight(float u)  // :91-96
{
    float m = gl_clamp(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m * m2 * m2;
}
static float F_Schlick(float f0, float f90, float u)  // :98-102
{
    float w = F_SchlickWeight(u);
    return f0 + (f90 - f0) * w;
}
static float F_Dielectric(float cosThetaI, float incidentIor)  // :105-119
{
    float sinThetaTSq = incidentIor * incidentIor * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = gl_sqrt(gl_max(1.0f - sinThetaTSq, 0.0f));
    float rs = (incidentIor * cosThetaT - cosThetaI) / (incidentIor * cosThetaT + cosThetaI);
    float rp = (incidentIor * cosThetaI - cosThetaT) / (incidentIor * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}
static float Fd_Burley(float roughness, float NdotV, float NdotL, float LdotH)  // :127-133
{
    float f90 = 0.5f + 2.0f * roughness * LdotH * LdotH;
    float lightScatter = F_Schlick(1.0f, f90, NdotL);
    float viewScatter = F_Schlick(1.0f, f90, NdotV);
    return lightScatter * viewScatter * ONE_OVER_PI;
}
static float Fd_HanrahanKrueger(float roughness, float NdotV, float NdotL, float LdotH)  // :137-144
{
    float Fss90 = roughness * LdotH * LdotH;
    float FLss = F_SchlickWeight(NdotL);
    float FVss = F_SchlickWeight(NdotV);
    float Fss = gl_mix(1.0f, Fss90, FLss) * gl_mix(1.0f, Fss90, FVss);
    return 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f) * ONE_OVER_PI;
}

// ---------------------------------------------------------- Library/utility.glsl
static vec3 OffsetRay(vec3 P, vec3 N)  // :9-26
{
    const float origin = 1.0f / 32.0f;
    const float floatScale = 1.0f / 65536.0f;
    const float intScale = 256.0f;
    int32_t ofx = (int32_t)(intScale * N.x), ofy = (int32_t)(intScale * N.y), ofz = (int32_t)(intScale * N.z);
    auto shift = [](float p, int32_t of) {
        // int addition on the bit pattern; wrap like GLSL int arithmetic
        uint32_t b = f2bits(p) + (uint32_t)((p < 0.0f) ? -of : of);
        return bits2f(b);
    };
    vec3 p_i{shift(P.x, ofx), shift(P.y, ofy), shift(P.z, ofz)};
    return {gl_abs(P.x) < origin ? P.x + floatScale * N.x : p_i.x, gl_abs(P.y) < origin ? P.y + floatScale * N.y : p_i.y,
            gl_abs(P.z) < origin ? P.z + floatScale * N.z : p_i.z};
}
static void GetTangentBitangent(vec3 N, vec3& tangent, vec3& bitangent)  // :28-36
{
    vec3 helper = gl_abs(N.x) > 0.999f ? vec3{0.0f, 0.0f, 1.0f} : vec3{1.0f, 0.0f, 0.0f};
    tangent = normalize(cross(N, helper));
    bitangent = normalize(cross(N, tangent));
}
static vec3 ToLocal(vec3 N, vec3 V)  // :49-53, transpose(mat3(T,B,N)) * V
{
    vec3 T, B;
    GetTangentBitangent(N, T, B);
    return {dot(T, V), dot(B, V), dot(N, V)};
}
static vec3 ToWorld(vec3 N, vec3 V)  // :55-58, mat3(T,B,N) * V
{
    vec3 T, B;
    GetTangentBitangent(N, T, B);
    return T * V.x + B * V.y + N * V.z;
}
static float Luminance(vec3 rgb) { return dot(rgb, vec3{0.2126f, 0.7152f, 0.0722f}); }  // :60-63

static vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }

static vec3 SampleNormalMap(const Ctx& c, uint64_t normalTextureHandle, const HitInfo& hitInfo)  // :69-92
{
    vec3 N = hitInfo.shadingNormal;
    vec3 T = hitInfo.tangent;
    T = normalize(T - dot(T, N) * N);
    vec3 B = cross(N, T);
    vec3 normalMap = xyz(SampleBindlessTexture(c, normalTextureHandle, hitInfo.uv));
    normalMap = normalize(normalMap * 2.0f - V3(1.0f));
    return normalize(T * normalMap.x + B * normalMap.y + N * normalMap.z);
}

static Material EvaluateMaterial(const Ctx& c, const Material& material, HitInfo& hitInfo)  // :94-169
{
    Material e = material;
    const Params& u = c.u;
    if (material.handle[0] != 0) e.emission *= xyz(SampleBindlessTexture(c, material.handle[0], hitInfo.uv));
    if (material.handle[1] != 0) e.albedo *= xyz(SampleBindlessTexture(c, material.handle[1], hitInfo.uv));
    if (material.handle[2] != 0) hitInfo.shadingNormal = SampleNormalMap(c, material.handle[2], hitInfo);
    if (material.handle[3] != 0) e.specular *= SampleBindlessTexture(c, material.handle[3], hitInfo.uv).w;
    if (material.handle[4] != 0) e.specularTint *= length(xyz(SampleBindlessTexture(c, material.handle[4], hitInfo.uv)));
    if (material.handle[5] != 0) {
        vec4 s = SampleBindlessTexture(c, material.handle[5], hitInfo.uv);
        e.metallic *= s.x;
        e.roughness *= s.y;
    }
    if (material.handle[6] != 0) e.sheenRoughness *= SampleBindlessTexture(c, material.handle[6], hitInfo.uv).w;
    if (material.handle[7] != 0) e.sheenTint *= length(xyz(SampleBindlessTexture(c, material.handle[7], hitInfo.uv)));
    if (material.handle[8] != 0) e.clearcoat *= SampleBindlessTexture(c, material.handle[8], hitInfo.uv).x;
    if (material.handle[9] != 0) e.clearcoatRoughness *= SampleBindlessTexture(c, material.handle[9], hitInfo.uv).y;
    if (material.handle[10] != 0) e.transmission *= SampleBindlessTexture(c, material.handle[10], hitInfo.uv).x;

    e.specular = gl_clamp(e.specular + u.SpecularModifier, 0.0f, 1.0f);
    e.specularTint = gl_clamp(e.specularTint + u.SpecularTintModifier, 0.0f, 1.0f);
    e.metallic = gl_clamp(e.metallic + u.MetallicModifier, 0.0f, 1.0f);
    e.roughness = gl_clamp(e.roughness + u.RoughnessModifier, 0.0f, 1.0f);
    e.subsurface = gl_clamp(e.subsurface + u.SubsurfaceModifier, 0.0f, 1.0f);
    e.anisotropy = gl_clamp(e.anisotropy + u.AnisotropyModifier, 0.0f, 1.0f);
    e.sheenRoughness = gl_clamp(e.sheenRoughness + u.SheenRoughnessModifier, 0.0f, 1.0f);
    e.sheenTint = gl_clamp(e.sheenTint + u.SheenTintModifier, 0.0f, 1.0f);
    e.clearcoat = gl_clamp(e.clearcoat + u.ClearcoatModifier, 0.0f, 1.0f);
    e.clearcoatRoughness = gl_clamp(e.clearcoatRoughness + u.ClearcoatRoughnessModifier, 0.0f, 1.0f);
    e.refraction = gl_clamp(e.refraction + u.RefractionModifier, 1.01f, 2.0f);
    e.transmission = gl_clamp(e.transmission + u.TransmissionModifier, 0.0f, 1.0f);
    return e;
}

static Material GetEvaluatedMaterial(const Ctx& c, HitInfo& hitInfo)  // :171-181
{
    const MaterialRecord& r = c.s->materials[hitInfo.primitiveIndex];
    Material m;
    std::memcpy(m.handle, r.handle, sizeof(m.handle));
    m.emission = {r.emission[0], r.emission[1], r.emission[2]};
    m.albedo = {r.albedo[0], r.albedo[1], r.albedo[2]};
    m.specular = r.specular;
    m.specularTint = r.specularTint;
    m.metallic = r.metallic;
    m.roughness = r.roughness;
    m.subsurface = r.subsurface;
    m.anisotropy = r.anisotropy;
    m.sheenRoughness = r.sheenRoughness;
    m.sheenTint = r.sheenTint;
    m.clearcoat = r.clearcoat;
    m.clearcoatRoughness = r.clearcoatRoughness;
    m.refraction = r.refraction;
    m.transmission = r.transmission;
    return EvaluateMaterial(c, m, hitInfo);
}

static float IorToF0(float transmittedIor, float incidentIor)  // :187-191
{
    float r = (transmittedIor - incidentIor) / (transmittedIor + incidentIor);
    return r * r;
}

static BrdfData PrepareEvaluationBrdfData(const Material& material, vec3 V, vec3 N, vec3 L)  // :199-262
{
    BrdfData data;
    vec3 X, Y;
    GetTangentBitangent(N, X, Y);
    data.roughness = gl_clamp(material.roughness, 0.045f, 1.0f);
    data.alpha = data.roughness * data.roughness;
    data.alphaSquared = data.alpha * data.alpha;
    data.roughnessClearcoat = gl_clamp(material.clearcoatRoughness, 0.045f, 1.0f);
    data.alphaClearcoat = data.roughnessClearcoat * data.roughnessClearcoat;
    data.alphaSquaredClearcoat = data.alphaClearcoat * data.alphaClearcoat;
    float aspect = gl_sqrt(1.0f - 0.9f * material.anisotropy);
    data.alphaAnisotropicX = gl_max(0.002025f, data.alpha / aspect);
    data.alphaAnisotropicY = gl_max(0.002025f, data.alpha * aspect);
    data.V = V;
    data.N = N;
    data.L = L;
    float NdotV = dot(data.N, data.V);
    float NdotL = dot(data.N, data.L);
    data.eta = NdotV <= 0.0f ? material.refraction : (1.0f / material.refraction);
    data.H = NdotL <= 0.0f ? normalize(L + V * data.eta) : normalize(L + V);
    float NdotH = dot(N, data.H);
    if (NdotH < 0.0f) data.H = -data.H;
    data.X = X;
    data.Y = Y;
    data.NdotV = NdotV;
    data.NdotL = NdotL;
    data.VdotH = dot(data.V, data.H);
    data.NdotH = dot(data.N, data.H);
    data.LdotH = dot(data.L, data.H);
    data.VdotX = dot(data.V, data.X);
    data.LdotX = dot(data.L, data.X);
    data.HdotX = dot(data.H, data.X);
    data.VdotY = dot(data.V, data.Y);
    data.LdotY = dot(data.L, data.Y);
    data.HdotY = dot(data.H, data.Y);
    data.f0 = IorToF0(1.0f, data.eta);
    data.f90 = gl_clamp(dot(V3(data.f0), V3(50.0f * 0.33f)), 0.0f, 1.0f);
    return data;
}

// ------------------------------------------------------------ Library/disney.glsl
static vec3 CalculateTint(const Material& material)  // :6-10
{
    float luminance = Luminance(material.albedo);
    return luminance > 0.0f ? material.albedo / luminance : V3(1.0f);
}
static void TintColors(const Material& material, const BrdfData& brdfData, vec3& Csheen, vec3& Cspec0)  // :12-18
{
    vec3 Ctint = CalculateTint(material);
    Cspec0 = brdfData.f0 * mix(V3(1.0f), Ctint, material.specularTint) * material.specular;
    Csheen = mix(V3(1.0f), Ctint, material.sheenTint);
}
static vec3 EvaluateDisneyDiffuse(const Material& material, const BrdfData& b, float& pdf)  // :20-36
{
    pdf = 0.0f;
    if (b.NdotL <= 0.0f) return V3(0.0f);
    float Fd = Fd_Burley(b.roughness, b.NdotV, b.NdotL, b.LdotH);
    float Fdss = Fd_HanrahanKrueger(b.roughness, b.NdotV, b.NdotL, b.LdotH);
    pdf = b.NdotL * ONE_OVER_PI;
    return gl_mix(Fd, Fdss, material.subsurface) * material.albedo;
}
static vec3 EvaluateMicrofacetReflection(const BrdfData& b, vec3 F, float& pdf)  // :38-52
{
    pdf = 0.0f;
    if (b.NdotL <= 0.0f) return V3(0.0f);
    float D = D_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.HdotX, b.HdotY, b.NdotH);
    float V1 = V_Smith_G1_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.VdotX, b.VdotY, gl_abs(b.NdotV));
    float V2 = V_Smith_G2_Correlated_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.VdotX, b.VdotY, b.LdotX,
                                                     b.LdotY, gl_abs(b.NdotV), gl_abs(b.NdotL));
    pdf = D * V1 / (4.0f * b.NdotV);
    return F * D * V2;
}
static vec3 EvaluateMicrofacetRefraction(const Material& material, const BrdfData& b, vec3 F, float& pdf)  // :54-73
{
    pdf = 0.0f;
    if (b.NdotL >= 0.0f) return V3(0.0f);
    float D = D_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.HdotX, b.HdotY, b.NdotH);
    float V1 = V_Smith_G1_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.VdotX, b.VdotY, gl_abs(b.NdotV));
    float V2 = V_Smith_G2_Correlated_GGX_Anisotropic(b.alphaAnisotropicX, b.alphaAnisotropicY, b.VdotX, b.VdotY, b.LdotX,
                                                     b.LdotY, gl_abs(b.NdotV), gl_abs(b.NdotL));
    float denom = b.LdotH + b.VdotH * b.eta;
    denom *= denom;
    float eta2 = b.eta * b.eta;
    float jacobian = gl_abs(b.LdotH) / denom;
    pdf = D * V1 * gl_max(0.0f, b.VdotH) * jacobian / b.NdotV;
    return pow3(material.albedo, V3(0.5f)) * (V3(1.0f) - F) * D * V2 * gl_abs(b.VdotH) * jacobian * eta2 * 4.0f;
}
static vec3 EvaluateSheen(const Material& material, const BrdfData& b, vec3 Csheen, float& pdf)  // :75-88
{
    pdf = 0.0f;
    if (b.NdotL <= 0.0f) return V3(0.0f);
    float D = D_Charlie(material.sheenRoughness, gl_clamp(b.NdotH, 0.0f, 1.0f));
    float V = V_Neubelt(b.NdotV, b.NdotL);
    pdf = b.NdotL * ONE_OVER_PI;
    return D * V * Csheen;
}
static vec3 EvaluateClearcoat(const BrdfData& b, float& pdf)  // :90-104
{
    pdf = 0.0f;
    if (b.NdotL <= 0.0f) return V3(0.0f);
    float D = D_GGX(b.alphaClearcoat, b.NdotH);
    float V = V_Kelemen(b.LdotH);
    float F = F_Schlick(MIN_DIELECTRICS_F0, 1.0f, b.LdotH);
    pdf = D * b.NdotH / (4.0f * b.LdotH);
    return V3(D) * V * F;
}

struct Lobes {
    float dielectricWeight, metalWeight, glassWeight, sheenWeight, clearcoatWeight;
    float diffuseProbability, dielectricProbability, metalProbability, glassProbability, sheenProbability,
        clearcoatProbability;
    vec3 Csheen, Cspec0;
};
// The block shared verbatim by SampleDisneyBrdf (:112-142) and EvaluateDisneyBrdf (:248-278).
static Lobes LobeWeights(const Material& material, const BrdfData& brdfData)
{
    Lobes l;
    TintColors(material, brdfData, l.Csheen, l.Cspec0);
    float clearcoatEnergyLoss = 1.0f - (F_Schlick(MIN_DIELECTRICS_F0, 1.0f, brdfData.LdotH) * material.clearcoat);
    l.dielectricWeight = (1.0f - material.metallic) * (1.0f - material.transmission) * clearcoatEnergyLoss;
    l.metalWeight = material.metallic * clearcoatEnergyLoss;
    l.glassWeight = (1.0f - material.metallic) * material.transmission;
    l.sheenWeight = material.sheenRoughness;
    l.clearcoatWeight = material.clearcoat;
    float schlickWeight = F_SchlickWeight(brdfData.NdotV);
    l.diffuseProbability = l.dielectricWeight * Luminance(material.albedo);
    l.dielectricProbability = l.dielectricWeight * Luminance(mix(l.Cspec0, V3(1.0f), schlickWeight));
    l.metalProbability = l.metalWeight * Luminance(mix(material.albedo, V3(1.0f), schlickWeight));
    l.glassProbability = l.glassWeight;
    l.sheenProbability = l.sheenWeight;
    l.clearcoatProbability = l.clearcoatWeight;
    float invTotalWeight = 1.0f / (l.diffuseProbability + l.dielectricProbability + l.metalProbability +
                                   l.glassProbability + l.sheenProbability + l.clearcoatProbability);
    l.diffuseProbability *= invTotalWeight;
    l.dielectricProbability *= invTotalWeight;
    l.metalProbability *= invTotalWeight;
    l.glassProbability *= invTotalWeight;
    l.sheenProbability *= invTotalWeight;
    l.clearcoatProbability *= invTotalWeight;
    return l;
}

static vec3 SampleDisneyBrdf(const Material& material, const BrdfData& brdfData, uint32_t& rngState)  // :110-236
{
    Lobes l = LobeWeights(material, brdfData);
    float cdf[6];
    cdf[0] = l.diffuseProbability;
    cdf[1] = cdf[0] + l.dielectricProbability;
    cdf[2] = cdf[1] + l.metalProbability;
    cdf[3] = cdf[2] + l.glassProbability;
    cdf[4] = cdf[3] + l.sheenProbability;
    cdf[5] = cdf[4] + l.clearcoatProbability;
    vec3 Xi = RandomValueVec3(rngState);
    float rd = Xi.z;
    if (rd < cdf[0]) {
        vec3 H = HemispherepointCos(Xi.x, Xi.y);
        H = ToWorld(brdfData.N, H);
        return H;
    } else if (rd < cdf[2]) {
        vec3 tangentV = ToLocal(brdfData.N, brdfData.V);
        vec3 H = ImportanceSampleVisibleGGX(tangentV, brdfData.alphaAnisotropicX, brdfData.alphaAnisotropicY, {Xi.x, Xi.y});
        if (H.z < 0.0f) H = -H;
        H = ToWorld(brdfData.N, H);
        return normalize(reflect(-brdfData.V, H));
    } else if (rd < cdf[3]) {
        vec3 tangentV = ToLocal(brdfData.N, brdfData.V);
        vec3 H = ImportanceSampleVisibleGGX(tangentV, brdfData.alphaAnisotropicX, brdfData.alphaAnisotropicY, {Xi.x, Xi.y});
        if (H.z < 0.0f) H = -H;
        H = ToWorld(brdfData.N, H);
        float F = F_Dielectric(gl_abs(dot(brdfData.V, H)), brdfData.eta);
        rd = RescaleRandomNumber(rd, cdf[2], cdf[3]);
        if (rd < F)
            return normalize(reflect(-brdfData.V, H));
        else
            return normalize(refract(-brdfData.V, H, brdfData.eta));
    } else if (rd < cdf[4]) {
        vec3 H = HemispherepointCos(Xi.x, Xi.y);
        H = ToWorld(brdfData.N, H);
        return H;
    } else {
        vec3 H = ImportanceSampleGGX(brdfData.alphaClearcoat, {Xi.x, Xi.y});
        if (H.z < 0.0f) H = -H;
        H = ToWorld(brdfData.N, H);
        return normalize(reflect(-brdfData.V, H));
    }
}

static vec3 EvaluateDisneyBrdf(const Material& material, const BrdfData& brdfData, float& pdf)  // :243-346
{
    vec3 f = V3(0.0f);
    pdf = 0.0f;
    Lobes l = LobeWeights(material, brdfData);
    bool reflection = brdfData.NdotL * brdfData.NdotV > 0.0f;
    float tmpPdf = 0.0f;
    float VdotH = gl_abs(brdfData.VdotH);
    if (l.diffuseProbability > 0.0f && reflection) {
        f += EvaluateDisneyDiffuse(material, brdfData, tmpPdf) * l.dielectricWeight;
        pdf += tmpPdf * l.diffuseProbability;
    }
    if (l.dielectricProbability > 0.0f && reflection) {
        float F = (F_Dielectric(VdotH, 1.0f / material.refraction) - brdfData.f0) / (1.0f - brdfData.f0);
        f += EvaluateMicrofacetReflection(brdfData, mix(l.Cspec0, V3(1.0f), F), tmpPdf) * l.dielectricWeight;
        pdf += tmpPdf * l.dielectricProbability;
    }
    if (l.metalProbability > 0.0f && reflection) {
        vec3 F = mix(material.albedo, V3(1.0f), F_SchlickWeight(VdotH));
        f += EvaluateMicrofacetReflection(brdfData, F, tmpPdf) * l.metalWeight;
        pdf += tmpPdf * l.metalProbability;
    }
    if (l.glassProbability > 0.0f) {
        float F = F_Dielectric(VdotH, brdfData.eta);
        if (reflection) {
            f += EvaluateMicrofacetReflection(brdfData, V3(F), tmpPdf) * l.glassWeight;
            pdf += tmpPdf * l.glassProbability * F;
        } else {
            f += EvaluateMicrofacetRefraction(material, brdfData, V3(F), tmpPdf) * l.glassWeight;
            pdf += tmpPdf * l.glassProbability * (1.0f - F);
        }
    }
    if (l.sheenProbability > 0.0f && reflection) {
        f += EvaluateSheen(material, brdfData, l.Csheen, tmpPdf) * l.sheenWeight;
        pdf += tmpPdf * l.sheenProbability;
    }
    if (l.clearcoatProbability > 0.0f && reflection) {
        f += EvaluateClearcoat(brdfData, tmpPdf) * l.clearcoatWeight;
        pdf += tmpPdf * l.clearcoatProbability;
    }
    return f * gl_abs(brdfData.NdotL);
}

// --------------------------------------------------------------- Library/hdri.glsl
static vec2 GetSphericalCoordinates(vec3 V)  // :6-12
{
    float u = gl_atan2(V.z, V.x) * ONE_OVER_TWO_PI + 0.5f;
    float v = 0.5f - gl_asin(V.y) * ONE_OVER_PI;
    return {u, v};
}
static vec3 SampleHdri(const Ctx& c, uint32_t& rngState)  // :15-31
{
    vec2 Xi = RandomValueVec2(rngState);
    vec3 t = envBilinear(c.s->cache, c.s->envW, c.s->envH, Xi);
    vec2 uv{t.x, t.y};
    uv.y = 1.0f - uv.y;
    float phi = TWO_PI * (uv.x - 0.5f);
    float theta = PI * (uv.y - 0.5f);
    return {gl_cos(theta) * gl_cos(phi), gl_sin(theta), gl_cos(theta) * gl_sin(phi)};
}
static vec3 EvaluateHdri(const Ctx& c, vec3 L, float& pdf)  // :34-54
{
    vec2 uv = GetSphericalCoordinates(normalize(L));
    vec3 color = envBilinear(c.s->hdri, c.s->envW, c.s->envH, uv);
    pdf = envBilinear(c.s->cache, c.s->envW, c.s->envH, uv).z;
    float theta = PI * (1.0f - uv.y);
    float sinTheta = gl_max(gl_sin(theta), 1e-10f);
    float pConvert = (float)c.s->envW * (float)c.s->envH / (TWO_PI * PI * sinTheta);
    pdf *= pConvert;
    return color;
}

// -------------------------------------------------------- Library/intersection.glsl
static HitInfo RayTriangle(Ray ray, const BvhPrimitive& primitive)  // :9-60
{
    vec3 posA{primitive.posAuvX[0], primitive.posAuvX[1], primitive.posAuvX[2]};
    vec3 posB{primitive.posBuvX[0], primitive.posBuvX[1], primitive.posBuvX[2]};
    vec3 posC{primitive.posCuvX[0], primitive.posCuvX[1], primitive.posCuvX[2]};
    vec3 norA{primitive.norAuvY[0], primitive.norAuvY[1], primitive.norAuvY[2]};
    vec3 norB{primitive.norBuvY[0], primitive.norBuvY[1], primitive.norBuvY[2]};
    vec3 norC{primitive.norCuvY[0], primitive.norCuvY[1], primitive.norCuvY[2]};
    vec2 uvA{primitive.posAuvX[3], primitive.norAuvY[3]};
    vec2 uvB{primitive.posBuvX[3], primitive.norBuvY[3]};
    vec2 uvC{primitive.posCuvX[3], primitive.norCuvY[3]};

    vec3 edgeAB = posB - posA;
    vec3 edgeAC = posC - posA;
    vec3 normalVector = cross(edgeAB, edgeAC);
    vec3 ao = ray.origin - posA;
    vec3 dao = cross(ao, ray.direction);

    float determinant = -dot(ray.direction, normalVector);
    float invDet = 1.0f / determinant;

    float dst = dot(ao, normalVector) * invDet;
    float u = dot(edgeAC, dao) * invDet;
    float v = -dot(edgeAB, dao) * invDet;
    float w = 1.0f - u - v;

    HitInfo hitInfo;
    hitInfo.didHit = determinant >= 1e-10f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
    hitInfo.hitPosition = ray.origin + ray.direction * dst;
    hitInfo.hitDirection = ray.direction;
    hitInfo.dst = dst;
    hitInfo.uv = uvA * w + uvB * u + uvC * v;
    hitInfo.shadingNormal = normalize(norA * w + norB * u + norC * v);
    hitInfo.geometryNormal = normalize(cross(edgeAC, edgeAB));

    vec2 deltaUVB = uvB - uvA;
    vec2 deltaUVC = uvC - uvA;
    float invTangentDeterminant = 1.0f / (deltaUVB.x * deltaUVC.y - deltaUVB.y * deltaUVC.x);
    hitInfo.tangent = (edgeAB * deltaUVC.y - edgeAC * deltaUVB.y) * invTangentDeterminant;
    hitInfo.bitangent = (edgeAC * deltaUVB.x - edgeAB * deltaUVC.x) * invTangentDeterminant;
    hitInfo.primitiveIndex = 0;
    return hitInfo;
}

static float HitAABB(Ray r, vec3 AA, vec3 BB)  // :63-77
{
    vec3 invdir = 1.0f / r.direction;
    vec3 f = (BB - r.origin) * invdir;
    vec3 n = (AA - r.origin) * invdir;
    vec3 tmax = vmax(f, n);
    vec3 tmin = vmin(f, n);
    float t1 = gl_min(tmax.x, gl_min(tmax.y, tmax.z));
    float t0 = gl_max(tmin.x, gl_max(tmin.y, tmin.z));
    return (t1 >= t0) ? ((t0 > 0.0f) ? (t0) : (t1)) : (-1.0f);
}

static HitInfo HitArrayClosest(Ctx& c, Ray ray, int L, int R)  // :83-112
{
    HitInfo closestHit{};
    closestHit.didHit = false;
    closestHit.dst = FLT_MAX_;
    uint32_t closestPrimitiveIndex = 0;
    for (int i = L; i <= R; i++) {
        const BvhPrimitive& primitive = c.s->prims[i];
        HitInfo hitInfo = RayTriangle(ray, primitive);
        c.c.triangleTests++;
        if (hitInfo.didHit && hitInfo.dst < closestHit.dst) {
            closestHit = hitInfo;
            closestPrimitiveIndex = primitive.meshIndex;
        }
    }
    if (closestHit.didHit) closestHit.primitiveIndex = closestPrimitiveIndex;
    return closestHit;
}

static HitInfo HitArrayAny(Ctx& c, Ray ray, int L, int R)  // :205-226
{
    HitInfo anyHit{};
    anyHit.didHit = false;
    anyHit.dst = FLT_MAX_;
    for (int i = L; i <= R; i++) {
        const BvhPrimitive& primitive = c.s->prims[i];
        HitInfo hitInfo = RayTriangle(ray, primitive);
        c.c.triangleTests++;
        if (hitInfo.didHit) {
            anyHit = hitInfo;
            break;
        }
    }
    return anyHit;
}

// HitBvhClosest (:115-199) and HitBvhAny (:230-315) share the traversal; `any` selects the
// leaf routine and the early exit.
static HitInfo HitBvh(Ctx& c, Ray ray, bool any)
{
    HitInfo best{};
    best.didHit = false;
    best.dst = FLT_MAX_;
    const std::vector<BvhNode>& bvhNodes = c.s->nodes;

    int stack[BVH_STACKSIZE + STACK_SPARE];
    int stackPointer = 0;
    stack[stackPointer++] = 1;
    if (any) c.c.raysAny++; else c.c.raysClosest++;
    while (stackPointer > 0) {
        int top = stack[--stackPointer];
        const BvhNode& node = bvhNodes[top];
        c.c.nodeFetches++;
        if (node.n > 0) {
            int L = node.index;
            int R = node.index + node.n - 1;
            if (any) {
                HitInfo hitInfo = HitArrayAny(c, ray, L, R);
                if (hitInfo.didHit) {
                    best = hitInfo;
                    break;
                }
            } else {
                HitInfo hitInfo = HitArrayClosest(c, ray, L, R);
                if (hitInfo.didHit && hitInfo.dst < best.dst) best = hitInfo;
            }
        } else {
            if (stackPointer < BVH_STACKSIZE) {
                float dstLeft = FLT_MAX_;
                float dstRight = FLT_MAX_;
                c.c.pairTests++;
                if (node.left > 0) {
                    const BvhNode& leftNode = bvhNodes[node.left];
                    c.c.nodeFetches++;
                    dstLeft = HitAABB(ray, {leftNode.AA[0], leftNode.AA[1], leftNode.AA[2]}, {leftNode.BB[0], leftNode.BB[1], leftNode.BB[2]});
                }
                if (node.right > 0) {
                    const BvhNode& rightNode = bvhNodes[node.right];
                    c.c.nodeFetches++;
                    dstRight = HitAABB(ray, {rightNode.AA[0], rightNode.AA[1], rightNode.AA[2]}, {rightNode.BB[0], rightNode.BB[1], rightNode.BB[2]});
                }
                if (dstLeft > 0.0f && dstRight > 0.0f) {
                    if (stackPointer + 2 > BVH_STACKSIZE) {
                        c.overflow = true;  // the shader would write stack[16]
                        c.c.stackOverflows++;
                    }
                    if (dstLeft < dstRight) {
                        stack[stackPointer++] = node.right;
                        stack[stackPointer++] = node.left;
                    } else {
                        stack[stackPointer++] = node.left;
                        stack[stackPointer++] = node.right;
                    }
                } else if (dstLeft > 0.0f) {
                    stack[stackPointer++] = node.left;
                } else if (dstRight > 0.0f) {
                    stack[stackPointer++] = node.right;
                }
                if ((uint32_t)stackPointer > c.c.maxStack) c.c.maxStack = (uint32_t)stackPointer;
                if (stackPointer >= BVH_STACKSIZE + STACK_SPARE - 2) break;  // cannot happen for a tree this shallow
            } else {
                c.overflow = true;  // traversal aborts here in the shader (:191-194)
                c.c.stackOverflows++;
                break;
            }
        }
    }
    if (!any && best.didHit) c.c.closestHits++;
    return best;
}

// ------------------------------------------------------------------ Library/debug.glsl
// The two views the shader writes instead of a frame when it is compiled with DEBUG_ENABLED
// (pathtracing.comp:216-223): both are pure functions of the pixel.
static vec4 DebugHdriCache(const Ctx& c, vec2 uv)  // :7-28
{
    vec3 hdriColor = envBilinear(c.s->hdri, c.s->envW, c.s->envH, vec2{uv.x, -uv.y});
    for (uint32_t i = 0u; i < 500u; i++) {
        uint32_t XiState = i;
        vec2 Xi = RandomValueVec2(XiState);
        vec3 t = envBilinear(c.s->cache, c.s->envW, c.s->envH, Xi);
        vec2 texcoord{t.x, t.y};
        texcoord.y = 1.0f - texcoord.y;
        vec2 d = uv - texcoord;
        if (gl_sqrt(d.x * d.x + d.y * d.y) < 0.005f) hdriColor = {1.0f, 0.0f, 0.0f};  // distance(uv, texcoord)
    }
    return {hdriColor.x, hdriColor.y, hdriColor.z, 1.0f};
}
static vec4 DebugBvh(const Ctx& c, Ray ray)  // :31-126
{
    uint32_t hitCounter = 0;
    const std::vector<BvhNode>& bvhNodes = c.s->nodes;
    int stack[BVH_STACKSIZE + STACK_SPARE];  // (the shader's seventeenth push lands in the spare part, as in HitBvh)
    int stackPointer = 0;
    stack[stackPointer++] = 1;
    while (stackPointer > 0) {
        int top = stack[--stackPointer];
        const BvhNode& node = bvhNodes[top];
        if (node.n > 0) {
            hitCounter++;
        } else {
            if (stackPointer < BVH_STACKSIZE) {
                float dstLeft = FLT_MAX_;
                float dstRight = FLT_MAX_;
                if (node.left > 0) {
                    const BvhNode& leftNode = bvhNodes[node.left];
                    dstLeft = HitAABB(ray, {leftNode.AA[0], leftNode.AA[1], leftNode.AA[2]}, {leftNode.BB[0], leftNode.BB[1], leftNode.BB[2]});
                }
                if (node.right > 0) {
                    const BvhNode& rightNode = bvhNodes[node.right];
                    dstRight = HitAABB(ray, {rightNode.AA[0], rightNode.AA[1], rightNode.AA[2]}, {rightNode.BB[0], rightNode.BB[1], rightNode.BB[2]});
                }
                if (dstLeft > 0.0f && dstRight > 0.0f) {
                    if (dstLeft < dstRight) {
                        stack[stackPointer++] = node.right;
                        stack[stackPointer++] = node.left;
                    } else {
                        stack[stackPointer++] = node.left;
                        stack[stackPointer++] = node.right;
                    }
                } else if (dstLeft > 0.0f) {
                    stack[stackPointer++] = node.left;
                } else if (dstRight > 0.0f) {
                    stack[stackPointer++] = node.right;
                }
            } else {
                hitCounter = 0xFFFFFFFFu;
                break;
            }
        }
    }
    if (hitCounter == 0xFFFFFFFFu) return {1.0f, 0.0f, 1.0f, 1.0f};  // stack overflow (purple)
    const vec3 mapTex[] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 1.0f, 1.0f},
                           {0.0f, 1.0f, 0.0f}, {1.0f, 1.0f, 0.0f}, {1.0f, 0.0f, 0.0f}};
    const float mapTexLen = 5.0f;
    const float maxHeat = 100.0f;
    float l = gl_clamp((float)hitCounter / maxHeat, 0.0f, 1.0f) * mapTexLen;
    vec3 a = mapTex[(uint32_t)std::floor(l)];
    vec3 b = mapTex[(uint32_t)std::ceil(l)];
    float t = l - std::floor(l);
    return {gl_mix(a.x, b.x, t), gl_mix(a.y, b.y, t), gl_mix(a.z, b.z, t), 1.0f};
}

// ---------------------------------------------------------------- pathtracing.comp
static void PathTrace(Ctx& c, Ray ray, vec3& radianceResult, vec3& primaryAlbedoResult, vec3& primaryNormalResult,
                      uint32_t& rngState)  // :16-143
{
    vec3 radiance = V3(0.0f);
    vec3 throughput = V3(1.0f);
    vec3 f = V3(1.0f);
    float pdfBRDF = 1.0f;
    const uint32_t MaxBounceCount = c.u.MaxBounceCount;
    for (uint32_t bounce = 0; bounce < MaxBounceCount; bounce++) {
        HitInfo hitInfo = HitBvh(c, ray, false);
        if (!hitInfo.didHit) {
            float pdfLight = 0.0f;
            vec3 colorLight = EvaluateHdri(c, ray.direction, pdfLight);
            float misWeight = 1.0f;
            if (bounce > 0u) misWeight = MisMixWeight(pdfBRDF, pdfLight);
            if (misWeight > 0.0f) radiance += misWeight * throughput * colorLight * f / pdfBRDF;
            break;
        }
        Material evaluatedMaterial = GetEvaluatedMaterial(c, hitInfo);
        vec3 V = -hitInfo.hitDirection;
        vec3 N = hitInfo.shadingNormal;
        vec3 emission = evaluatedMaterial.emission;
        radiance += throughput * emission;

        Ray hdriRay;
        hdriRay.origin = OffsetRay(hitInfo.hitPosition, hitInfo.geometryNormal);
        hdriRay.direction = SampleHdri(c, rngState);
        if (dot(N, hdriRay.direction) > 0.0f) {
            HitInfo hdriHitInfo = HitBvh(c, hdriRay, true);
            if (!hdriHitInfo.didHit) {
                vec3 L = hdriRay.direction;
                float pdfLight = 0.0f;
                vec3 colorLight = EvaluateHdri(c, L, pdfLight);
                BrdfData evaluationBrdfData = PrepareEvaluationBrdfData(evaluatedMaterial, V, N, L);
                pdfBRDF = 0.0f;
                f = EvaluateDisneyBrdf(evaluatedMaterial, evaluationBrdfData, pdfBRDF);
                if (pdfBRDF > 0.0f) {
                    float misWeight = MisMixWeight(pdfLight, pdfBRDF);
                    if (misWeight > 0.0f) radiance += misWeight * throughput * colorLight * f / pdfLight;
                }
            }
        }
        BrdfData samplingBrdfData = PrepareEvaluationBrdfData(evaluatedMaterial, V, N, V3(0.0f));
        vec3 L = SampleDisneyBrdf(evaluatedMaterial, samplingBrdfData, rngState);
        BrdfData evaluationBrdfData = PrepareEvaluationBrdfData(evaluatedMaterial, V, N, L);
        pdfBRDF = 0.0f;
        f = EvaluateDisneyBrdf(evaluatedMaterial, evaluationBrdfData, pdfBRDF);
        if (pdfBRDF <= 0.0f) break;  // NaN pdf continues, as `NaN <= 0` is false in GLSL too

        float p = gl_max(throughput.x, gl_max(throughput.y, throughput.z));
        if (RandomValue(rngState) >= p) break;
        throughput *= 1.0f / p;
        throughput *= f / pdfBRDF;

        ray.origin = OffsetRay(hitInfo.hitPosition, hitInfo.geometryNormal);
        ray.direction = L;
        if (hitInfo.didHit && bounce == 0) {
            primaryAlbedoResult = evaluatedMaterial.albedo;
            primaryNormalResult = hitInfo.shadingNormal;
        }
    }
    radianceResult = radiance;
}

static Ray GeneratePinholeCameraRay(const Ctx& c, const mat4& invProj, const mat4& invView, vec2 uv)  // :150-163
{
    (void)c;
    vec4 viewPos = mul(invProj, vec4{uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f, 0.0f, 1.0f});
    vec3 origin = vec3{viewPos.x, viewPos.y, viewPos.z} / viewPos.w;
    vec3 direction = normalize(origin);
    Ray ray;
    vec4 o = mul(invView, vec4{origin.x, origin.y, origin.z, 1.0f});
    vec4 d = mul(invView, vec4{direction.x, direction.y, direction.z, 0.0f});
    ray.origin = {o.x, o.y, o.z};
    ray.direction = {d.x, d.y, d.z};
    return ray;
}
static Ray GenerateThinLensCameraRay(const Ctx& c, const mat4& invProj, const mat4& invView, vec2 uv,
                                     uint32_t& rngState)  // :166-187
{
    const float* VM = c.u.ViewMatrix;  // ViewMatrix[col][row] = VM[col*4 + row]
    vec3 cameraForward = -vec3{VM[2 * 4 + 0], VM[2 * 4 + 1], VM[2 * 4 + 2]};
    vec3 cameraUp{VM[1 * 4 + 0], VM[1 * 4 + 1], VM[1 * 4 + 2]};
    vec3 cameraRight = cross(cameraUp, cameraForward);
    Ray ray = GeneratePinholeCameraRay(c, invProj, invView, uv);
    vec3 focalPoint = ray.origin + ray.direction * c.u.FocalLength;
    float r1 = RandomValue(rngState);  // argument order: left to right
    float r2 = RandomValue(rngState);
    vec3 h = HemispherepointCos(r1, r2);
    vec2 apertureSample = vec2{h.x, h.y} * vec2{c.u.ApertureShape[0], c.u.ApertureShape[1]} * c.u.ApertureSize;
    ray.origin = ray.origin + cameraRight * apertureSample.x + cameraUp * apertureSample.y;
    ray.direction = normalize(focalPoint - ray.origin);
    return ray;
}

struct Images {
    float *radiance, *albedo, *normal;  // RGBA32F, width*height*4
    uint8_t* overflowMask;              // width*height or null
};

// main() (:206-266) for one pixel and one FrameCount value.
static void ShadePixel(Ctx& c, const mat4& invProj, const mat4& invView, int px, int py, uint32_t FrameCount,
                       const Images& img)
{
    const Params& u = c.u;
    vec2 FrameDimensions{u.FrameDimensions[0], u.FrameDimensions[1]};
    vec2 uv = vec2{(float)px + 0.5f, (float)py + 0.5f} / FrameDimensions;
    uint32_t pixelIndex = (uint32_t)py * (uint32_t)FrameDimensions.x + (uint32_t)px;
    uint32_t rngState = pixelIndex + FrameCount * 719393u;
    if (u.AntiAliasingEnabled != 0u) {
        vec2 offset = RandomValueVec2(rngState);
        uv = uv + (offset - vec2{0.5f, 0.5f}) / FrameDimensions;
    }
    Ray ray = (u.ApertureSize == 0.0f) ? GeneratePinholeCameraRay(c, invProj, invView, uv)
                                       : GenerateThinLensCameraRay(c, invProj, invView, uv, rngState);
    vec3 radianceResult = V3(0.0f), primaryAlbedoResult = V3(0.0f), primaryNormalResult = V3(0.0f);
    c.overflow = false;
    c.c.samples++;
    PathTrace(c, ray, radianceResult, primaryAlbedoResult, primaryNormalResult, rngState);

    size_t o = ((size_t)py * (size_t)FrameDimensions.x + (size_t)px) * 4;
    if (img.overflowMask && c.overflow) img.overflowMask[o / 4] = 1;
    const vec3 res[3] = {radianceResult, primaryAlbedoResult, primaryNormalResult};
    float* dst[3] = {img.radiance, img.albedo, img.normal};
    if (u.AccumulateMode == 0u) {
        float weight = 1.0f / (float)FrameCount;
        for (int k = 0; k < 3; k++) {
            float* p = dst[k] + o;
            p[0] = gl_mix(p[0], res[k].x, weight);
            p[1] = gl_mix(p[1], res[k].y, weight);
            p[2] = gl_mix(p[2], res[k].z, weight);
            p[3] = gl_mix(p[3], 1.0f, weight);
        }
    } else {
        for (int k = 0; k < 3; k++) {
            float* p = dst[k] + o;
            p[0] = p[0] + res[k].x;
            p[1] = p[1] + res[k].y;
            p[2] = p[2] + res[k].z;
            p[3] = p[3] + 1.0f;
        }
    }
}

}  // namespace oracle

// =============================================================================== C interface
using namespace oracle;

extern "C" {

void* oracle_create() { return new Scene(); }
void oracle_destroy(void* s) { delete (Scene*)s; }

void oracle_set_scene(void* sp, const void* nodes48, long nnodes, const void* prims112, long nprims, const void* mats176,
                      long nmats)
{
    Scene& s = *(Scene*)sp;
    s.nodes.assign((const BvhNode*)nodes48, (const BvhNode*)nodes48 + nnodes);
    s.prims.assign((const BvhPrimitive*)prims112, (const BvhPrimitive*)prims112 + nprims);
    s.materials.assign((const MaterialRecord*)mats176, (const MaterialRecord*)mats176 + nmats);
}

void oracle_set_texture(void* sp, int slot, int w, int h, int channels, int isSrgb, int isFloat, const void* texels)
{
    Scene& s = *(Scene*)sp;
    if ((size_t)slot >= s.textures.size()) s.textures.resize(slot + 1);
    Texture& t = s.textures[slot];
    t.w = w;
    t.h = h;
    t.channels = channels;
    t.srgb = isSrgb != 0;
    t.isFloat = isFloat != 0;
    size_t n = (size_t)w * h * channels;
    if (isFloat)
        t.f32.assign((const float*)texels, (const float*)texels + n);
    else
        t.u8.assign((const uint8_t*)texels, (const uint8_t*)texels + n);
}

// CalculateHdriCache, Source/PathTracer/PathTracingRenderer.cpp:803-901.  The reference
// indexes cdf_y_condiciton[x] with x == width when lower_bound runs off the end; that read is
// out of bounds there and clamped to the last column here.
void oracle_calculate_hdri_cache(const float* textureData, int width, int height, float* cache)
{
    float lumSum = 0.0f;
    std::vector<std::vector<float>> pdf(height, std::vector<float>(width));
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            int index = 3 * (i * width + j);
            float R = textureData[index + 0], G = textureData[index + 1], B = textureData[index + 2];
            float lum = 0.212671f * R + 0.715160f * G + 0.072169f * B;
            pdf[i][j] = lum;
            lumSum += lum;
        }
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) pdf[i][j] /= lumSum;
    std::vector<float> pdf_x_margin(width, 0.0f);
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) pdf_x_margin[j] += pdf[i][j];
    std::vector<float> cdf_x_margin = pdf_x_margin;
    for (int i = 1; i < width; i++) cdf_x_margin[i] += cdf_x_margin[i - 1];
    std::vector<std::vector<float>> pdf_y_cond = pdf;
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) pdf_y_cond[i][j] /= pdf_x_margin[j];
    std::vector<std::vector<float>> cdf_y_cond = pdf_y_cond;
    for (int j = 0; j < width; j++)
        for (int i = 1; i < height; i++) cdf_y_cond[i][j] += cdf_y_cond[i - 1][j];
    std::vector<std::vector<float>> byColumn(width, std::vector<float>(height));
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) byColumn[j][i] = cdf_y_cond[i][j];
    for (int j = 0; j < width; j++)
        for (int i = 0; i < height; i++) {
            float xi_1 = float(i) / height;
            float xi_2 = float(j) / width;
            // std::lower_bound written out: first position whose value is not < xi
            long lo = 0, hi = width;
            while (lo < hi) {
                long mid = lo + (hi - lo) / 2;
                if (cdf_x_margin[mid] < xi_1) lo = mid + 1; else hi = mid;
            }
            long x = lo;
            long xc = x < width ? x : width - 1;
            lo = 0, hi = height;
            while (lo < hi) {
                long mid = lo + (hi - lo) / 2;
                if (byColumn[xc][mid] < xi_2) lo = mid + 1; else hi = mid;
            }
            long y = lo;
            cache[3 * (i * width + j) + 0] = float(x) / width;
            cache[3 * (i * width + j) + 1] = float(y) / height;
            cache[3 * (i * width + j) + 2] = pdf[i][j];
        }
}

void oracle_set_environment(void* sp, const float* hdri, const float* cache, int w, int h)
{
    Scene& s = *(Scene*)sp;
    s.envW = w;
    s.envH = h;
    s.hdri.assign(hdri, hdri + (size_t)w * h * 3);
    s.cache.assign(cache, cache + (size_t)w * h * 3);
}

// Renders FrameCount = frameFirst, frameFirst + frameStride, ... (frameCount values) for rows
// [y0, y1) into the three RGBA32F images (read-modify-write like the shader's imageLoad/Store).
// counters: 10 x uint64 (see Counters).  threads <= 0 -> hardware concurrency.
void oracle_render(void* sp, const void* params, int width, int height, uint32_t frameFirst, uint32_t frameCount,
                   uint32_t frameStride, int y0, int y1, float* radiance, float* albedo, float* normal,
                   uint8_t* overflowMask, void* countersOut, int threads)
{
    const Scene& s = *(Scene*)sp;
    Params u;
    std::memcpy(&u, params, sizeof(Params));
    if (u.MaxBounceCount == 0) u.MaxBounceCount = 3;
    u.FrameDimensions[0] = (float)width;
    u.FrameDimensions[1] = (float)height;
    mat4 view, invProj;
    std::memcpy(view.m, u.ViewMatrix, 64);
    std::memcpy(invProj.m, u.InvProjMatrix, 64);
    const mat4 invView = inverse(view);  // pathtracing.comp:159-160, same value for every thread
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    Images img{radiance, albedo, normal, overflowMask};
    std::vector<Counters> perThread(threads);
    std::atomic<int> nextRow(y0);
    auto worker = [&](int t) {
        Ctx c;
        c.s = &s;
        c.u = u;
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= y1) break;
            for (int x = 0; x < width; x++)
                for (uint32_t k = 0; k < frameCount; k++) ShadePixel(c, invProj, invView, x, y, frameFirst + k * frameStride, img);
        }
        perThread[t] = c.c;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    Counters total{};
    for (const Counters& c : perThread) {
        total.samples += c.samples;
        total.raysClosest += c.raysClosest;
        total.raysAny += c.raysAny;
        total.nodeFetches += c.nodeFetches;
        total.pairTests += c.pairTests;
        total.triangleTests += c.triangleTests;
        total.closestHits += c.closestHits;
        total.stackOverflows += c.stackOverflows;
        if (c.maxStack > total.maxStack) total.maxStack = c.maxStack;
    }
    if (countersOut) std::memcpy(countersOut, &total, sizeof(Counters));
}

// The shader's DEBUG_ENABLED output (pathtracing.comp:216-223: debug value of the pixel's pinhole ray stored
// to the radiance image, nothing accumulated).  view 1 = DEBUG_BVH, 2 = DEBUG_HDRI_CACHE.
void oracle_debug_view(void* sp, const void* params, int width, int height, int view, float* radiance, int threads)
{
    const Scene& s = *(Scene*)sp;
    Params u;
    std::memcpy(&u, params, sizeof(Params));
    u.FrameDimensions[0] = (float)width;
    u.FrameDimensions[1] = (float)height;
    mat4 viewM, invProj;
    std::memcpy(viewM.m, u.ViewMatrix, 64);
    std::memcpy(invProj.m, u.InvProjMatrix, 64);
    const mat4 invView = inverse(viewM);
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    std::atomic<int> nextRow(0);
    auto worker = [&]() {
        Ctx c;
        c.s = &s;
        c.u = u;
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= height) break;
            for (int x = 0; x < width; x++) {
                vec2 uv = vec2{(float)x + 0.5f, (float)y + 0.5f} / vec2{u.FrameDimensions[0], u.FrameDimensions[1]};
                vec4 v = view == 2 ? DebugHdriCache(c, uv) : DebugBvh(c, GeneratePinholeCameraRay(c, invProj, invView, uv));
                float* o = radiance + 4 * ((size_t)y * width + x);
                o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
}

// Element-wise evaluation of the built-ins defined in oracle_math.h, for checking the device
// implementations bit for bit.  fn: 0 sin, 1 cos, 2 asin, 3 atan2(a,b), 4 pow(a,b), 5 min, 6 max,
// 7 a/b, 8 sqrt, 9 1/sqrt.
void oracle_eval_math(int fn, const float* a, const float* b, float* out, long n)
{
    for (long i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = gl_sin(a[i]); break;
        case 1: out[i] = gl_cos(a[i]); break;
        case 2: out[i] = gl_asin(a[i]); break;
        case 3: out[i] = gl_atan2(a[i], b[i]); break;
        case 4: out[i] = gl_pow(a[i], b[i]); break;
        case 5: out[i] = gl_min(a[i], b[i]); break;
        case 6: out[i] = gl_max(a[i], b[i]); break;
        case 7: out[i] = a[i] / b[i]; break;
        case 8: out[i] = gl_sqrt(a[i]); break;
        default: out[i] = gl_inversesqrt(a[i]); break;
        }
    }
}

// The display pass: tonemapping.frag:36-71 (main -> ACESFitted -> RRTAndODTFit; `color * mat3(a, b, c)`
// is a row vector times a matrix with columns a, b, c; sums left to right) followed by the sRGB
// encoding of GL_FRAMEBUFFER_SRGB (Runtime/Renderer/Renderer.cpp:25; OpenGL 4.6 section 17.3.7) and
// round-to-nearest conversion to 8-bit unorm.  rgba: n RGBA32F texels; out: n x 3 bytes.
static float display_channel(float c)
{
    c = gl_clamp(c, 0.0f, 1.0f);
    const float e = c <= 0.0031308f ? c * 12.92f : 1.055f * gl_pow(c, 0.41666f) - 0.055f;
    return std::floor(gl_clamp(e, 0.0f, 1.0f) * 255.0f + 0.5f);
}
void oracle_display(const float* rgba, long n, float exposure, unsigned char* out)
{
    for (long i = 0; i < n; i++) {
        const float r = rgba[4 * i] * exposure, g = rgba[4 * i + 1] * exposure, b = rgba[4 * i + 2] * exposure;
        float c[3] = {(r * 0.59719f + g * 0.35458f) + b * 0.04823f, (r * 0.07600f + g * 0.90834f) + b * 0.01566f,
                      (r * 0.02840f + g * 0.13383f) + b * 0.83777f};
        for (int k = 0; k < 3; k++) {  // RRTAndODTFit, tonemapping.frag:36-41
            const float v = c[k];
            const float a = v * (v + 0.0245786f) - 0.000090537f;
            const float d = v * (0.983729f * v + 0.4329510f) + 0.238081f;
            c[k] = a / d;
        }
        const float x = (c[0] * 1.60475f + c[1] * -0.53108f) + c[2] * -0.07367f;
        const float y = (c[0] * -0.10208f + c[1] * 1.10813f) + c[2] * -0.00605f;
        const float z = (c[0] * -0.00327f + c[1] * -0.07276f) + c[2] * 1.07602f;
        out[3 * i] = (unsigned char)display_channel(x);
        out[3 * i + 1] = (unsigned char)display_channel(y);
        out[3 * i + 2] = (unsigned char)display_channel(z);
    }
}

// FNV-1a-64 over a byte range, chained through `h` (start: 1469598103934665603).
uint64_t oracle_fnv1a64(const void* data, long n, uint64_t h)
{
    const unsigned char* p = (const unsigned char*)data;
    for (long i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

int oracle_sizeof_params() { return (int)sizeof(Params); }
int oracle_sizeof_counters() { return (int)sizeof(Counters); }

}  // extern "C"
