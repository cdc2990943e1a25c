// Device-side surface shading: material evaluation, the six-lobe Disney-style BSDF
// (sample + evaluate), HDRI lookup and importance sampling, ray offsetting.
//
// Results are bit-identical to the reference shader's formulas
// (Shaders/Library/{utility,disney,brdf,montecarlo,hdri}.glsl) evaluated in IEEE binary32;
// the organisation is not the shader's.  The shader rebuilds its whole BrdfData block three
// times per bounce (NEE direction, the dummy L = 0 used for sampling, the sampled direction)
// and re-derives the lobe mixture in each of SampleDisneyBrdf / EvaluateDisneyBrdf.  Here the
// direction-independent part (tangent frame, roughness remaps, eta, f0, tints, luminances,
// view-side dot products) is computed once per hit into `Surface`, and only the
// direction-dependent part (`Dir`) is recomputed.  The three microfacet reflection lobes share
// one D / G1 / G2 evaluation.
#pragma once
#include "device_math.h"
#include "layouts.h"

namespace ptcore {

constexpr float kPi = 3.141592653589f;  // Library/common.glsl:6-9 (the shader's own constants)
constexpr float kTwoPi = 6.283185307178f;
constexpr float kOneOverPi = 0.318309886183f;
constexpr float kOneOverTwoPi = 0.159154943091f;
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kMinDielectricsF0 = 0.04f;

struct DevScene {
    const DevNode* nodes;
    const DevNodeQ* wide;  // the four-wide tree the traversal walks (boxes quantised outward)
    const DevTriangle* triangles;
    const DevShading* shading;
    const DevMaterial* materials;
    const DevTexture* textures;
    const unsigned char* texels;
    const float* srgbLut;  // 512 entries: 8-bit channel -> float, unsigned-normalised | sRGB-decoded
    const float4* envColorPdf;  // per HDRI texel: rgb = radiance, w = cache pdf (cache.b)
    const float2* envSample;    // per cache texel: (cache.r, cache.g)
};

// ---- RNG: montecarlo.glsl:8-37
PT_DEV unsigned nextRandom(unsigned& state)
{
    state = state * 747796405u + 2891336453u;
    unsigned result = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
PT_DEV float randomValue(unsigned& state) { return (float)nextRandom(state) / 4294967295.0f; }

// ---- texture(): bilinear, REPEAT, base level, float weights
// i mod n (REPEAT addressing).  Nearly every index lies within one period of [0, n): one add or
// subtract settles those, and the integer division (some 35 instructions) is left to the lanes that
// really are further out.
PT_DEV int wrapIndex(int i, int n)
{
    int m = i < 0 ? i + n : (i >= n ? i - n : i);
    if (__builtin_expect((unsigned)m >= (unsigned)n, 0)) {
        m = i % n;
        m = m < 0 ? m + n : m;
    }
    return m;
}
struct Bilerp {
    int x0, x1, y0, y1;
    float a, b;
};
PT_DEV Bilerp bilerpSetup(float u, float v, int w, int h)
{
    Bilerp r;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
    r.a = x - fx;
    r.b = y - fy;
    r.x0 = wrapIndex((int)fx, w);
    r.y0 = wrapIndex((int)fy, h);
    r.x1 = wrapIndex(r.x0 + 1, w);
    r.y1 = wrapIndex(r.y0 + 1, h);
    return r;
}
PT_DEV float bilerp(const Bilerp& q, float t00, float t10, float t01, float t11)
{
    return mix_(mix_(t00, t10, q.a), mix_(t01, t11, q.a), q.b);
}

// One tap: the arena holds every texel with four channels (absent ones as 0, 0, 1), so a tap of an 8-bit texture is
// one 32-bit read and four table lookups (table entries 0..255: b / 255; 256..511: sRGB decode; alpha stays linear),
// a tap of a float texture one 16-byte read.
PT_DEV float4 fetchTexel(const DevScene& S, const DevTexture& t, int x, int y)
{
    // (32-bit byte offsets from the arena's base, which the host keeps below 4 GB: a scalar base and a lane offset)
    const unsigned texel = (unsigned)y * (unsigned)t.width + (unsigned)x;
    if (t.format == 2) return *reinterpret_cast<const float4*>(S.texels + ((unsigned)t.offset + texel * 16u));
    const unsigned w = *reinterpret_cast<const unsigned*>(S.texels + ((unsigned)t.offset + texel * 4u));
    const unsigned colour = t.format == 1 ? 256u : 0u;
    return make_float4(S.srgbLut[colour + (w & 255u)], S.srgbLut[colour + ((w >> 8) & 255u)], S.srgbLut[colour + ((w >> 16) & 255u)],
                       S.srgbLut[w >> 24]);
}
PT_DEV float4 sampleTexture(const DevScene& S, int texIndex, float u, float v)
{
    const DevTexture t = S.textures[texIndex];
    Bilerp q = bilerpSetup(u, v, t.width, t.height);
    float4 t00 = fetchTexel(S, t, q.x0, q.y0), t10 = fetchTexel(S, t, q.x1, q.y0);
    float4 t01 = fetchTexel(S, t, q.x0, q.y1), t11 = fetchTexel(S, t, q.x1, q.y1);
    return make_float4(bilerp(q, t00.x, t10.x, t01.x, t11.x), bilerp(q, t00.y, t10.y, t01.y, t11.y),
                       bilerp(q, t00.z, t10.z, t01.z, t11.z), bilerp(q, t00.w, t10.w, t01.w, t11.w));
}

// ---- HDRI: hdri.glsl:6-54
// SampleHdri: one bilinear tap of the (discontinuous) inverse-CDF table at two randoms, in two halves -- the first
// draws the randoms and requests the four table entries, the second turns them into a direction -- so that a caller
// can put work that needs neither between the request and the use.
struct SampleTaps {
    Bilerp q;
    float2 t00, t10, t01, t11;
};
PT_DEV SampleTaps requestSampleHdri(const DevScene& S, int envW, int envH, unsigned& rng)
{
    SampleTaps e;
    float xi1 = randomValue(rng);
    float xi2 = randomValue(rng);
    e.q = bilerpSetup(xi1, xi2, envW, envH);
    e.t00 = S.envSample[(size_t)e.q.y0 * envW + e.q.x0], e.t10 = S.envSample[(size_t)e.q.y0 * envW + e.q.x1];
    e.t01 = S.envSample[(size_t)e.q.y1 * envW + e.q.x0], e.t11 = S.envSample[(size_t)e.q.y1 * envW + e.q.x1];
    return e;
}
PT_DEV v3 finishSampleHdri(const SampleTaps& e)
{
    float ux = bilerp(e.q, e.t00.x, e.t10.x, e.t01.x, e.t11.x);
    float uy = bilerp(e.q, e.t00.y, e.t10.y, e.t01.y, e.t11.y);
    uy = 1.0f - uy;
    float phi = kTwoPi * (ux - 0.5f);
    float theta = kPi * (uy - 0.5f);
    float st, ct, sp, cp;
    sincos_(theta, st, ct);
    sincos_(phi, sp, cp);
    return mk3(ct * cp, st, ct * sp);
}
PT_DEV v3 sampleHdri(const DevScene& S, int envW, int envH, unsigned& rng)
{
    const SampleTaps e = requestSampleHdri(S, envW, envH, rng);
    return finishSampleHdri(e);
}
// EvaluateHdri: radiance and solid-angle pdf towards L (colour and pdf share one 4-texel fetch).
PT_DEV v3 evaluateHdri(const DevScene& S, int envW, int envH, float hdriW, float hdriH, v3 L, float& pdf)
{
    v3 n = normalize3(L);
    float u = atan2_(n.z, n.x) * kOneOverTwoPi + 0.5f;
    float v = 0.5f - asin_(n.y) * kOneOverPi;
    Bilerp q = bilerpSetup(u, v, envW, envH);
    float4 t00 = S.envColorPdf[(size_t)q.y0 * envW + q.x0], t10 = S.envColorPdf[(size_t)q.y0 * envW + q.x1];
    float4 t01 = S.envColorPdf[(size_t)q.y1 * envW + q.x0], t11 = S.envColorPdf[(size_t)q.y1 * envW + q.x1];
    v3 color = mk3(bilerp(q, t00.x, t10.x, t01.x, t11.x), bilerp(q, t00.y, t10.y, t01.y, t11.y),
                   bilerp(q, t00.z, t10.z, t01.z, t11.z));
    pdf = bilerp(q, t00.w, t10.w, t01.w, t11.w);
    float theta = kPi * (1.0f - v);
    float sinTheta = fmax_(sin_(theta), 1e-10f);
    float pConvert = hdriW * hdriH / (kTwoPi * kPi * sinTheta);
    pdf = pdf * pConvert;
    return color;
}

// ---- OffsetRay: utility.glsl:9-26
PT_DEV float offsetComponent(float p, float n)
{
    int of = (int)(256.0f * n);
    float pi = __int_as_float(__float_as_int(p) + ((p < 0.0f) ? -of : of));
    return abs_(p) < (1.0f / 32.0f) ? p + (1.0f / 65536.0f) * n : pi;
}
PT_DEV v3 offsetRay(v3 P, v3 N) { return mk3(offsetComponent(P.x, N.x), offsetComponent(P.y, N.y), offsetComponent(P.z, N.z)); }

// ---- small BRDF terms: brdf.glsl
PT_DEV float luminance(v3 rgb) { return dot(rgb, mk3(0.2126f, 0.7152f, 0.0722f)); }
PT_DEV float schlickWeight(float u)
{
    float m = clamp_(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m * m2 * m2;
}
PT_DEV float schlick(float f0, float f90, float u) { return f0 + (f90 - f0) * schlickWeight(u); }
PT_DEV float fresnelDielectric(float cosThetaI, float ior)
{
    float sinThetaTSq = ior * ior * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = sqrt_(fmax_(1.0f - sinThetaTSq, 0.0f));
    float rs = (ior * cosThetaT - cosThetaI) / (ior * cosThetaT + cosThetaI);
    float rp = (ior * cosThetaI - cosThetaT) / (ior * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}

// ---- evaluated material + direction-independent BRDF state for one hit
struct Surface {
    // evaluated material (utility.glsl:94-169)
    v3 emission, albedo;
    float specular, specularTint, metallic, subsurface, sheenRoughness, sheenTint, clearcoat, refraction, transmission;
    // frame
    v3 N, V, X, Y;
    // PrepareEvaluationBrdfData, direction-independent part (utility.glsl:203-232,250,253,258)
    float roughness, alphaClearcoat, ax, ay;
    float NdotV, VdotX, VdotY, eta, f0;
    // lobe mixture, direction-independent part (disney.glsl:12-18,127-130)
    v3 Cspec0, Csheen;
    float lumAlbedo, lumDielectric, lumMetal;
    // direction-independent terms of the lobe evaluators, taken once per hit instead of once per evaluated direction:
    // the view-side Smith terms (brdf.glsl:50-56,70-75) and the refraction tint pow(albedo, 0.5) (disney.glsl:70)
    float smithV1, lenV;
    v3 glassTint;
};

struct Dir {
    v3 L, H;
    float NdotL, NdotH, VdotH, LdotH, LdotX, LdotY, HdotX, HdotY;
};

PT_DEV void tangentFrame(v3 N, v3& T, v3& B)  // utility.glsl:28-36
{
    v3 helper = abs_(N.x) > 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    T = normalize3(cross(N, helper));
    B = normalize3(cross(N, T));
}

struct MaterialEval {
    v3 emission, albedo;
    float specular, specularTint, metallic, roughness, subsurface, anisotropy, sheenRoughness, sheenTint, clearcoat,
        clearcoatRoughness, refraction, transmission;
    v3 glassTint;       // pow(albedo, 0.5) as the material record holds it ...
    bool tintOnRecord;  // ... which stands unless a texture has modulated the albedo
};

// The material record's scalar part as loaded (requested as early as the hit's material index is known, so that
// the reads are under way while the caller does other work).
struct MaterialWords {
    float4 m0, m1, m2, m3, m4, m5;
};
PT_DEV MaterialWords loadMaterial(const DevScene& S, unsigned meshIndex)
{
    const float4* mp = (const float4*)(S.materials + meshIndex);
    return MaterialWords{mp[0], mp[1], mp[2], mp[3], mp[4], mp[5]};
}

// `shadingNormal` is rewritten when a normal map is bound (utility.glsl:108-111).
PT_DEV MaterialEval evaluateMaterial(const DevScene& S, const DevParams& P, const MaterialWords& mw, unsigned meshIndex, float u,
                                     float v, v3& shadingNormal, v3 tangent)
{
    const float4 m0 = mw.m0, m1 = mw.m1, m2 = mw.m2, m3 = mw.m3, m4 = mw.m4;
    MaterialEval e;
    e.emission = mk3(m0.x, m0.y, m0.z);
    e.specular = m0.w;
    e.albedo = mk3(m1.x, m1.y, m1.z);
    e.specularTint = m1.w;
    e.metallic = m2.x;
    e.roughness = m2.y;
    e.subsurface = m2.z;
    e.anisotropy = m2.w;
    e.sheenRoughness = m3.x;
    e.sheenTint = m3.y;
    e.clearcoat = m3.z;
    e.clearcoatRoughness = m3.w;
    e.refraction = m4.x;
    e.transmission = m4.y;
    e.glassTint = mk3(mw.m5.x, mw.m5.y, mw.m5.z);
    e.tintOnRecord = true;
    if (__float_as_uint(m4.z) != 0u) {  // anyTexture
        const int* tex = S.materials[meshIndex].tex;
        if (tex[kTexEmission] >= 0) {
            float4 s = sampleTexture(S, tex[kTexEmission], u, v);
            e.emission = e.emission * mk3(s.x, s.y, s.z);
        }
        if (tex[kTexAlbedo] >= 0) {
            float4 s = sampleTexture(S, tex[kTexAlbedo], u, v);
            e.albedo = e.albedo * mk3(s.x, s.y, s.z);
            e.tintOnRecord = false;
        }
        if (tex[kTexNormal] >= 0) {  // SampleNormalMap, utility.glsl:69-92
            v3 N = shadingNormal;
            v3 T = tangent;
            T = normalize3(T - dot(T, N) * N);
            v3 B = cross(N, T);
            float4 s = sampleTexture(S, tex[kTexNormal], u, v);
            v3 nm = normalize3(mk3(s.x, s.y, s.z) * 2.0f - splat(1.0f));
            shadingNormal = normalize3(T * nm.x + B * nm.y + N * nm.z);
        }
        if (tex[kTexSpecular] >= 0) e.specular = e.specular * sampleTexture(S, tex[kTexSpecular], u, v).w;
        if (tex[kTexSpecularColor] >= 0) {
            float4 s = sampleTexture(S, tex[kTexSpecularColor], u, v);
            e.specularTint = e.specularTint * length3(mk3(s.x, s.y, s.z));
        }
        if (tex[kTexMetallicRoughness] >= 0) {
            float4 s = sampleTexture(S, tex[kTexMetallicRoughness], u, v);
            e.metallic = e.metallic * s.x;
            e.roughness = e.roughness * s.y;
        }
        if (tex[kTexSheenRoughness] >= 0) e.sheenRoughness = e.sheenRoughness * sampleTexture(S, tex[kTexSheenRoughness], u, v).w;
        if (tex[kTexSheenColor] >= 0) {
            float4 s = sampleTexture(S, tex[kTexSheenColor], u, v);
            e.sheenTint = e.sheenTint * length3(mk3(s.x, s.y, s.z));
        }
        if (tex[kTexClearcoat] >= 0) e.clearcoat = e.clearcoat * sampleTexture(S, tex[kTexClearcoat], u, v).x;
        if (tex[kTexClearcoatRoughness] >= 0)
            e.clearcoatRoughness = e.clearcoatRoughness * sampleTexture(S, tex[kTexClearcoatRoughness], u, v).y;
        if (tex[kTexTransmission] >= 0) e.transmission = e.transmission * sampleTexture(S, tex[kTexTransmission], u, v).x;
    }
    // UI modifiers (utility.glsl:155-166); P.modifiers follows the uniform declaration order
    e.specular = clamp_(e.specular + P.modifiers[0], 0.0f, 1.0f);
    e.specularTint = clamp_(e.specularTint + P.modifiers[1], 0.0f, 1.0f);
    e.metallic = clamp_(e.metallic + P.modifiers[2], 0.0f, 1.0f);
    e.roughness = clamp_(e.roughness + P.modifiers[3], 0.0f, 1.0f);
    e.subsurface = clamp_(e.subsurface + P.modifiers[4], 0.0f, 1.0f);
    e.anisotropy = clamp_(e.anisotropy + P.modifiers[5], 0.0f, 1.0f);
    e.sheenRoughness = clamp_(e.sheenRoughness + P.modifiers[6], 0.0f, 1.0f);
    e.sheenTint = clamp_(e.sheenTint + P.modifiers[7], 0.0f, 1.0f);
    e.clearcoat = clamp_(e.clearcoat + P.modifiers[8], 0.0f, 1.0f);
    e.clearcoatRoughness = clamp_(e.clearcoatRoughness + P.modifiers[9], 0.0f, 1.0f);
    e.refraction = clamp_(e.refraction + P.modifiers[10], 1.01f, 2.0f);
    e.transmission = clamp_(e.transmission + P.modifiers[11], 0.0f, 1.0f);
    return e;
}

PT_DEV Surface makeSurface(const MaterialEval& m, v3 N, v3 V)
{
    Surface s;
    s.emission = m.emission;
    s.albedo = m.albedo;
    s.specular = m.specular;
    s.specularTint = m.specularTint;
    s.metallic = m.metallic;
    s.subsurface = m.subsurface;
    s.sheenRoughness = m.sheenRoughness;
    s.sheenTint = m.sheenTint;
    s.clearcoat = m.clearcoat;
    s.refraction = m.refraction;
    s.transmission = m.transmission;
    s.N = N;
    s.V = V;
    tangentFrame(N, s.X, s.Y);
    s.roughness = clamp_(m.roughness, 0.045f, 1.0f);
    float alpha = s.roughness * s.roughness;
    float rc = clamp_(m.clearcoatRoughness, 0.045f, 1.0f);
    s.alphaClearcoat = rc * rc;
    float aspect = sqrt_(1.0f - 0.9f * m.anisotropy);
    s.ax = fmax_(0.002025f, alpha / aspect);
    s.ay = fmax_(0.002025f, alpha * aspect);
    s.NdotV = dot(N, V);
    s.eta = s.NdotV <= 0.0f ? m.refraction : (1.0f / m.refraction);
    s.VdotX = dot(V, s.X);
    s.VdotY = dot(V, s.Y);
    float r = (1.0f - s.eta) / (1.0f + s.eta);  // IorToF0(1, eta)
    s.f0 = r * r;
    // tints (disney.glsl:6-18)
    s.lumAlbedo = luminance(m.albedo);
    v3 Ctint = s.lumAlbedo > 0.0f ? m.albedo / s.lumAlbedo : splat(1.0f);
    s.Cspec0 = s.f0 * mix3(splat(1.0f), Ctint, m.specularTint) * m.specular;
    s.Csheen = mix3(splat(1.0f), Ctint, m.sheenTint);
    float sw = schlickWeight(s.NdotV);
    s.lumDielectric = luminance(mix3(s.Cspec0, splat(1.0f), sw));
    s.lumMetal = luminance(mix3(m.albedo, splat(1.0f), sw));
    const float absNdotV = abs_(s.NdotV);
    const float ga = s.VdotX * s.ax, gb = s.VdotY * s.ay;
    s.lenV = sqrt_(ga * ga + gb * gb + absNdotV * absNdotV);
    s.smithV1 = (2.0f * absNdotV) / (absNdotV + s.lenV);
    // (read only when the glass lobe has weight; the three pow are taken here only where a texture has changed the albedo)
    s.glassTint = m.glassTint;
    if (!m.tintOnRecord && (1.0f - m.metallic) * m.transmission > 0.0f)
        s.glassTint = mk3(pow_(m.albedo.x, 0.5f), pow_(m.albedo.y, 0.5f), pow_(m.albedo.z, 0.5f));
    return s;
}

PT_DEV Dir makeDir(const Surface& s, v3 L)  // utility.glsl:229-255, direction-dependent part
{
    Dir d;
    d.L = L;
    d.NdotL = dot(s.N, L);
    v3 H = d.NdotL <= 0.0f ? normalize3(L + s.V * s.eta) : normalize3(L + s.V);
    float NdotH = dot(s.N, H);
    if (NdotH < 0.0f) H = -H;
    d.H = H;
    d.VdotH = dot(s.V, H);
    d.NdotH = dot(s.N, H);
    d.LdotH = dot(L, H);
    d.LdotX = dot(L, s.X);
    d.HdotX = dot(H, s.X);
    d.LdotY = dot(L, s.Y);
    d.HdotY = dot(H, s.Y);
    return d;
}

struct LobeMix {
    float dielectricWeight, metalWeight, glassWeight;
    float pDiffuse, pDielectric, pMetal, pGlass, pSheen, pClearcoat;
};
// disney.glsl:117-142 == :253-278, given LdotH
PT_DEV LobeMix lobeMix(const Surface& s, float LdotH)
{
    LobeMix l;
    float clearcoatEnergyLoss = 1.0f - (schlick(kMinDielectricsF0, 1.0f, LdotH) * s.clearcoat);
    l.dielectricWeight = (1.0f - s.metallic) * (1.0f - s.transmission) * clearcoatEnergyLoss;
    l.metalWeight = s.metallic * clearcoatEnergyLoss;
    l.glassWeight = (1.0f - s.metallic) * s.transmission;
    l.pDiffuse = l.dielectricWeight * s.lumAlbedo;
    l.pDielectric = l.dielectricWeight * s.lumDielectric;
    l.pMetal = l.metalWeight * s.lumMetal;
    l.pGlass = l.glassWeight;
    l.pSheen = s.sheenRoughness;
    l.pClearcoat = s.clearcoat;
    float inv = 1.0f / (l.pDiffuse + l.pDielectric + l.pMetal + l.pGlass + l.pSheen + l.pClearcoat);
    l.pDiffuse *= inv;
    l.pDielectric *= inv;
    l.pMetal *= inv;
    l.pGlass *= inv;
    l.pSheen *= inv;
    l.pClearcoat *= inv;
    return l;
}

PT_DEV v3 toWorld(const Surface& s, v3 v) { return s.X * v.x + s.Y * v.y + s.N * v.z; }
PT_DEV v3 reflect3(v3 I, v3 N) { return I - 2.0f * dot(N, I) * N; }
PT_DEV v3 refract3(v3 I, v3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return splat(0.0f);
    return eta * I - (eta * d + sqrt_(k)) * N;
}

PT_DEV v3 hemispherePointCos(float u, float v)  // montecarlo.glsl:53-59
{
    float phi = v * kTwoPi;
    float cosTheta = sqrt_(1.0f - u);
    float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    float sp, cp;
    sincos_(phi, sp, cp);
    return mk3(cp * sinTheta, sp * sinTheta, cosTheta);
}
PT_DEV v3 sampleVisibleGGX(v3 Ve, float ax, float ay, float xi1, float xi2)  // montecarlo.glsl:83-106
{
    v3 Vh = normalize3(mk3(ax * Ve.x, ay * Ve.y, Ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * rsqrt_(lensq) : mk3(1.0f, 0.0f, 0.0f);
    v3 T2 = cross(Vh, T1);
    float r = sqrt_(xi1);
    float phi = kTwoPi * xi2;
    float sp, cp;
    sincos_(phi, sp, cp);
    float t1 = r * cp;
    float t2 = r * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = mix_(sqrt_(1.0f - t1 * t1), t2, s);
    v3 Nh = t1 * T1 + t2 * T2 + sqrt_(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize3(mk3(ax * Nh.x, ay * Nh.y, fmax_(0.0f, Nh.z)));
}

// SampleDisneyBrdf (disney.glsl:110-236).  The shader feeds it a BrdfData built from L = 0, so
// its lobe mixture sees LdotH = 0; the other inputs it reads are direction-independent.
PT_DEV v3 sampleBsdf(const Surface& s, unsigned& rng)
{
    LobeMix l = lobeMix(s, 0.0f);
    float c0 = l.pDiffuse;
    float c1 = c0 + l.pDielectric;
    float c2 = c1 + l.pMetal;
    float c3 = c2 + l.pGlass;
    float c4 = c3 + l.pSheen;
    float xi1 = randomValue(rng);
    float xi2 = randomValue(rng);
    float rd = randomValue(rng);
    // The lobes' samplers share their building blocks: a sine / cosine pair of one azimuth, a direction built in the
    // tangent frame and taken to the world, and (all but the cosine lobe) a reflection about that direction.  Each is
    // written once here, with the per-lobe operands selected beforehand, so that a wave whose lanes picked different
    // lobes runs one copy of them; every lane still performs its lobe's operations in its lobe's order.
    const bool cosine = rd < c0 || (!(rd < c2) && !(rd < c3) && rd < c4);  // diffuse or sheen: cosine hemisphere
    const bool ggx = !cosine && (rd < c2 || rd < c3);  // dielectric + metal reflection (rd < c2), or glass
    const bool coat = !cosine && !ggx;                 // clearcoat: GGX NDF (montecarlo.glsl:63-76)
    float sp, cp;
    sincos_(kTwoPi * (coat ? xi1 : xi2), sp, cp);  // (montecarlo.glsl:55 has the factors the other way round: same product)
    v3 local;
    if (cosine) {  // montecarlo.glsl:53-59
        float cosTheta = sqrt_(1.0f - xi1);
        float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
        local = mk3(cp * sinTheta, sp * sinTheta, cosTheta);
    } else if (ggx) {  // visible normals of the anisotropic GGX (montecarlo.glsl:83-106)
        v3 Ve = mk3(dot(s.X, s.V), dot(s.Y, s.V), dot(s.N, s.V));
        v3 Vh = normalize3(mk3(s.ax * Ve.x, s.ay * Ve.y, Ve.z));
        float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
        v3 T1 = lensq > 0.0f ? mk3(-Vh.y, Vh.x, 0.0f) * rsqrt_(lensq) : mk3(1.0f, 0.0f, 0.0f);
        v3 T2 = cross(Vh, T1);
        float r = sqrt_(xi1);
        float t1 = r * cp;
        float t2 = r * sp;
        float sh = 0.5f * (1.0f + Vh.z);
        t2 = mix_(sqrt_(1.0f - t1 * t1), t2, sh);
        v3 Nh = t1 * T1 + t2 * T2 + sqrt_(fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
        local = normalize3(mk3(s.ax * Nh.x, s.ay * Nh.y, fmax_(0.0f, Nh.z)));
    } else {
        float a2 = s.alphaClearcoat * s.alphaClearcoat;
        float cosThetaH = sqrt_((1.0f - xi2) / (1.0f + (a2 - 1.0f) * xi2));
        float sinThetaH = sqrt_(fmax_(0.0f, 1.0f - cosThetaH * cosThetaH));
        local = mk3(sinThetaH * cp, sinThetaH * sp, cosThetaH);
    }
    if (!cosine && local.z < 0.0f) local = -local;
    const v3 w = toWorld(s, local);
    if (cosine) return w;
    // `w` is the half vector: reflect about it, or (glass, when the rescaled random falls beyond the Fresnel term) refract
    bool refracts = false;
    if (ggx && !(rd < c2)) {
        float F = fresnelDielectric(abs_(dot(s.V, w)), s.eta);
        float rr = fmin_((rd - c2) / (c3 - c2), 0.99999994f);  // RescaleRandomNumber, montecarlo.glsl:118-122
        refracts = !(rr < F);
    }
    v3 out = reflect3(-s.V, w);
    if (refracts) out = refract3(-s.V, w, s.eta);
    return normalize3(out);
}

// EvaluateDisneyBrdf (disney.glsl:243-346) with its lobe evaluators (:20-104).
PT_DEV v3 evaluateBsdf(const Surface& s, const Dir& d, float& pdfOut)
{
    v3 f = splat(0.0f);
    float pdf = 0.0f;
    LobeMix l = lobeMix(s, d.LdotH);
    const bool reflection = d.NdotL * s.NdotV > 0.0f;
    const float VdotH = abs_(d.VdotH);
    const bool front = d.NdotL > 0.0f;  // the lobes' own `NdotL <= 0 -> 0` guards

    // shared microfacet terms (brdf.glsl:23-30,50-56,70-75)
    float D = 0.0f, V1 = 0.0f, V2 = 0.0f;
    const bool needMicrofacet = (reflection && (l.pDielectric > 0.0f || l.pMetal > 0.0f)) || l.pGlass > 0.0f;
    if (needMicrofacet) {
        float a2 = s.ax * s.ay;
        v3 dd = mk3(s.ay * d.HdotX, s.ax * d.HdotY, a2 * d.NdotH);
        float d2 = dot(dd, dd);
        float b2 = a2 / d2;
        D = a2 * b2 * b2 * kOneOverPi;
        float absNdotV = abs_(s.NdotV), absNdotL = abs_(d.NdotL);
        V1 = s.smithV1;
        float lambdaV = absNdotL * s.lenV;
        float lambdaL = absNdotV * length3(mk3(s.ax * d.LdotX, s.ay * d.LdotY, absNdotL));
        V2 = 0.5f / (lambdaV + lambdaL);
    }
    const float pdfReflect = D * V1 / (4.0f * s.NdotV);

    if (l.pDiffuse > 0.0f && reflection) {
        if (front) {
            // Fd_Burley / Fd_HanrahanKrueger (brdf.glsl:127-144)
            float f90 = 0.5f + 2.0f * s.roughness * d.LdotH * d.LdotH;
            float lightScatter = schlick(1.0f, f90, d.NdotL);
            float viewScatter = schlick(1.0f, f90, s.NdotV);
            float Fd = lightScatter * viewScatter * kOneOverPi;
            float Fss90 = s.roughness * d.LdotH * d.LdotH;
            float FLss = schlickWeight(d.NdotL);
            float FVss = schlickWeight(s.NdotV);
            float Fss = mix_(1.0f, Fss90, FLss) * mix_(1.0f, Fss90, FVss);
            float Fdss = 1.25f * (Fss * (1.0f / (d.NdotL + s.NdotV) - 0.5f) + 0.5f) * kOneOverPi;
            f = f + (mix_(Fd, Fdss, s.subsurface) * s.albedo) * l.dielectricWeight;
            pdf = pdf + (d.NdotL * kOneOverPi) * l.pDiffuse;
        } else {
            f = f + splat(0.0f) * l.dielectricWeight;
            pdf = pdf + 0.0f * l.pDiffuse;
        }
    }
    // The dielectric lobe wants the Fresnel term at 1 / ior, the glass lobe at eta, which IS 1 / ior seen from outside
    // and ior seen from inside.  A lane with one of the two lobes (transmission 0 or 1), or seen from outside, needs
    // one term: those share the first evaluation below, each lane at its own index; only a lane with both lobes seen
    // from inside takes the second.
    const bool wantDielectric = l.pDielectric > 0.0f && reflection, wantGlass = l.pGlass > 0.0f;
    const bool inside = s.NdotV <= 0.0f;  // (eta = ior)
    float fresnelFirst = 0.0f, fresnelGlass = 0.0f;
    if (wantDielectric || wantGlass) {
        fresnelFirst = fresnelDielectric(VdotH, wantDielectric ? 1.0f / s.refraction : s.eta);
        fresnelGlass = fresnelFirst;
        if (wantDielectric && wantGlass && inside) fresnelGlass = fresnelDielectric(VdotH, s.eta);
    }
    if (wantDielectric) {
        float F = (fresnelFirst - s.f0) / (1.0f - s.f0);
        v3 Fc = mix3(s.Cspec0, splat(1.0f), F);
        f = f + (front ? Fc * D * V2 : splat(0.0f)) * l.dielectricWeight;
        pdf = pdf + (front ? pdfReflect : 0.0f) * l.pDielectric;
    }
    if (l.pMetal > 0.0f && reflection) {
        v3 Fc = mix3(s.albedo, splat(1.0f), schlickWeight(VdotH));
        f = f + (front ? Fc * D * V2 : splat(0.0f)) * l.metalWeight;
        pdf = pdf + (front ? pdfReflect : 0.0f) * l.pMetal;
    }
    if (wantGlass) {
        float F = fresnelGlass;
        if (reflection) {
            f = f + (front ? splat(F) * D * V2 : splat(0.0f)) * l.glassWeight;
            pdf = pdf + (front ? pdfReflect : 0.0f) * l.pGlass * F;
        } else {
            // EvaluateMicrofacetRefraction (disney.glsl:54-73)
            v3 fr = splat(0.0f);
            float pr = 0.0f;
            if (!(d.NdotL >= 0.0f)) {
                float denom = d.LdotH + d.VdotH * s.eta;
                denom = denom * denom;
                float eta2 = s.eta * s.eta;
                float jacobian = abs_(d.LdotH) / denom;
                pr = D * V1 * fmax_(0.0f, d.VdotH) * jacobian / s.NdotV;
                fr = s.glassTint * (splat(1.0f) - splat(F)) * D * V2 * abs_(d.VdotH) * jacobian * eta2 * 4.0f;
            }
            f = f + fr * l.glassWeight;
            pdf = pdf + pr * l.pGlass * (1.0f - F);
        }
    }
    if (l.pSheen > 0.0f && reflection) {
        v3 fs = splat(0.0f);
        float ps = 0.0f;
        if (front) {  // EvaluateSheen (disney.glsl:75-88): D_Charlie x V_Neubelt
            float invR = 1.0f / s.sheenRoughness;
            float nh = clamp_(d.NdotH, 0.0f, 1.0f);
            float sin2h = 1.0f - nh * nh;
            float Dc = (2.0f + invR) * pow_(sin2h, invR * 0.5f) * kOneOverTwoPi;
            float Vn = 1.0f / (4.0f * (d.NdotL + s.NdotV - d.NdotL * s.NdotV));
            ps = d.NdotL * kOneOverPi;
            fs = Dc * Vn * s.Csheen;
        }
        f = f + fs * s.sheenRoughness;
        pdf = pdf + ps * l.pSheen;
    }
    if (l.pClearcoat > 0.0f && reflection) {
        v3 fc = splat(0.0f);
        float pc = 0.0f;
        if (front) {  // EvaluateClearcoat (disney.glsl:90-104): D_GGX x V_Kelemen x Schlick(0.04)
            float om = 1.0f - d.NdotH * d.NdotH;
            float a = d.NdotH * s.alphaClearcoat;
            float k = s.alphaClearcoat / (om + a * a);
            float Dg = k * k * kOneOverPi;
            float Vk = 0.25f / (d.LdotH * d.LdotH + 1e-5f);
            float Fk = schlick(kMinDielectricsF0, 1.0f, d.LdotH);
            pc = Dg * d.NdotH / (4.0f * d.LdotH);
            fc = splat(Dg) * Vk * Fk;
        }
        f = f + fc * s.clearcoat;
        pdf = pdf + pc * l.pClearcoat;
    }
    pdfOut = pdf;
    return f * abs_(d.NdotL);
}

}  // namespace ptcore
