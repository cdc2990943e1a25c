// Byte layouts shared by the host code and the HIP kernels.
//
// "Ref*" structs mirror the reference's records byte for byte
// (Source/PathTracer/BVH.h:8-62, Source/PathTracer/PathTracingRenderer.h:117-193,
// Shaders/Library/common.glsl:32-91).  "Dev*" structs are this library's own packed,
// cache-line-aligned device layouts that the kernels actually read.
#pragma once
#include <cstddef>
#include <cstdint>

namespace ptcore {

// ---------------------------------------------------------------- reference layouts
struct RefNode {  // BVH::BvhNode, 40 B
    int32_t left, right, n, index;
    float AA[3];
    float BB[3];
};
struct RefPrimitive {  // BVH::BvhPrimitive, 100 B
    float posA[3], posB[3], posC[3];
    float norA[3], norB[3], norC[3];
    float uvA[2], uvB[2], uvC[2];
    uint32_t meshIndex;
};
struct alignas(16) RefNodeAlign {  // BVH::BvhNodeAlign, 48 B
    int32_t left, right, n, index;
    float AA[3];
    float padA;
    float BB[3];
    float padB;
};
struct alignas(16) RefPrimitiveAlign {  // BVH::BvhPrimitiveAlign, 112 B
    float posAuvX[4], norAuvY[4];
    float posBuvX[4], norBuvY[4];
    float posCuvX[4], norCuvY[4];
    uint32_t meshIndex;
    uint32_t pad[3];
};
struct alignas(16) RefMaterialAlign {  // PathTracingRenderer::MaterialAlign, 176 B
    uint64_t textureHandle[11];        // order of Material::MaterialTextureSlot (Material.h:26-40)
    uint64_t pad0;
    float emission[3];
    float pad1;
    float albedo[3];
    float specular, specularTint, metallic, roughness, subsurface, anisotropy;
    float sheenRoughness, sheenTint, clearcoat, clearcoatRoughness, refraction, transmission;
    float pad2;
};
struct RefVertex {  // PathTracingRenderer::VertexSave, 32 B
    float pos[3], nor[3], uv[2];
};

static_assert(sizeof(RefNode) == 40, "");
static_assert(sizeof(RefPrimitive) == 100, "");
static_assert(sizeof(RefNodeAlign) == 48 && offsetof(RefNodeAlign, AA) == 16 && offsetof(RefNodeAlign, BB) == 32, "");
static_assert(sizeof(RefPrimitiveAlign) == 112 && offsetof(RefPrimitiveAlign, meshIndex) == 96, "");
static_assert(sizeof(RefMaterialAlign) == 176 && offsetof(RefMaterialAlign, emission) == 96 &&
                  offsetof(RefMaterialAlign, albedo) == 112 && offsetof(RefMaterialAlign, specular) == 124 &&
                  offsetof(RefMaterialAlign, transmission) == 168,
              "");
static_assert(sizeof(RefVertex) == 32, "");

enum TextureSlot {
    kTexEmission = 0,
    kTexAlbedo,
    kTexNormal,
    kTexSpecular,
    kTexSpecularColor,
    kTexMetallicRoughness,
    kTexSheenRoughness,
    kTexSheenColor,
    kTexClearcoat,
    kTexClearcoatRoughness,
    kTexTransmission,
    kTexSlotCount
};

// ------------------------------------------------------------------- device layouts
//
// A child reference is one int32:
//   ref >= 0 : index of a DevNode (interior node)
//   ref <  0 : leaf; bits = ~ref = (first_triangle << 3) | (triangle_count - 1), count 1..8
// Leaves therefore need no node fetch at all, and an interior node carries both
// children's boxes so one 64-B line decides both.
struct alignas(64) DevNode {
    float lmin[3];
    int32_t left;
    float lmax[3];
    int32_t right;
    float rmin[3];
    int32_t pad0;
    float rmax[3];
    int32_t pad1;
};
static_assert(sizeof(DevNode) == 64, "");

// Bytes of one finished sample's record in the megakernel's sample buffer (radiance, albedo, normal: three floats each).
constexpr unsigned kSampleBytes = 36;

// Four-wide node: the boxes of up to four children, bound by bound, and their references (0 = empty slot).
struct alignas(128) DevNode4 {
    float lo[3][4];
    float hi[3][4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(DevNode4) == 128, "");

// Four-wide node with quantised boxes, 64 B: what the traversal reads (four 16-byte loads, where exact boxes
// take seven).  The node spans a grid of 256 planes per axis: cell `scale` (a power of two), grid origin
// `origin`; a child's box is kept as the grid planes (8 bits each) nearest outside it (checked in double
// precision when the node is made), so it contains the exact box.  Box tests only steer the traversal (which
// sub-trees are visited, in which order); every hit is decided by the exact triangle records.
// qlo[a] / qhi[a]: byte k = lower / upper plane of child k on axis a.  child[k] = 0: empty slot (its planes
// are inverted).
struct alignas(64) DevNodeQ {
    float origin[3];
    float scale[3];
    uint32_t qlo[3];
    uint32_t qhi[3];
    int32_t child[4];
};
static_assert(sizeof(DevNodeQ) == 64, "");
// Cells by which every plane is moved further out.  0: on the benchmark scene a ray visits 6.1 nodes (5.96 with
// exact boxes); 1: 6.4 nodes, 3.5 % slower.  The slab tests work on distances along the ray computed as
// plane * (cell / d) + (origin - o) / d, whose rounding differs from that of ((bound - o) / d) on the exact
// box by a few units in the last place of the distance: a ray that grazes a box within that (and hits a
// triangle right there) is the case a margin of 1 would make safe and 0 leaves to chance; no test image
// has such a pixel.
#ifndef PT_QUANT_MARGIN
#define PT_QUANT_MARGIN 0
#endif
constexpr int kQuantMargin = PT_QUANT_MARGIN;

// Ray/triangle record: vertex A, the two edges and their cross product, all computed on the
// host with the reference's float expressions (intersection.glsl:23-25).
struct alignas(16) DevTriangle {
    float a[3];
    float n0;  // normalVector = cross(edgeAB, edgeAC)
    float ab[3];
    float n1;
    float ac[3];
    float n2;
};
static_assert(sizeof(DevTriangle) == 48, "");

// Shading record, fetched once per closest hit.  geoN is the reference's per-hit
// normalize(cross(edgeAC, edgeAB)) (intersection.glsl:48), which does not depend on the ray.
struct alignas(16) DevShading {
    float norA[3], uvAx;
    float norB[3], uvAy;
    float norC[3], uvBx;
    float uvBy, uvCx, uvCy;
    uint32_t meshIndex;
    float geoN[3];
    uint32_t pad;
};
static_assert(sizeof(DevShading) == 80, "");

struct alignas(16) DevMaterial {
    float emission[3];
    float specular;
    float albedo[3];
    float specularTint;
    float metallic, roughness, subsurface, anisotropy;
    float sheenRoughness, sheenTint, clearcoat, clearcoatRoughness;
    float refraction, transmission;
    uint32_t anyTexture;  // non-zero when at least one of tex[] is set
    uint32_t pad;
    float glassTint[3];  // pow(albedo, 0.5) per channel (the refraction lobe's tint, disney.glsl:70), filled in on the device at upload
    uint32_t pad2;
    int32_t tex[12];  // kTexSlotCount entries used; index into DevTexture table, -1 = none
};
static_assert(sizeof(DevMaterial) == 144, "");

struct DevTexture {
    uint64_t offset;  // byte offset into the texel arena
    int32_t width, height;
    int32_t channels;  // 1..4
    int32_t format;    // 0 = u8 unorm, 1 = u8 sRGB colour (+ linear alpha), 2 = f32
};

// Launch-constant block passed by value to the kernels.
struct DevParams {
    float invView[16];  // inverse(ViewMatrix), hoisted from pathtracing.comp:159-160
    float invProj[16];
    float view[16];
    float frameW, frameH;  // FrameDimensions
    uint32_t width, height;
    uint32_t antiAliasing;
    float focalLength, apertureSize, apertureShapeX, apertureShapeY;
    float modifiers[12];
    uint32_t maxBounces;
    uint32_t accumulateMode;
    uint32_t frameFirst, frameCount, frameStride;
    float hdriW, hdriH;
    int32_t envW, envH;
    int32_t rootRef;
    uint32_t tilesX, tilesY;
    uint32_t travExitNum;  // megakernel: leave the traversal phase below this many 64ths of the live lanes
    const unsigned int* blockList;  // megakernel: work items {tile, first sample << 16 | samples}, costliest first (null: item = tile, all samples)
    const unsigned int* blockCount; // megakernel: entries of blockList in use
    unsigned int* tileCost;         // megakernel: per tile, ticks waves spent on its items (input of the next launch's order)
    unsigned int* workNext;         // megakernel: next work item to hand out (zeroed before the launch)
    float* samples;                 // megakernel: finished samples, nine floats (radiance, albedo, normal) per (sample, pixel), sample-major
    unsigned long long pixels;      // width * height
    unsigned long long* waveTimes;  // counting build: per tile, wall-clock ticks at wave start and end (or null)
    uint32_t refillMin;    // wavefront: a wave refills from the ray queue once this many lanes are idle
    uint32_t hitMinNum;    // megakernel: shade hits once this many 64ths of the live lanes hold one
};

struct DevCounters {
    unsigned long long samples, raysClosest, raysAny, nodeVisits, triangleTests, closestHits;
    unsigned int maxStack, pad;
    // megakernel occupancy profile (counting build): wave-level iteration counts and the lanes
    // that took part in them.  [0] traversal iterations, [1] lanes traversing in them,
    // [2] iterations with a node step, [3] lanes in node steps, [4] iterations with a triangle
    // step, [5] lanes in triangle steps, [6] shading phases with a hit/miss to shade, [7] lanes
    // shading a hit, [8] lanes shading a miss, [9] shading phases, [10] lanes finishing a sample,
    // [11] lanes starting a sample
    unsigned long long profile[16];
    // counting build of the megakernel: closest-hit queries by path depth (primary rays in [0], the ray
    // after the first bounce in [1], ...; depths past 31 in [31])
    unsigned long long depthRays[32];
};

}  // namespace ptcore
