// Host-side (CPU) scene processing of the path: declarations.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "layouts.h"

namespace ptcore {

// host_bvh.cpp
// uncapped = false: the reference's builder bit for bit.  uncapped = true: the same sweep without the
// reference's cost cap (every node takes its best SAH split).
size_t BuildBvhWithSah(RefPrimitive* prims, size_t count, int leafSize, std::vector<RefNode>& nodes,
                       int threads = 0, bool uncapped = false);
void AssemblePrimitives(const RefVertex* v, const void* indices, size_t indexCount, int indexBytes,
                        uint32_t meshIndex, RefPrimitive* out);
void AlignNodes(const RefNode* in, size_t count, RefNodeAlign* out);
void AlignPrimitives(const RefPrimitive* in, size_t count, RefPrimitiveAlign* out);

// host_env.cpp
void CalculateHdriCache(const float* rgb, int width, int height, float* cacheOut);

// host_camera.cpp
void LookAt(const float eye[3], const float center[3], const float up[3], float out[16]);
void Perspective(float fovy, float aspect, float zNear, float zFar, float out[16]);
void Inverse4(const float in[16], float out[16]);

// host_pack.cpp: reference layouts -> packed device layouts, with validation.
struct PackedScene {
    std::vector<DevNode> nodes;
    std::vector<DevTriangle> triangles;
    std::vector<DevShading> shading;
    int32_t rootRef = 0;
    int maxDepth = 0;  // interior levels below (and including) the root; bounds the traversal stack
};
// Returns empty string on success, else a description of what is malformed.  leafCut > 0: a leaf of more triangles than
// that is cut into a small sub-tree of leaves of at most leafCut (the triangles keep their order, the new boxes are the
// tight boxes of the sub-ranges), so that the traversal, which tests one triangle per step, scans short leaves.
std::string PackScene(const RefNodeAlign* nodes, size_t nodeCount, const RefPrimitiveAlign* prims,
                      size_t primCount, size_t materialCount, PackedScene& out, int leafCut = 0);
std::string PackMaterials(const RefMaterialAlign* mats, size_t count, size_t textureSlots,
                          const std::vector<int>& slotToTexture, std::vector<DevMaterial>& out);

// host_cache.cpp: on-disk cache of a built BVH, keyed by a hash of the unsorted primitive soup.
uint64_t Fnv1a64(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull);
std::string SaveBvhCache(const std::string& path, uint64_t key, const RefNodeAlign* nodes, size_t nodeCount,
                         const RefPrimitiveAlign* prims, size_t primCount);
std::string LoadBvhCache(const std::string& path, uint64_t key, std::vector<RefNodeAlign>& nodes,
                         std::vector<RefPrimitiveAlign>& prims);

// host_gltf.cpp: glTF 2.0 / GLB reader (ModelLoader::Load + GenerateMaterial without assimp).
struct GltfMaterial {  // Material::MaterialAttributes (Material.h:43-59), same order and defaults
    float emission[3] = {0.0f, 0.0f, 0.0f};
    float albedo[3] = {1.0f, 1.0f, 1.0f};
    float specular = 1.0f, specularTint = 0.0f, metallic = 0.0f, roughness = 1.0f, subsurface = 0.0f, anisotropy = 0.0f,
          sheenRoughness = 0.0f, sheenTint = 0.0f, clearcoat = 0.0f, clearcoatRoughness = 0.0f, refraction = 1.5f,
          transmission = 0.0f;
    int32_t image[11] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // per Material::MaterialTextureSlot
};
static_assert(offsetof(GltfMaterial, image) == 18 * sizeof(float), "attributes are 18 packed floats");
struct GltfSubmesh {
    std::vector<RefVertex> vertices;
    std::vector<uint32_t> indices;
    GltfMaterial material;
};
struct GltfImage {
    std::string path;            // file to decode, or
    std::vector<uint8_t> bytes;  // the encoded image itself (GLB / data URI)
};
struct GltfScene {
    std::vector<GltfSubmesh> submeshes;
    std::vector<GltfImage> images;
};
// Returns empty string on success, else what is wrong with the file.
std::string LoadGltf(const std::string& path, GltfScene& scene);

}  // namespace ptcore
