// TEST INFRASTRUCTURE ONLY -- never linked into, or called from, the product library.
//
// Thin extern "C" shim around the reference's own BVH builder so tests can call
// it through ctypes.  It is compiled TOGETHER WITH the reference's unmodified
// Source/PathTracer/BVH.cpp (where that file lies under /root/reference; see
// oracle/Makefile) into oracle/_ref/libref_shim.so.  No reference source text
// lives in this file: it only includes the reference header and forwards.
//
// Follows the call site Source/PathTracer/PathTracingRenderer.cpp:575-584:
//   nodes = { zeroed dummy }; BuildBvhWithSah(prims, nodes, 0, n-1, 4)
#include "BVH.h"
#include "glm/gtc/matrix_transform.hpp"

#include <cstring>
#include <vector>

static_assert(sizeof(BVH::BvhPrimitive) == 100, "host primitive record is 100 B");
static_assert(sizeof(BVH::BvhNode) == 40, "host node record is 40 B");

extern "C" {

// prims: in/out, `count` records of 100 B (reordered in place like the reference does).
// nodes_out: capacity `node_capacity` records of 40 B. Returns node count (incl. dummy
// node 0) or -1 if the capacity is too small.
long ref_build_bvh_sah(void* prims, long count, int leaf_size, void* nodes_out, long node_capacity)
{
    std::vector<BVH::BvhPrimitive> p(count);
    std::memcpy(p.data(), prims, sizeof(BVH::BvhPrimitive) * count);
    BVH::BvhNode init{};
    std::vector<BVH::BvhNode> nodes{init};
    BVH::BuildBvhWithSah(p, nodes, 0, (int)count - 1, leaf_size);
    if ((long)nodes.size() > node_capacity) return -1;
    std::memcpy(prims, p.data(), sizeof(BVH::BvhPrimitive) * count);
    std::memcpy(nodes_out, nodes.data(), sizeof(BVH::BvhNode) * nodes.size());
    return (long)nodes.size();
}

// The unused median-split builder (Source/PathTracer/BVH.cpp:26-89), for completeness.
long ref_build_bvh_median(void* prims, long count, int leaf_size, void* nodes_out, long node_capacity)
{
    std::vector<BVH::BvhPrimitive> p(count);
    std::memcpy(p.data(), prims, sizeof(BVH::BvhPrimitive) * count);
    BVH::BvhNode init{};
    std::vector<BVH::BvhNode> nodes{init};
    BVH::BuildBvh(p, nodes, 0, (int)count - 1, leaf_size);
    if ((long)nodes.size() > node_capacity) return -1;
    std::memcpy(prims, p.data(), sizeof(BVH::BvhPrimitive) * count);
    std::memcpy(nodes_out, nodes.data(), sizeof(BVH::BvhNode) * nodes.size());
    return (long)nodes.size();
}

// sizeof/offsetof probes of the reference's GPU mirror structs (BVH.h:40-62).
int ref_sizeof_node_align() { return (int)sizeof(BVH::BvhNodeAlign); }
int ref_sizeof_prim_align() { return (int)sizeof(BVH::BvhPrimitiveAlign); }
int ref_offsetof_node_AA() { return (int)offsetof(BVH::BvhNodeAlign, AA); }
int ref_offsetof_node_BB() { return (int)offsetof(BVH::BvhNodeAlign, BB); }
int ref_offsetof_prim_mesh() { return (int)offsetof(BVH::BvhPrimitiveAlign, meshIndex); }

// glm 0.9.9.9 camera matrices exactly as the reference host computes them
// (Source/Runtime/Camera/Camera.cpp:10-18; Source/PathTracer/PathTracingApplication.cpp:233-235,250).
// Matrices are column-major float[16].
void ref_glm_look_at(const float* eye, const float* center, const float* up, float* out16)
{
    glm::mat4 m = glm::lookAt(glm::vec3(eye[0], eye[1], eye[2]), glm::vec3(center[0], center[1], center[2]),
                              glm::vec3(up[0], up[1], up[2]));
    std::memcpy(out16, &m[0][0], 64);
}
void ref_glm_perspective(float fovy, float aspect, float zn, float zf, float* out16)
{
    glm::mat4 m = glm::perspective(fovy, aspect, zn, zf);
    std::memcpy(out16, &m[0][0], 64);
}
void ref_glm_inverse(const float* in16, float* out16)
{
    glm::mat4 m;
    std::memcpy(&m[0][0], in16, 64);
    glm::mat4 r = glm::inverse(m);
    std::memcpy(out16, &r[0][0], 64);
}

}  // extern "C"
