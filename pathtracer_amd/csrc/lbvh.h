// BVH construction on the GPU for scenes where the reference's O(n log^2 n) sweep-SAH builder
// (Source/PathTracer/BVH.cpp:91-238) takes minutes.  The result is a different tree than BVH.cpp
// builds, so images agree with the reference only up to exact-tie breaking; it is emitted both in
// the reference's node/primitive layouts (for download and checking) and in the packed device
// layouts the render kernels read.
//
// Stages (all on the caller's stream):
//   bounds   centroid bounds of the soup (wave reduction + order-preserving integer atomics)
//   morton   63-bit code per triangle (21 bits per axis)
//   sort     hipcub radix sort of (code, triangle index)
//   tree     topology over the sorted triangles: locally-ordered clustering (ploc.h, the default),
//            or the radix tree of this file (Karras 2012: one thread per internal node, equal codes
//            split by index) followed by
//   fit      bottom-up boxes and heights: the second thread to arrive at a node continues upward
//   emit     sub-trees of <= 4 triangles collapse into leaves (the reference's leaf size); kept
//            nodes are numbered (depth-first rank for the clustered tree, prefix sums over the index
//            order for the radix tree); reference-layout nodes, packed DevNodes and the reordered
//            triangle / shading records are written
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#include "layouts.h"

namespace ptcore {

// Most triangles a leaf of the device-built tree may hold (LbvhTree::leafMax, PlocNodes::leafMax), chosen per build:
// 2 for scenes of up to eight million triangles (the triangle half of the traversal step runs at a third of the lanes: a
// cheap box more and a triangle test fewer is 2-4 % faster on every such scene measured), 3 for larger ones, which stream
// from HBM and pay more for the extra nodes; never more than the reference's 4 (BVH::BuildBvhWithSah is called with leaf
// size 4, PathTracingRenderer.cpp:583).
constexpr int kLbvhLeafLimit = 4;

struct LbvhBounds {
    unsigned int lo[3], hi[3];  // order-preserving encodings of floats
    unsigned int maxMeshIndex, pad;
};

__device__ __forceinline__ unsigned int lbvhEncode(float f)
{
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lbvhDecode(unsigned int e)
{
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

__device__ __forceinline__ void lbvhCentroid(const RefPrimitiveAlign& p, float c[3])
{
    for (int k = 0; k < 3; k++) c[k] = (p.posAuvX[k] + p.posBuvX[k] + p.posCuvX[k]) / 3.0f;  // BVH.cpp:28-31
}

__global__ void lbvhInitBoundsKernel(LbvhBounds* b)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int k = 0; k < 3; k++) {
            b->lo[k] = 0xffffffffu;
            b->hi[k] = 0u;
        }
        b->maxMeshIndex = 0;
        b->pad = 0;
    }
}

__global__ void lbvhBoundsKernel(const RefPrimitiveAlign* __restrict__ prims, unsigned int n, LbvhBounds* out)
{
    // grid-stride: a few thousand blocks whatever n is, so the seven atomics per wave below stay few
    unsigned int lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u}, mesh = 0u;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float c[3];
        lbvhCentroid(prims[i], c);
        for (int k = 0; k < 3; k++) {
            const unsigned int e = lbvhEncode(c[k]);
            lo[k] = e < lo[k] ? e : lo[k];
            hi[k] = e > hi[k] ? e : hi[k];
        }
        mesh = prims[i].meshIndex > mesh ? prims[i].meshIndex : mesh;
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; k++) {
            const unsigned int a = __shfl_down(lo[k], off, 64), b = __shfl_down(hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
        const unsigned int m = __shfl_down(mesh, off, 64);
        mesh = m > mesh ? m : mesh;
    }
    if ((threadIdx.x & 63u) == 0u) {
        for (int k = 0; k < 3; k++) {
            atomicMin(&out->lo[k], lo[k]);
            atomicMax(&out->hi[k], hi[k]);
        }
        atomicMax(&out->maxMeshIndex, mesh);
    }
}

__device__ __forceinline__ unsigned long long lbvhSpread21(unsigned int v)
{
    unsigned long long x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void lbvhMortonKernel(const RefPrimitiveAlign* __restrict__ prims, unsigned int n, const LbvhBounds* bounds,
                                 unsigned long long* keys, unsigned int* values)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c[3];
    lbvhCentroid(prims[i], c);
    unsigned int q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = lbvhDecode(bounds->lo[k]), hi = lbvhDecode(bounds->hi[k]);
        const float extent = hi - lo;
        float t = extent > 0.0f ? (c[k] - lo) / extent : 0.0f;
        t = t >= 0.0f ? t : 0.0f;  // NaN -> 0
        t = t <= 1.0f ? t : 1.0f;
        q[k] = (unsigned int)(t * 2097151.0f);
    }
    keys[i] = (lbvhSpread21(q[0]) << 2) | (lbvhSpread21(q[1]) << 1) | lbvhSpread21(q[2]);
    values[i] = i;
}

// Length of the common prefix of the keys at sorted positions i and j; equal keys continue into
// the positions themselves, so no two distinct positions ever compare as identical.
__device__ __forceinline__ int lbvhDelta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned int)i ^ (unsigned int)j);
    return __clzll((long long)(a ^ b));
}

struct LbvhTree {
    int* parent;      // 2n-1 entries: internal nodes 0..n-2, then leaf k at n-1+k
    int* left;        // n-1 entries, unified indices
    int* right;
    int* first;       // n-1: first and last sorted position covered by the internal node
    int* last;
    float* box;       // (2n-1) x 6: min xyz, max xyz
    int* height;      // n-1: interior levels below and including the node
    unsigned int* visits;  // n-1 arrival counters of the fit pass
    int leafMax;      // sub-trees of at most this many triangles become leaves
};

__global__ void lbvhTreeKernel(const unsigned long long* __restrict__ keys, int n, LbvhTree T)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = lbvhDelta(keys, n, i, i + 1) - lbvhDelta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvhDelta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvhDelta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvhDelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvhDelta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvhDelta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? (n - 1 + gamma) : gamma;
    const int right = hi == gamma + 1 ? (n - 1 + gamma + 1) : (gamma + 1);
    T.left[i] = left;
    T.right[i] = right;
    T.first[i] = lo;
    T.last[i] = hi;
    T.parent[left] = i;
    T.parent[right] = i;
    if (i == 0) T.parent[0] = -1;
}

__global__ void lbvhFitKernel(const RefPrimitiveAlign* __restrict__ prims, const unsigned int* __restrict__ order, int n, LbvhTree T)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    const RefPrimitiveAlign& p = prims[order[k]];
    float b[6];
    for (int c = 0; c < 3; c++) {
        b[c] = fminf(p.posAuvX[c], fminf(p.posBuvX[c], p.posCuvX[c]));
        b[3 + c] = fmaxf(p.posAuvX[c], fmaxf(p.posBuvX[c], p.posCuvX[c]));
    }
    float* mine = T.box + (size_t)(n - 1 + k) * 6;
    for (int c = 0; c < 6; c++) mine[c] = b[c];
    if (n == 1) return;
    int cur = T.parent[n - 1 + k];
    // Every node is completed by whichever of its two children's threads arrives second; the fences
    // make the first child's box visible to it across compute units and XCDs.
    for (;;) {
        __threadfence();
        const unsigned int seen = atomicAdd(&T.visits[cur], 1u);
        if (seen == 0u) return;
        __threadfence();
        const int l = T.left[cur], r = T.right[cur];
        const float* bl = T.box + (size_t)l * 6;
        const float* br = T.box + (size_t)r * 6;
        float* bo = T.box + (size_t)cur * 6;
        for (int c = 0; c < 3; c++) {
            bo[c] = fminf(__builtin_nontemporal_load(bl + c), __builtin_nontemporal_load(br + c));
            bo[3 + c] = fmaxf(__builtin_nontemporal_load(bl + 3 + c), __builtin_nontemporal_load(br + 3 + c));
        }
        const int hl = l >= n - 1 ? 0 : __builtin_nontemporal_load(T.height + l);
        const int hr = r >= n - 1 ? 0 : __builtin_nontemporal_load(T.height + r);
        T.height[cur] = 1 + (hl > hr ? hl : hr);
        if (cur == 0) return;
        cur = T.parent[cur];
    }
}

__device__ __forceinline__ int lbvhCount(const LbvhTree& T, int n, int u) { return u >= n - 1 ? 1 : T.last[u] - T.first[u] + 1; }
__device__ __forceinline__ int lbvhFirst(const LbvhTree& T, int n, int u) { return u >= n - 1 ? u - (n - 1) : T.first[u]; }

// Per internal node: how many reference-layout nodes it emits (itself plus its leaf children) and
// whether it is kept as an interior node.
__global__ void lbvhCountKernel(int n, LbvhTree T, unsigned int* emit, unsigned int* kept)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const bool keep = lbvhCount(T, n, i) > T.leafMax;
    unsigned int e = 0;
    if (keep) {
        e = 1u;
        if (lbvhCount(T, n, T.left[i]) <= T.leafMax) e++;
        if (lbvhCount(T, n, T.right[i]) <= T.leafMax) e++;
    }
    emit[i] = e;
    kept[i] = keep ? 1u : 0u;
}

__device__ __forceinline__ void lbvhWriteRefNode(RefNodeAlign* out, int left, int right, int count, int first, const float* box)
{
    RefNodeAlign r;
    r.left = left;
    r.right = right;
    r.n = count;
    r.index = first;
    for (int c = 0; c < 3; c++) {
        r.AA[c] = box[c];
        r.BB[c] = box[3 + c];
    }
    r.padA = r.padB = 0.0f;
    *out = r;
}

// emitAt / keptAt: exclusive prefix sums of lbvhCountKernel's outputs.
__global__ void lbvhEmitNodesKernel(int n, LbvhTree T, const unsigned int* __restrict__ emitAt,
                                    const unsigned int* __restrict__ keptAt, const unsigned int* __restrict__ kept,
                                    RefNodeAlign* refNodes, DevNode* devNodes)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1 || !kept[i]) return;
    const int id = 1 + (int)emitAt[i];      // reference node id (0 is the dummy)
    const int dev = 1 + (int)keptAt[i];     // packed node index (0 is "no node")
    const int child[2] = {T.left[i], T.right[i]};
    int refId[2], devRef[2];
    int nextLeaf = id + 1;
    for (int c = 0; c < 2; c++) {
        const int u = child[c];
        const int count = lbvhCount(T, n, u);
        if (count <= T.leafMax) {
            const int first = lbvhFirst(T, n, u);
            refId[c] = nextLeaf;
            lbvhWriteRefNode(refNodes + nextLeaf, 0, 0, count, first, T.box + (size_t)u * 6);
            nextLeaf++;
            devRef[c] = ~(int)(((unsigned int)first << 3) | (unsigned int)(count - 1));
        } else {
            refId[c] = 1 + (int)emitAt[u];
            devRef[c] = 1 + (int)keptAt[u];
        }
    }
    lbvhWriteRefNode(refNodes + id, refId[0], refId[1], 0, 0, T.box + (size_t)i * 6);
    const float* bl = T.box + (size_t)child[0] * 6;
    const float* br = T.box + (size_t)child[1] * 6;
    DevNode d;
    for (int c = 0; c < 3; c++) {
        d.lmin[c] = bl[c];
        d.lmax[c] = bl[3 + c];
        d.rmin[c] = br[c];
        d.rmax[c] = br[3 + c];
    }
    d.left = devRef[0];
    d.right = devRef[1];
    d.pad0 = d.pad1 = 0;
    devNodes[dev] = d;
}

// The soup in sorted order, in the reference's layout and in the packed device layouts; the packed
// records hold the ray-independent terms of the triangle test, evaluated with the same float
// expressions as the host packer uses (intersection.glsl:23-25,48).
__global__ void lbvhEmitTrianglesKernel(const RefPrimitiveAlign* __restrict__ prims, const unsigned int* __restrict__ order,
                                        unsigned int n, RefPrimitiveAlign* sorted, DevTriangle* tris, DevShading* shading)
{
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const RefPrimitiveAlign p = prims[order[k]];
    sorted[k] = p;
    float ab[3], ac[3];
    for (int c = 0; c < 3; c++) {
        ab[c] = p.posBuvX[c] - p.posAuvX[c];
        ac[c] = p.posCuvX[c] - p.posAuvX[c];
    }
    DevTriangle t;
    for (int c = 0; c < 3; c++) {
        t.a[c] = p.posAuvX[c];
        t.ab[c] = ab[c];
        t.ac[c] = ac[c];
    }
    t.n0 = ab[1] * ac[2] - ac[1] * ab[2];
    t.n1 = ab[2] * ac[0] - ac[2] * ab[0];
    t.n2 = ab[0] * ac[1] - ac[0] * ab[1];
    tris[k] = t;
    DevShading s;
    for (int c = 0; c < 3; c++) {
        s.norA[c] = p.norAuvY[c];
        s.norB[c] = p.norBuvY[c];
        s.norC[c] = p.norCuvY[c];
    }
    s.uvAx = p.posAuvX[3];
    s.uvAy = p.norAuvY[3];
    s.uvBx = p.posBuvX[3];
    s.uvBy = p.norBuvY[3];
    s.uvCx = p.posCuvX[3];
    s.uvCy = p.norCuvY[3];
    s.meshIndex = p.meshIndex;
    const float g0 = ac[1] * ab[2] - ab[1] * ac[2];  // cross(edgeAC, edgeAB)
    const float g1 = ac[2] * ab[0] - ab[2] * ac[0];
    const float g2 = ac[0] * ab[1] - ab[0] * ac[1];
    const float inv = 1.0f / sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    s.geoN[0] = g0 * inv;
    s.geoN[1] = g1 * inv;
    s.geoN[2] = g2 * inv;
    s.pad = 0;
    shading[k] = s;
}

// A scene of at most leafMax triangles: the root is a leaf (reference node 1).
__global__ void lbvhEmitRootLeafKernel(int n, LbvhTree T, RefNodeAlign* refNodes, DevNode* devNodes)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float box[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int k = 0; k < n; k++) {
        const float* b = T.box + (size_t)(n - 1 + k) * 6;
        for (int c = 0; c < 3; c++) {
            box[c] = fminf(box[c], b[c]);
            box[3 + c] = fmaxf(box[3 + c], b[3 + c]);
        }
    }
    lbvhWriteRefNode(refNodes + 1, 0, 0, n, 0, box);
    devNodes[0] = DevNode{};
}

}  // namespace ptcore
