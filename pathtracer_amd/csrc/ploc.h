// Topology of the device-built BVH by parallel locally-ordered clustering (Meister and Bittner
// 2018) instead of the Morton radix tree: the triangles, in Morton order, start as one cluster each;
// in every round a cluster looks at its `radius` neighbours on either side in the cluster array,
// picks the one whose box united with its own has the smallest surface area, and two clusters that
// pick each other merge into a new node.  Big triangles (a floor under a detailed mesh) are nobody's
// cheapest partner and stay single until the top of the tree, which is exactly where a surface-area
// builder puts them and where a radix tree, which only sees centroids, does not.
//
// Everything is a data-parallel function of the previous round (node numbers come from prefix sums,
// ties are broken by position), so the tree does not depend on scheduling.  The result is converted
// into the LbvhTree arrays that the emit kernels of lbvh.h read: internal node 0 is the root, every
// internal node covers a contiguous range of the final triangle order (the leaf order of the tree).
#pragma once
#include "lbvh.h"

namespace ptcore {

struct PlocNodes {
    int* left;      // n-1: children of internal node k (ids: leaf m = m, internal k = n + k)
    int* right;
    int* size;      // 2n-1: triangles below the node
    int* height;    // n-1: interior levels below and including the node
    float* box;     // (2n-1) x 6
    int* position;  // 2n-1: first position of the node's triangles in the final order
    int* kept;      // 2n-1: nodes of the sub-tree that stay interior nodes (more than leafMax triangles)
    int* preorder;  // 2n-1: rank of the node among the kept nodes in depth-first order, left before right
    int leafMax;    // sub-trees of at most this many triangles become leaves
};

__device__ __forceinline__ float plocArea(const float* a, const float* b)
{
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
    const float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
    const float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void plocInitKernel(const RefPrimitiveAlign* __restrict__ prims, const unsigned int* __restrict__ order, int n,
                               int* clusterId, float* clusterBox, PlocNodes N)
{
    const int m = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (m >= n) return;
    const RefPrimitiveAlign& p = prims[order[m]];
    for (int c = 0; c < 3; c++) {
        const float lo = fminf(p.posAuvX[c], fminf(p.posBuvX[c], p.posCuvX[c]));
        const float hi = fmaxf(p.posAuvX[c], fmaxf(p.posBuvX[c], p.posCuvX[c]));
        clusterBox[(size_t)m * 6 + c] = lo;
        clusterBox[(size_t)m * 6 + 3 + c] = hi;
        N.box[(size_t)m * 6 + c] = lo;
        N.box[(size_t)m * 6 + 3 + c] = hi;
    }
    clusterId[m] = m;
    N.size[m] = 1;
    N.kept[m] = 0;
}

// nearest[i] = the position within `radius` of i whose union with cluster i has the smallest area.
// Ties go to the nearer position, then to the partner i ^ 1, so that equal boxes pair up instead of
// all pointing the same way.
__global__ void plocNearestKernel(const float* __restrict__ clusterBox, int count, int radius, int* nearest)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count) return;
    float mine[6];
    for (int c = 0; c < 6; c++) mine[c] = clusterBox[(size_t)i * 6 + c];
    const int lo = i - radius < 0 ? 0 : i - radius, hi = i + radius >= count ? count - 1 : i + radius;
    float best = FLT_MAX;
    int bestJ = -1, bestDist = 0x7fffffff;
    for (int j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float a = plocArea(mine, clusterBox + (size_t)j * 6);
        const int dist = j > i ? j - i : i - j;
        const bool better = !(a >= best) || (a == best && (dist < bestDist || (dist == bestDist && j == (i ^ 1))));
        if (better || bestJ < 0) {
            best = a;
            bestJ = j;
            bestDist = dist;
        }
    }
    nearest[i] = bestJ;
}

// flag: low word 1 = the position survives into the next round (alone, or as the merged cluster),
// high word 1 = it creates a node.
__global__ void plocFlagKernel(const int* __restrict__ nearest, int count, unsigned long long* flags)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count) return;
    const int j = nearest[i];
    const bool mutual = j >= 0 && nearest[j] == i;
    flags[i] = !mutual ? 1ull : (i < j ? (1ull | (1ull << 32)) : 0ull);
}

__global__ void plocMergeKernel(const int* __restrict__ idIn, const float* __restrict__ boxIn, const int* __restrict__ nearest,
                                const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ scan,
                                int count, int created, int n, int* idOut, float* boxOut, PlocNodes N)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= count) return;
    const unsigned long long f = flags[i];
    if ((f & 1ull) == 0ull) return;
    const int at = (int)(unsigned int)scan[i];
    float b[6];
    for (int c = 0; c < 6; c++) b[c] = boxIn[(size_t)i * 6 + c];
    int id = idIn[i];
    if (f >> 32) {
        const int j = nearest[i];
        const int k = created + (int)(scan[i] >> 32);
        const int a = id, o = idIn[j];
        for (int c = 0; c < 3; c++) {
            b[c] = fminf(b[c], boxIn[(size_t)j * 6 + c]);
            b[3 + c] = fmaxf(b[3 + c], boxIn[(size_t)j * 6 + 3 + c]);
        }
        id = n + k;
        N.left[k] = a;
        N.right[k] = o;
        const int size = N.size[a] + N.size[o];
        N.size[id] = size;
        N.kept[id] = (size > N.leafMax ? 1 : 0) + N.kept[a] + N.kept[o];
        const int ha = a < n ? 0 : N.height[a - n], ho = o < n ? 0 : N.height[o - n];
        N.height[k] = 1 + (ha > ho ? ha : ho);
        for (int c = 0; c < 6; c++) N.box[(size_t)id * 6 + c] = b[c];
    }
    idOut[at] = id;
    for (int c = 0; c < 6; c++) boxOut[(size_t)at * 6 + c] = b[c];
}

// Top-down over the nodes of one round (later rounds first): a node's triangles start where the
// node starts, the right child's after the left child's; likewise its rank in depth-first order,
// which becomes the node's index in the packed array: the left child sits next to its parent and
// a sub-tree is one contiguous run of records, as in the host packer's pre-order layout.
__global__ void plocPositionKernel(int first, int count, int n, PlocNodes N)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= count) return;
    const int k = first + t;
    const int p = N.position[n + k];
    const int a = N.left[k], o = N.right[k];
    N.position[a] = p;
    N.position[o] = p + N.size[a];
    const int d = N.preorder[n + k];
    N.preorder[a] = d + 1;
    N.preorder[o] = d + 1 + N.kept[a];
}

// Into the arrays of lbvh.h: internal node k becomes index n-2-k (the root, created last, is 0),
// leaf m becomes n-1 + its position in the final order.
__global__ void plocRelabelKernel(int n, PlocNodes N, const unsigned int* __restrict__ order, unsigned int* finalOrder, LbvhTree T,
                                  unsigned int* packedAt)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) {  // leaf t
        const int p = N.position[t];
        finalOrder[p] = order[t];
        for (int c = 0; c < 6; c++) T.box[(size_t)(n - 1 + p) * 6 + c] = N.box[(size_t)t * 6 + c];
    }
    if (t < n - 1) {  // internal node t
        const int id = n + t, f = n - 2 - t;
        const int a = N.left[t], o = N.right[t];
        const int ua = a < n ? n - 1 + N.position[a] : n - 2 - (a - n);
        const int uo = o < n ? n - 1 + N.position[o] : n - 2 - (o - n);
        T.left[f] = ua;
        T.right[f] = uo;
        T.first[f] = N.position[id];
        T.last[f] = N.position[id] + N.size[id] - 1;
        T.height[f] = N.height[t];
        packedAt[f] = (unsigned int)N.preorder[id];
        for (int c = 0; c < 6; c++) T.box[(size_t)f * 6 + c] = N.box[(size_t)id * 6 + c];
        T.parent[ua] = f;
        T.parent[uo] = f;
        if (f == 0) T.parent[0] = -1;
    }
}

}  // namespace ptcore
