// Host-side BVH construction: the sweep-SAH builder of the reference
// (Source/PathTracer/BVH.cpp:91-238), restated so that its output is bit-identical while
// running several times faster:
//
//  * The reference sorts 100-byte primitive records four times per interior node with
//    std::sort and a comparator that recomputes both centroids on every call
//    (BVH.cpp:6-24,138-140,226-228).  std::sort is not stable, so the final permutation
//    depends on the exact sequence of comparator outcomes.  Here the centroid of every
//    primitive is computed once with the same float expression, and std::sort runs the
//    SAME call sequence (X, Y, Z, winner) over 16-byte {centroid, id} keys.  Identical
//    comparator outcomes on an identical algorithm give the identical permutation; the
//    100-byte records are gathered once at the end.
//  * Per-primitive bounds are computed once instead of inside every sweep.
//  * Sub-trees are independent (disjoint primitive ranges), and the reference numbers nodes
//    in pre-order (BVH.cpp:95-96,231-232).  Large sub-trees are therefore built on separate
//    threads into private node arrays with local ids and spliced in pre-order afterwards.
//
// All float arithmetic is kept in the reference's order; this file must be compiled without
// FMA contraction (-ffp-contract=off): the SAH cost 2*(lx*ly + lx*lz + ly*lz)*count decides
// `totalCost < cost` and a 1-ulp change flips splits.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <future>
#include <cstdlib>
#include <thread>
#include <vector>

#include "host.h"

namespace ptcore {
namespace {

constexpr float kRefInf = 114514.0f;  // BVH.cpp:4 -- also the cost cap that triggers the median fallback

struct Key {
    float c[3];   // centroid, (A + B + C) / 3 per component (BVH.cpp:8)
    uint32_t id;  // index into the caller's primitive array
};

struct Bounds {
    float lo[3], hi[3];
};

// glm::min(a, b) = (b < a) ? b : a ; glm::max(a, b) = (a < b) ? b : a  (NaN behaviour included)
inline float gmin(float a, float b) { return (b < a) ? b : a; }
inline float gmax(float a, float b) { return (a < b) ? b : a; }

// Surface area of `b` times `count`, every operation in binary32 and in this order.
inline float surfaceTimesCount(const Bounds& b, int count)
{
    const float ex = b.hi[0] - b.lo[0];
    const float ey = b.hi[1] - b.lo[1];
    const float ez = b.hi[2] - b.lo[2];
    const float area = 2.0f * ((ex * ey) + (ex * ez) + (ey * ez));
    return area * (float)count;
}

struct Builder {
    const std::vector<Bounds>& bounds;  // per original primitive id
    std::vector<Key>& keys;             // current order
    int leafSize;
    int parallelCutoff;
    std::atomic<int>& threadsLeft;
    // The reference's "infinity": the seed of its running bounds and, more importantly, the cost a
    // split has to beat.  With 114514 (BVH.cpp:4) a node whose best area x count reaches that value
    // keeps the default split, the median along X (BVH.cpp:132-134,209-222): that is what the top
    // levels of every large scene get.  With FLT_MAX every node takes its best sweep split.
    float kInf;

    // Builds the sub-tree over keys[l..r] into `out` with LOCAL pre-order ids (root = 0,
    // child ids relative to this vector).  `index` fields are absolute primitive positions.
    void build(int l, int r, std::vector<RefNode>& out) const
    {
        const size_t id = out.size();
        out.push_back(RefNode{});
        RefNode node{};
        node.left = node.right = node.n = node.index = 0;
        for (int k = 0; k < 3; k++) {
            node.AA[k] = 1145141919.0f;
            node.BB[k] = -1145141919.0f;
        }
        for (int i = l; i <= r; i++) {
            const Bounds& b = bounds[keys[i].id];
            for (int k = 0; k < 3; k++) {
                node.AA[k] = gmin(node.AA[k], b.lo[k]);
                node.BB[k] = gmax(node.BB[k], b.hi[k]);
            }
        }
        const int count = r - l + 1;
        if (count <= leafSize) {
            node.n = count;
            node.index = l;
            out[id] = node;
            return;
        }

        float bestCost = kInf;
        int bestAxis = 0;
        int bestSplit = (l + r) / 2;
        std::vector<Bounds> suffix(count);
        for (int axis = 0; axis < 3; axis++) {
            sortAxis(l, r, axis);

            // rightMin/rightMax of BVH.cpp:162-180: running bounds of [i, r], seeded with -+INF
            Bounds run;
            for (int k = 0; k < 3; k++) {
                run.lo[k] = kInf;
                run.hi[k] = -kInf;
            }
            for (int i = r; i >= l; i--) {
                const Bounds& b = bounds[keys[i].id];
                for (int k = 0; k < 3; k++) {
                    run.hi[k] = gmax(run.hi[k], b.hi[k]);
                    run.lo[k] = gmin(run.lo[k], b.lo[k]);
                }
                suffix[i - l] = run;
            }
            // leftMin/leftMax of BVH.cpp:142-160, fused with the split scan of :183-214
            for (int k = 0; k < 3; k++) {
                run.lo[k] = kInf;
                run.hi[k] = -kInf;
            }
            float cost = kInf;
            int split = l;
            for (int i = l; i <= r - 1; i++) {
                const Bounds& b = bounds[keys[i].id];
                for (int k = 0; k < 3; k++) {
                    run.hi[k] = gmax(run.hi[k], b.hi[k]);
                    run.lo[k] = gmin(run.lo[k], b.lo[k]);
                }
                // surface area x triangle count on either side of the cut, in the reference's order of
                // operations (BVH.cpp:187-213: extents, 2 * (xy + xz + yz), times the count, summed)
                const float sideL = surfaceTimesCount(run, i - l + 1);
                const float sideR = surfaceTimesCount(suffix[i + 1 - l], r - i);
                const float both = sideL + sideR;
                if (both < cost) {
                    cost = both;
                    split = i;
                }
            }
            if (cost < bestCost) {
                bestCost = cost;
                bestAxis = axis;
                bestSplit = split;
            }
        }
        sortAxis(l, r, bestAxis);
        suffix = std::vector<Bounds>();  // release before recursing

        // Children: left sub-tree occupies ids [id+1, id+1+|left|), right follows (pre-order).
        const int nl = bestSplit - l + 1, nr = r - bestSplit;
        bool spawned = false;
        std::vector<RefNode> rightNodes;
        std::future<void> fut;
        if (nl >= parallelCutoff && nr >= parallelCutoff && threadsLeft.fetch_sub(1) > 0) {
            spawned = true;
            fut = std::async(std::launch::async, [&] { build(bestSplit + 1, r, rightNodes); });
        } else if (nl >= parallelCutoff && nr >= parallelCutoff) {
            threadsLeft.fetch_add(1);
        }
        const size_t leftRoot = out.size();
        build(l, bestSplit, out);
        const size_t rightRoot = out.size();
        if (spawned) {
            fut.get();
            threadsLeft.fetch_add(1);
            // splice: shift the private sub-tree's local ids by rightRoot
            out.reserve(out.size() + rightNodes.size());
            for (RefNode rn : rightNodes) {
                if (rn.n == 0) {
                    rn.left += (int32_t)rightRoot;
                    rn.right += (int32_t)rightRoot;
                }
                out.push_back(rn);
            }
        } else {
            build(bestSplit + 1, r, out);
        }
        node.left = (int32_t)leftRoot;
        node.right = (int32_t)rightRoot;
        out[id] = node;
    }

    // Large ranges of the uncapped builder: 16 equal chunks sorted on their own threads, then merged
    // pairwise.  The chunking does not depend on the machine, so the order of equal centroids (and
    // with it the tree) does not either.  The capped builder keeps the single std::sort whose handling
    // of equal keys the reference's tree depends on.
    template <class Less>
    static void chunkedSort(Key* first, Key* last, Less less)
    {
        constexpr int kChunks = 16;
        const size_t n = (size_t)(last - first);
        size_t edge[kChunks + 1];
        for (int k = 0; k <= kChunks; k++) edge[k] = n * (size_t)k / kChunks;
        std::vector<std::thread> pool;
        for (int k = 0; k < kChunks; k++) pool.emplace_back([=] { std::sort(first + edge[k], first + edge[k + 1], less); });
        for (std::thread& t : pool) t.join();
        for (int width = 1; width < kChunks; width *= 2) {
            pool.clear();
            for (int k = 0; k + width < kChunks; k += 2 * width) {
                const int hi = k + 2 * width < kChunks ? k + 2 * width : kChunks;
                pool.emplace_back([=] { std::inplace_merge(first + edge[k], first + edge[k + width], first + edge[hi], less); });
            }
            for (std::thread& t : pool) t.join();
        }
    }

    void sortAxis(int l, int r, int axis) const
    {
        Key* first = keys.data() + l;
        Key* last = keys.data() + r + 1;
        if (kInf > kRefInf && r - l + 1 >= (1 << 18)) {
            switch (axis) {
            case 0: chunkedSort(first, last, [](const Key& a, const Key& b) { return a.c[0] < b.c[0]; }); break;
            case 1: chunkedSort(first, last, [](const Key& a, const Key& b) { return a.c[1] < b.c[1]; }); break;
            default: chunkedSort(first, last, [](const Key& a, const Key& b) { return a.c[2] < b.c[2]; }); break;
            }
            return;
        }
        switch (axis) {
        case 0: std::sort(first, last, [](const Key& a, const Key& b) { return a.c[0] < b.c[0]; }); break;
        case 1: std::sort(first, last, [](const Key& a, const Key& b) { return a.c[1] < b.c[1]; }); break;
        default: std::sort(first, last, [](const Key& a, const Key& b) { return a.c[2] < b.c[2]; }); break;
        }
    }
};

}  // namespace

// Local ids inside `build` are relative to the vector the sub-tree was built into.  When a
// sub-tree is built in place (same vector as its parent) its "local" ids are already the
// parent's ids, so only spliced sub-trees are shifted.  The top-level call builds into a
// vector that already holds the dummy node 0, which makes the root id 1 like the reference.
size_t BuildBvhWithSah(RefPrimitive* prims, size_t count, int leafSize, std::vector<RefNode>& nodes, int threads,
                       bool uncapped)
{
    nodes.clear();
    nodes.push_back(RefNode{});  // zeroed dummy node 0 (PathTracingRenderer.cpp:575-576)
    if (count == 0) return nodes.size();  // BuildBvhWithSah returns 0 for l > r (BVH.cpp:93)

    std::vector<Key> keys(count);
    std::vector<Bounds> bounds(count);
    for (size_t i = 0; i < count; i++) {
        const RefPrimitive& p = prims[i];
        for (int k = 0; k < 3; k++) {
            keys[i].c[k] = ((p.posA[k] + p.posB[k]) + p.posC[k]) / 3.0f;
            bounds[i].lo[k] = gmin(p.posA[k], gmin(p.posB[k], p.posC[k]));
            bounds[i].hi[k] = gmax(p.posA[k], gmax(p.posB[k], p.posC[k]));
        }
        keys[i].id = (uint32_t)i;
    }
    if (threads <= 0) {
        // PT_HOST_THREADS: a caller that runs several processes on one host (one per GPU) gives each its share
        const char* env = std::getenv("PT_HOST_THREADS");
        threads = env ? std::atoi(env) : (int)std::thread::hardware_concurrency();
        if (threads > 16) threads = 16;
        if (threads < 1) threads = 1;
    }
    std::atomic<int> threadsLeft(threads - 1);
    Builder b{bounds, keys, leafSize, 4096, threadsLeft, uncapped ? 3.402823466e38f : kRefInf};
    b.build(0, (int)count - 1, nodes);

    std::vector<RefPrimitive> sorted(count);
    for (size_t i = 0; i < count; i++) sorted[i] = prims[keys[i].id];
    std::memcpy(prims, sorted.data(), sizeof(RefPrimitive) * count);
    return nodes.size();
}

void AssemblePrimitives(const RefVertex* v, const void* indices, size_t indexCount, int indexBytes,
                        uint32_t meshIndex, RefPrimitive* out)
{
    auto at = [&](size_t i) -> uint32_t {
        switch (indexBytes) {
        case 1: return ((const uint8_t*)indices)[i];
        case 2: return ((const uint16_t*)indices)[i];
        default: return ((const uint32_t*)indices)[i];
        }
    };
    for (size_t i = 0, t = 0; i + 2 < indexCount; i += 3, t++) {
        const RefVertex& a = v[at(i)];
        const RefVertex& b = v[at(i + 1)];
        const RefVertex& c = v[at(i + 2)];
        RefPrimitive p{};
        std::memcpy(p.posA, a.pos, 12);
        std::memcpy(p.posB, b.pos, 12);
        std::memcpy(p.posC, c.pos, 12);
        std::memcpy(p.norA, a.nor, 12);
        std::memcpy(p.norB, b.nor, 12);
        std::memcpy(p.norC, c.nor, 12);
        std::memcpy(p.uvA, a.uv, 8);
        std::memcpy(p.uvB, b.uv, 8);
        std::memcpy(p.uvC, c.uv, 8);
        p.meshIndex = meshIndex;
        out[t] = p;
    }
}

void AlignNodes(const RefNode* in, size_t count, RefNodeAlign* out)
{
    for (size_t i = 0; i < count; i++) {
        RefNodeAlign a{};
        a.left = in[i].left;
        a.right = in[i].right;
        a.n = in[i].n;
        a.index = in[i].index;
        std::memcpy(a.AA, in[i].AA, 12);
        std::memcpy(a.BB, in[i].BB, 12);
        out[i] = a;
    }
}

void AlignPrimitives(const RefPrimitive* in, size_t count, RefPrimitiveAlign* out)
{
    for (size_t i = 0; i < count; i++) {
        const RefPrimitive& p = in[i];
        RefPrimitiveAlign a{};
        std::memcpy(a.posAuvX, p.posA, 12);
        a.posAuvX[3] = p.uvA[0];
        std::memcpy(a.norAuvY, p.norA, 12);
        a.norAuvY[3] = p.uvA[1];
        std::memcpy(a.posBuvX, p.posB, 12);
        a.posBuvX[3] = p.uvB[0];
        std::memcpy(a.norBuvY, p.norB, 12);
        a.norBuvY[3] = p.uvB[1];
        std::memcpy(a.posCuvX, p.posC, 12);
        a.posCuvX[3] = p.uvC[0];
        std::memcpy(a.norCuvY, p.norC, 12);
        a.norCuvY[3] = p.uvC[1];
        a.meshIndex = p.meshIndex;
        out[i] = a;
    }
}

}  // namespace ptcore
