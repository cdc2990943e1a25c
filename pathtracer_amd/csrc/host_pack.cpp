// Reference-layout scene buffers -> this library's packed device layouts, with the
// validation the kernels rely on (every index in range, the node graph a tree, leaf ranges
// inside the primitive array).  A malformed buffer is rejected here instead of faulting on
// the GPU.
//
// Ray-independent terms of the reference's per-test triangle maths
// (Shaders/Library/intersection.glsl:23-25,48) are evaluated once here with the same float
// expressions, so the kernel's results stay bit-identical while it loads 48 B per test
// instead of 112 B.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "host.h"

namespace ptcore {
namespace {

inline void Cross(const float x[3], const float y[3], float o[3])
{
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}

constexpr int kMaxLeaf = 8;  // leaf reference encodes (count - 1) in 3 bits

}  // namespace

std::string PackScene(const RefNodeAlign* nodes, size_t nodeCount, const RefPrimitiveAlign* prims,
                      size_t primCount, size_t materialCount, PackedScene& out, int leafCut)
{
    out = PackedScene{};
    if (nodeCount < 2) return "BVH needs at least the dummy node 0 and a root node 1";
    if (primCount == 0) return "no primitives";
    if (primCount >= (1u << 27)) return "too many primitives (limit 2^27)";

    // ---- triangles
    out.triangles.resize(primCount);
    out.shading.resize(primCount);
    for (size_t i = 0; i < primCount; i++) {
        const RefPrimitiveAlign& p = prims[i];
        if (p.meshIndex >= materialCount)
            return "primitive " + std::to_string(i) + " has meshIndex " + std::to_string(p.meshIndex) +
                   " but only " + std::to_string(materialCount) + " materials are uploaded";
        DevTriangle t{};
        float ab[3], ac[3], n[3];
        for (int k = 0; k < 3; k++) {
            ab[k] = p.posBuvX[k] - p.posAuvX[k];
            ac[k] = p.posCuvX[k] - p.posAuvX[k];
        }
        Cross(ab, ac, n);
        std::memcpy(t.a, p.posAuvX, 12);
        std::memcpy(t.ab, ab, 12);
        std::memcpy(t.ac, ac, 12);
        t.n0 = n[0];
        t.n1 = n[1];
        t.n2 = n[2];
        out.triangles[i] = t;

        DevShading s{};
        std::memcpy(s.norA, p.norAuvY, 12);
        std::memcpy(s.norB, p.norBuvY, 12);
        std::memcpy(s.norC, p.norCuvY, 12);
        s.uvAx = p.posAuvX[3];
        s.uvAy = p.norAuvY[3];
        s.uvBx = p.posBuvX[3];
        s.uvBy = p.norBuvY[3];
        s.uvCx = p.posCuvX[3];
        s.uvCy = p.norCuvY[3];
        s.meshIndex = p.meshIndex;
        float g[3];
        Cross(ac, ab, g);  // geometryNormal = normalize(cross(edgeAC, edgeAB))
        const float inv = 1.0f / std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        s.geoN[0] = g[0] * inv;
        s.geoN[1] = g[1] * inv;
        s.geoN[2] = g[2] * inv;
        out.shading[i] = s;
    }

    // ---- nodes: iterative pre-order walk from the root (node 1)
    auto leafRef = [&](const RefNodeAlign& n, size_t id, std::string& err) -> int32_t {
        if (n.n > kMaxLeaf) {
            err = "leaf node " + std::to_string(id) + " holds " + std::to_string(n.n) + " primitives (limit 8)";
            return 0;
        }
        if (n.index < 0 || (size_t)n.index + (size_t)n.n > primCount) {
            err = "leaf node " + std::to_string(id) + " range [" + std::to_string(n.index) + ", +" +
                  std::to_string(n.n) + ") is outside the primitive array";
            return 0;
        }
        return ~(int32_t)(((uint32_t)n.index << 3) | (uint32_t)(n.n - 1));
    };

    // A validated leaf range, cut into leaves of at most leafCut triangles when it holds more: returns the reference of
    // the range (a leaf, or the interior node on top of its halves) and reports the levels it added.
    auto rangeBox = [&](size_t first, size_t count, float lo[3], float hi[3]) {
        for (int k = 0; k < 3; k++) lo[k] = std::numeric_limits<float>::max(), hi[k] = -std::numeric_limits<float>::max();
        for (size_t i = first; i < first + count; i++)
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], std::min(prims[i].posAuvX[k], std::min(prims[i].posBuvX[k], prims[i].posCuvX[k])));
                hi[k] = std::max(hi[k], std::max(prims[i].posAuvX[k], std::max(prims[i].posBuvX[k], prims[i].posCuvX[k])));
            }
    };
    std::function<int32_t(size_t, size_t, int&)> cutRange = [&](size_t first, size_t count, int& levels) -> int32_t {
        if (leafCut <= 0 || count <= (size_t)leafCut) {
            levels = 0;
            return ~(int32_t)(((uint32_t)first << 3) | (uint32_t)(count - 1));
        }
        const size_t leftCount = (count + 1) / 2, rightCount = count - leftCount;
        const int32_t at = (int32_t)out.nodes.size();
        out.nodes.push_back(DevNode{});
        int levelsL = 0, levelsR = 0;
        DevNode d{};
        rangeBox(first, leftCount, d.lmin, d.lmax);
        rangeBox(first + leftCount, rightCount, d.rmin, d.rmax);
        d.left = cutRange(first, leftCount, levelsL);
        d.right = cutRange(first + leftCount, rightCount, levelsR);
        out.nodes[(size_t)at] = d;
        levels = 1 + std::max(levelsL, levelsR);
        return at;
    };

    std::string err;
    const RefNodeAlign& root = nodes[1];
    if (root.n > 0 && leafCut > 0 && root.n > leafCut) {  // a root leaf that is cut becomes a small tree of its own
        out.nodes.push_back(DevNode{});
        (void)leafRef(root, 1, err);
        if (!err.empty()) return err;
        int levels = 0;
        out.rootRef = cutRange((size_t)root.index, (size_t)root.n, levels);
        out.maxDepth = levels;
        return err;
    }
    if (root.n > 0) {
        out.nodes.push_back(DevNode{});  // node 0 must exist: idle lanes fetch it
        out.rootRef = leafRef(root, 1, err);
        out.maxDepth = 0;
        return err;
    }
    std::vector<uint8_t> seen(nodeCount, 0);
    struct Item {
        int32_t ref;   // reference node id
        int32_t dev;   // device node index
        int32_t depth;
    };
    std::vector<Item> stack;
    out.nodes.reserve(nodeCount / 2 + 1);
    out.nodes.push_back(DevNode{});  // index 0 stays unused: the kernels read reference 0 as "no node"
    out.nodes.push_back(DevNode{});
    stack.push_back({1, 1, 1});
    seen[1] = 1;
    out.rootRef = 1;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        const RefNodeAlign& n = nodes[it.ref];
        if (it.depth > out.maxDepth) out.maxDepth = it.depth;
        int32_t child[2] = {n.left, n.right};
        int32_t refs[2];
        Item pending[2];
        int npending = 0;
        for (int c = 0; c < 2; c++) {
            if (child[c] <= 0 || (size_t)child[c] >= nodeCount)
                return "interior node " + std::to_string(it.ref) + " has child index " + std::to_string(child[c]) +
                       " outside [1, " + std::to_string(nodeCount) + ")";
            if (seen[child[c]]) return "node " + std::to_string(child[c]) + " is reachable twice: not a tree";
            seen[child[c]] = 1;
            const RefNodeAlign& ch = nodes[child[c]];
            if (ch.n > 0) {
                refs[c] = leafRef(ch, (size_t)child[c], err);
                if (!err.empty()) return err;
                int levels = 0;
                refs[c] = cutRange((size_t)ch.index, (size_t)ch.n, levels);
                if (it.depth + levels > out.maxDepth) out.maxDepth = it.depth + levels;
            } else {
                refs[c] = (int32_t)out.nodes.size();
                out.nodes.push_back(DevNode{});
                pending[npending++] = {child[c], refs[c], it.depth + 1};
            }
        }
        DevNode d{};
        const RefNodeAlign& l = nodes[n.left];
        const RefNodeAlign& r = nodes[n.right];
        std::memcpy(d.lmin, l.AA, 12);
        std::memcpy(d.lmax, l.BB, 12);
        std::memcpy(d.rmin, r.AA, 12);
        std::memcpy(d.rmax, r.BB, 12);
        d.left = refs[0];
        d.right = refs[1];
        out.nodes[it.dev] = d;
        // push right first so the left sub-tree is laid out next (pre-order, left-adjacent)
        for (int k = npending - 1; k >= 0; k--) stack.push_back(pending[k]);
    }
    return std::string();
}

std::string PackMaterials(const RefMaterialAlign* mats, size_t count, size_t textureSlots,
                          const std::vector<int>& slotToTexture, std::vector<DevMaterial>& out)
{
    out.resize(count);
    for (size_t i = 0; i < count; i++) {
        const RefMaterialAlign& m = mats[i];
        DevMaterial d{};
        std::memcpy(d.emission, m.emission, 12);
        std::memcpy(d.albedo, m.albedo, 12);
        d.specular = m.specular;
        d.specularTint = m.specularTint;
        d.metallic = m.metallic;
        d.roughness = m.roughness;
        d.subsurface = m.subsurface;
        d.anisotropy = m.anisotropy;
        d.sheenRoughness = m.sheenRoughness;
        d.sheenTint = m.sheenTint;
        d.clearcoat = m.clearcoat;
        d.clearcoatRoughness = m.clearcoatRoughness;
        d.refraction = m.refraction;
        d.transmission = m.transmission;
        d.anyTexture = 0;
        for (int s = 0; s < 12; s++) d.tex[s] = -1;
        for (int s = 0; s < kTexSlotCount; s++) {
            const uint64_t h = m.textureHandle[s];
            if (h == 0) continue;
            if (h >= textureSlots || slotToTexture[(size_t)h] < 0)
                return "material " + std::to_string(i) + " references texture slot " + std::to_string(h) +
                       " which was never uploaded";
            d.tex[s] = slotToTexture[(size_t)h];
            d.anyTexture = 1;
        }
        out[i] = d;
    }
    return std::string();
}

}  // namespace ptcore
