// On-disk cache of a built BVH.  The reference rebuilds its tree at every start
// (PathTracingRenderer.cpp:566-613 calls BVH::BuildBvhWithSah from ProcessBvhNodeBuffer); for the
// multi-million-triangle scenes this library is also meant for, that sweep takes seconds to minutes,
// so the aligned node and primitive arrays can be stored next to the scene and reloaded.  The file
// is keyed by a hash of the unsorted input soup: a stale cache is detected, never used.
//
// File layout (little endian): char[8] "PTBVH\0\0\1", u64 key, u64 node count, u64 primitive count,
// u64 FNV-1a-64 of the payload, then the BvhNodeAlign[] (48 B each) and BvhPrimitiveAlign[] (112 B each).
#include <cstdio>
#include <cstring>

#include "host.h"

namespace ptcore {

uint64_t Fnv1a64(const void* data, size_t bytes, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

namespace {
constexpr char kMagic[8] = {'P', 'T', 'B', 'V', 'H', 0, 0, 1};
struct Header {
    char magic[8];
    uint64_t key, nodeCount, primCount, payloadHash;
};
}  // namespace

std::string SaveBvhCache(const std::string& path, uint64_t key, const RefNodeAlign* nodes, size_t nodeCount,
                         const RefPrimitiveAlign* prims, size_t primCount)
{
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.key = key;
    h.nodeCount = nodeCount;
    h.primCount = primCount;
    h.payloadHash = Fnv1a64(prims, primCount * sizeof(RefPrimitiveAlign), Fnv1a64(nodes, nodeCount * sizeof(RefNodeAlign)));
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return "cannot write '" + tmp + "'";
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 &&
                    (nodeCount == 0 || std::fwrite(nodes, sizeof(RefNodeAlign), nodeCount, f) == nodeCount) &&
                    (primCount == 0 || std::fwrite(prims, sizeof(RefPrimitiveAlign), primCount, f) == primCount);
    const bool closed = std::fclose(f) == 0;
    if (!ok || !closed) {
        std::remove(tmp.c_str());
        return "short write to '" + tmp + "'";
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        return "cannot rename '" + tmp + "' to '" + path + "'";
    }
    return std::string();
}

std::string LoadBvhCache(const std::string& path, uint64_t key, std::vector<RefNodeAlign>& nodes,
                         std::vector<RefPrimitiveAlign>& prims)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "no cache file '" + path + "'";
    Header h{};
    std::string err;
    if (std::fread(&h, sizeof(h), 1, f) != 1 || std::memcmp(h.magic, kMagic, 8) != 0) {
        err = "'" + path + "' is not a BVH cache file of this version";
    } else if (h.key != key) {
        err = "'" + path + "' was built from different primitives (stale cache)";
    } else if (h.nodeCount >= (1ull << 31) || h.primCount >= (1ull << 27)) {
        err = "'" + path + "' holds implausible counts";
    } else {
        nodes.resize((size_t)h.nodeCount);
        prims.resize((size_t)h.primCount);
        if ((h.nodeCount && std::fread(nodes.data(), sizeof(RefNodeAlign), nodes.size(), f) != nodes.size()) ||
            (h.primCount && std::fread(prims.data(), sizeof(RefPrimitiveAlign), prims.size(), f) != prims.size()))
            err = "'" + path + "' is truncated";
        else if (std::fgetc(f) != EOF)
            err = "'" + path + "' has trailing bytes";
        else if (Fnv1a64(prims.data(), prims.size() * sizeof(RefPrimitiveAlign),
                         Fnv1a64(nodes.data(), nodes.size() * sizeof(RefNodeAlign))) != h.payloadHash)
            err = "'" + path + "' is corrupt (payload hash mismatch)";
    }
    std::fclose(f);
    if (!err.empty()) {
        nodes.clear();
        prims.clear();
    }
    return err;
}

}  // namespace ptcore
