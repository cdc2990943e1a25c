// libptcore.so: the C ABI declared in include/ptcore.h.
//
// Host-side plumbing only: argument checks, copies of the caller's reference-layout buffers,
// repacking into the device layouts (host_pack.cpp), uploads, launches.  There is no CPU
// rendering path in this library: without a HIP device pt_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "host.h"
#include "kernels.h"
#include "lbvh.h"
#include "reinsert.h"

#include <hipcub/hipcub.hpp>

using namespace ptcore;

namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorName(e_) + " (" +        \
                                           hipGetErrorString(e_) + ")");                               \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t capacity = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= capacity && ptr) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        capacity = 0;
        hipError_t e = hipMalloc(&ptr, bytes ? bytes : 16);
        if (e == hipSuccess) capacity = bytes ? bytes : 16;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        capacity = 0;
    }
};

// A buffer that lives as long as its scope (error returns included).
struct ScopedBuffer : DeviceBuffer {
    ScopedBuffer() = default;
    ScopedBuffer(const ScopedBuffer&) = delete;
    ScopedBuffer& operator=(const ScopedBuffer&) = delete;
    ~ScopedBuffer() { release(); }
};

struct HostTexture {
    int width = 0, height = 0, channels = 0, format = 0;
    std::vector<unsigned char> bytes;
    bool present = false;
};

struct PendingModel {
    std::vector<RefVertex> vertices;
    std::vector<uint32_t> indices;
    RefMaterialAlign material;
};

// Scratch of pt_dispatch_static: the one entry point that allocates nothing.  Every GPU the library
// runs on holds its own copy of the array, at its own address, until the process ends; host code
// reaches it only through the per-device pointer taken once in pt_create (the array's name is a
// host-side stand-in, not a device address).
constexpr size_t kStaticScratchBytes = PT_STATIC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;
constexpr int kLaunchLanes = 3;  // dispatch calls whose launches may be in flight together
void* g_scratchPtr[kMaxDevices] = {};

}  // namespace

static __device__ __align__(256) unsigned char scratch[PT_STATIC_SCRATCH_BYTES];

struct pt_ctx {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipEvent_t evTrace0 = nullptr, evTrace1 = nullptr, evResolve1 = nullptr;  // around the last megakernel launch
    bool kernelTimed = false;
    bool timed = false;

    // caller-supplied reference-layout buffers (copied at upload, like glBufferData)
    std::vector<RefNodeAlign> nodes;
    std::vector<RefPrimitiveAlign> prims;
    std::vector<RefMaterialAlign> materials;
    std::vector<HostTexture> textures;  // index = slot
    std::vector<float> envColorPdf;     // float4 per texel
    std::vector<float> envSample;       // float2 per texel
    int envW = 0, envH = 0;
    bool geometryDirty = true, materialsDirty = true, texturesDirty = true, envDirty = true;
    bool haveParams = false;
    pt_params params{};
    std::vector<PendingModel> models;

    // device side
    DeviceBuffer dNodes4;
    int wideLevels = 0;
    bool foldedLevelByLevel = false;  // folding by surface area left the tree too deep
    DeviceBuffer dNodes, dTriangles, dShading, dMaterials, dTextures, dTexels, dSrgbLut, dEnvColorPdf, dEnvSample;
    DeviceBuffer dImages[3];
    DeviceBuffer dCounters[kLaunchLanes];  // one set per launch lane (a call's launches may still run when the next calls start)
    int counterSet = 0;         // the set of the most recent dispatch call: its lane's
    DeviceBuffer dRgbStaging;  // pt_read_image_rgb
    DeviceBuffer dWaveTimes;   // counting dispatches: start / end ticks per wave
    // What a megakernel launch owns besides the scene and the images, in several sets.  Dispatch calls of few frames
    // take the sets in turn and trace on the set's own stream, so that the waves of one call's launch
    // fill the chip while the last waves of the call before run dry (with one sample per pixel that tail is two
    // fifths of a launch).  The resolve passes all run on the context's stream, in call order; everything else the
    // library does is on that stream too and therefore behind every trace launched before it.
    struct LaunchLane {
        // launch plan from the ticks each tile's waves lived in the lane's previous launch: items per tile, per-item
        // cost (the sort key), tiles in descending key order, item list
        DeviceBuffer dTileCost, dTileParts, dTileKey, dTileKeySorted, dTileIds, dTileOrder, dTileSortTemp;
        DeviceBuffer dPartsInOrder, dBlockStart, dBlockList, dBlockCount, dScanTemp;
        size_t tileCostTiles = 0, tileSortBytes = 0, scanBytes = 0, blockCapacity = 0;
        bool tileCostValid = false;
        // the item list on record: made for launches of this many frames, used by this many launches since (0: none)
        uint32_t planFrames = 0, planUses = 0;
        size_t planGrid = 0;
        // finished samples of one launch, 36 B per (sample, pixel); resolved into the images after the launch
        DeviceBuffer dSamples;
        hipStream_t stream = nullptr;    // traces of overlapping calls
        hipEvent_t evTraced = nullptr;   // on `stream`, behind the trace kernel
        hipEvent_t evFree = nullptr;     // on the context's stream, behind the resolve pass that read the lane's samples
        bool used = false;               // evFree has been recorded
        uint64_t forkSeen = 0;           // the last evFork `stream` has waited for
    } lane[kLaunchLanes];
    int lastLane = 0;
    int overlap = -1;            // calls overlap: never (0), whenever allowed (1), calls of at most 8 frames (-1)
    hipEvent_t evFork = nullptr; // on the context's stream, behind the last work other than dispatches that the traces must see
    uint64_t forkCount = 0;
    bool forkDirty = true;
    bool tileOrdering = true;
    int overlapShare = -1; // overlapping calls launch slots / overlapShare waves each (0: all slots; -1: 2 for calls of at most two frames, else all)
    int planEvery = 8;   // a launch lane re-plans on every planEvery-th launch and reuses the list in between
    int tileSplit = -1;  // no block may cost more than 1/tileSplit of the launch's lower bound (0: never split, -1: by image size)
    uint32_t computeUnits = 256;
    std::vector<std::pair<void (*)(DevParams, DevScene, Images, DevCounters*), uint32_t>> residentWaves;  // per trace kernel variant
    uint32_t lastSlots = 0;  // resident waves of the most recent megakernel launch's variant
    uint64_t sampleBudgetBytes = 8ull << 30;
    int wavesPerSimd = 0;      // megakernel variant: 3 or 4 waves per SIMD (0: the default, 3)
    bool rebuildOnUpload = true;  // pt_upload_nodes' tree is checked, then replaced by the device builder's over the same primitives
    bool builderChosen = false;   // the tree on record comes from pt_process_buffers / _full_sweep: the caller named its builder, it stays
    int bvhLeafMax = 0;   // device BVH builder: most triangles per leaf (0: by scene size)
    int leafRecut = 0;    // a traversed caller's tree: leaves of more triangles than this are cut into leaves of at most this many (0: kept)
    int bvhOptimise = -1;  // device BVH builder: reinsertion passes over the clustered tree (reinsert.h; 0: none, -1: by scene size)
    int bvhOptimiseMaxHeight = 60;  // ... a tree the passes left deeper than this is put back as the clustering built it
    int bvhOptimiseMod = 2;  // ... every how-manyth node is a candidate in a pass (the offset advances pass by pass)
    size_t textureDecodeBytes = 256u << 20;  // 8-bit textures are kept as decoded floats while they fit in this many bytes in all (the Infinity Cache's size)
    int plocRadius = 32;  // device BVH builder: neighbours searched on either side per clustering round (0: Morton radix tree)
    bool pullItems = true;  // megakernel: resident waves pull the work items (else one block per item)
    size_t waveTimeCount = 0;
    // geometry built on the device (pt_build_bvh_device): the tree in the reference's layouts, kept for
    // pt_download_bvh; dNodes / dTriangles / dShading are filled by the builder itself
    DeviceBuffer dRefNodes, dRefPrims;
    bool deviceGeometry = false;
    size_t deviceNodeCount = 0, devicePrimCount = 0;
    uint32_t deviceMaxMeshIndex = 0;
    float4* bound[3] = {nullptr, nullptr, nullptr};
    int32_t rootRef = 0;
    int maxDepth = 0;
    uint32_t travExitNum = 28;  // of 64
    uint32_t refillMin = 16;
    uint32_t hitMinNum = 28;
    int pipeline = 0;    // 0 = megakernel, 1 = wavefront
    int checkEvery = 2;  // wavefront: read the live-ray count back every this many iterations
    int traverseBlocks = 0;
    int lastIterations = 0;

    // wavefront path state (one slot per pixel) and queues
    DeviceBuffer dWfState[10], dRayQ[2], dShadeQ, dCtrl;
    unsigned* hCtrl = nullptr;  // pinned mirror of one control word
    size_t nodeBytes = 0, triangleBytes = 0, shadingBytes = 0;
};

namespace {

struct DeviceGuard {
    int previous = -1;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        if (previous != device) (void)hipSetDevice(device);
    }
    ~DeviceGuard()
    {
        if (previous >= 0) (void)hipSetDevice(previous);
    }
};

void forgetTileCosts(pt_ctx* c)
{
    for (pt_ctx::LaunchLane& L : c->lane) L.tileCostValid = false, L.planFrames = 0;
}

float4* imagePtr(pt_ctx* c, int which) { return c->bound[which] ? c->bound[which] : (float4*)c->dImages[which].ptr; }

int buildLbvh(pt_ctx* c, const RefPrimitiveAlign* hostPrims, size_t count);

// Makes the four-wide tree the kernels walk (dNodes4) of the binary one in dNodes, level by level from the root,
// and drops the binary one.  A query's stack holds at most three entries per level of the wide tree.
constexpr int kStackLds = 48;      // megakernel: entries in LDS (40 ... 48 measured alike: twelve waves per CU fit either way) ...
constexpr int kStackPrivate = 48;  // ... and, for deeper trees, in lane-private memory
constexpr int kStackWavefront = 63;  // wavefront pipeline: all in LDS (64 rows of 256 lanes = 64 KB, next to 2 KB of its other tables)
int collapseTree(pt_ctx* c, size_t nodeCount)
{
    c->wideLevels = 0;
    c->nodeBytes = 2 * sizeof(DevNodeQ);
    HIP_TRY(c->dNodes4.ensure((nodeCount + 2) * sizeof(DevNodeQ)));
    HIP_TRY(hipMemsetAsync(c->dNodes4.ptr, 0, 2 * sizeof(DevNodeQ), c->stream));  // node 0: idle lanes fetch it
    if (c->rootRef <= 0) {  // the root is a leaf
        c->dNodes.release();
        return PT_OK;
    }
    if ((nodeCount + 2) * sizeof(DevNodeQ) >= (1ull << 32))
        return fail(PT_ERR_SCENE, "node array of 4 GB or more (the kernels address records by 32-bit byte offsets)");
    ScopedBuffer frontier[2], counters;
    HIP_TRY(frontier[0].ensure(nodeCount * sizeof(int2)));
    HIP_TRY(frontier[1].ensure(nodeCount * sizeof(int2)));
    HIP_TRY(counters.ensure(8));
    unsigned wideCount = 0;
    // by surface area first; should that leave the tree deeper than the stacks hold, both children of every node are
    // opened instead, which at least halves the binary tree's depth (at most 64)
    for (int uniform = 0; uniform < 2; uniform++) {
        const int2 root = make_int2(c->rootRef, 1);
        unsigned h[2] = {2u, 0u};  // next free wide node (0: the dummy, 1: the root); size of the next frontier
        HIP_TRY(hipMemcpyAsync(frontier[0].ptr, &root, sizeof(root), hipMemcpyHostToDevice, c->stream));
        unsigned count = 1;
        int cur = 0, levels = 0;
        while (count > 0) {
            HIP_TRY(hipMemcpyAsync(counters.ptr, h, 8, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(collapseKernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, (const DevNode*)c->dNodes.ptr,
                               (DevNodeQ*)c->dNodes4.ptr, (const int2*)frontier[cur].ptr, count, (int2*)frontier[cur ^ 1].ptr,
                               (unsigned*)counters.ptr, uniform);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h, counters.ptr, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            count = h[1];
            h[1] = 0;
            cur ^= 1;
            levels++;
        }
        c->wideLevels = levels;
        c->foldedLevelByLevel = uniform != 0;
        wideCount = h[0];
        // (with the wavefront pipeline selected at upload time its shorter stack decides; selected later, a tree folded
        // to between 22 and 32 levels is refused by its dispatch)
        if (3 * levels <= (c->pipeline == 1 ? kStackWavefront : kStackLds + kStackPrivate)) break;
    }
    if (3 * c->wideLevels > kStackLds + kStackPrivate)
        return fail(PT_ERR_SCENE, "BVH too deep: " + std::to_string(c->wideLevels) + " levels of four-wide nodes, the traversal stack holds " +
                                      std::to_string((kStackLds + kStackPrivate) / 3));
    c->nodeBytes = (size_t)wideCount * sizeof(DevNodeQ);
    c->dNodes.release();
    return PT_OK;
}

int syncScene(pt_ctx* c)
{
    if (!c->deviceGeometry && (c->nodes.empty() || c->prims.empty()))
        return fail(PT_ERR_STATE, "no BVH nodes / primitives uploaded");
    if (c->materials.empty()) return fail(PT_ERR_STATE, "no materials uploaded");
    if (c->envW == 0) return fail(PT_ERR_STATE, "no environment uploaded");
    if (!c->haveParams) return fail(PT_ERR_STATE, "pt_set_params has not been called");

    if (c->deviceGeometry && c->deviceMaxMeshIndex >= c->materials.size())
        return fail(PT_ERR_SCENE, "a primitive has meshIndex " + std::to_string(c->deviceMaxMeshIndex) + " but only " +
                                      std::to_string(c->materials.size()) + " materials are uploaded");
    if (c->geometryDirty && !c->deviceGeometry) {
        PackedScene packed;
        const bool rebuilt = c->rebuildOnUpload && !c->builderChosen;  // (then the uploaded tree is only validated here)
        std::string err = PackScene(c->nodes.data(), c->nodes.size(), c->prims.data(), c->prims.size(),
                                    c->materials.size(), packed, rebuilt ? 0 : c->leafRecut);
        if (!err.empty()) return fail(PT_ERR_SCENE, err);
        if (c->rebuildOnUpload && !c->builderChosen) {
            // the uploaded tree was only validated; the device builder's tree over the same primitives is used
            std::vector<RefPrimitiveAlign> prims;
            std::vector<RefNodeAlign> nodes;
            prims.swap(c->prims);  // buildLbvh drops the host copies
            nodes.swap(c->nodes);
            const int rc = buildLbvh(c, prims.data(), prims.size());
            if (rc != PT_OK) {  // keep what the caller uploaded: the next dispatch fails the same way, not with "nothing uploaded"
                c->prims.swap(prims);
                c->nodes.swap(nodes);
                return rc;
            }
            return syncScene(c);
        }
        if (packed.maxDepth > 64)
            return fail(PT_ERR_SCENE, "BVH is " + std::to_string(packed.maxDepth) + " interior levels deep; at most 64 are taken");
        const size_t nb = packed.nodes.size() * sizeof(DevNode), tb = packed.triangles.size() * sizeof(DevTriangle),
                     sb = packed.shading.size() * sizeof(DevShading);
        if (nb >= (1ull << 32) || tb >= (1ull << 32))
            return fail(PT_ERR_SCENE, "node or triangle array of 4 GB or more (the kernels address records by 32-bit byte offsets)");
        HIP_TRY(c->dNodes.ensure(nb));
        HIP_TRY(c->dTriangles.ensure(tb));
        HIP_TRY(c->dShading.ensure(sb));
        if (nb) HIP_TRY(hipMemcpyAsync(c->dNodes.ptr, packed.nodes.data(), nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dTriangles.ptr, packed.triangles.data(), tb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dShading.ptr, packed.shading.data(), sb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // `packed` dies at the end of this scope
        c->rootRef = packed.rootRef;
        c->maxDepth = packed.maxDepth;
        {
            const int rcw = collapseTree(c, packed.nodes.size());  // sets nodeBytes
            if (rcw != PT_OK) return rcw;
        }
        c->triangleBytes = tb;
        c->shadingBytes = sb;
        c->geometryDirty = false;
        c->forkDirty = true;
    }
    if (c->texturesDirty) {
        std::vector<DevTexture> table;
        std::vector<unsigned char> arena;
        // 8-bit channel -> float: entries 0..255 unsigned-normalised (b / 255, one correctly rounded binary32
        // division), entries 256..511 the GL_SRGB8 decode
        float lut[512];
        for (int i = 0; i < 256; i++) {
            lut[i] = (float)i / 255.0f;
            double sd = i / 255.0;
            lut[256 + i] = (float)(sd <= 0.04045 ? sd / 12.92 : std::pow((sd + 0.055) / 1.055, 2.4));
        }
        // Every texel goes into the arena with four channels -- (r, 0, 0, 1), (r, g, 0, 1), (r, g, b, 1) for
        // textures of fewer, which is what a sampler returns for them -- so that a tap is ONE read.  8-bit textures
        // are decoded here, once, into four floats a texel (the table above; alpha is never sRGB) as long as all of
        // them together stay within `textureDecodeBytes`: a tap is then 16 bytes and needs no table lookups behind it.
        // Beyond that they stay 8-bit, a tap is a 32-bit word and the kernels look the four channels up in the table.
        size_t decoded = 0;
        for (const HostTexture& t : c->textures)
            if (t.present && t.format != 2) decoded += (size_t)t.width * (size_t)t.height * 16;
        const bool decode = decoded <= c->textureDecodeBytes;
        for (const HostTexture& t : c->textures) {
            if (!t.present) continue;
            DevTexture d{};
            while (arena.size() % 16) arena.push_back(0);
            d.offset = arena.size();
            d.width = t.width;
            d.height = t.height;
            d.channels = t.channels;
            d.format = (t.format != 2 && decode) ? 2 : t.format;
            const size_t texels = (size_t)t.width * (size_t)t.height;
            const size_t at = arena.size();
            if (t.format == 2 || decode) {
                arena.resize(at + texels * 16);
                float* o = reinterpret_cast<float*>(arena.data() + at);
                const float* sf = reinterpret_cast<const float*>(t.bytes.data());
                const unsigned char* sb = t.bytes.data();
                const float* colour = lut + (t.format == 1 ? 256 : 0);
                for (size_t i = 0; i < texels; i++)
                    for (int ch = 0; ch < 4; ch++) {
                        float v = ch == 3 ? 1.0f : 0.0f;
                        if (ch < t.channels) {
                            const size_t k = i * (size_t)t.channels + (size_t)ch;
                            v = t.format == 2 ? sf[k] : (ch == 3 ? lut[sb[k]] : colour[sb[k]]);
                        }
                        o[i * 4 + (size_t)ch] = v;
                    }
            } else {
                arena.resize(at + texels * 4);
                unsigned char* o = arena.data() + at;
                const unsigned char* sb = t.bytes.data();
                for (size_t i = 0; i < texels; i++)
                    for (int ch = 0; ch < 4; ch++)
                        o[i * 4 + (size_t)ch] = ch < t.channels ? sb[i * (size_t)t.channels + (size_t)ch] : (ch == 3 ? 255 : 0);
            }
            table.push_back(d);
        }
        if (arena.size() >= (1ull << 32))
            return fail(PT_ERR_SCENE, "textures of 4 GB or more in all (four channels a texel; the kernels address texels by 32-bit byte offsets)");
        HIP_TRY(c->dTextures.ensure(table.size() * sizeof(DevTexture)));
        HIP_TRY(c->dTexels.ensure(arena.size()));
        HIP_TRY(c->dSrgbLut.ensure(sizeof(lut)));
        if (!table.empty()) {
            HIP_TRY(hipMemcpyAsync(c->dTextures.ptr, table.data(), table.size() * sizeof(DevTexture), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->dTexels.ptr, arena.data(), arena.size(), hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(c->dSrgbLut.ptr, lut, sizeof(lut), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->texturesDirty = false;
        c->forkDirty = true;
        c->materialsDirty = true;  // texture table indices may have moved
    }
    if (c->materialsDirty) {
        std::vector<int> slotToTexture(c->textures.size(), -1);
        int next = 0;
        for (size_t s = 0; s < c->textures.size(); s++)
            if (c->textures[s].present) slotToTexture[s] = next++;
        std::vector<DevMaterial> packed;
        std::string err = PackMaterials(c->materials.data(), c->materials.size(), c->textures.size(), slotToTexture, packed);
        if (!err.empty()) return fail(PT_ERR_SCENE, err);
        HIP_TRY(c->dMaterials.ensure(packed.size() * sizeof(DevMaterial)));
        HIP_TRY(hipMemcpyAsync(c->dMaterials.ptr, packed.data(), packed.size() * sizeof(DevMaterial), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(materialTintKernel, dim3(((unsigned)packed.size() + 63u) / 64u), dim3(64), 0, c->stream,
                           (DevMaterial*)c->dMaterials.ptr, (unsigned)packed.size());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->materialsDirty = false;
        c->forkDirty = true;
    }
    if (c->envDirty) {
        HIP_TRY(c->dEnvColorPdf.ensure(c->envColorPdf.size() * sizeof(float)));
        HIP_TRY(c->dEnvSample.ensure(c->envSample.size() * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(c->dEnvColorPdf.ptr, c->envColorPdf.data(), c->envColorPdf.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dEnvSample.ptr, c->envSample.data(), c->envSample.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->envDirty = false;
        c->forkDirty = true;
    }
    return PT_OK;
}

// Control words of the wavefront pipeline (device): ray counts and shade counts are double
// buffered by iteration parity; each kernel zeroes the word the next kernel will append to.
struct WfCtrl {
    unsigned rayCount[2];
    unsigned shadeCount[2];
    unsigned rayHead;
    unsigned pad[3];
};

__global__ void wfResetKernel(WfCtrl* ctrl)
{
    ctrl->rayCount[0] = ctrl->rayCount[1] = 0u;
    ctrl->shadeCount[0] = ctrl->shadeCount[1] = 0u;
    ctrl->rayHead = 0u;
}

// Zeroes the control words nobody uses during the current kernel, for the kernel after it.
__global__ void wfPrepareKernel(unsigned* a, unsigned* b)
{
    if (a) *a = 0u;
    if (b) *b = 0u;
}

int wavefrontDispatch(pt_ctx* c, const DevParams& P, const DevScene& S, const Images& img, DevCounters* ctr, bool counted)
{
    const size_t pixels = (size_t)c->width * c->height;
    if (pixels > kEntrySlotMask) return fail(PT_ERR_INVALID, "image too large for the ray queue encoding");
    if (P.frameCount > 0x00ffffffu) return fail(PT_ERR_INVALID, "more than 2^24 samples per pixel in one dispatch");
    if (P.maxBounces > 255u) return fail(PT_ERR_INVALID, "the wavefront pipeline packs the bounce count into 8 bits");
    for (DeviceBuffer& b : c->dWfState) HIP_TRY(b.ensure(pixels * sizeof(float4)));
    HIP_TRY(c->dRayQ[0].ensure(2 * pixels * sizeof(unsigned)));
    HIP_TRY(c->dRayQ[1].ensure(2 * pixels * sizeof(unsigned)));
    HIP_TRY(c->dShadeQ.ensure(pixels * sizeof(unsigned)));
    HIP_TRY(c->dCtrl.ensure(sizeof(WfCtrl)));
    if (!c->hCtrl) HIP_TRY(hipHostMalloc((void**)&c->hCtrl, 64));
    if (c->traverseBlocks == 0) {
        int perCu = 0;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, traverseKernel<32, false>, kBlockThreads, 0));
        if (perCu < 1) perCu = 1;
        c->traverseBlocks = perCu * prop.multiProcessorCount;
    }
    WfState W{};
    float4** fields[10] = {&W.rayO, &W.rayD, &W.hit, &W.thrPdf, &W.radRng, &W.fMeta, &W.shDir, &W.pend, &W.aovA, &W.aovN};
    for (int i = 0; i < 10; i++) *fields[i] = (float4*)c->dWfState[i].ptr;
    WfCtrl* ctrl = (WfCtrl*)c->dCtrl.ptr;
    unsigned* rayQ[2] = {(unsigned*)c->dRayQ[0].ptr, (unsigned*)c->dRayQ[1].ptr};
    const dim3 block(kBlockThreads);
    if (3 * c->wideLevels > kStackWavefront)
        return fail(PT_ERR_SCENE, "BVH too deep for the wavefront pipeline (" + std::to_string(c->wideLevels) +
                                      " levels of four-wide nodes, its stack holds " + std::to_string(kStackWavefront / 3) + "): use the megakernel");
    const bool deep = 3 * c->wideLevels > 32;

    hipLaunchKernelGGL(wfResetKernel, dim3(1), dim3(1), 0, c->stream, ctrl);
    hipLaunchKernelGGL(wfStartKernel, dim3((unsigned)((pixels + kBlockThreads - 1) / kBlockThreads)), block, 0, c->stream, P, W,
                       rayQ[0], &ctrl->rayCount[0], ctr);
    HIP_TRY(hipGetLastError());
    const unsigned shadeBlocks = (unsigned)std::min<size_t>((pixels + kBlockThreads - 1) / kBlockThreads, 4096);
    // every pixel shades at most max_bounces times per sample, plus the pass that ends its last sample
    const uint64_t maxIter = (uint64_t)P.frameCount * P.maxBounces + 2;
    int iter = 0;
    for (uint64_t i = 0; i < maxIter; i++, iter++) {
        const int cur = (int)(i & 1u), nxt = cur ^ 1;
        WfQueues Q{};
        Q.rayIn = rayQ[cur];
        Q.rayInCount = &ctrl->rayCount[cur];
        Q.rayHead = &ctrl->rayHead;
        Q.shadeQ = (unsigned*)c->dShadeQ.ptr;
        Q.shadeCount = &ctrl->shadeCount[cur];
        Q.rayOut = rayQ[nxt];
        Q.rayOutCount = &ctrl->rayCount[nxt];
        const dim3 tgrid((unsigned)c->traverseBlocks);
        if (counted) {
            if (deep)
                hipLaunchKernelGGL((traverseKernel<kStackWavefront, true>), tgrid, block, 0, c->stream, P, S, W, img, Q, ctr);
            else
                hipLaunchKernelGGL((traverseKernel<32, true>), tgrid, block, 0, c->stream, P, S, W, img, Q, ctr);
        } else {
            if (deep)
                hipLaunchKernelGGL((traverseKernel<kStackWavefront, false>), tgrid, block, 0, c->stream, P, S, W, img, Q, ctr);
            else
                hipLaunchKernelGGL((traverseKernel<32, false>), tgrid, block, 0, c->stream, P, S, W, img, Q, ctr);
        }
        // the ray head and the other parity's words are idle now: zero them for the next users
        hipLaunchKernelGGL(wfPrepareKernel, dim3(1), dim3(1), 0, c->stream, &ctrl->rayCount[nxt], (unsigned*)nullptr);
        hipLaunchKernelGGL(shadeKernel, dim3(shadeBlocks), block, 0, c->stream, P, S, W, img, Q, ctr);
        hipLaunchKernelGGL(wfPrepareKernel, dim3(1), dim3(1), 0, c->stream, &ctrl->rayHead, &ctrl->shadeCount[nxt]);
        HIP_TRY(hipGetLastError());
        if ((i + 1) % (uint64_t)c->checkEvery == 0) {
            HIP_TRY(hipMemcpyAsync(c->hCtrl, &ctrl->rayCount[nxt], sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (*c->hCtrl == 0u) {
                iter++;
                break;
            }
        }
    }
    c->lastIterations = iter;
    return PT_OK;
}

// The launch-constant block and the scene pointers of a dispatch, from the context's state.
void fillLaunchBlocks(const pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride, DevParams& P, DevScene& S)
{
    const pt_params& u = c->params;
    P = DevParams{};
    Inverse4(u.view_matrix, P.invView);
    std::memcpy(P.invProj, u.inv_proj_matrix, 64);
    std::memcpy(P.view, u.view_matrix, 64);
    P.frameW = (float)c->width;
    P.frameH = (float)c->height;
    P.width = (uint32_t)c->width;
    P.height = (uint32_t)c->height;
    P.antiAliasing = u.anti_aliasing_enabled;
    P.focalLength = u.focal_length;
    P.apertureSize = u.aperture_size;
    P.apertureShapeX = u.aperture_shape[0];
    P.apertureShapeY = u.aperture_shape[1];
    const float mods[12] = {u.specular_modifier,   u.specular_tint_modifier,  u.metallic_modifier,
                            u.roughness_modifier,  u.subsurface_modifier,     u.anisotropy_modifier,
                            u.sheen_roughness_modifier, u.sheen_tint_modifier, u.clearcoat_modifier,
                            u.clearcoat_roughness_modifier, u.refraction_modifier, u.transmission_modifier};
    std::memcpy(P.modifiers, mods, sizeof(mods));
    P.maxBounces = u.max_bounces ? u.max_bounces : 3u;
    P.accumulateMode = u.accumulate_mode;
    P.frameFirst = frameFirst;
    P.frameCount = frameCount;
    P.frameStride = frameStride;
    P.hdriW = (float)c->envW;
    P.hdriH = (float)c->envH;
    P.envW = c->envW;
    P.envH = c->envH;
    P.rootRef = c->rootRef;
    P.tilesX = (uint32_t)((c->width + kMegaTile - 1) / kMegaTile);
    P.tilesY = (uint32_t)((c->height + kMegaTile - 1) / kMegaTile);

    S = DevScene{};
    S.nodes = (const DevNode*)c->dNodes.ptr;
    S.wide = (const DevNodeQ*)c->dNodes4.ptr;
    S.triangles = (const DevTriangle*)c->dTriangles.ptr;
    S.shading = (const DevShading*)c->dShading.ptr;
    S.materials = (const DevMaterial*)c->dMaterials.ptr;
    S.textures = (const DevTexture*)c->dTextures.ptr;
    S.texels = (const unsigned char*)c->dTexels.ptr;
    S.srgbLut = (const float*)c->dSrgbLut.ptr;
    S.envColorPdf = (const float4*)c->dEnvColorPdf.ptr;
    S.envSample = (const float2*)c->dEnvSample.ptr;

    P.travExitNum = c->travExitNum;
    P.refillMin = c->refillMin;
    P.hitMinNum = c->hitMinNum;
    P.waveTimes = nullptr;
    P.workNext = nullptr;
}

// The trace kernel variant that fits the tree and the settings in use.  `pull`: a grid of resident waves that take
// the work items off a counter; else one block per item.
using TraceKernel = void (*)(DevParams, DevScene, Images, DevCounters*);
TraceKernel traceVariant(const pt_ctx* c, bool counted, bool pull)
{
    // A query's stack holds at most three entries per level of the four-wide tree.  Up to 48 entries it lives in
    // LDS (12.5 KB per wave, which twelve resident waves per CU still fit); deeper trees keep 48 entries there
    // and the rest in lane-private memory, which very few rays touch.
    const bool deep = 3 * c->wideLevels > kStackLds;
    // Three waves per SIMD (168 VGPRs, next to nothing spilled) unless four are asked for (128 VGPRs, some 70
    // values spilled in the shading blocks; LDS for 32 stack entries per lane only, the rest always in private
    // memory): three are faster on every tree measured.
    const bool three = c->wavesPerSimd != 4;
    static const TraceKernel table[2][2][3] = {
        // [counted][pull][three waves, shallow | three waves, deep | four waves]
        {{renderKernel<kStackLds, 0, false, 3, false>, renderKernel<kStackLds, kStackPrivate, false, 3, false>, renderKernel<32, 64, false, 4, false>},
         {renderKernel<kStackLds, 0, false, 3, true>, renderKernel<kStackLds, kStackPrivate, false, 3, true>, renderKernel<32, 64, false, 4, true>}},
        {{renderKernel<kStackLds, 0, true, 3, false>, renderKernel<kStackLds, kStackPrivate, true, 3, false>, renderKernel<32, 64, true, 4, false>},
         {renderKernel<kStackLds, 0, true, 3, true>, renderKernel<kStackLds, kStackPrivate, true, 3, true>, renderKernel<32, 64, true, 4, true>}},
    };
    return table[counted][pull][three ? (deep ? 1 : 0) : 2];
}

// Waves of `kernel` the chip keeps resident at once (registers and LDS both count), asked of the runtime once per variant.
int residentWaves(pt_ctx* c, TraceKernel kernel, uint32_t& waves)
{
    for (const auto& known : c->residentWaves)
        if (known.first == kernel) {
            waves = known.second;
            return PT_OK;
        }
    int perCu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, kMegaThreads, 0));
    waves = (uint32_t)std::max(1, perCu) * c->computeUnits;
    c->residentWaves.emplace_back(kernel, waves);
    return PT_OK;
}

// One launch (of a dispatch call's one or more) on launch lane `k`; with `overlap` its trace side goes to the
// lane's stream.
int dispatchImpl(pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride, bool counted, int k, bool overlap,
                 bool firstPart = true, bool lastPart = true)
{
    DevParams P;
    DevScene S;
    fillLaunchBlocks(c, frameFirst, frameCount, frameStride, P, S);
    pt_ctx::LaunchLane& L = c->lane[k];
    const hipStream_t st = overlap ? L.stream : c->stream;  // the trace side's stream
    if (overlap) {
        // the lane's stream sees what the context's stream did for the scene, and the lane's buffers are free again
        if (c->forkDirty) {
            HIP_TRY(hipEventRecord(c->evFork, c->stream));
            c->forkCount++;
            c->forkDirty = false;
        }
        if (L.forkSeen != c->forkCount) {
            HIP_TRY(hipStreamWaitEvent(st, c->evFork, 0));
            L.forkSeen = c->forkCount;
        }
        if (L.used) HIP_TRY(hipStreamWaitEvent(st, L.evFree, 0));
    }
    DevCounters* ctr = (DevCounters*)c->dCounters[c->counterSet].ptr;
    if (firstPart) HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(DevCounters), st));
    Images img{imagePtr(c, 0), imagePtr(c, 1), imagePtr(c, 2)};
    if (firstPart) HIP_TRY(hipEventRecord(c->evStart, st));
    if (c->pipeline == 0) {
        // Items are taken costliest first: the cost of a tile (ticks its waves lived in the lane's previous
        // launch) varies by an order of magnitude, and a launch that reaches its expensive tiles last
        // ends in a long, nearly empty tail.  A tile whose cost exceeds a fraction of the launch's
        // lower bound has its samples spread over several items, so that one wave is never the
        // critical path (small images, a few expensive tiles in a cheap frame).
        const size_t tiles = (size_t)P.tilesX * P.tilesY;
        if (frameCount > 65535u) return fail(PT_ERR_INVALID, "a megakernel launch takes at most 65535 frames");
        const bool pull = c->pullItems;
        const TraceKernel kernel = traceVariant(c, counted, pull);
        uint32_t slots = 0;  // waves the kernel variant in use keeps resident
        int rcs = residentWaves(c, kernel, slots);
        if (rcs != PT_OK) return rcs;
        c->lastSlots = slots;
        // The cap on an item's cost is 1/split of the launch's lower bound.  Waves that pull their items lose nothing
        // at an item's end, so the cap can be fine: swept over 1, 2, 4 ... 64 with 64 frames per launch, a 32nd is 11 %
        // faster than a half at 512 x 512 and 3 % at 720p, makes no difference at 1080p, and is 4 % slower than an
        // eighth at 256 x 256, where there are fewer tiles than waves.  Unless set by hand.
        const unsigned split = c->tileSplit >= 0 ? (unsigned)c->tileSplit : (tiles < slots ? 8u : 32u);
        // planPartsKernel adds fewer than max(1, split) * slots items to the one per tile: the list holds them all, whatever
        // number of resident waves the runtime reports for the variant in use
        const size_t capacity = tiles + (size_t)std::max(1u, split) * slots;
        if (L.tileCostTiles != tiles || L.blockCapacity != capacity) {
            for (DeviceBuffer* b : {&L.dTileCost, &L.dTileParts, &L.dTileKey, &L.dTileKeySorted, &L.dTileIds, &L.dTileOrder,
                                    &L.dPartsInOrder, &L.dBlockStart})
                HIP_TRY(b->ensure((tiles + 1) * 4));  // (the cost words are followed by the launch's work counter)
            HIP_TRY(L.dBlockList.ensure(capacity * 8));
            HIP_TRY(L.dBlockCount.ensure(4));
            std::vector<uint32_t> ids(tiles);
            for (size_t i = 0; i < tiles; i++) ids[i] = (uint32_t)i;
            HIP_TRY(hipMemcpyAsync(L.dTileIds.ptr, ids.data(), tiles * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));  // `ids` dies with this scope
            L.tileSortBytes = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, L.tileSortBytes, (const uint32_t*)L.dTileKey.ptr,
                                                                 (uint32_t*)L.dTileKeySorted.ptr, (const uint32_t*)L.dTileIds.ptr,
                                                                 (uint32_t*)L.dTileOrder.ptr, (int)tiles, 0, 32, st));
            HIP_TRY(L.dTileSortTemp.ensure(L.tileSortBytes));
            L.scanBytes = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, L.scanBytes, (const uint32_t*)L.dPartsInOrder.ptr,
                                                     (uint32_t*)L.dBlockStart.ptr, (int)tiles, st));
            HIP_TRY(L.dScanTemp.ensure(L.scanBytes));
            if (L.tileCostTiles != tiles) L.tileCostValid = false;
            L.planFrames = 0;
            L.tileCostTiles = tiles;
            L.blockCapacity = capacity;
        }
        HIP_TRY(L.dSamples.ensure((size_t)frameCount * (size_t)c->width * c->height * kSampleBytes));
        P.samples = (float*)L.dSamples.ptr;
        P.pixels = (unsigned long long)c->width * c->height;
        P.tileCost = (unsigned int*)L.dTileCost.ptr;
        P.workNext = P.tileCost + tiles;
        P.blockList = nullptr;
        P.blockCount = nullptr;
        size_t gridBlocks = tiles;
        // A launch lane plans on every planEvery-th launch and lets the list serve the launches of the same frame count in
        // between as it stands: a one-frame call (the viewer's pattern) is a fifth planning (0.768 -> 0.720 ms), a 64-frame
        // launch 0.4 % (0.08 ms).  A stale list only costs balance; new geometry or a new image size drops it.
        const bool reuse = L.tileCostValid && c->tileOrdering && L.planFrames == frameCount &&
                           L.planUses < (uint32_t)std::max(1, c->planEvery);
        if (reuse) {
            L.planUses++;
            P.blockList = (const unsigned int*)L.dBlockList.ptr;
            P.blockCount = (const unsigned int*)L.dBlockCount.ptr;
            gridBlocks = L.planGrid;
        } else if (L.tileCostValid && c->tileOrdering) {
            // at least four samples per item, at most 255 items per tile
            const unsigned maxParts = split == 0 ? 1u : std::max(1u, std::min(255u, frameCount / 4u));
            const unsigned tb = 256, tg = (unsigned)((tiles + tb - 1) / tb);
            hipLaunchKernelGGL(planPartsKernel, dim3(1), dim3(1024), 0, st, (const unsigned*)L.dTileCost.ptr, (unsigned)tiles,
                               slots, std::max(1u, split), maxParts, (unsigned*)L.dTileParts.ptr,
                               (unsigned*)L.dTileKey.ptr);
            HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(L.dTileSortTemp.ptr, L.tileSortBytes, (const uint32_t*)L.dTileKey.ptr,
                                                                 (uint32_t*)L.dTileKeySorted.ptr, (const uint32_t*)L.dTileIds.ptr,
                                                                 (uint32_t*)L.dTileOrder.ptr, (int)tiles, 0, 32, st));
            hipLaunchKernelGGL(gatherPartsKernel, dim3(tg), dim3(tb), 0, st, (const unsigned*)L.dTileOrder.ptr,
                               (const unsigned*)L.dTileParts.ptr, (unsigned)tiles, (unsigned*)L.dPartsInOrder.ptr);
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(L.dScanTemp.ptr, L.scanBytes, (const uint32_t*)L.dPartsInOrder.ptr,
                                                     (uint32_t*)L.dBlockStart.ptr, (int)tiles, st));
            hipLaunchKernelGGL(fillBlocksKernel, dim3(tg), dim3(tb), 0, st, (const unsigned*)L.dTileOrder.ptr,
                               (const unsigned*)L.dPartsInOrder.ptr, (const unsigned*)L.dBlockStart.ptr, (unsigned)tiles,
                               (unsigned)capacity, frameCount, (unsigned*)L.dBlockList.ptr, (unsigned*)L.dBlockCount.ptr);
            HIP_TRY(hipGetLastError());
            P.blockList = (const unsigned int*)L.dBlockList.ptr;
            P.blockCount = (const unsigned int*)L.dBlockCount.ptr;
            gridBlocks = split == 0 ? tiles : capacity;  // the most items the list can hold (blocks past its end find none)
            L.planFrames = frameCount;
            L.planUses = 1;
            L.planGrid = gridBlocks;
        }
        // The items are pulled by a grid of as many waves as the chip keeps resident (a block per item spends the end
        // of its life with idle lanes, and a launch of few samples per pixel is nothing but such ends: 1080p, benchmark
        // scene, one sample per pixel 2.5 -> 1.3 ms, 16 samples 12.1 -> 10.8 ms, 64 samples 43.7 -> 41.9 ms).
        if (pull) gridBlocks = std::min<size_t>(gridBlocks, slots);
        // Calls that overlap share the chip: each takes its share of the wave slots, so that the calls in flight fill
        // it together and every wave lives through more items (a wave's last item ends in a tail of idle lanes,
        // and a launch of one sample per pixel is ten items a wave when it has the whole chip).
        // 1080p, one / two / four / eight frames per call, ms per call with all slots -> with half: 0.717 -> 0.594, 1.086 -> 0.999,
        // 1.82 -> 1.84, 3.36 -> 3.64 (a third: 0.607, 1.05, 2.20, 4.23): half the slots for calls of one or two frames.
        const int share = c->overlapShare >= 0 ? c->overlapShare : (frameCount <= 2u ? 2 : 0);
        if (pull && overlap && share > 0) gridBlocks = std::min<size_t>(gridBlocks, std::max<size_t>(1, slots / (size_t)share));
        HIP_TRY(hipMemsetAsync(L.dTileCost.ptr, 0, (tiles + 1) * 4, st));  // waves add the ticks they spend; work counter
        L.tileCostValid = true;
        if (counted) {
            c->waveTimeCount = gridBlocks;
            HIP_TRY(c->dWaveTimes.ensure(c->waveTimeCount * 16));
            HIP_TRY(hipMemsetAsync(c->dWaveTimes.ptr, 0, c->waveTimeCount * 16, st));
            P.waveTimes = (unsigned long long*)c->dWaveTimes.ptr;
        }
        const dim3 grid((unsigned)gridBlocks);
        HIP_TRY(hipEventRecord(c->evTrace0, st));
        hipLaunchKernelGGL(kernel, grid, dim3(kMegaThreads), 0, st, P, S, img, ctr);
        HIP_TRY(hipEventRecord(c->evTrace1, st));
        if (overlap) {
            HIP_TRY(hipEventRecord(L.evTraced, st));
            HIP_TRY(hipStreamWaitEvent(c->stream, L.evTraced, 0));
        }
        // samples are folded into the images in FrameCount order: all resolve passes on the one stream
        hipLaunchKernelGGL(resolveKernel, dim3((unsigned)((P.pixels + 255) / 256)), dim3(256), 0, c->stream, P, img);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->evResolve1, c->stream));
        HIP_TRY(hipEventRecord(L.evFree, c->stream));
        L.used = true;
        c->kernelTimed = true;
    } else {
        int rc2 = wavefrontDispatch(c, P, S, img, ctr, counted);
        if (rc2 != PT_OK) return rc2;
    }
    if (lastPart) HIP_TRY(hipEventRecord(c->evStop, c->stream));
    c->timed = true;
    return PT_OK;
}

// A dispatch call.  One of many frames with no tile costs on record yet (first launch, new image size, new
// geometry) is split: a short probe launch measures the costs, the rest runs from the plan made of them.
// Frame order per pixel is unchanged, so are the results.
int dispatchSplit(pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride, bool counted)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (frameCount == 0) return PT_OK;
    if (frameFirst == 0) return fail(PT_ERR_INVALID, "FrameCount starts at 1 (weight 1/FrameCount)");
    if (frameStride == 0) return fail(PT_ERR_INVALID, "frame_stride must be >= 1");
    DeviceGuard guard(c->device);
    int rc = syncScene(c);
    if (rc != PT_OK) return rc;
    // Calls may overlap on the library's own stream only (a caller's stream may be under graph capture, where work
    // forked onto other streams has to be joined before the capture ends), with the megakernel, uncounted.  1080p,
    // benchmark scene, ms per call without / with: one frame 1.39 / 1.02, two 2.03 / 1.69, four 3.27 / 2.99, eight
    // 5.83 / 5.60, sixteen 11.08 / 10.87, sixty-four 42.2 / 41.9: long launches gain a per cent or two and would no
    // longer be timed one by one, so it is calls of few frames that do.
    const bool allowed = c->pipeline == 0 && !counted && c->stream == c->ownStream;
    const bool overlap = allowed && (c->overlap < 0 ? frameCount <= 8u : c->overlap != 0);
    // (three calls in flight are 7 % faster than two with one frame per call, 2 % with two, and slower from eight on)
    const int k = overlap ? (c->lastLane = (c->lastLane + 1) % (frameCount <= 2u ? kLaunchLanes : 2)) : 0;
    c->counterSet = k;
    if (c->pipeline != 0) return dispatchImpl(c, frameFirst, frameCount, frameStride, counted, 0, false);
    // The megakernel writes one 36-byte record per (sample, pixel) and a resolve pass folds them into
    // the images; a launch takes as many frames as the sample buffer's budget holds.
    const uint64_t perFrame = (uint64_t)c->width * c->height * kSampleBytes;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(frameCount, 65535u), c->sampleBudgetBytes / perFrame));
    uint32_t done = 0;
    while (done < frameCount) {
        uint32_t n = std::min(chunk, frameCount - done);
        if (c->tileOrdering && !c->lane[k].tileCostValid && n >= 8) n = n >= 64 ? 4u : 2u;  // probe: measure tile costs first
        rc = dispatchImpl(c, frameFirst + done * frameStride, n, frameStride, counted, k, overlap, done == 0, done + n == frameCount);
        if (rc != PT_OK) return rc;
        done += n;
    }
    return PT_OK;
}

}  // namespace

// ================================================================================ C ABI
extern "C" {

const char* pt_last_error(void) { return g_lastError.c_str(); }

int pt_create(int device, int width, int height, pt_ctx** out)
{
    if (!out) return fail(PT_ERR_INVALID, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
        return fail(PT_ERR_INVALID, "image size must be within 1..32768");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(PT_ERR_DEVICE, "no HIP device available (this library has no CPU rendering path)");
    if (device < 0 || device >= count) return fail(PT_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PT_ERR_DEVICE, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
    DeviceGuard guard(device);
    if (device >= kMaxDevices) return fail(PT_ERR_INVALID, "device index above 63");
    if (!g_scratchPtr[device]) {  // once per device, with that device current
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratchPtr[device] = p;
    }
    std::unique_ptr<pt_ctx> c(new (std::nothrow) pt_ctx());
    if (!c) return fail(PT_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->width = width;
    c->height = height;
    c->computeUnits = (uint32_t)std::max(1, prop.multiProcessorCount);
    HIP_TRY(hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking));
    c->stream = c->ownStream;
    HIP_TRY(hipEventCreate(&c->evStart));
    HIP_TRY(hipEventCreate(&c->evStop));
    HIP_TRY(hipEventCreate(&c->evTrace0));
    HIP_TRY(hipEventCreate(&c->evTrace1));
    HIP_TRY(hipEventCreate(&c->evResolve1));
    const size_t imageBytes = (size_t)width * height * sizeof(float4);
    for (int i = 0; i < 3; i++) {
        HIP_TRY(c->dImages[i].ensure(imageBytes));
        HIP_TRY(hipMemsetAsync(c->dImages[i].ptr, 0, imageBytes, c->stream));
    }
    for (DeviceBuffer& b : c->dCounters) {
        HIP_TRY(b.ensure(sizeof(DevCounters)));
        HIP_TRY(hipMemsetAsync(b.ptr, 0, sizeof(DevCounters), c->stream));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    for (pt_ctx::LaunchLane& L : c->lane) {
        HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&L.evTraced, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.evFree, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    // pt_dispatch_static must not query the runtime (its callers may be under graph capture): the occupancy of the
    // variants it can launch is taken here
    for (int deep = 0; deep < 2; deep++) {
        c->wideLevels = deep ? kStackLds : 1;
        uint32_t waves = 0;
        int rcw = residentWaves(c.get(), traceVariant(c.get(), false, true), waves);
        if (rcw != PT_OK) return rcw;
    }
    c->wideLevels = 0;
    *out = c.release();
    return PT_OK;
}

void pt_destroy(pt_ctx* c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (pt_ctx::LaunchLane& L : c->lane) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        for (DeviceBuffer* b : {&L.dSamples, &L.dTileCost, &L.dTileParts, &L.dTileKey, &L.dTileKeySorted, &L.dTileIds, &L.dTileOrder, &L.dTileSortTemp,
                                &L.dPartsInOrder, &L.dBlockStart, &L.dBlockList, &L.dBlockCount, &L.dScanTemp})
            b->release();
        if (L.evTraced) (void)hipEventDestroy(L.evTraced);
        if (L.evFree) (void)hipEventDestroy(L.evFree);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    DeviceBuffer* all[] = {&c->dWaveTimes, &c->dRefNodes, &c->dRefPrims, &c->dRgbStaging, &c->dNodes, &c->dNodes4, &c->dTriangles, &c->dShading, &c->dMaterials, &c->dTextures, &c->dTexels,
                           &c->dSrgbLut, &c->dEnvColorPdf, &c->dEnvSample, &c->dImages[0], &c->dImages[1],
                           &c->dImages[2]};
    for (DeviceBuffer* b : all) b->release();
    for (DeviceBuffer& b : c->dCounters) b.release();
    for (DeviceBuffer& b : c->dWfState) b.release();
    c->dRayQ[0].release();
    c->dRayQ[1].release();
    c->dShadeQ.release();
    c->dCtrl.release();
    if (c->hCtrl) (void)hipHostFree(c->hCtrl);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->evTrace0) (void)hipEventDestroy(c->evTrace0);
    if (c->evTrace1) (void)hipEventDestroy(c->evTrace1);
    if (c->evResolve1) (void)hipEventDestroy(c->evResolve1);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

int pt_set_stream(pt_ctx* c, void* hipStream)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hipStream ? (hipStream_t)hipStream : c->ownStream;
    c->forkDirty = true;
    return PT_OK;
}

int pt_bind_images(pt_ctx* c, void* dRadiance, void* dAlbedo, void* dNormal)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    const int given = (dRadiance != nullptr) + (dAlbedo != nullptr) + (dNormal != nullptr);
    if (given != 0 && given != 3) return fail(PT_ERR_INVALID, "bind all three images or none");
    void* ptrs[3] = {dRadiance, dAlbedo, dNormal};
    for (int i = 0; i < 3; i++) {
        if (ptrs[i] && ((uintptr_t)ptrs[i] % 16) != 0) return fail(PT_ERR_INVALID, "image pointers must be 16-byte aligned");
        if (ptrs[i]) {
            hipPointerAttribute_t attr;
            hipError_t e = hipPointerGetAttributes(&attr, ptrs[i]);
            if (e != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != c->device) {
                (void)hipGetLastError();
                return fail(PT_ERR_INVALID, "image pointer is not device memory of the context's GPU");
            }
        }
    }
    for (int i = 0; i < 3; i++) c->bound[i] = (float4*)ptrs[i];
    return PT_OK;
}

int pt_upload_nodes(pt_ctx* c, const void* nodes48, size_t count)
{
    if (!c || !nodes48) return fail(PT_ERR_INVALID, "null argument");
    if (count < 2) return fail(PT_ERR_INVALID, "need the dummy node 0 and the root node 1");
    const RefNodeAlign* p = (const RefNodeAlign*)nodes48;
    c->nodes.assign(p, p + count);
    c->geometryDirty = true;
    c->deviceGeometry = false;
    c->builderChosen = false;
    forgetTileCosts(c);
    return PT_OK;
}

int pt_upload_primitives(pt_ctx* c, const void* prims112, size_t count)
{
    if (!c || !prims112) return fail(PT_ERR_INVALID, "null argument");
    if (count == 0) return fail(PT_ERR_INVALID, "no primitives");
    const RefPrimitiveAlign* p = (const RefPrimitiveAlign*)prims112;
    c->prims.assign(p, p + count);
    c->geometryDirty = true;
    c->deviceGeometry = false;
    forgetTileCosts(c);
    return PT_OK;
}

int pt_set_bvh_rebuild(pt_ctx* c, int enable)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    c->rebuildOnUpload = enable != 0;
    return PT_OK;
}

int pt_upload_materials(pt_ctx* c, const void* mats176, size_t count)
{
    if (!c || !mats176) return fail(PT_ERR_INVALID, "null argument");
    if (count == 0) return fail(PT_ERR_INVALID, "no materials");
    const RefMaterialAlign* p = (const RefMaterialAlign*)mats176;
    c->materials.assign(p, p + count);
    c->materialsDirty = true;
    c->geometryDirty = true;  // meshIndex range is validated against the material count
    return PT_OK;
}

int pt_upload_texture(pt_ctx* c, int slot, int width, int height, int channels, int isSrgb, int isFloat, const void* texels)
{
    if (!c || !texels) return fail(PT_ERR_INVALID, "null argument");
    if (slot < 1 || slot > 65535) return fail(PT_ERR_INVALID, "texture slot must be within 1..65535");
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) return fail(PT_ERR_INVALID, "bad texture size");
    if (channels < 1 || channels > 4) return fail(PT_ERR_INVALID, "channels must be 1..4");
    if (isSrgb && isFloat) return fail(PT_ERR_INVALID, "sRGB decoding applies to 8-bit textures only");
    if ((size_t)slot >= c->textures.size()) c->textures.resize((size_t)slot + 1);
    HostTexture& t = c->textures[(size_t)slot];
    t.width = width;
    t.height = height;
    t.channels = channels;
    t.format = isFloat ? 2 : (isSrgb ? 1 : 0);
    const size_t bytes = (size_t)width * height * channels * (isFloat ? 4 : 1);
    t.bytes.assign((const unsigned char*)texels, (const unsigned char*)texels + bytes);
    t.present = true;
    c->texturesDirty = true;
    return PT_OK;
}

int pt_upload_environment_with_cache(pt_ctx* c, const float* hdri, const float* cache, int width, int height)
{
    if (!c || !hdri || !cache) return fail(PT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0 || width > 32768 || height > 32768) return fail(PT_ERR_INVALID, "bad HDRI size");
    const size_t n = (size_t)width * height;
    c->envColorPdf.resize(n * 4);
    c->envSample.resize(n * 2);
    for (size_t i = 0; i < n; i++) {
        c->envColorPdf[4 * i + 0] = hdri[3 * i + 0];
        c->envColorPdf[4 * i + 1] = hdri[3 * i + 1];
        c->envColorPdf[4 * i + 2] = hdri[3 * i + 2];
        c->envColorPdf[4 * i + 3] = cache[3 * i + 2];
        c->envSample[2 * i + 0] = cache[3 * i + 0];
        c->envSample[2 * i + 1] = cache[3 * i + 1];
    }
    c->envW = width;
    c->envH = height;
    c->envDirty = true;
    return PT_OK;
}

int pt_upload_environment(pt_ctx* c, const float* hdri, int width, int height)
{
    if (!c || !hdri) return fail(PT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0 || width > 32768 || height > 32768) return fail(PT_ERR_INVALID, "bad HDRI size");
    std::vector<float> cache((size_t)width * height * 3);
    CalculateHdriCache(hdri, width, height, cache.data());
    return pt_upload_environment_with_cache(c, hdri, cache.data(), width, height);
}

int pt_add_model(pt_ctx* c, const void* vertices32, size_t vertexCount, const void* indices, size_t indexCount,
                 int indexBytes, const void* material176)
{
    if (!c || !vertices32 || !indices || !material176) return fail(PT_ERR_INVALID, "null argument");
    if (indexBytes != 1 && indexBytes != 2 && indexBytes != 4)
        return fail(PT_ERR_INVALID, "EBO Type not supported!");  // PathTracingRenderer.cpp:281
    if (indexCount % 3 != 0) return fail(PT_ERR_INVALID, "index count is not a multiple of 3");
    PendingModel m;
    m.vertices.assign((const RefVertex*)vertices32, (const RefVertex*)vertices32 + vertexCount);
    m.indices.resize(indexCount);
    for (size_t i = 0; i < indexCount; i++) {
        uint32_t v = indexBytes == 1 ? ((const uint8_t*)indices)[i]
                     : indexBytes == 2 ? ((const uint16_t*)indices)[i]
                                       : ((const uint32_t*)indices)[i];
        if (v >= vertexCount) return fail(PT_ERR_INVALID, "index " + std::to_string(v) + " exceeds the vertex count");
        m.indices[i] = v;
    }
    std::memcpy(&m.material, material176, sizeof(RefMaterialAlign));
    c->models.push_back(std::move(m));
    return PT_OK;
}

int pt_clear_models(pt_ctx* c)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    c->models.clear();
    return PT_OK;
}

namespace {
int processBuffers(pt_ctx* c, bool uncapped);
}
int pt_process_buffers(pt_ctx* c) { return processBuffers(c, false); }
int pt_process_buffers_full_sweep(pt_ctx* c) { return processBuffers(c, true); }

namespace {
int processBuffers(pt_ctx* c, bool uncapped)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (c->models.empty()) return fail(PT_ERR_STATE, "no models added");
    size_t triangles = 0;
    for (const PendingModel& m : c->models) triangles += m.indices.size() / 3;
    if (triangles == 0) return fail(PT_ERR_STATE, "models contain no triangles");
    std::vector<RefPrimitive> soup(triangles);
    std::vector<RefMaterialAlign> mats;
    size_t at = 0;
    uint32_t meshIndex = 0;
    for (const PendingModel& m : c->models) {
        AssemblePrimitives(m.vertices.data(), m.indices.data(), m.indices.size(), 4, meshIndex, soup.data() + at);
        at += m.indices.size() / 3;
        mats.push_back(m.material);
        meshIndex++;
    }
    std::vector<RefNode> nodes;
    BuildBvhWithSah(soup.data(), soup.size(), 4, nodes, 0, uncapped);
    std::vector<RefNodeAlign> nodesAligned(nodes.size());
    std::vector<RefPrimitiveAlign> primsAligned(soup.size());
    AlignNodes(nodes.data(), nodes.size(), nodesAligned.data());
    AlignPrimitives(soup.data(), soup.size(), primsAligned.data());
    int rc = pt_upload_materials(c, mats.data(), mats.size());
    if (rc != PT_OK) return rc;
    rc = pt_upload_nodes(c, nodesAligned.data(), nodesAligned.size());
    if (rc != PT_OK) return rc;
    rc = pt_upload_primitives(c, primsAligned.data(), primsAligned.size());
    if (rc == PT_OK) c->builderChosen = true;  // (pt_upload_nodes cleared it)
    return rc;
}

int buildBvhSah(void* prims100, size_t count, int leafSize, void* nodes40Out, size_t nodeCapacity, size_t* nodeCountOut,
                bool uncapped)
{
    if (!prims100 || !nodes40Out || !nodeCountOut) return fail(PT_ERR_INVALID, "null argument");
    if (leafSize < 1) return fail(PT_ERR_INVALID, "leaf_size must be >= 1");
    if (count > 0x7fffffffu) return fail(PT_ERR_INVALID, "too many primitives");
    std::vector<RefNode> nodes;
    BuildBvhWithSah((RefPrimitive*)prims100, count, leafSize, nodes, 0, uncapped);
    if (nodes.size() > nodeCapacity) return fail(PT_ERR_INVALID, "node_capacity too small: need " + std::to_string(nodes.size()));
    std::memcpy(nodes40Out, nodes.data(), nodes.size() * sizeof(RefNode));
    *nodeCountOut = nodes.size();
    return PT_OK;
}
}  // namespace

int pt_build_bvh_sah(void* prims100, size_t count, int leafSize, void* nodes40Out, size_t nodeCapacity, size_t* nodeCountOut)
{
    return buildBvhSah(prims100, count, leafSize, nodes40Out, nodeCapacity, nodeCountOut, false);
}
int pt_build_bvh_full_sweep(void* prims100, size_t count, int leafSize, void* nodes40Out, size_t nodeCapacity, size_t* nodeCountOut)
{
    return buildBvhSah(prims100, count, leafSize, nodes40Out, nodeCapacity, nodeCountOut, true);
}

int pt_assemble_primitives(const void* vertices32, size_t vertexCount, const void* indices, size_t indexCount,
                           int indexBytes, uint32_t meshIndex, void* prims100Out)
{
    if (!vertices32 || !indices || !prims100Out) return fail(PT_ERR_INVALID, "null argument");
    if (indexBytes != 1 && indexBytes != 2 && indexBytes != 4) return fail(PT_ERR_INVALID, "EBO Type not supported!");
    for (size_t i = 0; i < indexCount; i++) {
        uint32_t v = indexBytes == 1 ? ((const uint8_t*)indices)[i]
                     : indexBytes == 2 ? ((const uint16_t*)indices)[i]
                                       : ((const uint32_t*)indices)[i];
        if (v >= vertexCount) return fail(PT_ERR_INVALID, "index " + std::to_string(v) + " exceeds the vertex count");
    }
    AssemblePrimitives((const RefVertex*)vertices32, indices, indexCount, indexBytes, meshIndex, (RefPrimitive*)prims100Out);
    return PT_OK;
}

int pt_align_nodes(const void* nodes40, size_t count, void* nodes48Out)
{
    if (!nodes40 || !nodes48Out) return fail(PT_ERR_INVALID, "null argument");
    AlignNodes((const RefNode*)nodes40, count, (RefNodeAlign*)nodes48Out);
    return PT_OK;
}

int pt_align_primitives(const void* prims100, size_t count, void* prims112Out)
{
    if (!prims100 || !prims112Out) return fail(PT_ERR_INVALID, "null argument");
    AlignPrimitives((const RefPrimitive*)prims100, count, (RefPrimitiveAlign*)prims112Out);
    return PT_OK;
}

int pt_calculate_hdri_cache(const float* hdri, int width, int height, float* cacheOut)
{
    if (!hdri || !cacheOut) return fail(PT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0) return fail(PT_ERR_INVALID, "bad HDRI size");
    CalculateHdriCache(hdri, width, height, cacheOut);
    return PT_OK;
}

int pt_bvh_cache_key(const void* primitives, size_t bytes, uint64_t* keyOut)
{
    if (!primitives || !keyOut) return fail(PT_ERR_INVALID, "null argument");
    *keyOut = Fnv1a64(primitives, bytes);
    return PT_OK;
}

int pt_bvh_cache_save(const char* path, uint64_t key, const void* nodes48, size_t nodeCount, const void* prims112, size_t primCount)
{
    if (!path || !nodes48 || !prims112) return fail(PT_ERR_INVALID, "null argument");
    const std::string err = SaveBvhCache(path, key, (const RefNodeAlign*)nodes48, nodeCount, (const RefPrimitiveAlign*)prims112, primCount);
    return err.empty() ? PT_OK : fail(PT_ERR_STATE, err);
}

int pt_bvh_cache_load(const char* path, uint64_t key, void* nodes48Out, size_t nodeCapacity, size_t* nodeCountOut,
                      void* prims112Out, size_t primCapacity, size_t* primCountOut)
{
    if (!path || !nodeCountOut || !primCountOut) return fail(PT_ERR_INVALID, "null argument");
    std::vector<RefNodeAlign> nodes;
    std::vector<RefPrimitiveAlign> prims;
    std::string err;
    try {
        err = LoadBvhCache(path, key, nodes, prims);
    } catch (const std::bad_alloc&) {
        return fail(PT_ERR_NOMEM, "out of host memory while reading the BVH cache");
    }
    if (!err.empty()) return fail(PT_ERR_STATE, err);
    *nodeCountOut = nodes.size();
    *primCountOut = prims.size();
    if (!nodes48Out && !prims112Out) return PT_OK;
    if (!nodes48Out || !prims112Out) return fail(PT_ERR_INVALID, "null output buffer");
    if (nodeCapacity < nodes.size() || primCapacity < prims.size()) return fail(PT_ERR_INVALID, "output capacity too small");
    std::memcpy(nodes48Out, nodes.data(), nodes.size() * sizeof(RefNodeAlign));
    std::memcpy(prims112Out, prims.data(), prims.size() * sizeof(RefPrimitiveAlign));
    return PT_OK;
}

struct pt_gltf {
    GltfScene scene;
};

int pt_gltf_load(const char* path, pt_gltf** out)
{
    if (!path || !out) return fail(PT_ERR_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<pt_gltf> g(new (std::nothrow) pt_gltf());
    if (!g) return fail(PT_ERR_NOMEM, "out of host memory");
    std::string err;
    try {
        err = LoadGltf(path, g->scene);
    } catch (const std::bad_alloc&) {
        return fail(PT_ERR_NOMEM, "out of host memory while reading the glTF file");
    }
    if (!err.empty()) return fail(PT_ERR_SCENE, err);
    *out = g.release();
    return PT_OK;
}

void pt_gltf_free(pt_gltf* g) { delete g; }

int pt_gltf_submesh_count(const pt_gltf* g, uint32_t* count)
{
    if (!g || !count) return fail(PT_ERR_INVALID, "null argument");
    *count = (uint32_t)g->scene.submeshes.size();
    return PT_OK;
}

int pt_gltf_submesh_size(const pt_gltf* g, uint32_t submesh, uint64_t* vertexCount, uint64_t* indexCount)
{
    if (!g || !vertexCount || !indexCount) return fail(PT_ERR_INVALID, "null argument");
    if (submesh >= g->scene.submeshes.size()) return fail(PT_ERR_INVALID, "sub-mesh index out of range");
    *vertexCount = g->scene.submeshes[submesh].vertices.size();
    *indexCount = g->scene.submeshes[submesh].indices.size();
    return PT_OK;
}

int pt_gltf_submesh_read(const pt_gltf* g, uint32_t submesh, void* vertices, uint32_t* indices)
{
    if (!g || !vertices || !indices) return fail(PT_ERR_INVALID, "null argument");
    if (submesh >= g->scene.submeshes.size()) return fail(PT_ERR_INVALID, "sub-mesh index out of range");
    const GltfSubmesh& sm = g->scene.submeshes[submesh];
    std::memcpy(vertices, sm.vertices.data(), sm.vertices.size() * sizeof(RefVertex));
    std::memcpy(indices, sm.indices.data(), sm.indices.size() * sizeof(uint32_t));
    return PT_OK;
}

int pt_gltf_submesh_material(const pt_gltf* g, uint32_t submesh, float attributes[18], int32_t images[11])
{
    if (!g || !attributes || !images) return fail(PT_ERR_INVALID, "null argument");
    if (submesh >= g->scene.submeshes.size()) return fail(PT_ERR_INVALID, "sub-mesh index out of range");
    const GltfMaterial& m = g->scene.submeshes[submesh].material;
    std::memcpy(attributes, &m, 18 * sizeof(float));
    std::memcpy(images, m.image, sizeof(m.image));
    return PT_OK;
}

int pt_gltf_image_count(const pt_gltf* g, uint32_t* count)
{
    if (!g || !count) return fail(PT_ERR_INVALID, "null argument");
    *count = (uint32_t)g->scene.images.size();
    return PT_OK;
}

int pt_gltf_image_source(const pt_gltf* g, uint32_t image, const char** path, const void** bytes, uint64_t* byteCount)
{
    if (!g || !path || !bytes || !byteCount) return fail(PT_ERR_INVALID, "null argument");
    if (image >= g->scene.images.size()) return fail(PT_ERR_INVALID, "image index out of range");
    const GltfImage& img = g->scene.images[image];
    *path = img.path.empty() ? nullptr : img.path.c_str();
    *bytes = img.bytes.empty() ? nullptr : img.bytes.data();
    *byteCount = img.bytes.size();
    return PT_OK;
}

void pt_look_at(const float eye[3], const float center[3], const float up[3], float out16[16]) { LookAt(eye, center, up, out16); }
void pt_perspective(float fovy, float aspect, float zNear, float zFar, float out16[16]) { Perspective(fovy, aspect, zNear, zFar, out16); }
void pt_inverse4(const float in16[16], float out16[16]) { Inverse4(in16, out16); }

void pt_default_params(int width, int height, pt_params* out)
{
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    const float eye[3] = {-2.5f, 1.0f, 0.0f}, center[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    LookAt(eye, center, up, out->view_matrix);
    Perspective(1.57f, (float)width / (float)height, 0.01f, 1000.0f, out->proj_matrix);
    Inverse4(out->proj_matrix, out->inv_proj_matrix);
    out->frame_count = 1;
    out->frame_dimensions[0] = (float)width;
    out->frame_dimensions[1] = (float)height;
    out->anti_aliasing_enabled = 1;
    out->focal_length = 3.5f;
    out->aperture_size = 0.0f;
    out->aperture_shape[0] = 1.0f;
    out->aperture_shape[1] = 1.0f;
    out->max_bounces = 3;
    out->accumulate_mode = PT_ACCUM_RUNNING_MEAN;
}

int pt_set_params(pt_ctx* c, const pt_params* p)
{
    if (!c || !p) return fail(PT_ERR_INVALID, "null argument");
    if ((p->frame_dimensions[0] != 0.0f || p->frame_dimensions[1] != 0.0f) &&
        (p->frame_dimensions[0] != (float)c->width || p->frame_dimensions[1] != (float)c->height))
        return fail(PT_ERR_INVALID, "FrameDimensions does not match the context size");
    if (p->accumulate_mode > PT_ACCUM_SUM) return fail(PT_ERR_INVALID, "unknown accumulate_mode");
    if (p->max_bounces > 1024) return fail(PT_ERR_INVALID, "max_bounces above 1024");
    c->params = *p;
    c->haveParams = true;
    return PT_OK;
}

int pt_clear(pt_ctx* c)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    DeviceGuard guard(c->device);
    const size_t imageBytes = (size_t)c->width * c->height * sizeof(float4);
    for (int i = 0; i < 3; i++) HIP_TRY(hipMemsetAsync(imagePtr(c, i), 0, imageBytes, c->stream));
    return PT_OK;
}

int pt_dispatch(pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride)
{
    return dispatchSplit(c, frameFirst, frameCount, frameStride, false);
}

int pt_dispatch_counted(pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride)
{
    return dispatchSplit(c, frameFirst, frameCount, frameStride, true);
}

// pt_dispatch for callers that cannot allocate.  Same kernels, same images; the sample records and
// the per-tile cost words live in the static scratch array of the context's GPU instead of buffers
// the context would (re)allocate, tiles are launched in row-major order (the launch plan needs sort
// scratch), nothing is uploaded and no event is recorded.
int pt_dispatch_static(pt_ctx* c, uint32_t frameFirst, uint32_t frameCount, uint32_t frameStride)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (frameCount == 0) return PT_OK;
    if (frameFirst == 0) return fail(PT_ERR_INVALID, "FrameCount starts at 1 (weight 1/FrameCount)");
    if (frameStride == 0) return fail(PT_ERR_INVALID, "frame_stride must be >= 1");
    if (frameCount > 65535u) return fail(PT_ERR_INVALID, "a megakernel launch takes at most 65535 frames");
    if (c->pipeline != 0) return fail(PT_ERR_STATE, "pt_dispatch_static runs the megakernel scheduler only");
    if (!c->haveParams) return fail(PT_ERR_STATE, "pt_set_params has not been called");
    if (c->geometryDirty || c->materialsDirty || c->texturesDirty || c->envDirty)
        return fail(PT_ERR_STATE, "scene buffers changed since the last pt_dispatch: this entry point uploads nothing, "
                                  "run one pt_dispatch first");
    unsigned char* base = (unsigned char*)g_scratchPtr[c->device];
    if (!base) return fail(PT_ERR_STATE, "no static scratch on this device");
    const size_t pixels = (size_t)c->width * c->height;
    const size_t tiles = (size_t)((c->width + kMegaTile - 1) / kMegaTile) * (size_t)((c->height + kMegaTile - 1) / kMegaTile);
    const size_t costBytes = ((tiles + 1) * 4 + 255) / 256 * 256;  // per-tile cost words and the work counter
    const size_t need = costBytes + (size_t)frameCount * pixels * kSampleBytes;
    if (need > kStaticScratchBytes)
        return fail(PT_ERR_INVALID, "pt_dispatch_static: " + std::to_string(frameCount) + " frames of " + std::to_string(c->width) + "x" +
                                        std::to_string(c->height) + " need " + std::to_string(need) + " bytes of scratch, the static array holds " +
                                        std::to_string(kStaticScratchBytes) + " (dispatch fewer frames per call)");
    DeviceGuard guard(c->device);
    DevParams P;
    DevScene S;
    fillLaunchBlocks(c, frameFirst, frameCount, frameStride, P, S);
    P.tileCost = (unsigned int*)base;
    P.workNext = P.tileCost + tiles;
    P.samples = (float*)(base + costBytes);
    P.pixels = pixels;
    P.blockList = nullptr;
    P.blockCount = nullptr;
    c->counterSet = 0;
    DevCounters* ctr = (DevCounters*)c->dCounters[c->counterSet].ptr;  // allocated by pt_create
    Images img{imagePtr(c, 0), imagePtr(c, 1), imagePtr(c, 2)};
    HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(DevCounters), c->stream));
    HIP_TRY(hipMemsetAsync(base, 0, costBytes, c->stream));  // the array keeps the previous call's contents
    const TraceKernel kernel = traceVariant(c, false, true);
    uint32_t slots = 0;
    const int rcs = residentWaves(c, kernel, slots);  // (known since pt_create unless four waves per SIMD were asked for)
    if (rcs != PT_OK) return rcs;
    hipLaunchKernelGGL(kernel, dim3((unsigned)std::min<size_t>(tiles, slots)), dim3(kMegaThreads), 0, c->stream, P, S, img, ctr);
    hipLaunchKernelGGL(resolveKernel, dim3((unsigned)((P.pixels + 255) / 256)), dim3(256), 0, c->stream, P, img);
    HIP_TRY(hipGetLastError());
    c->timed = false;  // no events around this launch
    c->kernelTimed = false;
    return PT_OK;
}

int pt_static_scratch_info(uint64_t* scratchBytes, uint32_t* sampleRecordBytes)
{
    if (scratchBytes) *scratchBytes = kStaticScratchBytes;
    if (sampleRecordBytes) *sampleRecordBytes = (uint32_t)kSampleBytes;
    return PT_OK;
}

int pt_scale_images(pt_ctx* c, float scale)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    DeviceGuard guard(c->device);
    const size_t n = (size_t)c->width * c->height;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(scaleImagesKernel, dim3(blocks), dim3(256), 0, c->stream, imagePtr(c, 0), imagePtr(c, 1), imagePtr(c, 2), n, scale);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_synchronize(pt_ctx* c)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_read_image(pt_ctx* c, int which, float* rgbaOut)
{
    if (!c || !rgbaOut) return fail(PT_ERR_INVALID, "null argument");
    if (which < 0 || which > 2) return fail(PT_ERR_INVALID, "image index must be 0..2");
    DeviceGuard guard(c->device);
    const size_t imageBytes = (size_t)c->width * c->height * sizeof(float4);
    HIP_TRY(hipMemcpyAsync(rgbaOut, imagePtr(c, which), imageBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_write_image(pt_ctx* c, int which, const float* rgba)
{
    if (!c || !rgba) return fail(PT_ERR_INVALID, "null argument");
    if (which < 0 || which > 2) return fail(PT_ERR_INVALID, "image index must be 0..2");
    DeviceGuard guard(c->device);
    const size_t imageBytes = (size_t)c->width * c->height * sizeof(float4);
    HIP_TRY(hipMemcpyAsync(imagePtr(c, which), rgba, imageBytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the caller's buffer is free again on return
    return PT_OK;
}

int pt_read_image_rgb(pt_ctx* c, int which, float* rgbOut)
{
    if (!c || !rgbOut) return fail(PT_ERR_INVALID, "null argument");
    if (which < 0 || which > 2) return fail(PT_ERR_INVALID, "image index must be 0..2");
    DeviceGuard guard(c->device);
    const size_t texels = (size_t)c->width * c->height;
    HIP_TRY(c->dRgbStaging.ensure(texels * 3 * sizeof(float)));
    const size_t quads = (texels + 3) / 4;
    hipLaunchKernelGGL(packRgbKernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, c->stream, imagePtr(c, which),
                       (float*)c->dRgbStaging.ptr, texels);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgbOut, c->dRgbStaging.ptr, texels * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_read_display(pt_ctx* c, float exposure, unsigned char* rgb8Out)
{
    if (!c || !rgb8Out) return fail(PT_ERR_INVALID, "null argument");
    DeviceGuard guard(c->device);
    const size_t texels = (size_t)c->width * c->height;
    HIP_TRY(c->dRgbStaging.ensure(texels * 3 * sizeof(float)));  // shared with pt_read_image_rgb
    const size_t quads = (texels + 3) / 4;
    hipLaunchKernelGGL(displayKernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, c->stream, imagePtr(c, PT_IMAGE_RADIANCE),
                       (unsigned char*)c->dRgbStaging.ptr, texels, exposure);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb8Out, c->dRgbStaging.ptr, texels * 3, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_debug_view(pt_ctx* c, int view)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (view != PT_DEBUG_VIEW_BVH && view != PT_DEBUG_VIEW_HDRI_CACHE) return fail(PT_ERR_INVALID, "unknown debug view");
    DeviceGuard guard(c->device);
    int rc = syncScene(c);
    if (rc != PT_OK) return rc;
    DevParams P;
    DevScene S;
    fillLaunchBlocks(c, 1, 1, 1, P, S);
    // the heat map walks the tree as the shader does, over the reference's node records: those of a device-built
    // tree are on the device already, uploaded ones are copied there for this call
    ScopedBuffer uploaded;
    const RefNodeAlign* nodes = (const RefNodeAlign*)c->dRefNodes.ptr;
    if (view == PT_DEBUG_VIEW_BVH && !c->deviceGeometry) {
        HIP_TRY(uploaded.ensure(c->nodes.size() * sizeof(RefNodeAlign)));
        HIP_TRY(hipMemcpyAsync(uploaded.ptr, c->nodes.data(), c->nodes.size() * sizeof(RefNodeAlign), hipMemcpyHostToDevice, c->stream));
        nodes = (const RefNodeAlign*)uploaded.ptr;
    }
    const dim3 block(16, 16), grid((unsigned)((c->width + 15) / 16), (unsigned)((c->height + 15) / 16));
    hipLaunchKernelGGL(debugViewKernel, grid, block, 0, c->stream, P, S, view, nodes, imagePtr(c, PT_IMAGE_RADIANCE));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));  // `uploaded` is freed with this scope
    return PT_OK;
}

int pt_get_stats(pt_ctx* c, pt_stats* out)
{
    if (!c || !out) return fail(PT_ERR_INVALID, "null argument");
    DeviceGuard guard(c->device);
    DevCounters h{};
    HIP_TRY(hipMemcpyAsync(&h, c->dCounters[c->counterSet].ptr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->samples = h.samples;
    out->rays_closest = h.raysClosest;
    out->rays_any = h.raysAny;
    out->node_visits = h.nodeVisits;
    out->triangle_tests = h.triangleTests;
    out->closest_hits = h.closestHits;
    out->max_stack_depth = h.maxStack;
    out->reserved = 0;
    return PT_OK;
}

int pt_last_dispatch_ms(pt_ctx* c, float* msOut)
{
    if (!c || !msOut) return fail(PT_ERR_INVALID, "null argument");
    if (!c->timed) return fail(PT_ERR_STATE, "no dispatch yet");
    DeviceGuard guard(c->device);
    HIP_TRY(hipEventSynchronize(c->evStop));
    HIP_TRY(hipEventElapsedTime(msOut, c->evStart, c->evStop));
    return PT_OK;
}

int pt_last_kernel_ms(pt_ctx* c, float* traceMs, float* resolveMs)
{
    if (!c || !traceMs || !resolveMs) return fail(PT_ERR_INVALID, "null argument");
    if (!c->kernelTimed) return fail(PT_ERR_STATE, "no megakernel launch yet");
    DeviceGuard guard(c->device);
    HIP_TRY(hipEventSynchronize(c->evResolve1));
    HIP_TRY(hipEventElapsedTime(traceMs, c->evTrace0, c->evTrace1));
    HIP_TRY(hipEventElapsedTime(resolveMs, c->evTrace1, c->evResolve1));
    return PT_OK;
}

int pt_scene_bytes(pt_ctx* c, uint64_t* nodeBytes, uint64_t* triangleBytes, uint64_t* shadingBytes)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (nodeBytes) *nodeBytes = c->nodeBytes;
    if (triangleBytes) *triangleBytes = c->triangleBytes;
    if (shadingBytes) *shadingBytes = c->shadingBytes;
    return PT_OK;
}

// Tuning hook.  key 0: pipeline (0 megakernel, 1 wavefront); 1: megakernel traversal-phase exit
// threshold in 64ths of the live lanes; 2: wavefront refill threshold (idle lanes); 3: wavefront
// iterations between read-backs of the live-ray count; 4: traverse grid size in blocks (0 = auto);
// 5: megakernel hit-shading threshold in 64ths of the live lanes; 6: launch plan from the previous launch's tile costs on/off;
// 7: tile split: no work item of the megakernel may cost more than 1/value of the launch's lower bound (0: one item per
// tile, -1: 8 for images with fewer tiles than resident waves, else 32); 8: sample-buffer budget in MiB (a dispatch is
// cut into launches whose sample records fit); 9: megakernel run by resident waves that pull the
// work items off a counter (1, the default) or by one block per item (0); 10: device BVH
// builder: clustering search radius (0 = Morton radix tree); 11: megakernel compiled for 3 or 4 waves per SIMD
// (0 = the default, 3); 12: the trace launches of consecutive dispatch calls may overlap (1), never do (0), or do for
// calls of at most 8 frames (-1, the default); 13: device BVH builder: most triangles per leaf (1..4; 0, the default: 2 for
// scenes of up to 8 M triangles, 3 above).
#ifdef PT_EXPERIMENT_SHADE_TIMERS
// Experiment builds only: the tick sums of shadeRay's sections since the last call (and resets them).
extern "C" int pt_experiment_shade_ticks(unsigned long long* out8)
{
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(ptcore::g_shadeTicks), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptcore::g_shadeTicks), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

int pt_debug_set_tuning(pt_ctx* c, int key, int value)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    switch (key) {
    case 0:
        if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "pipeline must be 0 or 1");
        c->pipeline = value;
        return PT_OK;
    case 1:
        if (value < 1 || value > 64) return fail(PT_ERR_INVALID, "threshold must be 1..64");
        c->travExitNum = (uint32_t)value;
        return PT_OK;
    case 2:
        if (value < 1 || value > 64) return fail(PT_ERR_INVALID, "refill threshold must be 1..64");
        c->refillMin = (uint32_t)value;
        return PT_OK;
    case 3:
        if (value < 1) return fail(PT_ERR_INVALID, "check interval must be >= 1");
        c->checkEvery = value;
        return PT_OK;
    case 4:
        if (value < 0 || value > 65535) return fail(PT_ERR_INVALID, "bad grid size");
        c->traverseBlocks = value;
        return PT_OK;
    case 5:
        if (value < 1 || value > 64) return fail(PT_ERR_INVALID, "hit threshold must be 1..64");
        c->hitMinNum = (uint32_t)value;
        return PT_OK;
    case 6:
        if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "tile ordering must be 0 or 1");
        c->tileOrdering = value != 0;
        return PT_OK;
    case 7:
        if (value < -1 || value > 64) return fail(PT_ERR_INVALID, "tile split must be -1..64");
        c->tileSplit = value;
        return PT_OK;
    case 9:
        if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "pull_items must be 0 or 1");
        c->pullItems = value != 0;
        return PT_OK;
    case 12:
        if (value < -1 || value > 1) return fail(PT_ERR_INVALID, "overlap_calls must be -1 (calls of few frames), 0 or 1");
        c->overlap = value;
        return PT_OK;
    case 11:
        if (value != 0 && value != 3 && value != 4) return fail(PT_ERR_INVALID, "waves per SIMD must be 0 (automatic), 3 or 4");
        c->wavesPerSimd = value;
        return PT_OK;
    case 10:
        if (value < 0 || value > 256) return fail(PT_ERR_INVALID, "clustering radius must be 0..256");
        c->plocRadius = value;
        return PT_OK;
    case 13:
        if (value < 0 || value > kLbvhLeafLimit) return fail(PT_ERR_INVALID, "triangles per leaf must be 0 (by scene size) or 1..4");
        c->bvhLeafMax = value;
        return PT_OK;
    case 17:
        if (value < -1 || value > 256) return fail(PT_ERR_INVALID, "reinsertion passes must be -1 (by scene size) or 0..256");
        c->bvhOptimise = value;
        return PT_OK;
    case 20:
        if (value < 0 || value > 8) return fail(PT_ERR_INVALID, "leaf re-cut must be 0 (keep the caller's leaves) or 1..8");
        c->leafRecut = value;  // (takes effect at the next upload)
        return PT_OK;
    case 19:
        if (value < 1 || value > 60) return fail(PT_ERR_INVALID, "height limit of the optimised tree must be 1..60");
        c->bvhOptimiseMaxHeight = value;
        return PT_OK;
    case 18:
        if (value < 1 || value > 64) return fail(PT_ERR_INVALID, "candidate spacing must be 1..64");
        c->bvhOptimiseMod = value;
        return PT_OK;
    case 16:
        if (value < -1 || value > 64) return fail(PT_ERR_INVALID, "share of the wave slots must be -1 (by frame count), 0 (all) or 1/1..1/64");
        c->overlapShare = value;
        return PT_OK;
    case 15:
        if (value < 1 || value > 1024) return fail(PT_ERR_INVALID, "launches per plan must be 1..1024");
        c->planEvery = value;
        return PT_OK;
    case 14:
        if (value < 0 || value > 4095) return fail(PT_ERR_INVALID, "decoded-texture budget must be 0..4095 MiB");
        c->textureDecodeBytes = (size_t)value << 20;
        c->texturesDirty = true;
        return PT_OK;
    case 8:
        if (value < 1) return fail(PT_ERR_INVALID, "sample budget must be >= 1 MiB");
        c->sampleBudgetBytes = (uint64_t)value << 20;
        return PT_OK;
    default: return fail(PT_ERR_INVALID, "unknown tuning key");
    }
}

// The four-wide tree in use: its depth in levels, and whether it had to be folded level by level (1) because folding
// by surface area left it deeper than the traversal stacks hold.  Valid after a dispatch or a device build.
int pt_debug_get_tree_info(pt_ctx* c, uint32_t* levelsOut, uint32_t* levelByLevelOut)
{
    if (!c || !levelsOut || !levelByLevelOut) return fail(PT_ERR_INVALID, "null argument");
    *levelsOut = (uint32_t)c->wideLevels;
    *levelByLevelOut = c->foldedLevelByLevel ? 1u : 0u;
    return PT_OK;
}

// Waves the trace kernel variant of the most recent megakernel launch keeps resident on the chip (its grid when
// the waves pull their items).
int pt_debug_get_resident_waves(pt_ctx* c, uint32_t* wavesOut)
{
    if (!c || !wavesOut) return fail(PT_ERR_INVALID, "null argument");
    if (!c->lastSlots) return fail(PT_ERR_STATE, "no megakernel launch yet");
    *wavesOut = c->lastSlots;
    return PT_OK;
}

// Profiling hook: wall-clock ticks at the start and the end of every wave of the most recent
// pt_dispatch_counted() with the megakernel scheduler, one pair (start, end) per block of the grid; blocks
// that found no work have start = 0.
int pt_debug_get_wave_times(pt_ctx* c, uint64_t* pairsOut, size_t capacity, size_t* countOut, int* clockKhzOut)
{
    if (!c || !countOut) return fail(PT_ERR_INVALID, "null argument");
    *countOut = c->waveTimeCount;
    DeviceGuard guard(c->device);
    if (clockKhzOut) HIP_TRY(hipDeviceGetAttribute(clockKhzOut, hipDeviceAttributeWallClockRate, c->device));
    if (!pairsOut) return PT_OK;
    if (capacity < c->waveTimeCount) return fail(PT_ERR_INVALID, "capacity too small");
    if (c->waveTimeCount == 0) return PT_OK;
    HIP_TRY(hipMemcpyAsync(pairsOut, c->dWaveTimes.ptr, c->waveTimeCount * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

// Profiling hook: the 16 wave-occupancy counters of the most recent pt_dispatch_counted() with the
// megakernel scheduler (see DevCounters::profile).
int pt_debug_get_profile(pt_ctx* c, uint64_t out16[16])
{
    if (!c || !out16) return fail(PT_ERR_INVALID, "null argument");
    DeviceGuard guard(c->device);
    DevCounters h{};
    HIP_TRY(hipMemcpyAsync(&h, c->dCounters[c->counterSet].ptr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 16; i++) out16[i] = h.profile[i];
    return PT_OK;
}

// Profiling hook: closest-hit queries by path depth of the most recent pt_dispatch_counted() with the
// megakernel scheduler (primary rays in [0]; depths past 31 in [31]).
int pt_debug_get_depth_histogram(pt_ctx* c, uint64_t out32[32])
{
    if (!c || !out32) return fail(PT_ERR_INVALID, "null argument");
    DeviceGuard guard(c->device);
    DevCounters h{};
    HIP_TRY(hipMemcpyAsync(&h, c->dCounters[c->counterSet].ptr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 32; i++) out32[i] = h.depthRays[i];
    return PT_OK;
}

// Test hook (not part of the drop-in surface): evaluates one float32 building block of the
// kernels element-wise on the device.  fn: 0 sin, 1 cos, 2 asin, 3 atan2(a,b), 4 pow(a,b),
// 5 min, 6 max, 7 a/b, 8 sqrt, 9 1/sqrt.
int pt_debug_eval_math(int device, int fn, const float* a, const float* b, float* out, size_t n)
{
    if (!a || !b || !out) return fail(PT_ERR_INVALID, "null argument");
    if (n == 0) return PT_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(PT_ERR_DEVICE, "no such HIP device");
    DeviceGuard guard(device);
    DeviceBuffer da, db, dout;
    int rc = PT_OK;
    auto body = [&]() -> int {
        HIP_TRY(da.ensure(n * 4));
        HIP_TRY(db.ensure(n * 4));
        HIP_TRY(dout.ensure(n * 4));
        HIP_TRY(hipMemcpy(da.ptr, a, n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(db.ptr, b, n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(evalMathKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, (const float*)da.ptr,
                           (const float*)db.ptr, (float*)dout.ptr, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, dout.ptr, n * 4, hipMemcpyDeviceToHost));
        return PT_OK;
    };
    rc = body();
    da.release();
    db.release();
    dout.release();
    return rc;
}

}  // extern "C"

namespace {

// Scratch of one device build, released when the build returns.
struct LbvhScratch {
    DeviceBuffer in, keys, keysSorted, values, order, sortTemp, parent, left, right, first, last, box, height, visits, emit,
        kept, emitAt, keptAt, scanTemp, bounds;
    // clustering builder (ploc.h)
    DeviceBuffer pLeft, pRight, pSize, pHeight, pBox, pPosition, pKept, pPreorder, packedAt, idA, idB, boxA, boxB, nearest, flags, scan, finalOrder;
    ~LbvhScratch()
    {
        DeviceBuffer* all[] = {&in, &keys, &keysSorted, &values, &order, &sortTemp, &parent, &left, &right, &first, &last,
                               &box, &height, &visits, &emit, &kept, &emitAt, &keptAt, &scanTemp, &bounds,
                               &pLeft, &pRight, &pSize, &pHeight, &pBox, &pPosition, &pKept, &pPreorder, &packedAt, &idA, &idB, &boxA, &boxB, &nearest,
                               &flags, &scan, &finalOrder};
        for (DeviceBuffer* b : all) b->release();
    }
};

// Tree topology by locally-ordered clustering over the Morton-sorted triangles (ploc.h), written
// into the arrays the emit stage reads.  One small read-back per round (the cluster count).
int buildPlocTopology(pt_ctx* c, LbvhScratch& w, const RefPrimitiveAlign* dIn, int n, LbvhTree T)
{
    const size_t count = (size_t)n, internal = count - 1;
    HIP_TRY(w.pLeft.ensure(internal * 4));
    HIP_TRY(w.pRight.ensure(internal * 4));
    HIP_TRY(w.pSize.ensure(2 * count * 4));
    HIP_TRY(w.pHeight.ensure(internal * 4));
    HIP_TRY(w.pBox.ensure(2 * count * 24));
    HIP_TRY(w.pPosition.ensure(2 * count * 4));
    HIP_TRY(w.pKept.ensure(2 * count * 4));
    HIP_TRY(w.pPreorder.ensure(2 * count * 4));
    HIP_TRY(w.packedAt.ensure(internal * 4));
    HIP_TRY(w.idA.ensure(count * 4));
    HIP_TRY(w.idB.ensure(count * 4));
    HIP_TRY(w.boxA.ensure(count * 24));
    HIP_TRY(w.boxB.ensure(count * 24));
    HIP_TRY(w.nearest.ensure(count * 4));
    HIP_TRY(w.flags.ensure(count * 8));
    HIP_TRY(w.scan.ensure(count * 8));
    HIP_TRY(w.finalOrder.ensure(count * 4));
    hipStream_t st = c->stream;
    const dim3 block(256);
    auto grid = [](size_t items) { return dim3((unsigned int)((items + 255) / 256)); };
    PlocNodes N{(int*)w.pLeft.ptr,     (int*)w.pRight.ptr, (int*)w.pSize.ptr,    (int*)w.pHeight.ptr, (float*)w.pBox.ptr,
                (int*)w.pPosition.ptr, (int*)w.pKept.ptr,  (int*)w.pPreorder.ptr, T.leafMax};
    int* idIn = (int*)w.idA.ptr;
    int* idOut = (int*)w.idB.ptr;
    float* boxIn = (float*)w.boxA.ptr;
    float* boxOut = (float*)w.boxB.ptr;
    hipLaunchKernelGGL(plocInitKernel, grid(count), block, 0, st, dIn, (const unsigned int*)w.order.ptr, n, idIn, boxIn, N);
    HIP_TRY(hipGetLastError());
    size_t scanBytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (const unsigned long long*)w.flags.ptr,
                                             (unsigned long long*)w.scan.ptr, n, st));
    HIP_TRY(w.scanTemp.ensure(scanBytes));
    std::vector<std::pair<int, int>> rounds;  // (first node created, nodes created) per round
    int clusters = n, created = 0;
    while (clusters > 1) {
        // the fewer clusters are left, the higher in the tree the merges are and the cheaper a wide search is
        const int radius = clusters < (1 << 16) ? std::max(c->plocRadius, 256) : (clusters < (1 << 20) ? std::max(c->plocRadius, 128) : c->plocRadius);
        hipLaunchKernelGGL(plocNearestKernel, grid((size_t)clusters), block, 0, st, (const float*)boxIn, clusters, radius,
                           (int*)w.nearest.ptr);
        hipLaunchKernelGGL(plocFlagKernel, grid((size_t)clusters), block, 0, st, (const int*)w.nearest.ptr, clusters,
                           (unsigned long long*)w.flags.ptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.scanTemp.ptr, scanBytes, (const unsigned long long*)w.flags.ptr,
                                                 (unsigned long long*)w.scan.ptr, clusters, st));
        hipLaunchKernelGGL(plocMergeKernel, grid((size_t)clusters), block, 0, st, (const int*)idIn, (const float*)boxIn,
                           (const int*)w.nearest.ptr, (const unsigned long long*)w.flags.ptr,
                           (const unsigned long long*)w.scan.ptr, clusters, created, n, idOut, boxOut, N);
        HIP_TRY(hipGetLastError());
        unsigned long long lastScan = 0, lastFlag = 0;
        HIP_TRY(hipMemcpyAsync(&lastScan, (unsigned long long*)w.scan.ptr + (clusters - 1), 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&lastFlag, (unsigned long long*)w.flags.ptr + (clusters - 1), 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const unsigned long long total = lastScan + lastFlag;
        const int survivors = (int)(unsigned int)total, merges = (int)(total >> 32);
        if (merges < 1 || survivors != clusters - merges)
            return fail(PT_ERR_DEVICE, "clustering builder made no progress (internal error)");
        rounds.push_back({created, merges});
        created += merges;
        clusters = survivors;
        std::swap(idIn, idOut);
        std::swap(boxIn, boxOut);
    }
    if (created != n - 1) return fail(PT_ERR_DEVICE, "clustering builder lost nodes (internal error)");
    // Reinsertion passes: 16 by default, which is where the trees of the benchmark scenes stop getting better (a ray of the
    // 576 k-triangle scene visits 3.53 nodes instead of 3.79, from inside the mesh 17.6 instead of 20.2; the build takes
    // 34 ms instead of 20), 8 beyond four million triangles, where a pass takes 25 ms and more (20 M triangles: 12.4
    // visits instead of 13.8, 0.34 s instead of 0.11 s).
    const int optimisePasses = c->bvhOptimise >= 0 ? c->bvhOptimise : (count > 4000000 ? 8 : 16);
    if (optimisePasses > 0 && n >= 8) {
        // Reinsertion passes (reinsert.h) over the finished topology; the nodes are then no longer numbered level by
        // level, so positions and depth-first ranks come from each node's path to the root.
        // The passes' arrays live in buffers the clustering rounds are done with: parent links in `flags`, a move's
        // destination in `scan`, gains and lock words in one cluster-box array, "is moving" words and old places in
        // the other, the refit's counters in `nearest` and the two cluster-id arrays, a copy of the child links as they
        // were in the unsorted key array.
        const size_t total = 2 * count - 1;
        int* rParent = (int*)w.flags.ptr;
        int* rOut = (int*)w.scan.ptr;
        float* rGain = (float*)w.boxA.ptr;
        unsigned long long* rLock = (unsigned long long*)((char*)w.boxA.ptr + count * 8);
        unsigned long long* rMoving = (unsigned long long*)w.boxB.ptr;
        int* rOld = (int*)((char*)w.boxB.ptr + count * 16);
        unsigned int* rVisits = (unsigned int*)w.nearest.ptr;
        unsigned int* rDirty = (unsigned int*)w.idA.ptr;
        unsigned int* rStamp = (unsigned int*)w.idB.ptr;  // (the refit's stamps)
        int* rLeft0 = (int*)w.keys.ptr;
        int* rRight0 = (int*)((char*)w.keys.ptr + count * 4);
        ReinsertTree R{rParent, N.left, N.right, N.box, n};
        HIP_TRY(hipMemcpyAsync(rLeft0, N.left, internal * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(rRight0, N.right, internal * 4, hipMemcpyDeviceToDevice, st));
        auto refit = [&]() -> int {
            HIP_TRY(hipMemsetAsync(rVisits, 0, internal * 4, st));
            hipLaunchKernelGGL(reinsertRefitKernel, grid(count), block, 0, st, R, N, rVisits);
            HIP_TRY(hipGetLastError());
            return PT_OK;
        };
        hipLaunchKernelGGL(reinsertParentKernel, grid(internal), block, 0, st, R);
        const int mod = std::max(1, c->bvhOptimiseMod);
        // moves applied per pass, counted in the array the relabelling fills afterwards; read back once every candidate
        // has had its turn (`mod` passes): a round without a move ends the passes
        unsigned int* rMoves = (unsigned int*)w.packedAt.ptr;
        const int passes = std::min(optimisePasses, (int)std::min<size_t>(internal, 256));
        HIP_TRY(hipMemsetAsync(rMoves, 0, (size_t)passes * 4, st));
        std::vector<unsigned int> moves((size_t)passes, 0u);
        int heightNow = 0;
        HIP_TRY(hipMemcpyAsync(&heightNow, N.height + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int pass = 0; pass < passes; pass++) {
            HIP_TRY(hipMemsetAsync(rLock, 0, total * 8, st));
            HIP_TRY(hipMemsetAsync(rMoving, 0, total * 8, st));
            HIP_TRY(hipMemsetAsync(rOut, 0xff, total * 4, st));  // "stays where it is"
            // (blocks of one wave: the searches of a block's lanes differ widely in length)
            hipLaunchKernelGGL(reinsertSearchKernel, dim3((unsigned int)((total + 63) / 64)), dim3(64), 0, st, R, mod, pass % mod, rOut, rGain, rLock);
            hipLaunchKernelGGL(reinsertClaimKernel, grid(total), block, 0, st, R, rOut, (const float*)rGain,
                               (const unsigned long long*)rLock, rMoving);
            hipLaunchKernelGGL(reinsertRingKernel, grid(total), block, 0, st, R, rOut, (const unsigned long long*)rMoving);
            hipLaunchKernelGGL(reinsertApplyKernel, grid(total), block, 0, st, R, (const int*)rOut, rOld, rMoves + pass);
            // boxes, counts and heights above the moves' old and new places (the rest of the tree has not changed)
            HIP_TRY(hipMemsetAsync(rDirty, 0, internal * 4, st));
            HIP_TRY(hipMemsetAsync(rStamp, 0, internal * 4, st));
            hipLaunchKernelGGL(reinsertMarkKernel, grid(total), block, 0, st, R, (const int*)rOut, (const int*)rOld, rDirty);
            // one launch per level of flagged nodes; the root's stamp says when the refit is complete (the tree's height,
            // known from the pass before, is how many launches that takes unless the pass made the tree deeper)
            unsigned int round = 0, rootStamp = 0;
            for (;;) {
                for (int k = 0; k < heightNow + 2; k++)
                    hipLaunchKernelGGL(reinsertRefitMarkedKernel, grid(internal), block, 0, st, R, N, (const unsigned int*)rDirty, rStamp, round++);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(&moves[(size_t)pass], rMoves + pass, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(&rootStamp, rStamp + (n - 2), 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(&heightNow, N.height + (n - 2), 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (moves[(size_t)pass] == 0u || rootStamp != 0u) break;
                if (round > 4096u) return fail(PT_ERR_DEVICE, "tree optimisation: the refit does not end (internal error)");
            }
            if ((pass + 1) % mod == 0) {  // a round of passes in which every candidate had its turn and none moved: done
                unsigned int sum = 0;
                for (int k = pass + 1 - mod; k <= pass; k++) sum += moves[(size_t)k];
                if (sum == 0u) break;
            }
        }
        if (heightNow > c->bvhOptimiseMaxHeight) {  // deeper than the traversal takes: the clustered tree as it was
            HIP_TRY(hipMemcpyAsync(N.left, rLeft0, internal * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(N.right, rRight0, internal * 4, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(reinsertParentKernel, grid(internal), block, 0, st, R);
            const int rc = refit();
            if (rc != PT_OK) return rc;
        }
        hipLaunchKernelGGL(reinsertPlaceKernel, grid(total), block, 0, st, R, N);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemsetAsync((int*)w.pPosition.ptr + (n + n - 2), 0, 4, st));  // the root's triangles start at 0
        HIP_TRY(hipMemsetAsync((int*)w.pPreorder.ptr + (n + n - 2), 0, 4, st));  // and the root is the first kept node
        for (size_t r = rounds.size(); r-- > 0;)
            hipLaunchKernelGGL(plocPositionKernel, grid((size_t)rounds[r].second), block, 0, st, rounds[r].first, rounds[r].second, n, N);
    }
    hipLaunchKernelGGL(plocRelabelKernel, grid(count), block, 0, st, n, N, (const unsigned int*)w.order.ptr,
                       (unsigned int*)w.finalOrder.ptr, T, (unsigned int*)w.packedAt.ptr);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int buildLbvh(pt_ctx* c, const RefPrimitiveAlign* hostPrims, size_t count)
{
    if (count == 0) return fail(PT_ERR_INVALID, "no primitives");
    // (the kernels address node and triangle records by 32-bit byte offsets: both arrays stay below 4 GB)
    if (count >= (1u << 26)) return fail(PT_ERR_INVALID, "too many primitives (limit 2^26)");
    DeviceGuard guard(c->device);
    // The build writes into the live geometry buffers and can still fail after that: from here until it
    // succeeds the context has no geometry (a dispatch returns PT_ERR_STATE, never an old tree over new
    // triangles).
    c->deviceGeometry = false;
    c->geometryDirty = true;
    forgetTileCosts(c);
    c->nodes.clear();
    c->prims.clear();
    const int n = (int)count;
    const unsigned int un = (unsigned int)count;
    const size_t internal = count > 1 ? count - 1 : 1;
    // leaves of two triangles up to eight million triangles, of three beyond.  On the trees the reinsertion passes leave,
    // 1.3 M and 5.2 M triangles at 1920 x 1080 trace 2.9 % and 2.2 % faster with two than with four (2.5 % and 1.1 % faster
    // than with three); 20 M triangles at 3840 x 2160, which stream from HBM (0.65 of its peak), trace 5 618 Mrays/s with
    // three, 5 482 with four, 5 308 with two: there the extra nodes of small leaves cost more than the tests they save.
    const int leafMax = c->bvhLeafMax > 0 ? c->bvhLeafMax : (count > 8000000 ? 3 : 2);
    LbvhScratch w;
    HIP_TRY(w.in.ensure(count * sizeof(RefPrimitiveAlign)));
    HIP_TRY(w.keys.ensure(count * 8));
    HIP_TRY(w.keysSorted.ensure(count * 8));
    HIP_TRY(w.values.ensure(count * 4));
    HIP_TRY(w.order.ensure(count * 4));
    HIP_TRY(w.parent.ensure((2 * count) * 4));
    HIP_TRY(w.left.ensure(internal * 4));
    HIP_TRY(w.right.ensure(internal * 4));
    HIP_TRY(w.first.ensure(internal * 4));
    HIP_TRY(w.last.ensure(internal * 4));
    HIP_TRY(w.box.ensure((2 * count) * 6 * 4));
    HIP_TRY(w.height.ensure(internal * 4));
    HIP_TRY(w.visits.ensure(internal * 4));
    HIP_TRY(w.emit.ensure(internal * 4));
    HIP_TRY(w.kept.ensure(internal * 4));
    HIP_TRY(w.emitAt.ensure(internal * 4));
    HIP_TRY(w.keptAt.ensure(internal * 4));
    HIP_TRY(w.bounds.ensure(sizeof(LbvhBounds)));
    HIP_TRY(c->dRefPrims.ensure(count * sizeof(RefPrimitiveAlign)));
    HIP_TRY(c->dTriangles.ensure(count * sizeof(DevTriangle)));
    HIP_TRY(c->dShading.ensure(count * sizeof(DevShading)));

    hipStream_t st = c->stream;
    const dim3 block(256), gridPrims((un + 255u) / 256u), gridNodes((unsigned int)((internal + 255) / 256));
    const RefPrimitiveAlign* dIn = (const RefPrimitiveAlign*)w.in.ptr;
    HIP_TRY(hipMemcpyAsync(w.in.ptr, hostPrims, count * sizeof(RefPrimitiveAlign), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lbvhInitBoundsKernel, dim3(1), dim3(64), 0, st, (LbvhBounds*)w.bounds.ptr);
    hipLaunchKernelGGL(lbvhBoundsKernel, dim3(std::min(gridPrims.x, 2048u)), block, 0, st, dIn, un, (LbvhBounds*)w.bounds.ptr);
    hipLaunchKernelGGL(lbvhMortonKernel, gridPrims, block, 0, st, dIn, un, (const LbvhBounds*)w.bounds.ptr,
                       (unsigned long long*)w.keys.ptr, (unsigned int*)w.values.ptr);
    HIP_TRY(hipGetLastError());
    size_t tempBytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, (const unsigned long long*)w.keys.ptr,
                                               (unsigned long long*)w.keysSorted.ptr, (const unsigned int*)w.values.ptr,
                                               (unsigned int*)w.order.ptr, n, 0, 63, st));
    HIP_TRY(w.sortTemp.ensure(tempBytes));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.sortTemp.ptr, tempBytes, (const unsigned long long*)w.keys.ptr,
                                               (unsigned long long*)w.keysSorted.ptr, (const unsigned int*)w.values.ptr,
                                               (unsigned int*)w.order.ptr, n, 0, 63, st));

    LbvhTree T{(int*)w.parent.ptr, (int*)w.left.ptr,   (int*)w.right.ptr,  (int*)w.first.ptr,
               (int*)w.last.ptr,   (float*)w.box.ptr,  (int*)w.height.ptr, (unsigned int*)w.visits.ptr, leafMax};
    HIP_TRY(hipMemsetAsync(w.visits.ptr, 0, internal * 4, st));
    HIP_TRY(hipMemsetAsync(w.height.ptr, 0, internal * 4, st));
    const unsigned int* finalOrder = (const unsigned int*)w.order.ptr;
    const bool clustered = c->plocRadius > 0 && n > leafMax;
    if (clustered) {
        int rc = buildPlocTopology(c, w, dIn, n, T);
        if (rc != PT_OK) return rc;
        finalOrder = (const unsigned int*)w.finalOrder.ptr;
    } else {
        if (n > 1) hipLaunchKernelGGL(lbvhTreeKernel, gridNodes, block, 0, st, (const unsigned long long*)w.keysSorted.ptr, n, T);
        hipLaunchKernelGGL(lbvhFitKernel, gridPrims, block, 0, st, dIn, (const unsigned int*)w.order.ptr, n, T);
    }
    hipLaunchKernelGGL(lbvhEmitTrianglesKernel, gridPrims, block, 0, st, dIn, finalOrder, un,
                       (RefPrimitiveAlign*)c->dRefPrims.ptr, (DevTriangle*)c->dTriangles.ptr, (DevShading*)c->dShading.ptr);
    HIP_TRY(hipGetLastError());

    size_t refNodeCount = 2, devNodeCount = 1;
    int height = 0;
    LbvhBounds hb{};
    if (n > leafMax) {
        hipLaunchKernelGGL(lbvhCountKernel, gridNodes, block, 0, st, n, T, (unsigned int*)w.emit.ptr, (unsigned int*)w.kept.ptr);
        HIP_TRY(hipGetLastError());
        size_t scanBytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, (const unsigned int*)w.emit.ptr, (unsigned int*)w.emitAt.ptr,
                                                 n - 1, st));
        HIP_TRY(w.scanTemp.ensure(scanBytes));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.scanTemp.ptr, scanBytes, (const unsigned int*)w.emit.ptr,
                                                 (unsigned int*)w.emitAt.ptr, n - 1, st));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.scanTemp.ptr, scanBytes, (const unsigned int*)w.kept.ptr,
                                                 (unsigned int*)w.keptAt.ptr, n - 1, st));
        unsigned int tail[4] = {0, 0, 0, 0};  // last elements of emit, emitAt, kept, keptAt
        HIP_TRY(hipMemcpyAsync(&tail[0], (unsigned int*)w.emit.ptr + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&tail[1], (unsigned int*)w.emitAt.ptr + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&tail[2], (unsigned int*)w.kept.ptr + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&tail[3], (unsigned int*)w.keptAt.ptr + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&height, w.height.ptr, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&hb, w.bounds.ptr, sizeof(hb), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        refNodeCount = 1 + (size_t)tail[0] + tail[1];
        devNodeCount = 1 + (size_t)tail[2] + tail[3];
        if (height > 64)
            return fail(PT_ERR_SCENE, "the device-built BVH is " + std::to_string(height) +
                                          " levels deep; at most 64 are taken (use the host builder)");
        HIP_TRY(c->dRefNodes.ensure(refNodeCount * sizeof(RefNodeAlign)));
        HIP_TRY(c->dNodes.ensure(devNodeCount * sizeof(DevNode)));
        HIP_TRY(hipMemsetAsync(c->dRefNodes.ptr, 0, sizeof(RefNodeAlign), st));
        HIP_TRY(hipMemsetAsync(c->dNodes.ptr, 0, sizeof(DevNode), st));
        // packed node index: depth-first rank for the clustered tree, index order for the radix tree
        hipLaunchKernelGGL(lbvhEmitNodesKernel, gridNodes, block, 0, st, n, T, (const unsigned int*)w.emitAt.ptr,
                           (const unsigned int*)(clustered ? w.packedAt.ptr : w.keptAt.ptr), (const unsigned int*)w.kept.ptr, (RefNodeAlign*)c->dRefNodes.ptr,
                           (DevNode*)c->dNodes.ptr);
        HIP_TRY(hipGetLastError());
        c->rootRef = 1;
    } else {
        HIP_TRY(c->dRefNodes.ensure(2 * sizeof(RefNodeAlign)));
        HIP_TRY(c->dNodes.ensure(sizeof(DevNode)));
        HIP_TRY(hipMemsetAsync(c->dRefNodes.ptr, 0, 2 * sizeof(RefNodeAlign), st));
        hipLaunchKernelGGL(lbvhEmitRootLeafKernel, dim3(1), dim3(64), 0, st, n, T, (RefNodeAlign*)c->dRefNodes.ptr,
                           (DevNode*)c->dNodes.ptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&hb, w.bounds.ptr, sizeof(hb), hipMemcpyDeviceToHost, st));
        c->rootRef = ~(int32_t)((0u << 3) | (unsigned int)(n - 1));
    }
    HIP_TRY(hipStreamSynchronize(st));  // the scratch buffers die with this scope
    c->maxDepth = height;
    {
        const int rcw = collapseTree(c, devNodeCount);  // sets nodeBytes
        if (rcw != PT_OK) return rcw;
    }
    forgetTileCosts(c);
    c->deviceGeometry = true;
    c->geometryDirty = false;
    c->forkDirty = true;  // (the builder's kernels ran on the context's stream)
    c->nodes.clear();
    c->prims.clear();
    c->deviceNodeCount = refNodeCount;
    c->devicePrimCount = count;
    c->deviceMaxMeshIndex = hb.maxMeshIndex;
    c->triangleBytes = count * sizeof(DevTriangle);
    c->shadingBytes = count * sizeof(DevShading);
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_build_bvh_device(pt_ctx* c, const void* prims112, size_t count)
{
    if (!c || !prims112) return fail(PT_ERR_INVALID, "null argument");
    return buildLbvh(c, (const RefPrimitiveAlign*)prims112, count);
}

int pt_process_buffers_device(pt_ctx* c)
{
    if (!c) return fail(PT_ERR_INVALID, "null context");
    if (c->models.empty()) return fail(PT_ERR_STATE, "no models added");
    size_t triangles = 0;
    for (const PendingModel& m : c->models) triangles += m.indices.size() / 3;
    if (triangles == 0) return fail(PT_ERR_STATE, "models contain no triangles");
    std::vector<RefPrimitive> soup(triangles);
    std::vector<RefMaterialAlign> mats;
    size_t at = 0;
    uint32_t meshIndex = 0;
    for (const PendingModel& m : c->models) {
        AssemblePrimitives(m.vertices.data(), m.indices.data(), m.indices.size(), 4, meshIndex, soup.data() + at);
        at += m.indices.size() / 3;
        mats.push_back(m.material);
        meshIndex++;
    }
    std::vector<RefPrimitiveAlign> aligned(soup.size());
    AlignPrimitives(soup.data(), soup.size(), aligned.data());
    int rc = pt_upload_materials(c, mats.data(), mats.size());
    if (rc != PT_OK) return rc;
    return buildLbvh(c, aligned.data(), aligned.size());
}

int pt_download_bvh(pt_ctx* c, void* nodes48Out, size_t nodeCapacity, size_t* nodeCountOut, void* prims112Out,
                    size_t primCapacity, size_t* primCountOut)
{
    if (!c || !nodeCountOut || !primCountOut) return fail(PT_ERR_INVALID, "null argument");
    if (!c->deviceGeometry) return fail(PT_ERR_STATE, "the current geometry was not built on the device");
    *nodeCountOut = c->deviceNodeCount;
    *primCountOut = c->devicePrimCount;
    if (!nodes48Out && !prims112Out) return PT_OK;  // size query
    if (!nodes48Out || !prims112Out) return fail(PT_ERR_INVALID, "null output buffer");
    if (nodeCapacity < c->deviceNodeCount || primCapacity < c->devicePrimCount)
        return fail(PT_ERR_INVALID, "output capacity too small: need " + std::to_string(c->deviceNodeCount) + " nodes and " +
                                        std::to_string(c->devicePrimCount) + " primitives");
    DeviceGuard guard(c->device);
    HIP_TRY(hipMemcpyAsync(nodes48Out, c->dRefNodes.ptr, c->deviceNodeCount * sizeof(RefNodeAlign), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(prims112Out, c->dRefPrims.ptr, c->devicePrimCount * sizeof(RefPrimitiveAlign), hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

}  // extern "C"
