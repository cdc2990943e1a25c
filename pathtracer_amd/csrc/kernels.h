// HIP kernels of the path-tracing core for gfx950 (wave64).
//
// Two schedulers drive the same per-ray and per-hit device code (same bits out of both):
//
//  * MEGAKERNEL (default): a persistent per-wave state machine in ONE launch.  A wave covers an
//    8x8 pixel tile and owns the tile's (pixel, sample) pairs of the launch as a pool; a lane that
//    needs work takes the next pair.  Every lane is in one of four modes -- NEW (take a sample),
//    SHADE (a closest-hit query finished), TRAV (a BVH query is in flight), FINISH (write the
//    sample out) -- and the wave alternates between
//      - a traversal phase, in which all TRAV lanes advance their query one step per iteration
//        (closest-hit and shadow queries side by side) until the share of lanes still traversing
//        falls below a threshold, and
//      - a shading phase, in which the lanes whose query finished shade their miss or (once enough
//        of them wait) their hit, write finished samples to the sample buffer, generate primary
//        rays for new ones, and re-enter the traversal phase next to the lanes whose query is still
//        under way (their stack and cursor persist).
//    Path state never leaves the registers.  resolveKernel then folds the sample buffer into the
//    images in FrameCount order.
//  * WAVEFRONT pipeline (kept for A/B measurements): path state in global memory (one slot per
//    pixel); per iteration a persistent traverseKernel pulls tagged ray entries from a queue
//    (ballot-and-prefix hand-out from per-wave chunks) and a shadeKernel shades one hit per lane
//    and appends the next rays with ballot-and-prefix compaction.  On this path's workloads
//    (short paths, a 160-byte state slot per pixel streamed through HBM every iteration, ~70
//    iterations of two launches for 16 samples) it measured 2.7x slower than the megakernel.
//
// Common rules
//   * A pixel's samples enter its running mean in FrameCount order (the megakernel's resolve pass
//     and the wavefront pipeline's sample-after-sample walk both see to it), so the images carry the
//     bits of frame-by-frame dispatches.
//   * One shading pass emits both rays of a bounce.  The shader traces the shadow ray in the
//     middle of its bounce; everything it computes after an unoccluded shadow ray (light
//     radiance, BSDF value, MIS weight) does not depend on the ray's outcome, so it is
//     evaluated up front into a pending contribution that the shadow query merely gates.
//   * The per-lane stack of postponed far children lives in LDS (entry k of thread t at word
//     k*blockDim + t: consecutive lanes hit consecutive banks) with its newest entry cached in a
//     register; the near child is followed directly.
//
// Same results as the reference shader's traversal (intersection.glsl:115-199,230-315):
// children are visited near-first with the shader's tie rule, leaves are scanned in order
// with a strict `<`, so the same triangle wins.  Two things differ without changing the
// result: sub-trees whose box entry distance lies beyond the best hit so far are skipped
// (the shader never prunes), and the stack never overflows (the shader's 16-entry stack
// aborts traversal when it does).
#pragma once
#include "shading.h"

namespace ptcore {

constexpr int kBlockThreads = 256;
// Waves per SIMD of a megakernel instantiation that does not say (the host's table names it for every variant it
// launches: 3, i.e. 168 VGPRs and next to nothing spilled; the 4-wave build, 128 VGPRs and some 55 values spilled
// in the shading blocks, is 14 % slower on the four-wide kernel, 2 waves 35 %, 5 waves 80 %).  What bounds the
// kernel is the vector L1's access rate (one access per cycle and CU, two for a lane's 16-byte read), which more
// resident waves do not raise.
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 4
#endif
// Threads per megakernel block: 64 (one wave, one 8x8 pixel tile) or 256 (a 16x16 tile).
#ifndef PT_MEGA_THREADS
#define PT_MEGA_THREADS 64
#endif
constexpr int kMegaThreads = PT_MEGA_THREADS;
constexpr unsigned kMegaTile = kMegaThreads == 64 ? 8u : 16u;  // tile edge in pixels
// Entry distance must exceed the best hit by this factor before a sub-tree is skipped: box and
// triangle distances are rounded independently, the margin keeps a triangle whose computed
// distance ties or narrowly beats the current best from being culled with its box.
constexpr float kPruneMargin = 1.0009765625f;  // 1 + 2^-10

// Slab test of intersection.glsl:63-77.  The shader returns the entry distance, or the exit distance
// when the origin is inside, or -1 on a miss, and its callers test that value for `> 0`; with t1 >= t0
// (no miss) that value is positive exactly when t1 is, so the hit decision is taken from t0 and t1
// directly.  `dist` is the shader's return value of a hit (the children are ordered by it), `t0out` the
// entry distance (for pruning).
PT_DEV bool hitAabb(v3 o, v3 inv, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& dist,
                    float& t0out)
{
    float fx = (maxx - o.x) * inv.x, fy = (maxy - o.y) * inv.y, fz = (maxz - o.z) * inv.z;
    float nx = (minx - o.x) * inv.x, ny = (miny - o.y) * inv.y, nz = (minz - o.z) * inv.z;
    float t1 = fmin_(fmax_(fx, nx), fmin_(fmax_(fy, ny), fmax_(fz, nz)));
    float t0 = fmax_(fmin_(fx, nx), fmax_(fmin_(fy, ny), fmin_(fz, nz)));
    t0out = t0;
    dist = (t0 > 0.0f) ? t0 : t1;
    return (t1 >= t0) & (t1 > 0.0f);
}

// ------------------------------------------------------------------------------ BVH query
struct Query {
    v3 ro, rd, inv;
    int cur, sp, top;  // `top` caches the newest stack entry; LDS holds the older ones
    unsigned leafPos, leafEnd;
    float bestT, bestU, bestV;
    int bestPrim;
    bool isShadow;
};

struct StepCounters {
    unsigned nodeVisits, triangleTests, maxStack;
};

PT_DEV v3 reciprocal3(v3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
// The reciprocal direction of a query, for the box tests alone (the triangle test divides exactly): the
// hardware's approximate reciprocal (1 ulp, one instruction where the correctly rounded division takes ten),
// kept finite (a component of 0 gives +-2^90, not infinity), because the grid-plane distances are
// multiply-adds in which 0 * infinity would be a NaN.  For a ray parallel to a pair of slab planes the two
// distances are then huge and of opposite sign inside the slab, huge and of the same sign outside it, which
// decides the box test as the infinities would.
PT_DEV float finiteReciprocal(float d)
{
    const float big = 1.2379400392853803e27f;  // 2^90
    return fmin_(fmax_(__builtin_amdgcn_rcpf(d), -big), big);
}
PT_DEV v3 queryReciprocal3(v3 d) { return mk3(finiteReciprocal(d.x), finiteReciprocal(d.y), finiteReciprocal(d.z)); }

// `inv` = queryReciprocal3(rd), for callers that have it at hand.
PT_DEV void beginQuery(Query& q, int rootRef, v3 ro, v3 rd, v3 inv, bool isShadow)
{
    q.ro = ro;
    q.rd = rd;
    q.inv = inv;
    q.bestT = kFltMax;
    q.bestPrim = -1;  // (bestU and bestV are read only once bestPrim says there is a hit)
    q.sp = 0;
    q.top = 0;
    q.cur = rootRef;
    q.leafPos = q.leafEnd;  // no leaf parked, whatever the previous query left
    q.isShadow = isShadow;
}
PT_DEV void beginQuery(Query& q, int rootRef, v3 ro, v3 rd, bool isShadow) { beginQuery(q, rootRef, ro, rd, queryReciprocal3(rd), isShadow); }

// What queryStep reports: the query goes on, it is complete (a closest-hit query's result is in best*,
// a shadow query found no occluder), or a shadow query met an occluder (and is complete).
enum QueryState : int { kQueryGoesOn = 0, kQueryDone = 1, kQueryOccluded = 2 };

// Advances the query by one iteration and reports its state.  A lane can
// hold a (four-wide) interior node to visit (`cur` > 0) and, next to it, a leaf whose triangles are still
// being scanned (`leafPos` < `leafEnd`): reaching a leaf does not interrupt the walk, the leaf is
// parked and its triangles are tested one per iteration while the lane goes on visiting nodes.  A
// second leaf reference waits in `cur` (< 0) until the parked leaf is done, so leaves are scanned
// in the order they were reached and triangles in leaf order.  The node and the triangle record
// are fetched together before either is used: one memory round trip per iteration.  (Holding the
// triangle half back until several lanes of the wave have a leaf was measured: every threshold
// above one lane is slower, the lanes that wait cost more than the fuller triangle steps save.)
// `stack` points at this thread's column of the block's LDS stack (entries STRIDE words apart).
// With LDS_DEPTH > 0 only the first LDS_DEPTH entries live there and deeper ones go to `deep`, an
// array in the lane's private memory: deep trees then cost no more LDS (and no fewer resident
// waves) than shallow ones, and the entries past LDS_DEPTH are touched by very few rays.
template <int STRIDE, int LDS_DEPTH>
PT_DEV void stackPut(int* stack, int* deep, int i, int v)
{
    if (LDS_DEPTH == 0 || i < LDS_DEPTH) stack[i * STRIDE] = v;
    else deep[i - LDS_DEPTH] = v;
}
// (With a private part, the LDS load is unconditional and the private one sits in a branch of its own
// behind it, volatile so that it stays there: given one plain load in each arm of a branch the compiler
// merges them into a single access through a selected generic pointer, a flat load slower than either.)
template <int STRIDE, int LDS_DEPTH>
PT_DEV int stackGet(const int* stack, const int* deep, int i)
{
    if (LDS_DEPTH == 0) return stack[i * STRIDE];
    int v = stack[(i < LDS_DEPTH ? i : LDS_DEPTH - 1) * STRIDE];
    if (i >= LDS_DEPTH) v = const_cast<const volatile int*>(deep)[i - LDS_DEPTH];
    return v;
}

// The stack: `top` is the newest entry (0 when there is none), row k of the column holds what was
// `top` when the k-th entry was pushed, so row 0 always holds 0 and a pop needs no test for an empty
// stack: it takes `top`, steps `sp` down (not below 0) and reloads `top` from row `sp`.  Row 0 is
// zeroed once by the kernel; the first push of every query rewrites it with the same 0.
template <int STRIDE, int LDS_DEPTH>
PT_DEV void stackPop(Query& q, const int* stack, const int* deep)
{
    q.cur = q.top;
    q.sp = q.sp > 0 ? q.sp - 1 : 0;
    q.top = stackGet<STRIDE, LDS_DEPTH>(stack, deep, q.sp);
}

// Fills in the material records' refraction tint, pow(albedo, 0.5) per channel, with the device's own pow: a hit then
// reads it instead of taking three pow (the shader takes them per evaluated direction, disney.glsl:70).
__global__ void materialTintKernel(DevMaterial* __restrict__ mats, unsigned count)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    mats[i].glassTint[0] = pow_(mats[i].albedo[0], 0.5f);
    mats[i].glassTint[1] = pow_(mats[i].albedo[1], 0.5f);
    mats[i].glassTint[2] = pow_(mats[i].albedo[2], 0.5f);
}

// The traversal kernels walk a four-wide tree made of the binary one, level by level from the root (one launch
// of collapseKernel per level: `frontier` holds (binary node, index of its wide node) pairs, `counters` the next
// free wide node and the size of the next frontier).  A wide node starts from the binary node's two children;
// while a slot is free, the interior entry with the largest surface area is replaced by its own two children
// (`uniform`: both children are always opened instead, which at least halves the depth).  On the reference's
// tree of the benchmark scene, whose builder caps its split cost, a ray then visits 6.0 nodes instead of 15.2.
// Empty slots hold NaN boxes, which no ray enters.
__global__ void collapseKernel(const DevNode* __restrict__ in, DevNodeQ* __restrict__ out, const int2* __restrict__ frontier,
                               unsigned count, int2* __restrict__ next, unsigned* __restrict__ counters, int uniform)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int2 item = frontier[i];
    const DevNode n = in[item.x];
    float mn[4][3], mx[4][3];
    int ref[4] = {0, 0, 0, 0};
    int used = 2;
    for (int a = 0; a < 3; a++) {
        mn[0][a] = n.lmin[a], mx[0][a] = n.lmax[a];
        mn[1][a] = n.rmin[a], mx[1][a] = n.rmax[a];
    }
    ref[0] = n.left, ref[1] = n.right;
    auto open = [&](int k) {  // entry k gives way to its two children
        const DevNode m = in[ref[k]];
        for (int a = 0; a < 3; a++) {
            mn[k][a] = m.lmin[a], mx[k][a] = m.lmax[a];
            mn[used][a] = m.rmin[a], mx[used][a] = m.rmax[a];
        }
        ref[k] = m.left;
        ref[used++] = m.right;
    };
    if (uniform) {
        const bool second = ref[1] > 0;
        if (ref[0] > 0) open(0);
        if (second) open(1);
    }
    while (!uniform && used < 4) {
        int pick = -1;
        float bestArea = -1.0f;
        for (int k = 0; k < used; k++) {
            if (ref[k] <= 0) continue;
            const float dx = mx[k][0] - mn[k][0], dy = mx[k][1] - mn[k][1], dz = mx[k][2] - mn[k][2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > bestArea) bestArea = area, pick = k;
        }
        if (pick < 0) break;
        open(pick);
    }
    // the interior entries get wide nodes of their own in the next level, side by side
    int interior = 0;
    for (int k = 0; k < used; k++) interior += ref[k] > 0;
    int dst = interior ? (int)atomicAdd(&counters[0], (unsigned)interior) : 0;
    int at = interior ? (int)atomicAdd(&counters[1], (unsigned)interior) : 0;
    DevNodeQ w;
    for (int k = 0; k < 4; k++) {
        int r = k < used ? ref[k] : 0;
        if (r > 0) {
            next[at++] = make_int2(r, dst);
            r = dst++;
        }
        w.child[k] = r;
    }
    // The grid: per axis the smallest power-of-two cell with which 256 planes span the children's boxes with
    // kQuantMargin cells to spare on either side; every bound goes to the nearest plane outside it, then that many
    // cells further out.  Containment is checked in double precision, where origin + q * cell is exact.
    for (int a = 0; a < 3; a++) {
        float lo = mn[0][a], hi = mx[0][a];
        for (int k = 1; k < used; k++) lo = fmin_(lo, mn[k][a]), hi = fmax_(hi, mx[k][a]);
        const double span = (double)hi - (double)lo;
        const int cells = 255 - 2 * kQuantMargin - 1;  // (one more cell for the rounding of the origin)
        float cell = 7.8886090522101181e-31f;          // 2^-100: any cell does for a flat node
        if (span > 0.0) {
            int e = 0;
            (void)frexp(span / (double)cells, &e);  // span / cells = m * 2^e, 0.5 <= m < 1
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            cell = (float)ldexp(1.0, e);
        }
        unsigned wlo = 0, whi = 0;
        for (int attempt = 0;; attempt++) {
            const float origin = lo - (float)(kQuantMargin + 1) * cell;  // (rounds towards lo at worst: the checks below see it)
            bool ok = (double)origin <= (double)lo;
            wlo = 0, whi = 0;
            for (int k = 0; k < 4 && ok; k++) {
                long long ql = 255, qh = 0;  // an empty slot: inverted, entered by no ray
                if (k < used) {
                    ql = (long long)floor(((double)mn[k][a] - (double)origin) / (double)cell) - kQuantMargin;
                    qh = (long long)ceil(((double)mx[k][a] - (double)origin) / (double)cell) + kQuantMargin;
                    ql = ql < 0 ? 0 : ql;
                    ok = qh <= 255 && (double)origin + (double)ql * (double)cell <= (double)mn[k][a] &&
                         (double)origin + (double)qh * (double)cell >= (double)mx[k][a];
                }
                wlo |= (unsigned)(ql & 255) << (8 * k);
                whi |= (unsigned)(qh & 255) << (8 * k);
            }
            if (ok || attempt > 300) {  // (boxes that are not finite never fit: the last grid tried stands)
                w.origin[a] = origin;
                break;
            }
            cell = cell * 2.0f;
        }
        w.scale[a] = cell;
        w.qlo[a] = wlo;
        w.qhi[a] = whi;
    }
    out[item.y] = w;
}

template <bool COUNT, int STRIDE = kBlockThreads, int LDS_DEPTH = 0>
PT_DEV int queryStep(const DevScene& S, Query& q, int* stack, StepCounters& sc, int* deep = nullptr)
{
    const bool hasNode = q.cur > 0;
    const bool doTri = q.leafPos < q.leafEnd;
    // Both records are requested up front by every calling lane -- record 0 of either array (always
    // allocated) stands in when the lane has no node or no leaf -- so the loads sit in straight-line
    // code and are all in flight before the first use.
    // (32-bit byte offsets from the array bases: the loads then take a scalar base and a 32-bit lane
    // offset, with no 64-bit address arithmetic in the loop; the host refuses arrays beyond 4 GB)
    const unsigned nodeOff = (hasNode ? (unsigned)q.cur : 0u) << 6;
    const unsigned triIdx16 = (doTri ? q.leafPos : 0u) << 4;
    unsigned triOff;  // x 48 as one full-rate shift-add (the compiler would fold it into a quarter-rate 32-bit multiply)
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(triOff) : "v"(triIdx16));
    static_assert(sizeof(DevNodeQ) == 64 && sizeof(DevTriangle) == 48, "record sizes are wired into the offsets above");
    const float4* nrec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.wide) + nodeOff);
    const float4* trec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.triangles) + triOff);
    const float4 n0 = nrec[0], n1 = nrec[1], n2 = nrec[2];  // origin xyz, cell x | cell yz, lower planes xy | lower z, upper xyz
    const int4 kids = *reinterpret_cast<const int4*>(nrec + 3);
#ifdef PT_EXPERIMENT_EXTRA_LOAD  // what one more 16-byte read per node visit costs
    const float4 extraRead = nrec[4 + PT_EXPERIMENT_EXTRA_LOAD];
#endif
    const float4 t0 = trec[0], t1 = trec[1], t2 = trec[2];
    bool finished = false;  // query complete
    bool pop = false;
    // Both halves run for every calling lane, unbranched -- a lane without a node visits record 0 and a lane without a
    // triangle tests record 0, their results masked -- which keeps all seven reads at the head of the step, the node's
    // ahead of the triangle's.  Reads return in order: the node half, first to use its data, then waits for the node
    // only, and the triangle's record has the node half's instructions to arrive in.  (Given a branch around either
    // half the compiler moves that half's reads into it, behind the other half: two round trips a step, or the node
    // waiting for the triangle.  1.5-3 % of the launch.)
    {
        if (COUNT) sc.nodeVisits += hasNode;
#ifdef PT_EXPERIMENT_EXTRA_LOAD
        asm volatile("" ::"v"(extraRead.x));
#endif
#ifdef PT_EXPERIMENT_EXTRA_VALU  // what this many more vector instructions per node visit cost
        {
            float e0 = q.ro.x, e1 = q.ro.y, e2 = q.ro.z, e3 = q.rd.x;
            for (int i = 0; i < PT_EXPERIMENT_EXTRA_VALU / 4; i++)
                asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3"
                             : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
            asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(e3));
        }
#endif
        const float limit = q.bestT * kPruneMargin;  // bestT stays FLT_MAX for shadow rays
        // Slab tests of the four children against grid planes (intersection.glsl:63-77 tests the exact boxes; these
        // contain them).  The ray is moved into the node's grid once -- t(plane p) = p * (cell * inv) + (origin - ro) * inv,
        // one multiply-add per plane -- and the planes a ray meets first on each axis are picked by the sign of its
        // direction, which also leaves the inverted box of an empty slot with its exit before its entry.
        const float ax = (n0.x - q.ro.x) * q.inv.x, ay = (n0.y - q.ro.y) * q.inv.y, az = (n0.z - q.ro.z) * q.inv.z;
        const float bx = n0.w * q.inv.x, by = n1.x * q.inv.y, bz = n1.y * q.inv.z;
        const unsigned lox = __float_as_uint(n1.z), loy = __float_as_uint(n1.w), loz = __float_as_uint(n2.x);
        const unsigned hix = __float_as_uint(n2.y), hiy = __float_as_uint(n2.z), hiz = __float_as_uint(n2.w);
        const bool negx = q.inv.x < 0.0f, negy = q.inv.y < 0.0f, negz = q.inv.z < 0.0f;
        const unsigned nearx = negx ? hix : lox, farx = negx ? lox : hix;
        const unsigned neary = negy ? hiy : loy, fary = negy ? loy : hiy;
        const unsigned nearz = negz ? hiz : loz, farz = negz ? loz : hiz;
#define PT_PLANE(word, k, b, a) __builtin_fmaf((float)(((word) >> (8 * (k))) & 255u), b, a)
        // A child is entered when the ray's segment [0, limit] and its three slab intervals overlap: the latest
        // entry (0 and the three near planes) is no later than the earliest exit (the limit and the three far
        // planes).  The children are ordered by that entry distance (the shader orders its two by the entry
        // distance, or the exit distance when the origin is inside the box: the order decides nothing but which
        // of two hits at exactly the same distance is found first).
#define PT_CHILD(k, hit, dist)                                                                                                        \
    {                                                                                                                                 \
        const float e = fmax_(fmax_(fmax_(PT_PLANE(nearx, k, bx, ax), PT_PLANE(neary, k, by, ay)), PT_PLANE(nearz, k, bz, az)), 0.0f); \
        const float x = fmin_(fmin_(fmin_(PT_PLANE(farx, k, bx, ax), PT_PLANE(fary, k, by, ay)), PT_PLANE(farz, k, bz, az)), limit);  \
        dist = e;                                                                                                                     \
        hit = e <= x;                                                                                                                 \
    }
        float d0, d1, d2, d3;
        bool h0, h1, h2, h3;
        PT_CHILD(0, h0, d0)
        PT_CHILD(1, h1, d1)
        PT_CHILD(2, h2, d2)
        PT_CHILD(3, h3, d3)
#undef PT_CHILD
#undef PT_PLANE
        // The nearest child entered is visited next and the others wait on the stack.  Three comparators over (distance,
        // reference) pairs, in which a child not entered is the farthest, bring the nearest to the front and leave
        // the rest half ordered (the larger of each first pair behind the smaller); finishing the sorting network
        // (two more comparators) visits no fewer nodes and costs 2 % on the reference's tree.  A push that is not one
        // writes the row above the stack's top, which nothing reads.
        // (an empty slot has reference 0 and an inverted box, which no ray enters; should a degenerate grid let one in
        // all the same, the slot sorts by its distance like a child, and a reference of 0 in front only means that the
        // step pops -- the children entered are on the stack by then, nothing is skipped)
        h0 &= hasNode, h1 &= hasNode, h2 &= hasNode, h3 &= hasNode;
        int r0 = h0 ? kids.x : 0, r1 = h1 ? kids.y : 0, r2 = h2 ? kids.z : 0, r3 = h3 ? kids.w : 0;
        float k0 = h0 ? d0 : kFltMax, k1 = h1 ? d1 : kFltMax, k2 = h2 ? d2 : kFltMax, k3 = h3 ? d3 : kFltMax;
#define PT_CSWAP(ka, ra, kb, rb)          \
    {                                     \
        const bool sw = kb < ka;          \
        const float kt = fmin_(ka, kb);   \
        const int rt = sw ? rb : ra;      \
        kb = fmax_(ka, kb);               \
        rb = sw ? ra : rb;                \
        ka = kt;                          \
        ra = rt;                          \
    }
        PT_CSWAP(k0, r0, k1, r1)
        PT_CSWAP(k2, r2, k3, r3)
        PT_CSWAP(k0, r0, k2, r2)
#undef PT_CSWAP
        const bool p3 = r3 != 0, p2 = r2 != 0, p1 = r1 != 0;
        stackPut<STRIDE, LDS_DEPTH>(stack, deep, q.sp, q.top);
        q.top = p3 ? r3 : q.top;
        q.sp += p3;
        stackPut<STRIDE, LDS_DEPTH>(stack, deep, q.sp, q.top);
        q.top = p2 ? r2 : q.top;
        q.sp += p2;
        stackPut<STRIDE, LDS_DEPTH>(stack, deep, q.sp, q.top);
        q.top = p1 ? r1 : q.top;
        q.sp += p1;
        if (COUNT) sc.maxStack = (unsigned)q.sp > sc.maxStack ? (unsigned)q.sp : sc.maxStack;
        q.cur = r0 != 0 ? r0 : q.cur;
        pop = hasNode & (r0 == 0);
    }
    {
        // Moller-Trumbore with the shader's expressions (intersection.glsl:26-40); the edges and
        // their cross product come precomputed from the host.
        if (COUNT) sc.triangleTests += doTri;
        const v3 n = mk3(t0.w, t1.w, t2.w);
        const v3 ab = mk3(t1.x, t1.y, t1.z), ac = mk3(t2.x, t2.y, t2.z);
        const v3 ao = q.ro - mk3(t0.x, t0.y, t0.z);
        const v3 dao = cross(ao, q.rd);
        const float det = -dot(q.rd, n);
        const float invDet = 1.0f / det;
        const float dst = dot(ao, n) * invDet;
        const float u = dot(ac, dao) * invDet;
        const float v = -dot(ab, dao) * invDet;
        const float w = 1.0f - u - v;
        const bool ok = doTri & (det >= 1e-10f) & (dst >= 0.0f) & (u >= 0.0f) & (v >= 0.0f) & (w >= 0.0f);
        // (for a shadow query only `occluded` below counts; what it keeps as its best hit is not read)
        const bool better = ok & (dst < q.bestT);
        q.bestT = better ? dst : q.bestT;
        q.bestU = better ? u : q.bestU;
        q.bestV = better ? v : q.bestV;
        q.bestPrim = better ? (int)q.leafPos : q.bestPrim;
        q.leafPos += doTri ? 1u : 0u;
        finished = q.isShadow & ok;  // occluded
    }
    if (pop) stackPop<STRIDE, LDS_DEPTH>(q, stack, deep);  // no child entered: next sub-tree off the stack
    if (q.cur < 0 && q.leafPos == q.leafEnd) {  // park the leaf `cur` refers to, walk on
        const unsigned bits = ~(unsigned)q.cur;
        q.leafPos = bits >> 3;
        q.leafEnd = q.leafPos + (bits & 7u) + 1u;
        stackPop<STRIDE, LDS_DEPTH>(q, stack, deep);
    }
    return finished ? kQueryOccluded : ((q.cur == 0 && q.leafPos == q.leafEnd) ? kQueryDone : kQueryGoesOn);
}

// ------------------------------------------------------------------------------ path logic
struct PathState {
    unsigned rng, bounce, k, frame;
    v3 radiance, throughput, f;
    float pdfBsdf;
    v3 albedoRes, normalRes;
};

struct BounceRays {
    bool hasShadow, hasExt;
    v3 origin, shadowDir, extDir, pending;
};

// Primary ray: pathtracing.comp:150-200 (inverse(ViewMatrix) arrives precomputed in P.invView).
template <class Matrix>
PT_DEV void mulMat4(Matrix M, float x, float y, float z, float w, float out[4])
{
    for (int i = 0; i < 4; i++) out[i] = ((M[0 + i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i] * w;
}
// The kernel's argument segment as floats (DevParams is every kernel's first argument: invView at float 0, invProj at 16).
// The persistent kernel reads the two camera matrices from there, through a pointer made opaque where a sample starts, so
// that they are fetched by scalar loads at that point instead of living in (spilled) scalar registers across the loop.
typedef const __attribute__((address_space(4))) float* KernargFloats;
static_assert(offsetof(DevParams, invView) == 0 && offsetof(DevParams, invProj) == 64, "KernargFloats offsets below");
PT_DEV void pinholeRay(const DevParams& P, float u, float v, v3& ro, v3& rd, KernargFloats K = nullptr)
{
    float vp[4];
    if (K) mulMat4(K + 16, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 0.0f, 1.0f, vp);
    else mulMat4(P.invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 0.0f, 1.0f, vp);
    v3 origin = mk3(vp[0], vp[1], vp[2]) / vp[3];
    v3 direction = normalize3(origin);
    float o4[4], d4[4];
    if (K) {
        mulMat4(K, origin.x, origin.y, origin.z, 1.0f, o4);
        mulMat4(K, direction.x, direction.y, direction.z, 0.0f, d4);
    } else {
        mulMat4(P.invView, origin.x, origin.y, origin.z, 1.0f, o4);
        mulMat4(P.invView, direction.x, direction.y, direction.z, 0.0f, d4);
    }
    ro = mk3(o4[0], o4[1], o4[2]);
    rd = mk3(d4[0], d4[1], d4[2]);
}
PT_DEV void primaryRay(const DevParams& P, float u, float v, unsigned& rng, v3& ro, v3& rd, KernargFloats K = nullptr)
{
    pinholeRay(P, u, v, ro, rd, K);
    if (P.apertureSize != 0.0f) {  // thin lens (:166-187)
        const float* VM = P.view;
        v3 cameraForward = -mk3(VM[8], VM[9], VM[10]);
        v3 cameraUp = mk3(VM[4], VM[5], VM[6]);
        v3 cameraRight = cross(cameraUp, cameraForward);
        v3 focalPoint = ro + rd * P.focalLength;
        float r1 = randomValue(rng);
        float r2 = randomValue(rng);
        v3 h = hemispherePointCos(r1, r2);
        float ax = h.x * P.apertureShapeX * P.apertureSize;
        float ay = h.y * P.apertureShapeY * P.apertureSize;
        ro = ro + cameraRight * ax + cameraUp * ay;
        rd = normalize3(focalPoint - ro);
    }
}

// main(): pathtracing.comp:206-245 for sample index ps.k of pixel (px, py).
PT_DEV void startSample(const DevParams& P, unsigned px, unsigned py, PathState& ps, v3& ro, v3& rd, KernargFloats K = nullptr)
{
    const unsigned pixelIndex = py * (unsigned)P.frameW + px;  // uint(FrameDimensions.x), :213
    ps.frame = P.frameFirst + ps.k * P.frameStride;
    ps.rng = pixelIndex + ps.frame * 719393u;
    float u = ((float)px + 0.5f) / P.frameW, v = ((float)py + 0.5f) / P.frameH;
    if (P.antiAliasing != 0u) {
        float r1 = randomValue(ps.rng);
        float r2 = randomValue(ps.rng);
        u = u + (r1 - 0.5f) / P.frameW;
        v = v + (r2 - 0.5f) / P.frameH;
    }
    primaryRay(P, u, v, ps.rng, ro, rd, K);
    ps.radiance = splat(0.0f);
    ps.throughput = splat(1.0f);
    ps.f = splat(1.0f);
    ps.pdfBsdf = 1.0f;
    ps.albedoRes = splat(0.0f);
    ps.normalRes = splat(0.0f);
    ps.bounce = 0;
}

// One ray's shading, for a lane whose closest-hit query ended in a miss (`isHit` false: the ray (.., rd) left the scene
// at bounce ps.bounce, environment radiance with MIS against the BSDF sample that produced the ray,
// pathtracing.comp:36-52, and the path ends) or in a hit (`isHit`: one iteration of PathTrace's bounce loop,
// pathtracing.comp:25-140, around the closest hit (prim, t, u, v) of the ray (ro, rd); updates the path state and
// returns the rays the bounce wants traced, none at all meaning that the path has ended).  Both kinds of lane go
// through ONE environment lookup -- a miss looks along its ray, a hit towards the light direction it has just
// sampled -- so that a wave with lanes of either kind pays for that code once, at the fill of both together.
#ifdef PT_EXPERIMENT_SHADE_TIMERS  // where a shading block's ticks go: wave clock between marks, summed per wave in LDS, flushed at the kernel's end
__device__ unsigned long long g_shadeTicks[8];
__shared__ unsigned long long s_shadeTicks[8];
#define PT_MARK(i)                                                              \
    {                                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter();           \
        if (threadIdx.x == __builtin_ctzll(__ballot(1))) s_shadeTicks[i] += now_ - mark_; \
        mark_ = __builtin_readcyclecounter();                                   \
    }
#else
#define PT_MARK(i)
#endif
PT_DEV BounceRays shadeRay(const DevParams& P, const DevScene& S, PathState& ps, v3 ro, v3 rd, int prim, float tHit, float uHit,
                           float vHit, bool isHit)
{
#ifdef PT_EXPERIMENT_SHADE_TIMERS
    unsigned long long mark_ = __builtin_readcyclecounter();
#endif
    BounceRays out;
    out.hasShadow = false;
    out.hasExt = false;
    out.origin = out.shadowDir = out.extDir = out.pending = splat(0.0f);
    // A hit's reads form two rounds instead of four: the shading record and the table entries of the light direction
    // (SampleHdri's two randoms are the bounce's first: pathtracing.comp:65) are requested together, then the material
    // record and the light's radiance and pdf (EvaluateHdri, taken for every hit: whether it counts is decided further
    // down by N.L, which needs the material's normal map first).
    float4 s0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), s1 = s0, s2 = s0, s3 = s0;
    v3 N = splat(0.0f), Ng = splat(0.0f), P0 = splat(0.0f);
    float uvx = 0.0f, uvy = 0.0f;
    unsigned meshIndex = 0u;
    int normalMap = -1;
    MaterialWords mw{s0, s0, s0, s0, s0, s0};
    v3 lookDir = rd;
    if (isHit) {
        // hit attributes of the winning triangle (intersection.glsl:41-57)
        const float w = 1.0f - uHit - vHit;
        P0 = ro + rd * tHit;
        const float4* shp = reinterpret_cast<const float4*>(S.shading + prim);
        s0 = shp[0], s1 = shp[1], s2 = shp[2], s3 = shp[3];
        const float4 s4 = shp[4];
        // (the light direction's table entries are requested first and used last: the hit's attributes and the
        // material's reads go in between)
        const SampleTaps taps = requestSampleHdri(S, P.envW, P.envH, ps.rng);
        uvx = s0.w * w + s2.w * uHit + s3.y * vHit;
        uvy = s1.w * w + s3.x * uHit + s3.z * vHit;
        N = normalize3(mk3(s0.x, s0.y, s0.z) * w + mk3(s1.x, s1.y, s1.z) * uHit + mk3(s2.x, s2.y, s2.z) * vHit);
        Ng = mk3(s4.x, s4.y, s4.z);
        meshIndex = __float_as_uint(s3.w);
        mw = loadMaterial(S, meshIndex);
        normalMap = S.materials[meshIndex].tex[kTexNormal];
        lookDir = finishSampleHdri(taps);
    }
    PT_MARK(0)
    float pdfLight = 0.0f;
    const v3 colorLight = evaluateHdri(S, P.envW, P.envH, P.hdriW, P.hdriH, lookDir, pdfLight);
    PT_MARK(1)
    if (!isHit) {
        float misWeight = 1.0f;
        if (ps.bounce > 0u) {
            float t = ps.pdfBsdf * ps.pdfBsdf;
            misWeight = t / (pdfLight * pdfLight + t);
        }
        if (misWeight > 0.0f) ps.radiance = ps.radiance + misWeight * ps.throughput * colorLight * ps.f / ps.pdfBsdf;
        return out;
    }
    const v3 Ll = lookDir;
    // (The two words of the material record that steer branches are made opaque here, after the environment lookup:
    // left alone, the compiler tests them where they are loaded, and the wave sits out the material's round trip
    // before it starts on the lookup's arithmetic.)
    asm volatile("" : "+v"(normalMap), "+v"(mw.m4.z));
    v3 tangent = splat(0.0f);
    if (normalMap >= 0) {
        const float4* tp = reinterpret_cast<const float4*>(S.triangles + prim);
        const float4 q1 = tp[1], q2 = tp[2];
        const float dBx = s2.w - s0.w, dBy = s3.x - s1.w, dCx = s3.y - s0.w, dCy = s3.z - s1.w;
        const float invTd = 1.0f / (dBx * dCy - dBy * dCx);
        tangent = (mk3(q1.x, q1.y, q1.z) * dCy - mk3(q2.x, q2.y, q2.z) * dBy) * invTd;
    }
    const MaterialEval me = evaluateMaterial(S, P, mw, meshIndex, uvx, uvy, N, tangent);
    PT_MARK(2)
    const v3 V = -rd;
    const Surface surf = makeSurface(me, N, V);
    PT_MARK(3)
    ps.radiance = ps.radiance + ps.throughput * surf.emission;
    out.origin = offsetRay(P0, Ng);  // shared by the shadow and the extension ray (:66,:131)

    // next-event estimation towards the HDRI (pathtracing.comp:64-102), everything but the
    // occlusion test itself
    if (dot(N, Ll) > 0.0f) {
        const Dir dl = makeDir(surf, Ll);
        float pdfNee = 0.0f;
        const v3 fNee = evaluateBsdf(surf, dl, pdfNee);
        if (pdfNee > 0.0f) {
            float t = pdfLight * pdfLight;
            float misWeight = t / (pdfNee * pdfNee + t);
            if (misWeight > 0.0f) out.pending = misWeight * ps.throughput * colorLight * fNee / pdfLight;
        }
        // a shadow query whose pending contribution is exactly zero cannot change the result and
        // is not issued (a NaN contribution still is)
        out.hasShadow = !(out.pending.x == 0.0f && out.pending.y == 0.0f && out.pending.z == 0.0f);
        out.shadowDir = Ll;
    }

    PT_MARK(4)
    // BSDF sampling, Russian roulette, next ray (pathtracing.comp:104-139)
    const v3 L = sampleBsdf(surf, ps.rng);
    PT_MARK(5)
    const Dir ds = makeDir(surf, L);
    ps.pdfBsdf = 0.0f;
    ps.f = evaluateBsdf(surf, ds, ps.pdfBsdf);
    if (!(ps.pdfBsdf <= 0.0f)) {
        const float p = fmax_(ps.throughput.x, fmax_(ps.throughput.y, ps.throughput.z));
        if (!(randomValue(ps.rng) >= p)) {
            ps.throughput = ps.throughput * (1.0f / p);
            ps.throughput = ps.throughput * (ps.f / ps.pdfBsdf);
            if (ps.bounce == 0u) {
                ps.albedoRes = surf.albedo;
                ps.normalRes = N;
            }
            ps.bounce++;
            out.hasExt = ps.bounce < P.maxBounces;
            out.extDir = L;
        }
    }
    PT_MARK(6)
    return out;
}

PT_DEV void shadeMiss(const DevParams& P, const DevScene& S, PathState& ps, v3 rd)
{
    (void)shadeRay(P, S, ps, rd, rd, -1, 0.0f, 0.0f, 0.0f, false);
}
PT_DEV BounceRays shadeHit(const DevParams& P, const DevScene& S, PathState& ps, v3 ro, v3 rd, int prim, float tHit,
                           float uHit, float vHit)
{
    return shadeRay(P, S, ps, ro, rd, prim, tHit, uHit, vHit, true);
}

// Miss or hit in one call (prim < 0 is a miss), for callers that shade one ray per lane.
PT_DEV BounceRays shadeBounce(const DevParams& P, const DevScene& S, PathState& ps, v3 ro, v3 rd, int prim, float tHit,
                              float uHit, float vHit)
{
    if (prim < 0) {
        shadeMiss(P, S, ps, rd);
        BounceRays out;
        out.hasShadow = false;
        out.hasExt = false;
        out.origin = out.shadowDir = out.extDir = out.pending = splat(0.0f);
        return out;
    }
    return shadeHit(P, S, ps, ro, rd, prim, tHit, uHit, vHit);
}

PT_DEV float4 accumulate(float4 prev, v3 x, float weight, unsigned mode)
{
    if (mode == 0u)  // mix(prev, vec4(x, 1), weight): pathtracing.comp:248-260
        return make_float4(mix_(prev.x, x.x, weight), mix_(prev.y, x.y, weight), mix_(prev.z, x.z, weight),
                           mix_(prev.w, 1.0f, weight));
    return make_float4(prev.x + x.x, prev.y + x.y, prev.z + x.z, prev.w + 1.0f);
}

// One finished sample in the sample buffer.
struct SampleRecord {
    v3 radiance, albedo, normal;
};
static_assert(sizeof(SampleRecord) == kSampleBytes, "the host sizes the sample buffer with kSampleBytes");

struct Images {
    float4* radiance;
    float4* albedo;
    float4* normal;
};

PT_DEV void accumulateSample(const DevParams& P, const Images& img, size_t texel, const PathState& ps)
{
    const float weight = 1.0f / (float)ps.frame;
    img.radiance[texel] = accumulate(img.radiance[texel], ps.radiance, weight, P.accumulateMode);
    img.albedo[texel] = accumulate(img.albedo[texel], ps.albedoRes, weight, P.accumulateMode);
    img.normal[texel] = accumulate(img.normal[texel], ps.normalRes, weight, P.accumulateMode);
}

PT_DEV unsigned waveSum(unsigned v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
PT_DEV unsigned waveMax(unsigned v)
{
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ================================================================================ megakernel
enum LaneMode : int { kModeNew = 0, kModeShade = 1, kModeTrav = 2, kModeFinish = 3, kModeDone = 4 };

// Number of set bits of `mask` below the calling lane.
PT_DEV unsigned laneRank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ------------------------------------------------------------------ launch plan of the megakernel
// From the ticks each tile's waves lived in the previous launch: how many blocks a tile's samples
// are spread over, and the per-block cost the blocks are then sorted by.  With `slots` waves
// resident on the chip the launch cannot end before total / slots ticks; a tile is split so that none
// of its blocks costs more than 1/splitN of that bound, which adds at most splitN * slots blocks.
// One block of 1024 threads.
__global__ void planPartsKernel(const unsigned* __restrict__ cost, unsigned tiles, unsigned slots, unsigned splitN,
                                unsigned maxParts, unsigned* __restrict__ parts, unsigned* __restrict__ key)
{
    __shared__ unsigned long long s_sum[1024];
    unsigned long long sum = 0;
    for (unsigned t = threadIdx.x; t < tiles; t += blockDim.x) sum += cost[t];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned off = blockDim.x / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    const unsigned long long total = s_sum[0];
    // largest cost a block may have: total / (slots * splitN), rounded up so that the extra blocks
    // (sum of ceil(c / cap) - 1 < total / cap) can never outnumber slots * splitN, the room the
    // block list has for them
    const unsigned long long share = (unsigned long long)slots * splitN;
    unsigned long long cap = (total + share - 1) / share;
    if (cap == 0) cap = 1;
    for (unsigned t = threadIdx.x; t < tiles; t += blockDim.x) {
        const unsigned long long c = cost[t];
        unsigned long long n = (c + cap - 1) / cap;
        n = n < 1 ? 1 : (n > maxParts ? maxParts : n);
        parts[t] = (unsigned)n;
        key[t] = (unsigned)(c / n);
    }
}

// parts of the tiles in sorted order (input of the prefix sum that places their blocks)
__global__ void gatherPartsKernel(const unsigned* __restrict__ order, const unsigned* __restrict__ parts, unsigned tiles,
                                  unsigned* __restrict__ partsInOrder)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tiles) partsInOrder[i] = parts[order[i]];
}

// block list: the blocks of the i-th tile in sorted order start at start[i]; a tile's `n` blocks take
// equal shares of the launch's frameCount (< 65536) samples
__global__ void fillBlocksKernel(const unsigned* __restrict__ order, const unsigned* __restrict__ partsInOrder,
                                 const unsigned* __restrict__ start, unsigned tiles, unsigned capacity, unsigned frameCount,
                                 unsigned* __restrict__ blockList, unsigned* __restrict__ blockCount)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tiles) return;
    const unsigned n = partsInOrder[i], first = start[i], tile = order[i];
    for (unsigned p = 0; p < n && first + p < capacity; p++) {
        const unsigned kBegin = p * frameCount / n, kEnd = (p + 1u) * frameCount / n;
        blockList[2u * (first + p)] = tile;
        blockList[2u * (first + p) + 1u] = (kBegin << 16) | (kEnd - kBegin);
    }
    if (i == tiles - 1) *blockCount = first + n < capacity ? first + n : capacity;
}

// STACK entries of traversal stack per lane in LDS, DEEP more in private memory (0: none); compiled
// for WAVES resident waves per SIMD (4: 128 VGPRs, 3: 168 VGPRs and a quarter of the spills).
// PULL: the grid is as many waves as the chip keeps resident and every wave takes work items off a
// global counter until none is left; else one block per item.
template <int STACK, int DEEP, bool COUNT, int WAVES = PT_WAVES_PER_SIMD, bool PULL = true>
__global__ __launch_bounds__(kMegaThreads, WAVES) void renderKernel(DevParams P, DevScene S, Images img, DevCounters* counters)
{
    static_assert(kMegaThreads == 64, "one wave per block: the sample pool below is wave-wide");
    __shared__ int s_stack[(STACK + 1) * kMegaThreads];  // (a spare row for the push that is not one)
    int deepStack[DEEP + 1];
    __shared__ unsigned s_pixel[64];  // the tile's pixels that lie inside the image, compacted
    __shared__ unsigned s_depth[COUNT ? 32 : 1];  // counting build: closest-hit queries by path depth
    const unsigned lane = threadIdx.x;
    if (COUNT && lane < 32u) s_depth[lane] = 0u;
    int* stack = s_stack + lane;
    stack[0] = 0;  // row 0 of the stack column: "no entry"
#ifdef PT_EXPERIMENT_SHADE_TIMERS
    if (lane < 8u) s_shadeTicks[lane] = 0ull;
    __syncthreads();
#endif

    // Work items.  An item is a tile's (pixel, sample) pairs for a range of the launch's samples: the
    // whole range, tile after tile, without cost information; with it, the entries of the list the
    // planning kernels made from the previous launch's per-tile cost: costliest first, an expensive
    // tile's samples spread over several entries.  Either every block takes the item of its index, or
    // (PULL) a wave takes the next item off a global counter whenever its lanes run out of samples:
    // lanes then go straight on from one item to the next, where a wave per item ends every item in a
    // tail of idle lanes -- and a launch of few samples per pixel is nothing but tails.  Consecutive
    // waves start on different XCDs, so the eight of them are equally loaded either way.
    //
    // Within an item the pairs form a pool handed out in sample-major order: a lane that finishes a path
    // takes the next pair, whichever pixel it belongs to, so an expensive pixel's samples are traced by
    // many lanes at once.  Finished samples are not accumulated here (the running mean of a pixel must
    // take its samples in FrameCount order): they go to the sample buffer, one record per (sample,
    // pixel), and resolveKernel folds them into the images in order afterwards.
    const unsigned itemCount = P.blockList ? *P.blockCount : P.tilesX * P.tilesY;
    unsigned tile = 0, tileOrigin = 0, kBegin = 0, nInside = 0;  // wave-uniform: the item the lanes are fed from
    unsigned poolNext = 0, poolSize = 0;
    bool haveItem = false;
    unsigned itemStart = (unsigned)wall_clock64();  // (differences of the low word: an item lives far less than its 43 s period)
    // Makes item `idx` the one the lanes are fed from; all lanes call it together.
    auto takeItem = [&](unsigned idx) {
        unsigned kCount = P.frameCount;
        tile = idx;
        kBegin = 0;
        if (P.blockList) {
            tile = __builtin_amdgcn_readfirstlane(P.blockList[2u * idx]);
            const unsigned w1 = __builtin_amdgcn_readfirstlane(P.blockList[2u * idx + 1u]);
            kBegin = w1 >> 16;
            kCount = w1 & 0xffffu;
        }
        const unsigned tileX = tile % P.tilesX, tileY = tile / P.tilesX;
        tileOrigin = (tileX * kMegaTile) | ((tileY * kMegaTile) << 16);
        const bool inside = tileX * kMegaTile + (lane & 7u) < P.width && tileY * kMegaTile + (lane >> 3) < P.height;
        const unsigned long long insideMask = __ballot(inside);
        nInside = (unsigned)__popcll(insideMask);
        __syncthreads();  // the previous item's table is no longer read
        if (inside) s_pixel[laneRank(insideMask)] = lane;
        __syncthreads();
        poolNext = 0;
        poolSize = nInside * kCount;
        haveItem = true;
    };
    // Books the ticks since the item was taken on its tile (the plan of the next launch is made from these sums).
    auto bookItem = [&]() {
        const unsigned now = (unsigned)wall_clock64();
        if (haveItem && lane == 0) {
            const unsigned lived = now - itemStart;
            atomicAdd(&P.tileCost[tile], lived > 0x0fffffffu ? 0x0fffffffu : lived);
        }
        itemStart = now;
        haveItem = false;
    };
    if (!PULL && blockIdx.x < itemCount) takeItem(blockIdx.x);  // (a block past the end of the list has no item)

    unsigned nRays = 0, nShadow = 0, nHits = 0, nSamples = 0;
    StepCounters sc{0u, 0u, 0u};

    PathState ps{};
    ps.k = 0;
    unsigned pixel = 0;  // the lane's current sample: x | y << 16 of its pixel in the image
    int mode = kModeNew;
    Query q{};
    v3 extDir = splat(0.0f);   // direction of the extension ray waiting behind a shadow ray ...
    v3 extInv = splat(0.0f);   // ... and, without PULL, its reciprocal: the divisions are then done in the hit block, not at the restart
                               // inside the traversal loop (with PULL the three registers cost more than the divisions)
    v3 pending = splat(0.0f);  // NEE contribution, added if the shadow ray is unoccluded
    bool hasExt = false;

    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform, counting build only
    const unsigned long long waveStart = wall_clock64();
    for (;;) {
        // ================================================================ shading phase
        // (at a raised issue priority: a wave in a shading phase has some two thousand arithmetic instructions ahead
        // of it and the waves it shares the SIMD with, mid-traversal, mostly wait for memory; letting it go first is
        // worth 1.8 % of the launch on the headline scene, 0.8-2.1 % on the others; levels 1 to 3 measure alike)
        __builtin_amdgcn_s_setprio(1);
        unsigned long long tick = 0;
        if (COUNT) {
            tick = __builtin_readcyclecounter();
            const int nShade = __popcll(__ballot(mode == kModeShade));
            prof[9]++;
            if (nShade) prof[6]++;
            prof[7] += (unsigned)__popcll(__ballot(mode == kModeShade && q.bestPrim >= 0));
            prof[8] += (unsigned)__popcll(__ballot(mode == kModeShade && q.bestPrim < 0));
        }
        // A miss ends its path with the environment's radiance.  Hits are shaded once enough lanes hold one (the
        // block is the most expensive of the phase and runs at the occupancy of the lanes that take part), or when too
        // few other lanes are left to keep the traversal phase busy.  Misses and the hits that are due go through the
        // shading code together: its environment lookup serves both.
#ifdef PT_EXPERIMENT_MISSES_WAIT  // misses wait for the shading block like hits: both kinds count towards its threshold
        const int nShade = __popcll(__ballot(mode == kModeShade));
        const int aliveNow = __popcll(__ballot(mode != kModeDone));
        const bool doHit = nShade >= ((aliveNow * (int)P.hitMinNum + 63) >> 6) ||
                           aliveNow - nShade < ((aliveNow * (int)P.travExitNum + 63) >> 6);
        const bool isMiss = doHit && mode == kModeShade && q.bestPrim < 0;
        const bool isHit = doHit && mode == kModeShade && q.bestPrim >= 0;
#else
        const bool isMiss = mode == kModeShade && q.bestPrim < 0;
        const int nHit = __popcll(__ballot(mode == kModeShade && q.bestPrim >= 0));
        const int aliveNow = __popcll(__ballot(mode != kModeDone)) - __popcll(__ballot(isMiss));  // (a miss is on its way to FINISH)
        const bool doHit = nHit >= ((aliveNow * (int)P.hitMinNum + 63) >> 6) ||
                           aliveNow - nHit < ((aliveNow * (int)P.travExitNum + 63) >> 6);
        const bool isHit = doHit && mode == kModeShade && q.bestPrim >= 0;
#endif
#ifdef PT_EXPERIMENT_COUNT_PHASES
        if (COUNT) {
            const bool anyMiss = __ballot(isMiss) != 0ull, anyHit = __ballot(isHit) != 0ull;
            prof[15] += (anyMiss && anyHit) ? 1u : 0u;
            prof[10] += anyHit ? 1u : 0u;   // (these two slots hold other tallies in a regular build)
            prof[11] += anyMiss ? 1u : 0u;
        }
#endif
        if (isMiss || isHit) {
            const BounceRays br = shadeRay(P, S, ps, q.ro, q.rd, q.bestPrim, q.bestT, q.bestU, q.bestV, isHit);
            nHits += isHit;
            mode = kModeFinish;
            hasExt = false;
            if (br.hasShadow) {
                nShadow++;
                pending = br.pending;
                hasExt = br.hasExt;
                extDir = br.extDir;
                if (!PULL) extInv = queryReciprocal3(br.extDir);
                beginQuery(q, P.rootRef, br.origin, br.shadowDir, true);
                mode = kModeTrav;
            } else if (br.hasExt) {
                nRays++;
                if (COUNT) atomicAdd(&s_depth[ps.bounce < 31u ? ps.bounce : 31u], 1u);
                beginQuery(q, P.rootRef, br.origin, br.extDir, false);
                mode = kModeTrav;
            }
        }
        if (COUNT) {
            const unsigned long long now = __builtin_readcyclecounter();
            prof[14] += now - tick;
            prof[12] += now - tick;
            tick = now;
        }
        // a finished sample goes to its record of the sample buffer
        if (COUNT) prof[10] += (unsigned)__popcll(__ballot(mode == kModeFinish));
        if (mode == kModeFinish) {
            const unsigned px = pixel & 0xffffu, py = pixel >> 16;
            SampleRecord* rec = reinterpret_cast<SampleRecord*>(P.samples) + ((size_t)ps.k * P.pixels + ((size_t)py * P.width + px));
            // (written once, read once by resolveKernel after the launch: stored past the caches' retention, so that
            // 36 bytes a sample do not push scene data out of the L2s)
            float* rf = reinterpret_cast<float*>(rec);
            __builtin_nontemporal_store(ps.radiance.x, rf + 0), __builtin_nontemporal_store(ps.radiance.y, rf + 1), __builtin_nontemporal_store(ps.radiance.z, rf + 2);
            __builtin_nontemporal_store(ps.albedoRes.x, rf + 3), __builtin_nontemporal_store(ps.albedoRes.y, rf + 4), __builtin_nontemporal_store(ps.albedoRes.z, rf + 5);
            __builtin_nontemporal_store(ps.normalRes.x, rf + 6), __builtin_nontemporal_store(ps.normalRes.y, rf + 7), __builtin_nontemporal_store(ps.normalRes.z, rf + 8);
            mode = kModeNew;
        }
        // lanes without a sample take the next (pixel, sample) pairs of the pool, and of the next item
        // when this one runs out
        unsigned long long startedMask = 0ull;
        for (;;) {
            const unsigned long long wantMask = __ballot(mode == kModeNew);
            if (wantMask == 0ull) break;
            if (poolNext == poolSize) {
                unsigned idx = itemCount;
                if (PULL) {
                    bookItem();
                    if (lane == 0) idx = atomicAdd(P.workNext, 1u);  // (goes on counting when the items are gone: once per wave)
                    idx = __builtin_amdgcn_readfirstlane(idx);
                }
                if (idx >= itemCount) {
                    if (mode == kModeNew) mode = kModeDone;
                    break;
                }
                takeItem(idx);
                if (poolSize == 0u) continue;  // an empty item
            }
            const unsigned item = poolNext + laneRank(wantMask);
            if (mode == kModeNew && item < poolSize) {
                const unsigned slot = nInside == 64u ? (item & 63u) : item % nInside;
                ps.k = kBegin + (nInside == 64u ? (item >> 6) : item / nInside);
                const unsigned inTile = s_pixel[slot];
                pixel = tileOrigin + ((inTile & 7u) | ((inTile >> 3) << 16));
                v3 ro, rd;
                // (the camera matrices by scalar loads from the argument segment, here: held in scalar registers across the
                // loop they are 32 of the 52 values the kernel spilled to vector lanes, and a sample's start read them
                // back one v_readlane each; 0.6 % of the launch on every scene)
                KernargFloats K = (KernargFloats)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(K));
                startSample(P, pixel & 0xffffu, pixel >> 16, ps, ro, rd, K);
                hasExt = false;
                nSamples++;
                nRays++;
                if (COUNT) atomicAdd(&s_depth[0], 1u);
                beginQuery(q, P.rootRef, ro, rd, false);
                mode = kModeTrav;
            }
            const unsigned want = (unsigned)__popcll(wantMask);
            poolNext = poolSize - poolNext < want ? poolSize : poolNext + want;
            if (COUNT) startedMask |= wantMask;
        }
        if (COUNT) prof[11] += (unsigned)__popcll(startedMask & __ballot(mode == kModeTrav));
        if (__ballot(mode == kModeTrav || mode == kModeShade) == 0ull) break;
        // Leave the traversal phase once fewer than `threshold` lanes are traversing; lanes that
        // are done for good or wait for the hit block do not count against the wave.
        const int alive = __popcll(__ballot(mode != kModeDone && mode != kModeShade));
        const int threshold = (alive * (int)P.travExitNum + 63) >> 6;

        // ============================================================== traversal phase
        __builtin_amdgcn_s_setprio(0);
        if (COUNT) {
            const unsigned long long now = __builtin_readcyclecounter();
            prof[12] += now - tick;
            tick = now;
        }
        for (;;) {
            if (COUNT) {
                const unsigned long long tm = __ballot(mode == kModeTrav);
                const unsigned long long nodeM = __ballot(mode == kModeTrav && q.cur > 0);
                const unsigned long long leafM = __ballot(mode == kModeTrav && q.leafPos < q.leafEnd);
                const unsigned long long tri = leafM;
                prof[0]++;
                prof[1] += (unsigned)__popcll(tm);
                if (nodeM) prof[2]++;
                prof[3] += (unsigned)__popcll(nodeM);
                if (tri) prof[4]++;
                prof[5] += (unsigned)__popcll(tri);
            }
            if (mode == kModeTrav) {
                const int state = queryStep<COUNT, kMegaThreads, (DEEP > 0 ? STACK : 0)>(S, q, stack, sc, deepStack);
                if (state != kQueryGoesOn) {
                    if (q.isShadow) {
                        if (state != kQueryOccluded) ps.radiance = ps.radiance + pending;
                        if (hasExt) {
                            nRays++;
                            if (COUNT) atomicAdd(&s_depth[ps.bounce < 31u ? ps.bounce : 31u], 1u);
                            hasExt = false;
                            if (PULL) beginQuery(q, P.rootRef, q.ro, extDir, false);
                            else beginQuery(q, P.rootRef, q.ro, extDir, extInv, false);
                        } else {
                            mode = kModeFinish;
                        }
                    } else {
                        mode = kModeShade;
                    }
                }
            }
            // (no lane traversing ends the phase whatever the threshold: lanes waiting for the hit
            // block can leave `alive`, and with it the threshold, at zero)
            const int nTrav = __popcll(__ballot(mode == kModeTrav));
            if (nTrav == 0 || nTrav < threshold) break;
        }
        if (COUNT) prof[13] += __builtin_readcyclecounter() - tick;
    }

#ifdef PT_EXPERIMENT_SHADE_TIMERS
    __syncthreads();
    if (lane < 8u) atomicAdd(&g_shadeTicks[lane], s_shadeTicks[lane]);
#endif
    // counters: wave-level reduction, one atomic per wave and counter
    const unsigned sRays = waveSum(nRays), sShadow = waveSum(nShadow), sHits = waveSum(nHits), sSamples = waveSum(nSamples);
    const unsigned nSamplesWave = __builtin_amdgcn_readfirstlane(sSamples);
    bookItem();  // the last item's share of the wave's life
    if (lane == 0) {
        atomicAdd(&counters->raysClosest, (unsigned long long)sRays);
        atomicAdd(&counters->raysAny, (unsigned long long)sShadow);
        atomicAdd(&counters->closestHits, (unsigned long long)sHits);
        atomicAdd(&counters->samples, (unsigned long long)sSamples);
    }
    if (COUNT) {
        const unsigned sNodes = waveSum(sc.nodeVisits), sTris = waveSum(sc.triangleTests), m = waveMax(sc.maxStack);
        if (lane == 0) {
            atomicAdd(&counters->nodeVisits, (unsigned long long)sNodes);
            atomicAdd(&counters->triangleTests, (unsigned long long)sTris);
            atomicMax(&counters->maxStack, m);
            for (int i = 0; i < 16; i++) atomicAdd(&counters->profile[i], prof[i]);
            for (int i = 0; i < 32; i++)
                if (s_depth[i]) atomicAdd(&counters->depthRays[i], (unsigned long long)s_depth[i]);
            if (P.waveTimes && nSamplesWave) {
                P.waveTimes[2 * (size_t)blockIdx.x] = waveStart;
                P.waveTimes[2 * (size_t)blockIdx.x + 1] = wall_clock64();
            }
        }
    }
}

// Folds the sample buffer into the images: one thread per pixel applies its samples in FrameCount
// order with the shader's accumulation expression (pathtracing.comp:248-260), i.e. the same
// rounding sequence as frame-by-frame dispatches.  Streams frameCount x 36 B per pixel, coalesced.
__global__ void resolveKernel(DevParams P, Images img)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.pixels) return;
    float4 r = img.radiance[p], a = img.albedo[p], n = img.normal[p];
    for (unsigned k = 0; k < P.frameCount; k++) {
        const SampleRecord rec = reinterpret_cast<const SampleRecord*>(P.samples)[(size_t)k * P.pixels + p];
        const float weight = 1.0f / (float)(P.frameFirst + k * P.frameStride);
        r = accumulate(r, rec.radiance, weight, P.accumulateMode);
        a = accumulate(a, rec.albedo, weight, P.accumulateMode);
        n = accumulate(n, rec.normal, weight, P.accumulateMode);
    }
    img.radiance[p] = r;
    img.albedo[p] = a;
    img.normal[p] = n;
}

// ================================================================================= wavefront
// Path state in global memory, one slot per pixel (slot = y * width + x).
struct WfState {
    float4* rayO;    // ray origin xyz (shared by a bounce's shadow and extension ray)
    float4* rayD;    // extension ray direction xyz
    float4* hit;     // closest hit: t, u, v, primitive index (int bits; -1 = miss)
    float4* thrPdf;  // throughput xyz, pdfBsdf
    float4* radRng;  // radiance xyz, rng state (uint bits)
    float4* fMeta;   // f xyz, (bounce << 24 | sample index k) as uint bits
    float4* shDir;   // shadow ray direction xyz
    float4* pend;    // pending NEE contribution xyz
    float4* aovA;    // primary albedo xyz
    float4* aovN;    // primary normal xyz
};

// Ray queue entry: slot | flags.
constexpr unsigned kEntryShadow = 0x80000000u;  // occlusion query along shDir (else extension along rayD)
constexpr unsigned kEntryLast = 0x40000000u;    // shadow query that is the last ray of its path
constexpr unsigned kEntrySlotMask = 0x3fffffffu;

struct WfQueues {
    const unsigned* rayIn;  // entries to trace this iteration
    const unsigned* rayInCount;
    unsigned* rayHead;    // next entry to hand out
    unsigned* shadeQ;     // slots whose closest-hit query finished (traverseKernel appends)
    unsigned* shadeCount;
    unsigned* rayOut;     // next iteration's entries (shadeKernel appends)
    unsigned* rayOutCount;
};

PT_DEV void loadPath(const DevParams& P, const WfState& W, unsigned slot, PathState& ps)
{
    const float4 a = W.thrPdf[slot], b = W.radRng[slot], c = W.fMeta[slot], d = W.aovA[slot], e = W.aovN[slot];
    ps.throughput = mk3(a.x, a.y, a.z);
    ps.pdfBsdf = a.w;
    ps.radiance = mk3(b.x, b.y, b.z);
    ps.rng = __float_as_uint(b.w);
    ps.f = mk3(c.x, c.y, c.z);
    const unsigned meta = __float_as_uint(c.w);
    ps.bounce = meta >> 24;
    ps.k = meta & 0x00ffffffu;
    ps.frame = P.frameFirst + ps.k * P.frameStride;
    ps.albedoRes = mk3(d.x, d.y, d.z);
    ps.normalRes = mk3(e.x, e.y, e.z);
}
PT_DEV void storePath(const WfState& W, unsigned slot, const PathState& ps)
{
    W.thrPdf[slot] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, ps.pdfBsdf);
    W.radRng[slot] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, __uint_as_float(ps.rng));
    W.fMeta[slot] = make_float4(ps.f.x, ps.f.y, ps.f.z, __uint_as_float((ps.bounce << 24) | ps.k));
    W.aovA[slot] = make_float4(ps.albedoRes.x, ps.albedoRes.y, ps.albedoRes.z, 0.0f);
    W.aovN[slot] = make_float4(ps.normalRes.x, ps.normalRes.y, ps.normalRes.z, 0.0f);
}

// Appends `value` for every lane with `want` set: one atomic per wave, slots by prefix count.
// Must be reached by all lanes of the wave that are still in the enclosing loop.
PT_DEV void waveAppend(bool want, unsigned value, unsigned* queue, unsigned* count)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const unsigned lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (unsigned)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) queue[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// Accumulates the finished sample of `slot`, then starts the pixel's next sample if there is one.
// Returns true with the primary ray in (ro, rd) and the fresh path state stored; false when the
// pixel has no samples left.
PT_DEV bool finishAndRestart(const DevParams& P, const WfState& W, const Images& img, unsigned slot, PathState& ps, v3& ro,
                             v3& rd)
{
    accumulateSample(P, img, slot, ps);
    ps.k++;
    if (ps.k >= P.frameCount) return false;
    const unsigned px = slot % P.width, py = slot / P.width;
    startSample(P, px, py, ps, ro, rd);
    storePath(W, slot, ps);
    W.rayO[slot] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    W.rayD[slot] = make_float4(rd.x, rd.y, rd.z, 0.0f);
    return true;
}

// First sample of every pixel: path state, primary ray, one extension entry per pixel.
__global__ __launch_bounds__(kBlockThreads) void wfStartKernel(DevParams P, WfState W, unsigned* rayOut,
                                                              unsigned* rayOutCount, DevCounters* counters)
{
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned total = P.width * P.height;
    if (slot < total) {
        PathState ps{};
        ps.k = 0;
        v3 ro, rd;
        startSample(P, slot % P.width, slot / P.width, ps, ro, rd);
        storePath(W, slot, ps);
        W.rayO[slot] = make_float4(ro.x, ro.y, ro.z, 0.0f);
        W.rayD[slot] = make_float4(rd.x, rd.y, rd.z, 0.0f);
        rayOut[slot] = slot;
    }
    if (slot == 0) {
        *rayOutCount = total;
        atomicAdd(&counters->samples, (unsigned long long)total);
        atomicAdd(&counters->raysClosest, (unsigned long long)total);
    }
}

template <int STACK, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void traverseKernel(DevParams P, DevScene S, WfState W, Images img, WfQueues Q,
                                                               DevCounters* counters)
{
    __shared__ int s_stack[(STACK + 1) * kBlockThreads];  // (a spare row for the push that is not one)
    // finished extension queries wait here until a full wave's worth can be appended to the shade
    // queue with one atomic (a contended global counter per finished ray would stall every step)
    __shared__ unsigned s_done[kBlockThreads / 64][128];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int* stack = s_stack + tid;
    stack[0] = 0;  // row 0 of the stack column: "no entry"
    volatile unsigned* done = s_done[wave];
    const unsigned total = *Q.rayInCount;
    // Queue entries are reserved a chunk at a time per wave and handed to idle lanes from the
    // wave's private range; chunks shrink with the queue so late, short iterations still spread.
    const unsigned wavesInGrid = gridDim.x * (kBlockThreads / 64);
    unsigned chunk = total / (wavesInGrid * 4u);
    chunk = chunk < 64u ? 64u : (chunk > 512u ? 512u : chunk);

    unsigned nRays = 0, nSamples = 0;
    StepCounters sc{0u, 0u, 0u};
    Query q{};
    bool busy = false;
    unsigned entry = 0;
    // wave-uniform bookkeeping
    bool exhausted = false;
    unsigned poolPos = 0, poolEnd = 0, doneCount = 0;

    auto flushDone = [&](unsigned n) {  // appends the first n (<= 64) buffered slots
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(Q.shadeCount, n);
        base = __shfl(base, 0, 64);
        if (lane < n) Q.shadeQ[base + lane] = done[lane];
    };

    for (;;) {
        // ---- refill idle lanes
        const unsigned long long idle = __ballot(!busy);
        const unsigned nIdle = (unsigned)__popcll(idle);
        if ((nIdle >= P.refillMin || nIdle == 64u) && !(exhausted && poolPos == poolEnd)) {
            if (poolPos == poolEnd) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(Q.rayHead, chunk);
                base = __shfl(base, 0, 64);
                if (base >= total) {
                    exhausted = true;
                    poolPos = poolEnd = 0;
                } else {
                    poolPos = base;
                    poolEnd = base + chunk < total ? base + chunk : total;
                    if (poolEnd == total) exhausted = true;
                }
            }
            const unsigned avail = poolEnd - poolPos;
            const unsigned take = nIdle < avail ? nIdle : avail;
            const unsigned rank = (unsigned)__popcll(idle & ((1ull << lane) - 1ull));
            if (!busy && rank < take) {
                entry = Q.rayIn[poolPos + rank];
                const unsigned slot = entry & kEntrySlotMask;
                const float4 o = W.rayO[slot];
                const bool shadow = (entry & kEntryShadow) != 0u;
                const float4 d = shadow ? W.shDir[slot] : W.rayD[slot];
                beginQuery(q, P.rootRef, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), shadow);
                busy = true;
            }
            poolPos += take;
        }
        if (__ballot(busy) == 0ull) {
            if (exhausted && poolPos == poolEnd) break;
            continue;
        }

        bool toShade = false;
        unsigned shadeSlot = 0;
        const int state = busy ? queryStep<COUNT>(S, q, stack, sc) : (int)kQueryGoesOn;
        if (state != kQueryGoesOn) {
            const unsigned slot = entry & kEntrySlotMask;
            const bool unoccluded = state != kQueryOccluded;
            busy = false;
            if (q.isShadow) {
                const bool last = (entry & kEntryLast) != 0u;
                if (unoccluded || last) {
                    float4 r = W.radRng[slot];
                    if (unoccluded) {  // the pending contribution counts
                        const float4 pc = W.pend[slot];
                        r.x = r.x + pc.x;
                        r.y = r.y + pc.y;
                        r.z = r.z + pc.z;
                        if (!last) W.radRng[slot] = r;
                    }
                    if (last) {  // the path is over: accumulate, next sample, trace its primary ray here
                        PathState ps;
                        loadPath(P, W, slot, ps);
                        ps.radiance = mk3(r.x, r.y, r.z);
                        v3 ro, rd;
                        if (finishAndRestart(P, W, img, slot, ps, ro, rd)) {
                            nSamples++;
                            nRays++;
                            entry = slot;
                            beginQuery(q, P.rootRef, ro, rd, false);
                            busy = true;
                        }
                    }
                }
                // occluded and not last: nothing to do, the path goes on through its extension entry
            } else {
                W.hit[slot] = make_float4(q.bestT, q.bestU, q.bestV, __int_as_float(q.bestPrim));
                toShade = true;
                shadeSlot = slot;
            }
        }
        // ---- buffer finished extension queries, flush 64 at a time
        const unsigned long long fin = __ballot(toShade);
        if (fin != 0ull) {
            if (toShade) done[doneCount + (unsigned)__popcll(fin & ((1ull << lane) - 1ull))] = shadeSlot;
            doneCount += (unsigned)__popcll(fin);
            __builtin_amdgcn_wave_barrier();
            if (doneCount >= 64u) {
                flushDone(64u);
                __builtin_amdgcn_wave_barrier();
                const unsigned rem = doneCount - 64u;
                const unsigned moved = lane < rem ? done[64u + lane] : 0u;
                __builtin_amdgcn_wave_barrier();
                if (lane < rem) done[lane] = moved;
                doneCount = rem;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (doneCount > 0u) flushDone(doneCount);

    const unsigned sRays = waveSum(nRays), sSamples = waveSum(nSamples);
    if (lane == 0 && (sRays | sSamples)) {
        atomicAdd(&counters->raysClosest, (unsigned long long)sRays);
        atomicAdd(&counters->samples, (unsigned long long)sSamples);
    }
    if (COUNT) {
        const unsigned sNodes = waveSum(sc.nodeVisits), sTris = waveSum(sc.triangleTests), m = waveMax(sc.maxStack);
        if (lane == 0) {
            atomicAdd(&counters->nodeVisits, (unsigned long long)sNodes);
            atomicAdd(&counters->triangleTests, (unsigned long long)sTris);
            atomicMax(&counters->maxStack, m);
        }
    }
}

__global__ __launch_bounds__(kBlockThreads) void shadeKernel(DevParams P, DevScene S, WfState W, Images img, WfQueues Q,
                                                            DevCounters* counters)
{
    const unsigned total = *Q.shadeCount;
    const unsigned lane = threadIdx.x & 63u;
    unsigned nRays = 0, nShadow = 0, nHits = 0, nSamples = 0;
    // whole waves iterate together so the queue appends below see every lane
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x; base < total; base += stride) {
        const unsigned idx = base + threadIdx.x;
        unsigned e0 = 0, e1 = 0;
        bool want0 = false, want1 = false;
        if (idx < total) {
            const unsigned slot = Q.shadeQ[idx];
            PathState ps;
            loadPath(P, W, slot, ps);
            const float4 o = W.rayO[slot], d = W.rayD[slot], h = W.hit[slot];
            const int prim = __float_as_int(h.w);
            if (prim >= 0) nHits++;
            const BounceRays br = shadeBounce(P, S, ps, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), prim, h.x, h.y, h.z);
            if (br.hasShadow || br.hasExt) {
                storePath(W, slot, ps);
                W.rayO[slot] = make_float4(br.origin.x, br.origin.y, br.origin.z, 0.0f);
                if (br.hasShadow) {
                    W.shDir[slot] = make_float4(br.shadowDir.x, br.shadowDir.y, br.shadowDir.z, 0.0f);
                    W.pend[slot] = make_float4(br.pending.x, br.pending.y, br.pending.z, 0.0f);
                    e0 = slot | kEntryShadow | (br.hasExt ? 0u : kEntryLast);
                    want0 = true;
                    nShadow++;
                }
                if (br.hasExt) {
                    W.rayD[slot] = make_float4(br.extDir.x, br.extDir.y, br.extDir.z, 0.0f);
                    e1 = slot;
                    want1 = true;
                    nRays++;
                }
            } else {
                v3 ro, rd;
                if (finishAndRestart(P, W, img, slot, ps, ro, rd)) {
                    e1 = slot;
                    want1 = true;
                    nRays++;
                    nSamples++;
                }
            }
        }
        // shadow entries first: a path's shadow query then tends to run before its extension query
        waveAppend(want0, e0, Q.rayOut, Q.rayOutCount);
        waveAppend(want1, e1, Q.rayOut, Q.rayOutCount);
    }
    const unsigned sRays = waveSum(nRays), sShadow = waveSum(nShadow), sHits = waveSum(nHits), sSamples = waveSum(nSamples);
    if (lane == 0 && (sRays | sShadow | sHits | sSamples)) {
        atomicAdd(&counters->raysClosest, (unsigned long long)sRays);
        atomicAdd(&counters->raysAny, (unsigned long long)sShadow);
        atomicAdd(&counters->closestHits, (unsigned long long)sHits);
        atomicAdd(&counters->samples, (unsigned long long)sSamples);
    }
}

__global__ void scaleImagesKernel(float4* a, float4* b, float4* c, size_t n, float s)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 x = a[i], y = b[i], z = c[i];
        a[i] = make_float4(x.x * s, x.y * s, x.z * s, x.w * s);
        b[i] = make_float4(y.x * s, y.y * s, y.z * s, y.w * s);
        c[i] = make_float4(z.x * s, z.y * s, z.z * s, z.w * s);
    }
}

// RGBA32F image -> tightly packed RGB32F (the layout glGetTexImage(GL_RGB, GL_FLOAT) returns and
// a denoiser's Float3 input expects).  Four texels per thread: 3 x 16-B stores for 4 x 16-B loads.
__global__ void packRgbKernel(const float4* __restrict__ in, float* __restrict__ out, size_t texels)
{
    const size_t quad = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t first = quad * 4;
    if (first >= texels) return;
    if (first + 4 <= texels) {
        const float4 a = in[first], b = in[first + 1], c = in[first + 2], d = in[first + 3];
        float4* o = reinterpret_cast<float4*>(out + first * 3);
        o[0] = make_float4(a.x, a.y, a.z, b.x);
        o[1] = make_float4(b.y, b.z, c.x, c.y);
        o[2] = make_float4(c.z, d.x, d.y, d.z);
    } else {
        for (size_t i = first; i < texels; i++) {
            const float4 a = in[i];
            out[i * 3] = a.x;
            out[i * 3 + 1] = a.y;
            out[i * 3 + 2] = a.z;
        }
    }
}

// Display transform of the reference's post-processing pass: exposure, the fitted ACES curve
// (Shaders/tonemapping.frag:21-71; `color * mat3(a, b, c)` is a row vector times a matrix whose
// columns are a, b, c, so component i is dot(color, column i)), then the sRGB encoding the reference
// gets from GL_FRAMEBUFFER_SRGB on its default framebuffer (Runtime/Renderer/Renderer.cpp:25) and
// the conversion to 8-bit unorm.  Sums run left to right; pow_ is the fixed-polynomial power of
// device_math.h, so the bytes are reproducible on a CPU.
PT_DEV float displayChannel(float c)
{
    c = clamp_(c, 0.0f, 1.0f);
    const float e = c <= 0.0031308f ? c * 12.92f : 1.055f * pow_(c, 0.41666f) - 0.055f;  // GL 4.6 section 17.3.7
    return __builtin_floorf(clamp_(e, 0.0f, 1.0f) * 255.0f + 0.5f);
}
PT_DEV unsigned displayTexel(float4 hdr, float exposure)
{
    const float r = hdr.x * exposure, g = hdr.y * exposure, b = hdr.z * exposure;
    v3 c = mk3((r * 0.59719f + g * 0.35458f) + b * 0.04823f, (r * 0.07600f + g * 0.90834f) + b * 0.01566f,
               (r * 0.02840f + g * 0.13383f) + b * 0.83777f);
    const v3 num = c * (c + splat(0.0245786f)) - splat(0.000090537f);
    const v3 den = c * (0.983729f * c + splat(0.4329510f)) + splat(0.238081f);
    c = mk3(num.x / den.x, num.y / den.y, num.z / den.z);
    const float x = (c.x * 1.60475f + c.y * -0.53108f) + c.z * -0.07367f;
    const float y = (c.x * -0.10208f + c.y * 1.10813f) + c.z * -0.00605f;
    const float z = (c.x * -0.00327f + c.y * -0.07276f) + c.z * 1.07602f;
    return (unsigned)displayChannel(x) | ((unsigned)displayChannel(y) << 8) | ((unsigned)displayChannel(z) << 16);
}
// RGBA32F radiance image -> tightly packed RGB8.  Four texels per thread: three 4-byte stores.
__global__ void displayKernel(const float4* __restrict__ in, unsigned char* __restrict__ out, size_t texels, float exposure)
{
    const size_t first = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (first >= texels) return;
    if (first + 4 <= texels) {
        const unsigned a = displayTexel(in[first], exposure), b = displayTexel(in[first + 1], exposure);
        const unsigned c = displayTexel(in[first + 2], exposure), d = displayTexel(in[first + 3], exposure);
        unsigned* o = reinterpret_cast<unsigned*>(out + first * 3);  // first * 3 is a multiple of 12
        o[0] = a | (b << 24);
        o[1] = (b >> 8) | (c << 16);
        o[2] = (c >> 16) | (d << 8);
    } else {
        for (size_t i = first; i < texels; i++) {
            const unsigned a = displayTexel(in[i], exposure);
            out[i * 3] = (unsigned char)a;
            out[i * 3 + 1] = (unsigned char)(a >> 8);
            out[i * 3 + 2] = (unsigned char)(a >> 16);
        }
    }
}

// The two views the reference's shader writes to the radiance image instead of a frame when it is built
// with DEBUG_ENABLED (pathtracing.comp:216-223, Library/debug.glsl): one thread per pixel, nothing
// accumulated.  view 1 (debug.glsl:31-126): the leaves the pixel's pinhole ray reaches when every child
// box it enters is followed (no triangle tests, no pruning), as a heat map over 0..100, purple where the
// shader's 16-entry stack overflows; walked over the reference's own node records.  view 2 (:7-28): the
// environment map with a red dot wherever one of 500 draws from the sampling cache lands.
__global__ void debugViewKernel(DevParams P, DevScene S, int view, const RefNodeAlign* __restrict__ nodes, float4* __restrict__ radiance)
{
    const unsigned x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= P.width || y >= P.height) return;
    const float u = ((float)x + 0.5f) / P.frameW, v = ((float)y + 0.5f) / P.frameH;
    float4 out;
    if (view == 2) {
        Bilerp q = bilerpSetup(u, -v, P.envW, P.envH);
        const float4 c00 = S.envColorPdf[(size_t)q.y0 * P.envW + q.x0], c10 = S.envColorPdf[(size_t)q.y0 * P.envW + q.x1];
        const float4 c01 = S.envColorPdf[(size_t)q.y1 * P.envW + q.x0], c11 = S.envColorPdf[(size_t)q.y1 * P.envW + q.x1];
        v3 color = mk3(bilerp(q, c00.x, c10.x, c01.x, c11.x), bilerp(q, c00.y, c10.y, c01.y, c11.y), bilerp(q, c00.z, c10.z, c01.z, c11.z));
        for (unsigned i = 0u; i < 500u; i++) {
            unsigned state = i;
            const float xi1 = randomValue(state);
            const float xi2 = randomValue(state);
            q = bilerpSetup(xi1, xi2, P.envW, P.envH);
            const float2 t00 = S.envSample[(size_t)q.y0 * P.envW + q.x0], t10 = S.envSample[(size_t)q.y0 * P.envW + q.x1];
            const float2 t01 = S.envSample[(size_t)q.y1 * P.envW + q.x0], t11 = S.envSample[(size_t)q.y1 * P.envW + q.x1];
            const float tx = bilerp(q, t00.x, t10.x, t01.x, t11.x);
            const float ty = 1.0f - bilerp(q, t00.y, t10.y, t01.y, t11.y);
            const float dx = u - tx, dy = v - ty;
            if (sqrt_(dx * dx + dy * dy) < 0.005f) color = mk3(1.0f, 0.0f, 0.0f);
        }
        out = make_float4(color.x, color.y, color.z, 1.0f);
    } else {
        v3 ro, rd;
        pinholeRay(P, u, v, ro, rd);
        const v3 inv = reciprocal3(rd);
        unsigned count = 0u;
        int stack[18];  // the shader's 16, and the two entries its last pushes write past them
        int sp = 0;
        stack[sp++] = 1;
        while (sp > 0) {
            const RefNodeAlign node = nodes[stack[--sp]];
            if (node.n > 0) {
                count++;
            } else if (sp < 16) {
                float dl = kFltMax, dr = kFltMax, t0;
                if (node.left > 0) {
                    const RefNodeAlign c = nodes[node.left];
                    if (!hitAabb(ro, inv, c.AA[0], c.AA[1], c.AA[2], c.BB[0], c.BB[1], c.BB[2], dl, t0)) dl = -1.0f;
                }
                if (node.right > 0) {
                    const RefNodeAlign c = nodes[node.right];
                    if (!hitAabb(ro, inv, c.AA[0], c.AA[1], c.AA[2], c.BB[0], c.BB[1], c.BB[2], dr, t0)) dr = -1.0f;
                }
                if (dl > 0.0f && dr > 0.0f) {
                    const bool leftFirst = dl < dr;
                    stack[sp++] = leftFirst ? node.right : node.left;
                    stack[sp++] = leftFirst ? node.left : node.right;
                } else if (dl > 0.0f) {
                    stack[sp++] = node.left;
                } else if (dr > 0.0f) {
                    stack[sp++] = node.right;
                }
            } else {
                count = 0xffffffffu;
                break;
            }
        }
        if (count == 0xffffffffu) {
            out = make_float4(1.0f, 0.0f, 1.0f, 1.0f);
        } else {
            const float map[6][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 1.0f, 1.0f},
                                     {0.0f, 1.0f, 0.0f}, {1.0f, 1.0f, 0.0f}, {1.0f, 0.0f, 0.0f}};
            const float l = clamp_((float)count / 100.0f, 0.0f, 1.0f) * 5.0f;
            const float fl = floorf(l);
            const unsigned ia = (unsigned)fl, ib = (unsigned)ceilf(l);
            const float t = l - fl;
            out = make_float4(mix_(map[ia][0], map[ib][0], t), mix_(map[ia][1], map[ib][1], t), mix_(map[ia][2], map[ib][2], t), 1.0f);
        }
    }
    radiance[(size_t)y * P.width + x] = out;
}

// Element-wise evaluation of the float32 building blocks, so tests can compare them bit for bit
// with a CPU evaluation of the same operation sequences.
__global__ void evalMathKernel(int fn, const float* a, const float* b, float* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r;
    switch (fn) {
    case 0: r = sin_(x); break;
    case 1: r = cos_(x); break;
    case 2: r = asin_(x); break;
    case 3: r = atan2_(x, y); break;
    case 4: r = pow_(x, y); break;
    case 5: r = fmin_(x, y); break;
    case 6: r = fmax_(x, y); break;
    case 7: r = x / y; break;
    case 8: r = sqrt_(x); break;
    default: r = rsqrt_(x); break;
    }
    out[i] = r;
}

}  // namespace ptcore
