// Optimisation of the clustered tree's topology by parallel reinsertion (after Meister and Bittner 2018, "Parallel
// Reinsertion for Bounding Volume Hierarchy Optimization"): a node -- a triangle or a whole sub-tree -- is taken out of
// the tree together with its parent and put back next to the node where that lowers the summed surface area of the
// tree's boxes the most.  One pass: every candidate node searches the tree for its best new neighbour, candidates
// whose moves touch the same nodes are thinned out by priority (the larger gain wins), the winners' links are
// rewritten, the boxes are refitted bottom-up.  Every step is a function of the tree before the pass alone (ties are
// broken by node number), so the result does not depend on scheduling.
//
// Node ids are those of ploc.h: triangle m of the Morton order is m, internal node k is n + k, the root n + (n - 2).
// The root and its two children are never moved, so the root stays the root.
#pragma once
#include "ploc.h"

namespace ptcore {

struct ReinsertTree {
    int* parent;  // 2n-1 (root: -1)
    int* left;    // n-1, by internal node: child ids
    int* right;
    float* box;   // (2n-1) x 6
    int n;
};

constexpr int kReinsertStack = 64;  // sub-trees the search may have pending (it never needs more than the tree is deep)

__device__ __forceinline__ float reinsertArea(const float* b)
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void reinsertParentKernel(ReinsertTree R)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= R.n - 1) return;
    R.parent[R.left[k]] = R.n + k;
    R.parent[R.right[k]] = R.n + k;
    if (k == R.n - 2) R.parent[R.n + k] = -1;
}

// The lock word of a move: its gain (a positive float orders like its bits) above the mover's number.
__device__ __forceinline__ unsigned long long reinsertKey(float gain, int id, int total)
{
    return ((unsigned long long)__float_as_uint(gain) << 32) | (unsigned int)(total - id);
}

// Best new neighbour `out` of candidate `id` (every mod-th node, offset `phase`).  Taking `id` and its parent P out
// lets the sibling S move up into P's place and shrinks the ancestors; putting them back at `out` makes P the parent
// of (id, out) and grows out's ancestors below the lowest common one.  The gain of a position is
//     area(P) + sum of the shrinkage of id's ancestors below the common one
//             - area(id + out) - sum of the growth of out's ancestors below the common one,
// searched sub-tree by sub-tree up the path to the root (first S's, then the sibling's of P, of P's parent ...), each
// depth-first with the bound that positions below `x` gain at most what is left after x's own growth and area(id).
__global__ void reinsertSearchKernel(ReinsertTree R, int mod, int phase, int* __restrict__ outOf, float* __restrict__ gainOf,
                                     unsigned long long* lock)
{
    const int n = R.n, total = 2 * n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total) return;
    outOf[id] = -1;
    if (id % mod != phase) return;
    const int P = R.parent[id];
    if (P < 0) return;
    const int G = R.parent[P];
    if (G < 0) return;
    const int pl = R.left[P - n];
    const int S = pl == id ? R.right[P - n] : pl;
    float in[6], acc[6];
    for (int c = 0; c < 6; c++) {
        in[c] = R.box[(size_t)id * 6 + c];
        acc[c] = R.box[(size_t)S * 6 + c];
    }
    const float aIn = reinsertArea(in);
    float base = reinsertArea(R.box + (size_t)P * 6);
    int best = -1;
    float dBest = 1e-5f * base;  // (moves that gain less than that of the parent's area are not worth a conflict)
    int pendingNode[kReinsertStack];
    float pendingGrowth[kReinsertStack];
    int budget = 1 << 16;  // positions looked at per candidate, at most (the bound leaves a few hundred)
    auto search = [&](int top) {
        int sp = 1;
        pendingNode[0] = top;
        pendingGrowth[0] = 0.0f;
        while (sp > 0 && budget > 0) {
            --sp;
            --budget;
            const int x = pendingNode[sp];
            const float g = pendingGrowth[sp];
            const float* bx = R.box + (size_t)x * 6;
            const float m = plocArea(in, bx);
            const float gain = base - g - m;
            if (gain > dBest) {
                dBest = gain;
                best = x;
            }
            if (x >= n) {
                const float below = g + (m - reinsertArea(bx));
                if (base - below - aIn > dBest && sp + 2 <= kReinsertStack) {
                    pendingNode[sp] = R.right[x - n];
                    pendingGrowth[sp++] = below;
                    pendingNode[sp] = R.left[x - n];
                    pendingGrowth[sp++] = below;
                }
            }
        }
    };
    search(S);  // (S itself gains exactly nothing: area(id + S) is area(P))
    int onPath = P;
    for (int level = 0; level < 4 * kReinsertStack; level++) {
        const int par = R.parent[onPath];
        if (par < 0) break;
        const int l = R.left[par - n];
        const int sib = l == onPath ? R.right[par - n] : l;
        search(sib);
        for (int c = 0; c < 3; c++) {
            acc[c] = fminf(acc[c], R.box[(size_t)sib * 6 + c]);
            acc[3 + c] = fmaxf(acc[3 + c], R.box[(size_t)sib * 6 + 3 + c]);
        }
        base += reinsertArea(R.box + (size_t)par * 6) - reinsertArea(acc);  // `par` without `id` below it
        onPath = par;
    }
    if (best < 0) return;
    outOf[id] = best;
    gainOf[id] = dBest;
    const unsigned long long key = reinsertKey(dBest, id, total);
    atomicMax(&lock[id], key);
    atomicMax(&lock[P], key);
    atomicMax(&lock[S], key);
    atomicMax(&lock[G], key);
    atomicMax(&lock[best], key);
    atomicMax(&lock[R.parent[best]], key);
}

// A move goes ahead only if it holds every node whose links it rewrites.
__global__ void reinsertClaimKernel(ReinsertTree R, int* outOf, const float* __restrict__ gainOf,
                                    const unsigned long long* __restrict__ lock, unsigned long long* moving)
{
    const int n = R.n, total = 2 * n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total) return;
    const int out = outOf[id];
    if (out < 0) return;
    const int P = R.parent[id], G = R.parent[P];
    const int pl = R.left[P - n];
    const int S = pl == id ? R.right[P - n] : pl;
    const unsigned long long key = reinsertKey(gainOf[id], id, total);
    const bool mine = lock[id] == key && lock[P] == key && lock[S] == key && lock[G] == key && lock[out] == key &&
                      lock[R.parent[out]] == key;
    if (mine) moving[id] = key;
    else outOf[id] = -1;
}

// Two moves whose destinations lie inside each other's sub-trees would tie the two sub-trees into a ring cut off from
// the root (and likewise longer chains): a move gives way when a sub-tree above its destination moves with a larger
// key.  The smallest key of any such ring gives way, which opens it.
__global__ void reinsertRingKernel(ReinsertTree R, int* outOf, const unsigned long long* __restrict__ moving)
{
    const int total = 2 * R.n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total) return;
    const int out = outOf[id];
    if (out < 0) return;
    const unsigned long long key = moving[id];
    int u = R.parent[out];
    for (int level = 0; level < 4 * kReinsertStack && u >= 0; level++) {
        if (u != id && moving[u] > key) {
            outOf[id] = -1;
            return;
        }
        u = R.parent[u];
    }
}

__device__ __forceinline__ void reinsertReplaceChild(ReinsertTree R, int node, int was, int now)
{
    if (R.left[node - R.n] == was) R.left[node - R.n] = now;
    else R.right[node - R.n] = now;
}

__global__ void reinsertApplyKernel(ReinsertTree R, const int* __restrict__ outOf, int* __restrict__ oldPlace, unsigned int* moves)
{
    const int n = R.n, total = 2 * n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total) return;
    const int out = outOf[id];
    if (out < 0) return;
    const int P = R.parent[id], G = R.parent[P];
    const int pl = R.left[P - n];
    const int S = pl == id ? R.right[P - n] : pl;
    oldPlace[id] = G;  // (where the boxes shrink from: reinsertMarkKernel)
    atomicAdd(moves, 1u);
    reinsertReplaceChild(R, G, P, S);  // S moves up into P's place
    R.parent[S] = G;
    const int Po = R.parent[out];      // (read after the step above: S may be out's parent, or G)
    reinsertReplaceChild(R, Po, out, P);  // P takes out's place ...
    R.parent[P] = Po;
    reinsertReplaceChild(R, P, S, out);   // ... as the parent of (id, out)
    R.parent[out] = P;
}

// A node's box, triangle count, kept-node count and height from its children's.
__device__ __forceinline__ void reinsertFitNode(ReinsertTree R, PlocNodes N, int cur)
{
    const int n = R.n, k = cur - n;
    const int a = R.left[k], o = R.right[k];
    const float* ba = R.box + (size_t)a * 6;
    const float* bo = R.box + (size_t)o * 6;
    float* mine = R.box + (size_t)cur * 6;
    for (int c = 0; c < 3; c++) {
        mine[c] = fminf(__builtin_nontemporal_load(ba + c), __builtin_nontemporal_load(bo + c));
        mine[3 + c] = fmaxf(__builtin_nontemporal_load(ba + 3 + c), __builtin_nontemporal_load(bo + 3 + c));
    }
    const int size = __builtin_nontemporal_load(N.size + a) + __builtin_nontemporal_load(N.size + o);
    N.size[cur] = size;
    N.kept[cur] = (size > N.leafMax ? 1 : 0) + __builtin_nontemporal_load(N.kept + a) + __builtin_nontemporal_load(N.kept + o);
    const int ha = a < n ? 0 : __builtin_nontemporal_load(N.height + (a - n));
    const int ho = o < n ? 0 : __builtin_nontemporal_load(N.height + (o - n));
    N.height[k] = 1 + (ha > ho ? ha : ho);
}

// Boxes, triangle counts, kept-node counts and heights bottom-up: the second thread to arrive at a node completes it.
__global__ void reinsertRefitKernel(ReinsertTree R, PlocNodes N, unsigned int* visits)
{
    const int n = R.n;
    const int m = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (m >= n) return;
    int cur = R.parent[m];
    while (cur >= 0) {
        __threadfence();
        const unsigned int seen = atomicAdd(&visits[cur - n], 1u);
        if (seen != 1u) return;  // (the first child to arrive leaves the node to the second)
        __threadfence();
        reinsertFitNode(R, N, cur);
        cur = R.parent[cur];
    }
}

// After a pass only the nodes above a move's old and new place have other boxes (triangle counts, heights) than
// before.  Mark: every move flags the two paths to the root (a path already flagged is being flagged to the top by
// whoever got there first).  Refit: one launch per level of flagged nodes, bottom-up -- in launch `round` a flagged node
// is fitted once each flagged child was fitted in an EARLIER launch (its stamp, the number of that launch plus one,
// says so; a stamp written in this launch is larger than `round` and does not count yet).  A child's box is thus
// always read in a later launch than the one that wrote it: no fences, no atomics.  The root's stamp tells the host
// when the refit is complete.
__global__ void reinsertMarkKernel(ReinsertTree R, const int* __restrict__ outOf, const int* __restrict__ oldPlace, unsigned int* dirty)
{
    const int n = R.n, total = 2 * n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total || outOf[id] < 0) return;
    const int from[2] = {R.parent[id], oldPlace[id]};
    for (int side = 0; side < 2; side++) {
        int u = from[side];
        for (int level = 0; level < 4 * kReinsertStack && u >= 0; level++) {
            if (atomicExch(&dirty[u - n], 1u) != 0u) break;
            u = R.parent[u];
        }
    }
}

__global__ void reinsertRefitMarkedKernel(ReinsertTree R, PlocNodes N, const unsigned int* __restrict__ dirty, unsigned int* stamp,
                                          unsigned int round)
{
    const int n = R.n;
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n - 1 || !dirty[k] || stamp[k] != 0u) return;
    const int a = R.left[k], o = R.right[k];
    const bool flaggedA = a >= n && dirty[a - n] != 0u, flaggedO = o >= n && dirty[o - n] != 0u;
    const unsigned int sa = flaggedA ? stamp[a - n] : 0u, so = flaggedO ? stamp[o - n] : 0u;
    const bool readyA = !flaggedA || (sa != 0u && sa <= round);  // (a child that is not flagged is as it was)
    const bool readyO = !flaggedO || (so != 0u && so <= round);
    if (!readyA || !readyO) return;
    reinsertFitNode(R, N, n + k);
    stamp[k] = round + 1u;
}

// Where a node's triangles start in the final order and the node's depth-first rank among the kept nodes, from the
// path to the root: a right child comes behind its sibling's triangles and kept nodes (plocPositionKernel's rule,
// for a tree whose nodes are no longer numbered by level).
__global__ void reinsertPlaceKernel(ReinsertTree R, PlocNodes N)
{
    const int n = R.n, total = 2 * n - 1;
    const int id = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (id >= total) return;
    int position = 0, rank = 0, u = id;
    for (int level = 0; level < 4 * kReinsertStack; level++) {
        const int p = R.parent[u];
        if (p < 0) break;
        const int l = R.left[p - n];
        rank += 1;
        if (l != u) {
            position += N.size[l];
            rank += N.kept[l];
        }
        u = p;
    }
    N.position[id] = position;
    N.preorder[id] = rank;
}

}  // namespace ptcore
