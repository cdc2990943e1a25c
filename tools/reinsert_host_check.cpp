// The reinsertion passes of the device builder (pathtracer_amd/csrc/reinsert.h) run thread by thread on the host over
// random triangle boxes, with the tree checked after every pass: links consistent, every node reached from the root
// exactly once, boxes the unions of their children's.  Build and run (AddressSanitizer on):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Itools/hostshim -Ipathtracer_amd/csrc tools/reinsert_host_check.cpp -o /tmp/reinsert_check && /tmp/reinsert_check
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "reinsert.h"

using namespace ptcore;

template <class K, class... A>
static void launch(K kernel, int items, int block, A... args)
{
    blockDim.x = (unsigned)block;
    gridDim.x = (unsigned)((items + block - 1) / block);
    for (unsigned b = 0; b < gridDim.x; b++)
        for (unsigned t = 0; t < (unsigned)block; t++) {
            blockIdx.x = b;
            threadIdx.x = t;
            kernel(args...);
        }
}

static double area(const float* b) { return (double)reinsertArea(b); }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 5000;
    const int passes = argc > 2 ? atoi(argv[2]) : 24;
    const int mod = argc > 3 ? atoi(argv[3]) : 4;
    const unsigned seed = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
    const int total = 2 * n - 1;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<int> left(n - 1), right(n - 1), size(total, 1), height(n - 1, 0), position(total), kept(total, 0), preorder(total), parent(total);
    std::vector<float> box((size_t)total * 6);
    for (int m = 0; m < n; m++) {  // small boxes on a sphere, a few large ones, some coincident
        float c[3];
        const float th = U(rng) * 6.2831853f, z = U(rng) * 2.0f - 1.0f, r = std::sqrt(1.0f - z * z);
        c[0] = r * std::cos(th), c[1] = r * std::sin(th), c[2] = z;
        const float e = (m % 97 == 0) ? 1.5f : 0.02f * U(rng);
        for (int k = 0; k < 3; k++) box[(size_t)m * 6 + k] = c[k] - e, box[(size_t)m * 6 + 3 + k] = c[k] + e;
        if (m % 50 == 1)
            for (int k = 0; k < 6; k++) box[(size_t)m * 6 + k] = box[(size_t)(m - 1) * 6 + k];
    }
    // a poor tree to start from: neighbours in this (random) order paired level by level; the root is created last
    std::vector<int> level(n), next;
    for (int m = 0; m < n; m++) level[m] = m;
    int created = 0;
    while (level.size() > 1) {
        next.clear();
        for (size_t i = 0; i + 1 < level.size(); i += 2) {
            const int k = created++, a = level[i], o = level[i + 1];
            left[k] = a, right[k] = o;
            next.push_back(n + k);
        }
        if (level.size() & 1) next.push_back(level.back());
        level = next;
    }
    if (created != n - 1) return printf("setup lost nodes\n"), 1;
    PlocNodes N{left.data(), right.data(), size.data(), height.data(), box.data(), position.data(), kept.data(), preorder.data(), 2};
    ReinsertTree R{parent.data(), left.data(), right.data(), box.data(), n};
    std::vector<int> outOf(total);
    std::vector<float> gainOf(total);
    std::vector<unsigned long long> lock(total), moving(total);
    std::vector<unsigned int> visits(n - 1), dirty(n - 1), pending(n - 1);
    std::vector<int> oldPlace(total);
    auto refit = [&]() {
        std::fill(visits.begin(), visits.end(), 0u);
        launch(reinsertRefitKernel, n, 256, R, N, visits.data());
    };
    auto check = [&](const char* when) -> double {
        std::vector<int> seen(total, 0), stack{2 * n - 2};
        if (parent[2 * n - 2] != -1) printf("%s: the root has a parent\n", when), exit(1);
        double cost = 0.0;
        int reached = 0;
        while (!stack.empty()) {
            const int u = stack.back();
            stack.pop_back();
            if (seen[u]++) printf("%s: node %d reached twice\n", when, u), exit(1);
            reached++;
            if (u < n) continue;
            const int a = left[u - n], o = right[u - n];
            if (parent[a] != u || parent[o] != u) printf("%s: child of %d has another parent\n", when, u), exit(1);
            for (int k = 0; k < 3; k++) {
                if (box[(size_t)u * 6 + k] != std::min(box[(size_t)a * 6 + k], box[(size_t)o * 6 + k]) ||
                    box[(size_t)u * 6 + 3 + k] != std::max(box[(size_t)a * 6 + 3 + k], box[(size_t)o * 6 + 3 + k]))
                    printf("%s: box of %d is not its children's union\n", when, u), exit(1);
            }
            if (size[u] != size[a] + size[o]) printf("%s: size of %d\n", when, u), exit(1);
            cost += area(&box[(size_t)u * 6]);
            stack.push_back(a), stack.push_back(o);
        }
        if (reached != total) printf("%s: %d of %d nodes reached\n", when, reached, total), exit(1);
        return cost;
    };
    launch(reinsertParentKernel, n - 1, 256, R);
    refit();
    double cost = check("start");
    printf("n %d, start: cost %.4f, height %d\n", n, cost, height[n - 2]);
    for (int pass = 0; pass < passes; pass++) {
        std::fill(lock.begin(), lock.end(), 0ull);
        std::fill(moving.begin(), moving.end(), 0ull);
        std::fill(outOf.begin(), outOf.end(), -1);
        launch(reinsertSearchKernel, total, 64, R, mod, pass % mod, outOf.data(), gainOf.data(), lock.data());
        int wanted = 0, claimed = 0, applied = 0;
        for (int v : outOf) wanted += v >= 0;
        launch(reinsertClaimKernel, total, 256, R, outOf.data(), (const float*)gainOf.data(), (const unsigned long long*)lock.data(), moving.data());
        for (int v : outOf) claimed += v >= 0;
        launch(reinsertRingKernel, total, 256, R, outOf.data(), (const unsigned long long*)moving.data());
        for (int v : outOf) applied += v >= 0;
        unsigned int moved = 0;
        launch(reinsertApplyKernel, total, 256, R, (const int*)outOf.data(), oldPlace.data(), &moved);
        if ((int)moved != applied) printf("pass %d: %u moves counted, %d applied\n", pass, moved, applied), exit(1);
        std::fill(dirty.begin(), dirty.end(), 0u);
        std::fill(pending.begin(), pending.end(), 0u);
        launch(reinsertMarkKernel, total, 256, R, (const int*)outOf.data(), (const int*)oldPlace.data(), dirty.data());
        for (unsigned int round = 0; moved && pending[n - 2] == 0u; round++) {  // (`pending` holds the stamps)
            if (round > 4u * kReinsertStack) printf("pass %d: the marked refit does not end\n", pass), exit(1);
            launch(reinsertRefitMarkedKernel, n - 1, 256, R, N, (const unsigned int*)dirty.data(), pending.data(), round);
        }
        {  // the marked refit leaves what a refit of the whole tree leaves
            std::vector<float> b2 = box;
            std::vector<int> s2 = size, k2 = kept, h2 = height;
            refit();
            if (b2 != box || s2 != size || k2 != kept || h2 != height) printf("pass %d: marked refit differs from the full one\n", pass), exit(1);
        }
        const double now = check("after a pass");
        printf("pass %2d: %6d moves wanted, %6d claimed, %6d applied, cost %.4f (%+.2f %%), height %d\n", pass, wanted, claimed, applied, now,
               100.0 * (now - cost) / cost, height[n - 2]);
        cost = now;
    }
    launch(reinsertPlaceKernel, total, 256, R, N);
    // positions: the leaves' are a permutation; a node's range holds its children's, left first
    std::vector<int> hit(n, 0);
    for (int m = 0; m < n; m++) {
        if (position[m] < 0 || position[m] >= n || hit[position[m]]++) return printf("leaf positions are no permutation\n"), 1;
    }
    std::vector<int> rankSeen(total + 1, 0);
    for (int k = 0; k < n - 1; k++) {
        const int u = n + k, a = left[k], o = right[k];
        if (position[a] != position[u] || position[o] != position[u] + size[a]) return printf("position of %d's children\n", u), 1;
        if (size[u] > 2) {
            if (preorder[u] < 0 || preorder[u] >= total || rankSeen[preorder[u]]++) return printf("depth-first rank of %d repeats\n", u), 1;
            if (size[a] > 2 && preorder[a] != preorder[u] + 1) return printf("left child of %d is not next to it\n", u), 1;
        }
    }
    if (position[2 * n - 2] != 0 || preorder[2 * n - 2] != 0) return printf("root placement\n"), 1;
    printf("ok\n");
    return 0;
}
