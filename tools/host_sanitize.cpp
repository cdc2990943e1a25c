// Sanitizer driver for the CPU side of the path: the host code of libptcore (glTF/GLB reader, sweep-SAH BVH
// builder, reference -> device packers, HDRI importance cache, camera matrices, BVH disk cache) compiled with
// g++ -fsanitize=address,undefined and exercised natively (no HIP, no Python), on well-formed inputs and on
// damaged ones: every path argument is a .glb/.gltf file that is read as it is and then in truncated and
// byte-flipped copies (written to $TMPDIR), which the reader must accept or refuse without touching memory it
// does not own.
//
//     make -C pathtracer_amd/csrc host-asan        (builds and runs; tests/test_host_sanitizers.py does the same)
//
// Exit code 0 and the line "host_sanitize ok" mean every call returned; the sanitizers abort the process otherwise.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../pathtracer_amd/csrc/host.h"

using namespace ptcore;

static std::vector<unsigned char> readFile(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void writeFile(const std::string& path, const std::vector<unsigned char>& bytes)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)bytes.data(), (std::streamsize)bytes.size());
}

static size_t g_loaded = 0, g_refused = 0;

static void loadOne(const std::string& path, bool mustLoad)
{
    GltfScene scene;
    const std::string err = LoadGltf(path, scene);
    if (err.empty()) {
        g_loaded++;
        // walk everything the reader produced (out-of-range indices would show in the assembly below)
        size_t tris = 0;
        for (const GltfSubmesh& m : scene.submeshes) {
            std::vector<RefPrimitive> prims(m.indices.size() / 3);
            if (!prims.empty())
                AssemblePrimitives(m.vertices.data(), m.indices.data(), m.indices.size() / 3 * 3, 4, 0, prims.data());
            tris += prims.size();
        }
        for (const GltfImage& im : scene.images) (void)(im.path.size() + im.bytes.size());
        (void)tris;
    } else {
        g_refused++;
        if (mustLoad) {
            std::fprintf(stderr, "host_sanitize: %s did not load: %s\n", path.c_str(), err.c_str());
            std::exit(3);
        }
    }
}

// The file as it is, then damaged copies of it next to it (so that relative buffer URIs still resolve).
static void exerciseGltf(const std::string& path, const std::string& tmpDir, int variants)
{
    loadOne(path, true);
    const std::vector<unsigned char> good = readFile(path);
    if (good.empty()) return;
    const std::string ext = path.size() > 4 && path.substr(path.size() - 4) == ".glb" ? ".glb" : ".gltf";
    std::mt19937 rng(12345u + (unsigned)good.size());
    for (int v = 0; v < variants; v++) {
        std::vector<unsigned char> bad = good;
        const int kind = v % 4;
        if (kind == 0) {  // truncated
            bad.resize(rng() % good.size());
        } else if (kind == 1) {  // a few bytes flipped anywhere
            for (int k = 0; k < 4; k++) bad[rng() % bad.size()] ^= (unsigned char)(1u << (rng() % 8));
        } else if (kind == 2) {  // header / JSON region damaged (where lengths, counts and offsets live)
            const size_t span = std::min<size_t>(bad.size(), 2048);
            for (int k = 0; k < 3; k++) bad[rng() % span] = (unsigned char)(rng() & 0xff);
        } else {  // a digit of the JSON turned into another digit: plausible but wrong counts and offsets
            const size_t span = std::min<size_t>(bad.size(), 16384);
            for (int tries = 0, done = 0; tries < 4096 && done < 2; tries++) {
                const size_t at = rng() % span;
                if (bad[at] >= '0' && bad[at] <= '9') bad[at] = (unsigned char)('0' + rng() % 10), done++;
            }
        }
        // damaged copies go to the scratch directory: a .gltf whose buffers are separate files finds them only when the
        // original lies there too (the test writes its own files there)
        const std::string out = tmpDir + "/damaged_" + std::to_string(v) + ext;
        writeFile(out, bad);
        loadOne(out, false);
        std::remove(out.c_str());
    }
}

static void exerciseBvh(const std::string& tmpDir)
{
    std::mt19937 rng(7u);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (size_t count : {size_t(0), size_t(1), size_t(3), size_t(4), size_t(5), size_t(257), size_t(20000)}) {
        std::vector<RefPrimitive> prims(count);
        for (size_t i = 0; i < count; i++) {
            RefPrimitive& p = prims[i];
            // clusters with exact ties (several triangles sharing a centroid) and a few huge ones
            const float cx = std::floor(U(rng) * 8.0f), cy = std::floor(U(rng) * 8.0f), cz = (i % 7 == 0) ? 0.0f : U(rng);
            const float s = (i % 97 == 0) ? 1.0e6f : 0.1f;
            for (int k = 0; k < 3; k++) {
                p.posA[k] = (k == 0 ? cx : k == 1 ? cy : cz);
                p.posB[k] = p.posA[k] + (k == 0 ? s : 0.0f);
                p.posC[k] = p.posA[k] + (k == 1 ? s : 0.0f);
                p.norA[k] = p.norB[k] = p.norC[k] = (k == 2 ? 1.0f : 0.0f);
            }
            p.uvA[0] = p.uvA[1] = p.uvB[1] = p.uvC[0] = 0.0f;
            p.uvB[0] = p.uvC[1] = 1.0f;
            p.meshIndex = (uint32_t)(i % 3);
        }
        for (int uncapped = 0; uncapped < 2; uncapped++) {
            std::vector<RefPrimitive> work = prims;
            std::vector<RefNode> nodes;
            const size_t n = BuildBvhWithSah(work.data(), work.size(), 4, nodes, count > 1000 ? 4 : 1, uncapped != 0);
            if (n != nodes.size()) std::exit(4);
            std::vector<RefNodeAlign> nodes48(nodes.size());
            std::vector<RefPrimitiveAlign> prims112(work.size());
            AlignNodes(nodes.data(), nodes.size(), nodes48.data());
            AlignPrimitives(work.data(), work.size(), prims112.data());
            PackedScene packed;
            const std::string err = PackScene(nodes48.data(), nodes48.size(), prims112.data(), prims112.size(), 3, packed);
            if (count > 0 && !err.empty()) {
                std::fprintf(stderr, "host_sanitize: PackScene refused a built tree: %s\n", err.c_str());
                std::exit(5);
            }
            if (count == 0) continue;
            // the same tree with its leaves cut to two and to one triangle: every triangle still in exactly one leaf
            for (int cut = 1; cut <= 2; cut++) {
                PackedScene recut;
                const std::string e2 = PackScene(nodes48.data(), nodes48.size(), prims112.data(), prims112.size(), 3, recut, cut);
                if (!e2.empty()) std::exit(6);
                std::vector<int> covered(prims112.size(), 0);
                std::vector<int32_t> walk{recut.rootRef};
                while (!walk.empty()) {
                    const int32_t r = walk.back();
                    walk.pop_back();
                    if (r < 0) {
                        const uint32_t bits = ~(uint32_t)r, first = bits >> 3, n2 = (bits & 7u) + 1u;
                        if ((int)n2 > cut) std::exit(7);
                        for (uint32_t t = first; t < first + n2; t++) covered[t]++;
                    } else {
                        if (r == 0 || (size_t)r >= recut.nodes.size()) std::exit(8);
                        walk.push_back(recut.nodes[(size_t)r].left);
                        walk.push_back(recut.nodes[(size_t)r].right);
                    }
                }
                for (int cvr : covered)
                    if (cvr != 1) std::exit(9);
            }
            // a material table that is too short, and trees with damaged links, must be refused, not walked
            PackedScene rejected;
            (void)PackScene(nodes48.data(), nodes48.size(), prims112.data(), prims112.size(), 1, rejected);
            std::vector<RefNodeAlign> broken = nodes48;
            for (int k = 0; k < 8 && broken.size() > 1; k++) {
                RefNodeAlign& b = broken[1 + rng() % (broken.size() - 1)];
                switch (rng() % 4) {
                case 0: b.left = (int32_t)rng(); break;
                case 1: b.right = -5; break;
                case 2: b.n = 1 << 20; break;
                default: b.index = (int32_t)(rng() % (2 * count + 8)); break;
                }
                PackedScene r2;
                (void)PackScene(broken.data(), broken.size(), prims112.data(), prims112.size(), 3, r2);
            }
            // disk cache: round trip, wrong key, truncated file
            const std::string path = tmpDir + "/tree.ptbvh";
            const uint64_t key = Fnv1a64(prims.data(), prims.size() * sizeof(RefPrimitive));
            std::string e2 = SaveBvhCache(path, key, nodes48.data(), nodes48.size(), prims112.data(), prims112.size());
            if (!e2.empty()) std::exit(6);
            std::vector<RefNodeAlign> n2;
            std::vector<RefPrimitiveAlign> p2;
            e2 = LoadBvhCache(path, key, n2, p2);
            if (!e2.empty() || n2.size() != nodes48.size() || p2.size() != prims112.size() ||
                std::memcmp(n2.data(), nodes48.data(), n2.size() * sizeof(RefNodeAlign)) != 0)
                std::exit(7);
            if (LoadBvhCache(path, key + 1, n2, p2).empty()) std::exit(8);
            std::vector<unsigned char> file = readFile(path);
            for (size_t cut : {size_t(0), size_t(7), file.size() / 2, file.size() - 1}) {
                std::vector<unsigned char> part(file.begin(), file.begin() + (std::ptrdiff_t)std::min(cut, file.size()));
                writeFile(path, part);
                (void)LoadBvhCache(path, key, n2, p2);
            }
            for (int k = 0; k < 16; k++) {  // damaged header fields (counts far beyond the file)
                std::vector<unsigned char> bad = file;
                bad[rng() % std::min<size_t>(bad.size(), 64)] ^= 0xff;
                writeFile(path, bad);
                (void)LoadBvhCache(path, key, n2, p2);
            }
            std::remove(path.c_str());
        }
    }
}

static void exerciseMaterials()
{
    std::vector<RefMaterialAlign> mats(3);
    std::memset(mats.data(), 0, mats.size() * sizeof(RefMaterialAlign));
    mats[1].textureHandle[kTexAlbedo] = 2;   // 1-based slot 2
    mats[2].textureHandle[kTexNormal] = 99;  // beyond the table: must be refused
    std::vector<int> slotToTexture = {-1, 0, -1};
    std::vector<DevMaterial> out;
    (void)PackMaterials(mats.data(), 2, slotToTexture.size(), slotToTexture, out);
    (void)PackMaterials(mats.data(), 3, slotToTexture.size(), slotToTexture, out);
    (void)PackMaterials(mats.data(), 0, 0, {}, out);
}

static void exerciseEnvironmentAndCamera()
{
    for (int w : {1, 2, 17, 64}) {
        const int h = w == 1 ? 1 : w / 2 + 1;
        std::vector<float> rgb((size_t)w * h * 3), cache((size_t)w * h * 3);
        for (size_t i = 0; i < rgb.size(); i++) rgb[i] = (float)((i * 2654435761u) % 1000u) / 250.0f;
        CalculateHdriCache(rgb.data(), w, h, cache.data());
        std::fill(rgb.begin(), rgb.end(), 0.0f);  // the 1 x 1 black HDRI of the reference's content: 0 / 0
        CalculateHdriCache(rgb.data(), w, h, cache.data());
        rgb[0] = 1.0e30f;  // one texel holding all the energy
        CalculateHdriCache(rgb.data(), w, h, cache.data());
    }
    const float eye[3] = {-2.5f, 1.0f, 0.0f}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    float view[16], proj[16], inv[16];
    LookAt(eye, at, up, view);
    Perspective(1.57f, 16.0f / 9.0f, 0.01f, 1000.0f, proj);
    Inverse4(proj, inv);
    Inverse4(view, inv);
    const float zero[16] = {};
    Inverse4(zero, inv);  // singular: infinities and NaNs, no trap
    LookAt(eye, eye, up, view);
}

int main(int argc, char** argv)
{
    const char* tmp = std::getenv("TMPDIR");
    const std::string tmpDir = tmp && *tmp ? tmp : "/tmp";
    int variants = 200;
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        if (std::strncmp(argv[i], "--variants=", 11) == 0) variants = std::atoi(argv[i] + 11);
        else files.push_back(argv[i]);
    }
    exerciseEnvironmentAndCamera();
    exerciseMaterials();
    exerciseBvh(tmpDir);
    for (const std::string& f : files) exerciseGltf(f, tmpDir, variants);
    loadOne(tmpDir + "/no_such_file.glb", false);
    std::printf("host_sanitize ok: %zu files, %zu loads accepted, %zu refused\n", files.size(), g_loaded, g_refused);
    return 0;
}
