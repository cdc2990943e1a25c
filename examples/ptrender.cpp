// Headless renderer in plain C++ on top of the C ABI (no Python, no torch): what a C++ host such
// as the reference application does with the library.  Reads glTF 2.0 / GLB files, builds the BVH
// (the reference's sweep SAH on the host, or the GPU builder), renders FrameCount 1..spp and writes
// the radiance image as PFM (float) and optionally as a tone-mapped PPM.
//
//   ptrender scene.glb [more.glb ...] [--width 1024] [--height 1024] [--spp 49] [--bounces 3]
//            [--eye x y z] [--look-at x y z] [--fovy 1.57] [--device-bvh | --full-sweep-bvh]
//            [--hdri raw_f32_rgb_file width height] [--out render.pfm] [--ppm display.ppm] [--exposure 1.0]
//
// Materials come from ModelLoader::GenerateMaterial's mapping (pt_gltf_submesh_material); image
// textures are skipped with a warning, because decoding PNG/JPEG is the host's job (stb_image in
// the reference) and this example carries no decoder.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptcore.h"

namespace {

[[noreturn]] void die(const char* what)
{
    std::fprintf(stderr, "ptrender: %s: %s\n", what, pt_last_error());
    std::exit(1);
}
#define PT(call)                  \
    do {                          \
        if ((call) != PT_OK) die(#call); \
    } while (0)

// MaterialAlign, 176 B (PathTracingRenderer.h:163-193): 11 handles, pad, emission, albedo, 12 floats
struct MaterialAlign {
    uint64_t handle[11];
    uint64_t pad0;
    float emission[3];
    float pad1;
    float albedo[3];
    float scalars[12];
    float pad2;
};
static_assert(sizeof(MaterialAlign) == 176, "MaterialAlign layout");

// Analytic equirectangular sky: gradient, grey ground, Gaussian sun lobe (row 0 = straight up).
std::vector<float> MakeSky(int width, int height)
{
    std::vector<float> rgb((size_t)width * height * 3);
    const double pi = 3.14159265358979323846;
    double sun[3] = {-0.45, 0.75, 0.48};
    const double sl = std::sqrt(sun[0] * sun[0] + sun[1] * sun[1] + sun[2] * sun[2]);
    for (double& s : sun) s /= sl;
    for (int i = 0; i < height; i++) {
        const double v = (i + 0.5) / height, elev = pi * (0.5 - v);
        for (int j = 0; j < width; j++) {
            const double u = (j + 0.5) / width, phi = 2.0 * pi * (u - 0.5);
            const double d[3] = {std::cos(elev) * std::cos(phi), std::sin(elev), std::cos(elev) * std::sin(phi)};
            const double up = d[1] < 0.0 ? 0.0 : (d[1] > 1.0 ? 1.0 : d[1]);
            const double sky[3] = {(1 - up) * 0.9 + up * 0.25, (1 - up) * 0.9 + up * 0.45, (1 - up) * 0.95 + up * 0.9};
            const double ground[3] = {0.18, 0.16, 0.14};
            double c = d[0] * sun[0] + d[1] * sun[1] + d[2] * sun[2];
            c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
            const double lobe = 60.0 * std::exp((c - 1.0) / 0.004);
            const double tint[3] = {1.0, 0.93, 0.82};
            for (int k = 0; k < 3; k++)
                rgb[((size_t)i * width + j) * 3 + k] = (float)((d[1] >= 0.0 ? sky[k] : ground[k]) + lobe * tint[k]);
        }
    }
    return rgb;
}

}  // namespace

int main(int argc, char** argv)
{
    std::vector<std::string> files;
    int width = 1024, height = 1024, spp = 49, bounces = 3;  // the reference's defaults
    float eye[3] = {-2.5f, 1.0f, 0.0f}, center[3] = {0.0f, 0.0f, 0.0f}, fovy = 1.57f, exposure = 1.0f;
    bool deviceBvh = false, fullSweepBvh = false;
    std::string out = "render.pfm", ppm, hdriPath;
    int hdriW = 1024, hdriH = 512;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) {
            if (i + n >= argc) {
                std::fprintf(stderr, "ptrender: %s needs %d value(s)\n", a.c_str(), n);
                std::exit(2);
            }
        };
        if (a == "--width") { need(1); width = std::atoi(argv[++i]); }
        else if (a == "--height") { need(1); height = std::atoi(argv[++i]); }
        else if (a == "--spp") { need(1); spp = std::atoi(argv[++i]); }
        else if (a == "--bounces") { need(1); bounces = std::atoi(argv[++i]); }
        else if (a == "--fovy") { need(1); fovy = (float)std::atof(argv[++i]); }
        else if (a == "--eye") { need(3); for (float& e : eye) e = (float)std::atof(argv[++i]); }
        else if (a == "--look-at") { need(3); for (float& e : center) e = (float)std::atof(argv[++i]); }
        else if (a == "--device-bvh") deviceBvh = true;
        else if (a == "--full-sweep-bvh") fullSweepBvh = true;
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--ppm") { need(1); ppm = argv[++i]; }
        else if (a == "--exposure") { need(1); exposure = (float)std::atof(argv[++i]); }
        else if (a == "--hdri") { need(3); hdriPath = argv[++i]; hdriW = std::atoi(argv[++i]); hdriH = std::atoi(argv[++i]); }
        else if (a.rfind("--", 0) == 0) { std::fprintf(stderr, "ptrender: unknown option %s\n", a.c_str()); return 2; }
        else files.push_back(a);
    }
    if (files.empty() || width <= 0 || height <= 0 || spp <= 0) {
        std::fprintf(stderr, "usage: ptrender scene.glb [...] [--width W] [--height H] [--spp N] [--bounces B] [--eye x y z] "
                             "[--look-at x y z] [--fovy f] [--device-bvh | --full-sweep-bvh] [--hdri file w h] [--out x.pfm] [--ppm x.ppm] [--exposure e]\n");
        return 2;
    }

    pt_ctx* ctx = nullptr;
    PT(pt_create(0, width, height, &ctx));
    size_t triangles = 0;
    for (const std::string& file : files) {
        pt_gltf* g = nullptr;
        PT(pt_gltf_load(file.c_str(), &g));
        uint32_t count = 0;
        PT(pt_gltf_submesh_count(g, &count));
        for (uint32_t s = 0; s < count; s++) {
            uint64_t nv = 0, ni = 0;
            PT(pt_gltf_submesh_size(g, s, &nv, &ni));
            std::vector<float> vertices(nv * 8);
            std::vector<uint32_t> indices(ni);
            PT(pt_gltf_submesh_read(g, s, vertices.data(), indices.data()));
            float attributes[18];
            int32_t images[11];
            PT(pt_gltf_submesh_material(g, s, attributes, images));
            MaterialAlign m{};
            std::memcpy(m.emission, attributes, 12);
            std::memcpy(m.albedo, attributes + 3, 12);
            std::memcpy(m.scalars, attributes + 6, 48);
            for (int slot = 0; slot < 11; slot++)
                if (images[slot] >= 0)
                    std::fprintf(stderr, "ptrender: %s sub-mesh %u: texture slot %d skipped (no image decoder here)\n",
                                 file.c_str(), s, slot);
            PT(pt_add_model(ctx, vertices.data(), nv, indices.data(), ni, 4, &m));
            triangles += ni / 3;
        }
        pt_gltf_free(g);
    }
    // the reference's builder bit for bit unless asked otherwise (both alternatives trace faster)
    PT(deviceBvh ? pt_process_buffers_device(ctx) : (fullSweepBvh ? pt_process_buffers_full_sweep(ctx) : pt_process_buffers(ctx)));

    std::vector<float> hdri;
    if (hdriPath.empty()) {
        hdri = MakeSky(hdriW, hdriH);
    } else {
        hdri.resize((size_t)hdriW * hdriH * 3);
        FILE* f = std::fopen(hdriPath.c_str(), "rb");
        if (!f || std::fread(hdri.data(), 4, hdri.size(), f) != hdri.size()) {
            std::fprintf(stderr, "ptrender: cannot read %d x %d float RGB texels from %s\n", hdriW, hdriH, hdriPath.c_str());
            return 1;
        }
        std::fclose(f);
    }
    PT(pt_upload_environment(ctx, hdri.data(), hdriW, hdriH));  // builds the importance-sampling cache

    pt_params p;
    pt_default_params(width, height, &p);
    const float up[3] = {0.0f, 1.0f, 0.0f};
    pt_look_at(eye, center, up, p.view_matrix);
    pt_perspective(fovy, (float)width / (float)height, 0.01f, 1000.0f, p.proj_matrix);
    pt_inverse4(p.proj_matrix, p.inv_proj_matrix);
    p.max_bounces = (uint32_t)bounces;
    PT(pt_set_params(ctx, &p));
    PT(pt_clear(ctx));
    PT(pt_dispatch(ctx, 1, (uint32_t)spp, 1));
    std::vector<float> rgb((size_t)width * height * 3);
    PT(pt_read_image_rgb(ctx, PT_IMAGE_RADIANCE, rgb.data()));
    float ms = 0.0f;
    PT(pt_last_dispatch_ms(ctx, &ms));
    pt_stats st;
    PT(pt_get_stats(ctx, &st));

    // PFM stores rows bottom-up, which is the order of the images (row 0 = bottom, as in GL)
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "ptrender: cannot write %s\n", out.c_str()); return 1; }
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    std::fwrite(rgb.data(), 4, rgb.size(), f);
    std::fclose(f);
    if (!ppm.empty()) {  // the reference's display pass (exposure 1, ACES fit, sRGB encoding), done on the device
        std::vector<unsigned char> display((size_t)width * height * 3);
        PT(pt_read_display(ctx, exposure, display.data()));
        f = std::fopen(ppm.c_str(), "wb");
        if (!f) { std::fprintf(stderr, "ptrender: cannot write %s\n", ppm.c_str()); return 1; }
        std::fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (int y = height - 1; y >= 0; y--)  // PPM rows run top to bottom
            std::fwrite(display.data() + (size_t)y * width * 3, 1, (size_t)width * 3, f);
        std::fclose(f);
    }
    std::printf("ptrender: %zu triangles, %dx%d, %d spp, %d bounces: %.2f ms on the GPU, %.0f Mrays/s -> %s\n", triangles, width,
                height, spp, bounces, ms, (double)(st.rays_closest + st.rays_any) / (ms * 1e3), out.c_str());
    pt_destroy(ctx);
    return 0;
}
