/*
 * ptcore.h -- C ABI of the MI355X-native path-tracing core (libptcore.so).
 *
 * The library replaces ONE path of Kliaxe/PathTracer: "upload scene buffers ->
 * dispatch the path-tracing compute shader -> accumulate into three RGBA32F
 * images".  The reference has no plugin/FFI layer; the seam is the set of plain
 * OpenGL calls its two host classes make.  Every entry point below names the
 * reference call(s) it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - Plain C: opaque context, pointers and sizes only.  No exceptions cross the
 *     boundary.  Every function returning int returns PT_OK (0) or a negative
 *     PT_ERR_* code; pt_last_error() gives a thread-local message for the last
 *     failure on the calling thread.
 *   - Inputs are borrowed for the duration of the call and copied (glBufferData
 *     semantics); the context owns all device memory it allocates.
 *   - One context = one HIP device + one stream.  Calls on a context must be
 *     serialised by the caller; different contexts may be driven from different
 *     threads (the reference is single-threaded: Application.cpp:51-64).
 *   - Matrices are column-major float[16] (glm / GLSL layout).
 *   - Image texel (x, y) lives at float offset (y * width + x) * 4; row 0 is the
 *     GL image origin (bottom row when displayed).
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID (-1) /* bad argument (null pointer, size, enum)               */
#define PT_ERR_STATE (-2)   /* call order: scene/params/environment missing          */
#define PT_ERR_DEVICE (-3)  /* HIP runtime failure (message carries hipGetErrorName) */
#define PT_ERR_NOMEM (-4)   /* host or device allocation failed                      */
#define PT_ERR_SCENE (-5)   /* malformed BVH / primitive / material table            */

#define PT_IMAGE_RADIANCE 0 /* image unit 0: radianceImage        (pathtracing.comp:8)  */
#define PT_IMAGE_ALBEDO 1   /* image unit 1: primaryAlbedoImage   (pathtracing.comp:9)  */
#define PT_IMAGE_NORMAL 2   /* image unit 2: primaryNormalImage   (pathtracing.comp:10) */

#define PT_ACCUM_RUNNING_MEAN 0 /* mix(prev, x, 1/FrameCount): pathtracing.comp:248-265 */
#define PT_ACCUM_SUM 1          /* prev + x: for sample-sharded multi-GPU renders       */

/* Record sizes of the reference's buffers. */
#define PT_SIZEOF_NODE 40       /* BVH::BvhNode            (BVH.h:8-21)   */
#define PT_SIZEOF_PRIMITIVE 100 /* BVH::BvhPrimitive       (BVH.h:23-38)  */
#define PT_SIZEOF_NODE_ALIGN 48 /* BVH::BvhNodeAlign       (BVH.h:40-48)  */
#define PT_SIZEOF_PRIMITIVE_ALIGN 112 /* BVH::BvhPrimitiveAlign (BVH.h:50-62) */
#define PT_SIZEOF_MATERIAL_ALIGN 176  /* MaterialAlign (PathTracingRenderer.h:163-193) */
#define PT_SIZEOF_VERTEX 32     /* VertexSave {pos, nor, uv} (PathTracingRenderer.h:117-122) */

typedef struct pt_ctx pt_ctx;

/*
 * The uniform block of the compute shader (Shaders/Library/resources.glsl:6-31) as
 * set each frame by PathTracingApplication::UpdateMaterial
 * (PathTracingApplication.cpp:244-274), plus two knobs the reference hard-codes.
 */
typedef struct pt_params {
    float view_matrix[16];     /* ViewMatrix                                        */
    float proj_matrix[16];     /* ProjMatrix (uploaded by the reference, unused)    */
    float inv_proj_matrix[16]; /* InvProjMatrix = inverse(ProjMatrix), host-side    */
    uint32_t frame_count;      /* FrameCount: ignored; pt_dispatch supplies it      */
    float frame_dimensions[2]; /* FrameDimensions: must equal the context size or 0 */
    uint32_t anti_aliasing_enabled;
    float focal_length;
    float aperture_size;
    float aperture_shape[2];
    float specular_modifier;
    float specular_tint_modifier;
    float metallic_modifier;
    float roughness_modifier;
    float subsurface_modifier;
    float anisotropy_modifier;
    float sheen_roughness_modifier;
    float sheen_tint_modifier;
    float clearcoat_modifier;
    float clearcoat_roughness_modifier;
    float refraction_modifier;
    float transmission_modifier;
    float debug_value_a;
    float debug_value_b;
    /* not uniforms in the reference: */
    uint32_t max_bounces;     /* `const uint MaxBounceCount = 3` (pathtracing.comp:24); 0 -> 3 */
    uint32_t accumulate_mode; /* PT_ACCUM_*                                                   */
} pt_params;

/* Counters of the most recent dispatch.  samples, rays_closest, rays_any and closest_hits are kept by
 * every dispatch (pt_dispatch, pt_dispatch_static, pt_dispatch_counted); node_visits, triangle_tests and
 * max_stack_depth only by pt_dispatch_counted() and are zero after the others. */
typedef struct pt_stats {
    uint64_t samples;         /* (pixel, FrameCount) pairs traced                        */
    uint64_t rays_closest;    /* closest-hit BVH queries                                 */
    uint64_t rays_any;        /* any-hit (shadow) BVH queries                            */
    uint64_t node_visits;     /* two-child node records fetched (child-AABB pair tests)  */
    uint64_t triangle_tests;  /* ray/triangle tests                                      */
    uint64_t closest_hits;    /* closest-hit queries that hit                            */
    uint32_t max_stack_depth; /* largest traversal-stack occupancy seen                  */
    uint32_t reserved;
} pt_stats;

/* ---------------------------------------------------------------- life cycle */

/* Replaces: GL context + PathTracingRenderPass::InitializeTextures
 * (PathTracingRenderPass.cpp:171-204: three RGBA32F width x height images) and
 * PathTracingRenderer::InitializeBuffers (PathTracingRenderer.cpp:89-95).
 * Fails with PT_ERR_DEVICE when no HIP device is usable: there is no CPU path. */
int pt_create(int device, int width, int height, pt_ctx** out);
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(void);

/* Run the context's work on a caller-owned hipStream_t (NULL = the context's own stream). */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);

/* Render into caller-owned device memory (each width*height*4 floats) instead of the
 * context's own images; pass NULLs to return to the internal images.  Stands in for
 * glBindImageTexture units 0-2 (PathTracingRenderPass.cpp:62-64), where the images are
 * owned by the render pass, not the shader. */
int pt_bind_images(pt_ctx* ctx, void* d_radiance, void* d_albedo, void* d_normal);

/* ------------------------------------------------- scene buffers (SSBO 0..3) */

/* SSBO binding 2 (PathTracingRenderer.cpp:566-613): `count` BvhNodeAlign records of 48 B,
 * node 0 the dummy, root = node 1.  The tree is taken as it is: at the next dispatch it is repacked
 * (leaves folded into child references) and folded, on the device, into the four-wide nodes the kernels
 * walk -- the same boxes and the same leaves, so the same hits; PT_ERR_SCENE at that dispatch if the
 * tree is more than 64 levels deep. */
int pt_upload_nodes(pt_ctx* ctx, const void* nodes48, size_t count);
/* SSBO binding 3 (PathTracingRenderer.cpp:615-662): BvhPrimitiveAlign records of 112 B in
 * post-build order. */
int pt_upload_primitives(pt_ctx* ctx, const void* prims112, size_t count);
/* No counterpart in the reference.  By default (enable != 0) the tree handed to pt_upload_nodes is checked
 * but not traversed: the GPU builder (pt_build_bvh_device) makes its own over the primitives of
 * pt_upload_primitives at the next dispatch (milliseconds).  The reference's BVH::BuildBvhWithSah caps its
 * split cost and cuts the top of any sizeable scene at the median along X, which costs 1.1-4x the node
 * visits per ray; a host that keeps calling it gets the better tree with no change on its side, and the
 * same images (the tree only decides where rays look; equal except where two hit distances tie exactly).
 * enable = 0: the uploaded tree itself is folded and traversed.  pt_download_bvh returns the tree in use.
 * A tree made by pt_process_buffers / pt_process_buffers_full_sweep is always traversed as built: there the
 * caller has named its builder (pt_process_buffers_device is the call that names the GPU builder). */
int pt_set_bvh_rebuild(pt_ctx* ctx, int enable);
/* SSBO binding 1 (PathTracingRenderer.cpp:509-564): MaterialAlign records of 176 B.  The 11
 * 64-bit bindless handles are reinterpreted as 1-based slots of pt_upload_texture
 * (0 = no texture, as in the reference). */
int pt_upload_materials(pt_ctx* ctx, const void* mats176, size_t count);
/* Stands in for a resident bindless sampler2D (Texture2DLoader.cpp:16-62;
 * PathTracingRenderPass.cpp:47-50).  slot >= 1.  channels 1..4; texels are u8 (unorm,
 * is_srgb decodes the colour channels like GL_SRGB8) or f32 when is_float.  Row 0 is the
 * GL row 0 (the caller applies the loader's vertical flip).  Sampling is bilinear, REPEAT,
 * base level; a texture of fewer than four channels reads as (r, 0, 0, 1) / (r, g, 0, 1) / (r, g, b, 1).
 * On the device every texel takes four channels; all textures together must stay below 4 GB in that form. */
int pt_upload_texture(pt_ctx* ctx, int slot, int width, int height, int channels, int is_srgb, int is_float,
                      const void* texels);
/* SSBO binding 0 + the two environment textures (PathTracingRenderer.cpp:463-507): RGB32F
 * HDRI of w x h (row 0 first, not flipped).  Builds the importance-sampling cache on the
 * host (CalculateHdriCache, PathTracingRenderer.cpp:776-919). */
int pt_upload_environment(pt_ctx* ctx, const float* hdri_rgb, int width, int height);
/* Same, with a caller-computed cache (w*h*3 floats: sample x, sample y, pdf). */
int pt_upload_environment_with_cache(pt_ctx* ctx, const float* hdri_rgb, const float* cache_rgb, int width,
                                     int height);

/* ------------------------- host-side scene processing (the reference's L4 CPU code) */

/* PathTracingRenderer::AddPathTracingModel / ClearPathTracingModels / ProcessBuffers
 * (PathTracingRenderer.cpp:156-168,170-461): queue sub-meshes (32-B vertices, indices of
 * index_bytes = 1, 2 or 4 each, one 176-B material per sub-mesh), then flatten to a triangle
 * soup in submission order, build the sweep-SAH BVH (leaf size 4), convert to the aligned
 * layouts and upload nodes, primitives and materials.  The environment is uploaded
 * separately.  World matrices are not applied (the reference stores but never uses them). */
int pt_add_model(pt_ctx* ctx, const void* vertices32, size_t vertex_count, const void* indices,
                 size_t index_count, int index_bytes, const void* material176);
int pt_clear_models(pt_ctx* ctx);
int pt_process_buffers(pt_ctx* ctx);

/* BVH::BuildBvhWithSah (BVH.cpp:91-238) as called at PathTracingRenderer.cpp:575-584:
 * prims100 (count records of 100 B) is reordered in place; nodes40_out receives the nodes
 * including the zeroed dummy node 0.  Bit-identical to the reference built with libstdc++.
 * *node_count_out = number of nodes written.  PT_ERR_INVALID if node_capacity is too small
 * (2*count+1 always suffices). */
int pt_build_bvh_sah(void* prims100, size_t count, int leaf_size, void* nodes40_out, size_t node_capacity,
                     size_t* node_count_out);
/* Triangle-soup assembly for one sub-mesh (PathTracingRenderer.cpp:414-446). */
int pt_assemble_primitives(const void* vertices32, size_t vertex_count, const void* indices, size_t index_count,
                           int index_bytes, uint32_t mesh_index, void* prims100_out);
/* BvhNode -> BvhNodeAlign, BvhPrimitive -> BvhPrimitiveAlign
 * (PathTracingRenderer.cpp:590-605,623-654). */
int pt_align_nodes(const void* nodes40, size_t count, void* nodes48_out);
int pt_align_primitives(const void* prims100, size_t count, void* prims112_out);
/* CalculateHdriCache (PathTracingRenderer.cpp:803-901): cache_out = w*h*3 floats. */
int pt_calculate_hdri_cache(const float* hdri_rgb, int width, int height, float* cache_out);

/* The same sweep builder without the reference's cost cap (no counterpart in the reference).
 * BVH.cpp accepts a split only while its cost, area x count, stays below the constant 114514
 * (BVH.cpp:4,132-134,209-222); a node that exceeds it is cut at the median along X, which is what the
 * top levels of every scene of more than a few thousand triangles get.  Here every node takes its
 * best sweep split: a different, better tree (a third of the node visits per ray on the benchmark
 * scene), same node and primitive layouts, images equal to the reference tree's except where
 * distances tie exactly.  Arguments as pt_build_bvh_sah / pt_process_buffers. */
int pt_build_bvh_full_sweep(void* prims100, size_t count, int leaf_size, void* nodes40_out, size_t node_capacity,
                            size_t* node_count_out);
int pt_process_buffers_full_sweep(pt_ctx* ctx);

/* BVH construction on the GPU (no counterpart in the reference): Morton-sorted triangles are merged
 * bottom-up by locally-ordered clustering (nearest partner by surface area within a window of the
 * sorted order), sub-trees of at most 4 triangles become leaves.  Built for scenes where
 * BVH::BuildBvhWithSah's O(n log^2 n) sweep takes minutes (20 M triangles in 0.11 s including the
 * upload); the tree differs from the one BVH.cpp builds (its SAH cost is close to an uncapped sweep's,
 * far below the reference's capped one), images differ from the reference's only where distances
 * tie exactly.  prims112: `count` unsorted BvhPrimitiveAlign records (host memory); the call uploads
 * them, builds and installs the geometry in place of pt_upload_nodes + pt_upload_primitives.
 * PT_ERR_SCENE if the tree comes out deeper than the traversal stack (64 levels); PT_ERR_INVALID for
 * 2^26 primitives or more (the kernels address node and triangle records by 32-bit byte offsets, so
 * either array stays below 4 GB; an uploaded tree of that size is refused with PT_ERR_SCENE at the
 * first dispatch). */
int pt_build_bvh_device(pt_ctx* ctx, const void* prims112, size_t count);
/* pt_process_buffers with the GPU builder in place of BuildBvhWithSah. */
int pt_process_buffers_device(pt_ctx* ctx);
/* The device-built tree in the reference's layouts (BvhNodeAlign incl. dummy node 0, root = node 1;
 * BvhPrimitiveAlign in leaf order).  Pass NULL output pointers to query the counts. */
int pt_download_bvh(pt_ctx* ctx, void* nodes48_out, size_t node_capacity, size_t* node_count_out, void* prims112_out,
                    size_t prim_capacity, size_t* prim_count_out);

/* On-disk cache of a built BVH (no counterpart in the reference, which rebuilds the tree at every
 * start: PathTracingRenderer.cpp:566-613).  The key is a hash of the UNSORTED primitive soup the tree
 * was built from, so a file built from other geometry is refused (PT_ERR_STATE), as is a missing,
 * truncated or corrupt one.  pt_bvh_cache_load with NULL outputs returns the counts of a valid file. */
int pt_bvh_cache_key(const void* primitives, size_t bytes, uint64_t* key_out);
int pt_bvh_cache_save(const char* path, uint64_t key, const void* nodes48, size_t node_count, const void* prims112,
                      size_t prim_count);
int pt_bvh_cache_load(const char* path, uint64_t key, void* nodes48_out, size_t node_capacity, size_t* node_count_out,
                      void* prims112_out, size_t prim_capacity, size_t* prim_count_out);

/* glTF 2.0 / GLB scene reader: what ModelLoader::Load (ModelLoader.cpp:83-121) hands to
 * PathTracingRenderer::AddPathTracingModel, without assimp.  One sub-mesh per mesh primitive, in file
 * order; node transforms are ignored as in the reference; vertices are {pos, nor, uv} with V flipped
 * (v -> 1 - v) as the reference's importer does for glTF.  PT_ERR_SCENE (message via
 * pt_last_error) for unreadable files, non-triangle primitives, sparse accessors, a primitive
 * without TEXCOORD_0 (the reference throws there too: ModelLoader.cpp:389-403). */
typedef struct pt_gltf pt_gltf;
int pt_gltf_load(const char* path, pt_gltf** out);
void pt_gltf_free(pt_gltf* scene);
int pt_gltf_submesh_count(const pt_gltf* scene, uint32_t* count_out);
int pt_gltf_submesh_size(const pt_gltf* scene, uint32_t submesh, uint64_t* vertex_count_out, uint64_t* index_count_out);
/* vertices32_out: vertex_count x 32 B {pos3, nor3, uv2}; indices_out: index_count x u32. */
int pt_gltf_submesh_read(const pt_gltf* scene, uint32_t submesh, void* vertices32_out, uint32_t* indices_out);
/* ModelLoader::GenerateMaterial (ModelLoader.cpp:150-287).  attributes18_out: the 18 floats of
 * Material::MaterialAttributes in declaration order (Material.h:43-59: emission3, albedo3, specular,
 * specularTint, metallic, roughness, subsurface, anisotropy, sheenRoughness, sheenTint, clearcoat,
 * clearcoatRoughness, refraction, transmission).  images11_out: per Material::MaterialTextureSlot the
 * index of the image to load, or -1.  Slot formats as the reference requests them
 * (ModelLoader.cpp:201-231): albedo sRGB8; specular and sheen-roughness RGBA8; the others RGB8;
 * all loaded flipped vertically (PathTracingApplication.cpp:212). */
int pt_gltf_submesh_material(const pt_gltf* scene, uint32_t submesh, float attributes18_out[18], int32_t images11_out[11]);
int pt_gltf_image_count(const pt_gltf* scene, uint32_t* count_out);
/* Where image `image` comes from: *path_out = file to decode (NULL when embedded), else
 * *bytes_out / *byte_count_out = the encoded (PNG/JPEG) bytes, owned by `scene`.  Decoding images is
 * the caller's job, as it is stb_image's in the reference. */
int pt_gltf_image_source(const pt_gltf* scene, uint32_t image, const char** path_out, const void** bytes_out,
                         uint64_t* byte_count_out);
/* glm::lookAt / glm::perspective (right-handed, -1..1 depth) / glm::inverse, as used by
 * Runtime/Camera/Camera.cpp:10-18 and PathTracingApplication.cpp:233-235,250. */
void pt_look_at(const float eye[3], const float center[3], const float up[3], float out16[16]);
void pt_perspective(float fovy, float aspect, float z_near, float z_far, float out16[16]);
void pt_inverse4(const float in16[16], float out16[16]);
/* Fills every field with the reference's defaults (PathTracingApplication.h:108-144,
 * PathTracingApplication.cpp:229-242) for a width x height frame. */
void pt_default_params(int width, int height, pt_params* out);

/* ------------------------------------------------------------------ per frame */

/* PathTracingApplication::UpdateMaterial (PathTracingApplication.cpp:244-274) +
 * Material::Use() uniform flush (PathTracingRenderPass.cpp:59). */
int pt_set_params(pt_ctx* ctx, const pt_params* params);
/* Zero the three images (the reference relies on weight 1 at FrameCount 1 instead:
 * pathtracing.comp:248; Texture2DObject.cpp:19-22). */
int pt_clear(pt_ctx* ctx);
/* glDispatchCompute + glMemoryBarrier (PathTracingRenderPass.cpp:66-72), for FrameCount =
 * frame_first, frame_first + frame_stride, ... (frame_count values) in one launch; every
 * pixel of the frame is rendered (the reference drops the last height % 16 rows).
 * Asynchronous on the context's stream. */
int pt_dispatch(pt_ctx* ctx, uint32_t frame_first, uint32_t frame_count, uint32_t frame_stride);
/* Same image result, and additionally fills the pt_stats counters (slower). */
int pt_dispatch_counted(pt_ctx* ctx, uint32_t frame_first, uint32_t frame_count, uint32_t frame_stride);
/* No counterpart in the reference.  pt_dispatch for callers that cannot allocate device memory: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and
 * bit-identical images, but nothing is allocated, freed, uploaded or timed: the sample records of the
 * launch and the per-tile cost words live in a static device array of PT_STATIC_SCRATCH_BYTES that every
 * GPU the library runs on holds until the process ends, and tiles launch in row-major order (no launch
 * plan).  PT_ERR_INVALID when 256 + 36 bytes x width x height x frame_count (+ 4 per 8x8 tile) exceed the
 * array: dispatch fewer frames per call.  PT_ERR_STATE when scene buffers or textures changed since the
 * last pt_dispatch (which uploads them), or under the wavefront scheduler.  All contexts on one GPU share
 * the array: calls that overlap on one GPU (different streams or threads) race.
 * The kernel's occupancy is asked of the runtime when the context is created, so a call itself makes no
 * runtime query either. */
#define PT_STATIC_SCRATCH_BYTES (128u << 20)
int pt_dispatch_static(pt_ctx* ctx, uint32_t frame_first, uint32_t frame_count, uint32_t frame_stride);
/* No counterpart in the reference.  The limits of pt_dispatch_static as built: size of the static array
 * and bytes of one sample record (either pointer may be null). */
int pt_static_scratch_info(uint64_t* scratch_bytes, uint32_t* sample_record_bytes);
/* Multiply the three images by `scale` (sum mode: 1/N after the cross-GPU reduce). */
int pt_scale_images(pt_ctx* ctx, float scale);
int pt_synchronize(pt_ctx* ctx);
/* glGetTexImage(GL_RGBA, GL_FLOAT) (PathTracingRenderPass.cpp:89-99 reads GL_RGB): copies
 * width*height*4 floats to host memory; synchronises the stream. */
int pt_read_image(pt_ctx* ctx, int which, float* rgba_out);
/* The inverse of pt_read_image: replaces an accumulation image by width*height RGBA floats from the
 * host, so that a progressive render can go on from saved images (continue with the FrameCount
 * that follows the last one accumulated).  The reference keeps its accumulation only in GL textures
 * and restarts at frame 1 after any change (PathTracingApplication.cpp:85-88,
 * PathTracingRenderPass.cpp:62-64); this entry point has no counterpart there.  Synchronises. */
int pt_write_image(pt_ctx* ctx, int which, const float* rgba);
/* The same read-back in the reference's own layout: glGetTexImage(GL_RGB, GL_FLOAT) into
 * glm::vec3[width*height] (PathTracingRenderPass.cpp:89-99), which is also the Float3 row-major
 * buffer the reference hands to its denoiser for colour, albedo and normal (:101-155).  Copies
 * width*height*3 floats; synchronises the stream. */
int pt_read_image_rgb(pt_ctx* ctx, int which, float* rgb_out);
/* The reference's display pass over the radiance image: PostFXRenderPass with Shaders/tonemapping.frag
 * (:21-71: exposure, fitted ACES curve, clamp) drawn into the default framebuffer with
 * GL_FRAMEBUFFER_SRGB enabled (Runtime/Renderer/Renderer.cpp:25; the pass is installed at
 * PathTracingRenderer.cpp:85-86), read back as 8-bit RGB: width*height*3 bytes, row 0 = GL row 0
 * (the bottom row when displayed).  `exposure` is the Exposure uniform (PathTracingApplication.h
 * default 1.0).  Runs on the device; synchronises the stream. */
int pt_read_display(pt_ctx* ctx, float exposure, unsigned char* rgb8_out);
/* The two diagnostic views the reference's compute shader writes to the radiance image, in place of a
 * frame, when it is compiled with DEBUG_ENABLED (Shaders/pathtracing.comp:216-223 calling
 * Shaders/Library/debug.glsl:129-137): PT_DEBUG_VIEW_BVH is DebugBvh (:31-126), the number of leaves the
 * pixel's pinhole ray reaches when every child box it enters is followed, as a heat map saturating at 100
 * and purple where the shader's 16-entry stack overflows; PT_DEBUG_VIEW_HDRI_CACHE is DebugHdriCache
 * (:7-28), the environment map with a red dot wherever one of 500 draws from the sampling cache lands.
 * Needs pt_set_params (camera matrices), the uploaded tree and the environment; overwrites the radiance
 * image (the bound one if images are bound), accumulates nothing; synchronises the stream. */
#define PT_DEBUG_VIEW_BVH 1
#define PT_DEBUG_VIEW_HDRI_CACHE 2
int pt_debug_view(pt_ctx* ctx, int view);
int pt_get_stats(pt_ctx* ctx, pt_stats* out);
/* Milliseconds spent by the device in the most recent dispatch (HIP events on the context's
 * stream); synchronises. */
int pt_last_dispatch_ms(pt_ctx* ctx, float* ms_out);
/* The same for the two kernels of the most recent megakernel launch, each on its own: the trace
 * kernel (renderKernel) and the pass that folds its sample records into the images (resolveKernel).
 * A dispatch that was split into several launches reports its last one.  Measurement only; the
 * reference has no counterpart (it times nothing). */
int pt_last_kernel_ms(pt_ctx* ctx, float* trace_ms_out, float* resolve_ms_out);
/* Device-side sizes of the packed scene, for bandwidth accounting. */
int pt_scene_bytes(pt_ctx* ctx, uint64_t* node_bytes, uint64_t* triangle_bytes, uint64_t* shading_bytes);

/* Tuning hook, not part of the drop-in surface (no counterpart in the reference).  key 0: scheduler
 * (0 = single-launch megakernel, the default; 1 = wavefront pipeline, which takes trees of at most 21
 * four-wide levels where the megakernel takes 32: select it before the upload so that the fold aims for that); 1: megakernel traversal-phase
 * exit threshold in 64ths of the lanes that entered the phase (default 28); 2: wavefront refill threshold in idle
 * lanes (default 16); 3: wavefront iterations between read-backs of the live-ray count (default 2);
 * 4: traverse grid size in blocks (0 = sized from the occupancy query); 5: megakernel hit-shading
 * threshold in 64ths of the live lanes (default 28): hits wait until that share of the wave holds one;
 * 6: costliest-block-first launch order from the previous launch's per-tile cost (default 1); 7: tile
 * split: a tile whose cost exceeds 1/value of the launch's lower bound (total cost / resident waves) has
 * its samples spread over several work items; 0 = one item per tile; -1 (default) = 8 for images with
 * fewer tiles than resident waves, else 32;
 * 8: budget of the megakernel's sample buffer in MiB (36 bytes per sample and pixel; a dispatch is cut
 * into launches that fit; default 8192); 9: megakernel run by a grid of resident waves that pull the work
 * items off a counter (1, the default) or by one block per work item (0); 10: device
 * BVH builder: clusters searched on either side per round (default 32; 0 = Morton radix tree instead);
 * 11: megakernel variant compiled for 3 or 4 resident waves per SIMD (default 0: 3); 12: the trace launches
 * of consecutive pt_dispatch calls may overlap, the next call's waves filling the chip while the last waves of
 * the call before run dry (1), never do (0), or do for calls of at most 8 frames (-1, the default); calls
 * overlap on the library's own stream only, never on one set with pt_set_stream; 13: device BVH builder: most
 * triangles per leaf, 1..4 (default 0: 2 for scenes of up to 8 M triangles, 3 above);
 * 14: 8-bit textures are kept on the device as decoded floats (four per texel: a bilinear tap is one 16-byte read) while
 * all of them together fit in this many MiB, and as 8-bit words with a decode table beyond (default 256; 0 = never
 * decode).  Either form gives the same bits; 15: a launch lane makes a new launch plan on every value-th launch
 * and reuses the item list for the launches of the same frame count in between (default 8; 1 = plan every launch); 16: a call whose trace overlaps
 * its neighbours' launches 1/value of the waves the chip keeps resident (0 = all of them; -1, the default: half for calls of
 * one or two frames, all otherwise): the calls in flight fill the chip together, and each wave lives through more work
 * items before its last one ends in idle lanes; 17: device BVH builder: reinsertion passes over the clustered tree,
 * 0..256 (default -1: 16, and 8 for scenes of more than four million triangles; 0 = the clustered tree as built) -- in a
 * pass every candidate node looks for the place in the tree where it, moved there with its parent, lowers the summed
 * surface area most, and the moves that do not touch each other's nodes are made; 18: every value-th node is a candidate
 * in a pass, the offset advancing pass by pass (default 2); 19: a tree that the passes left more than this many levels
 * deep is put back as the clustering built it (default 60, the most the traversal takes with room to spare); 20: when a
 * caller's tree is traversed as uploaded (pt_set_bvh_rebuild(ctx, 0), pt_process_buffers), its leaves of more than value
 * triangles are cut into leaves of at most value, in place and in order, under new interior nodes with the tight boxes
 * of the sub-ranges (default 0 = the caller's leaves as they are: cutting them to two gains 0.1-1.5 % on four of the
 * benchmark scenes and loses 1.9 % on the interior; set before the upload).  Trees
 * differ between settings, images only where two triangles are hit at exactly the same distance. */
int pt_debug_set_tuning(pt_ctx* ctx, int key, int value);

/* Profiling hook, not part of the drop-in surface (no counterpart in the reference): wave-level
 * occupancy counters of the most recent pt_dispatch_counted() under the megakernel scheduler:
 * [0] traversal iterations, [1] lanes traversing in them, [2] iterations with a node step, [3] lanes
 * in node steps, [4] iterations with a triangle step, [5] lanes in triangle steps, [6] shading phases
 * that shaded something, [7] lanes shading a hit, [8] lanes shading a miss, [9] shading phases,
 * [10] lanes finishing a sample, [11] lanes starting a sample, [12] shader clock ticks waves spent in
 * shading phases, [13] in traversal phases, [14] of [12] in the hit block, [15] of [12] in the miss block. */
int pt_debug_get_profile(pt_ctx* ctx, uint64_t out16[16]);
/* Profiling hook (no counterpart in the reference): closest-hit queries of the most recent
 * pt_dispatch_counted() under the megakernel scheduler by path depth: [0] primary rays, [1] the rays
 * after the first bounce, ...; depths past 31 are counted in [31]. */
int pt_debug_get_depth_histogram(pt_ctx* ctx, uint64_t out32[32]);
/* The four-wide tree the kernels walk (made of the uploaded or device-built binary tree at the first dispatch
 * after it changed): its depth in levels, and whether it was folded level by level (1) because folding by
 * surface area had left it deeper than the traversal stacks hold (32 levels). */
int pt_debug_get_tree_info(pt_ctx* ctx, uint32_t* levels_out, uint32_t* level_by_level_out);
/* Waves the trace kernel variant of the most recent megakernel launch keeps resident on the GPU (registers
 * and LDS both limit it; the runtime is asked once per variant): the launch's grid when the waves pull
 * their work items. */
int pt_debug_get_resident_waves(pt_ctx* ctx, uint32_t* waves_out);
/* Profiling hook (no counterpart in the reference): device wall-clock ticks at the start and end of every
 * wave of the most recent pt_dispatch_counted() under the megakernel scheduler, as (start, end) pairs
 * per block of the grid (start = 0: the block found no work); *clock_khz_out = tick rate.  NULL pairs_out
 * queries the count. */
int pt_debug_get_wave_times(pt_ctx* ctx, uint64_t* pairs_out, size_t capacity, size_t* count_out, int* clock_khz_out);

/* Test hook, not part of the drop-in surface (no counterpart in the reference): evaluates one
 * float32 building block of the kernels element-wise on the device so that tests can compare it
 * bit for bit with a CPU evaluation of the same operation sequence.  fn: 0 sin, 1 cos, 2 asin,
 * 3 atan2(a,b), 4 pow(a,b), 5 min, 6 max, 7 a/b, 8 sqrt, 9 1/sqrt.  a, b, out: host arrays of n. */
int pt_debug_eval_math(int device, int fn, const float* a, const float* b, float* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
