"""GPU parity on the multi-material scenes: the reference's AreaLight and Mill glTF scenes (read by
the product's glTF reader, stored as .npz fixtures) and the synthetic sphere field.  Same bar as
tests/test_gpu_parity.py: bit-exact float32 images against the CPU oracle.

The Mill textures are .jpg files decoded with Pillow and box-filtered to 256 texels when the
fixture was made: texel values are not claimed to equal what stb_image gives the reference
(parity of the decode is unpinned); oracle and GPU see the same texels.
"""
import os

import numpy as np
import pytest

from pathtracer_amd import gltf
from pathtracer_amd import renderer as R
from pathtracer_amd import scenes
from tests.common import built_meshes, environment
from tests.test_gpu_parity import PIPELINES, _assert_images_equal, _pair

pytestmark = pytest.mark.gpu

SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def _render_both(scene_tuple, textures, width, height, eye, center, bounces, frames, pipeline, what):
    o, g = _pair(scene_tuple, width, height, textures=textures, pipeline=pipeline)
    p = R.default_params(width, height)
    p.max_bounces = bounces
    R.set_camera(p, eye, center, width, height, fovy=1.0)
    g.set_params(p)
    g.clear()
    g.dispatch(1, frames, 1, counted=True)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, width, height, 1, frames, 1)
    assert cpu[4].stack_overflows == 0
    _assert_images_equal(gpu, cpu[:3], cpu[3], what)
    st = g.stats()
    assert (st.rays_closest, st.closest_hits) == (cpu[4].rays_closest, cpu[4].closest_hits)
    assert np.isfinite(gpu[0]).all()
    return gpu, st


@PIPELINES
def test_arealight_scene_emission_and_glass(pipeline):
    # AreaLight.gltf: emissive disc (strength 50), three transmissive solids, a plane without a
    # material (glTF default material: white, metallic 1, roughness 1)
    s = gltf.GltfScene.from_npz(os.path.join(SCENES, "arealight.npz"))
    mats, textures = s.materials()
    assert not textures and len(mats) == 5
    assert tuple(mats[3]["emission"]) == (np.float32(50.0), np.float32(0.6932815909385681) * np.float32(50.0),
                                          np.float32(0.2317374348640442) * np.float32(50.0))
    assert (mats[0]["metallic"], mats[0]["roughness"], mats[0]["specular"]) == (1.0, 1.0, 1.0)
    assert (mats[1]["transmission"], mats[1]["specular"], mats[1]["roughness"]) == (1.0, 0.0, 0.0)
    gpu, st = _render_both(built_meshes(s.submeshes, mats), None, 160, 120, (0.5, 3.0, -8.0), (0.0, 1.2, 2.0), 6, 2,
                           pipeline, "AreaLight 160x120 2spp")
    assert gpu[0][..., :3].max() > 10.0  # the light is seen directly somewhere


@PIPELINES
def test_mill_scene_textured(pipeline):
    # Mill.gltf: three base-colour textures (sRGB decode, bilinear, REPEAT), 16 361 triangles
    s = gltf.GltfScene.from_npz(os.path.join(SCENES, "mill.npz"))
    mats, textures = s.materials()
    assert len(textures) == 3 and all(srgb and t.shape[2] == 3 for _, t, srgb in textures)
    assert [int(m["handles"][1]) for m in mats] == [1, 2, 3]
    tex = {slot: (texels, srgb) for slot, texels, srgb in textures}
    gpu, st = _render_both(built_meshes(s.submeshes, mats), tex, 160, 120, (14.0, 9.0, 16.0), (0.0, 5.0, 0.0), 3, 2,
                           pipeline, "Mill 160x120 2spp")
    assert gpu[1][..., :3].std() > 0.02  # the albedo image shows the textures


@PIPELINES
def test_sphere_field_sixteen_bounces(pipeline):
    meshes, mats = scenes.sphere_field(grid=6, seed=1, levels=1)
    assert len(meshes) == 37 and len(mats) == 37
    gpu, st = _render_both(built_meshes(meshes, mats), None, 192, 128, (0.0, 6.0, -14.0), (0.0, 0.5, 0.0), 16, 2,
                           pipeline, "sphere field 192x128 2spp")
    assert st.max_stack_depth > 0


def test_bunny_subdivided_half_million_triangles():
    # the benchmark's larger stand-in for the missing dragon mesh: bunny subdivided once (576 184 triangles)
    v, i = scenes.load_ptm("bunny")
    v, i = scenes.subdivide(v, i, 1)
    meshes = [(v, i), scenes.load_ptm("floor")]
    mats = np.array([m for _, m in scenes.recipe("DragonGlass")], dtype=scenes.MATERIAL_DTYPE)
    _render_both(built_meshes(meshes, mats), None, 128, 96, (-2.5, 1.0, 0.0), (0.0, 0.0, 0.0), 8, 1, 0,
                 "subdivided bunny 128x96")


def test_converged_images_agree_within_rmse():
    # Converged-image check with a stated tolerance: 256 spp rendered by the GPU from FrameCount
    # 1..256 against 256 spp rendered by the oracle from the DISJOINT FrameCount 257..512 (other
    # random streams entirely).  Both estimate the same image, so they may differ only by Monte-Carlo
    # noise: relative RMSE of the tone-mapped radiance below 2 %, mean within 0.5 %.
    from pathtracer_amd import image_io
    width, height, n = 96, 96, 256
    o, g = _pair("BunnyClearcoat", width, height, env=(512, 256))
    p = R.default_params(width, height)
    g.set_params(p)
    g.clear()
    g.dispatch(1, n, 1)
    gpu = g.read_image(0)[..., :3]
    p.accumulate_mode = 1  # sums: the oracle's frames do not start at 1, a running mean would weight them wrongly
    cpu = o.render(p, width, height, n + 1, n, 1)[0][..., :3] / np.float32(n)
    a, b = image_io.tonemap(gpu, 1.0).astype(np.float64), image_io.tonemap(cpu, 1.0).astype(np.float64)
    rmse = np.sqrt(np.mean((a - b) ** 2)) / max(np.mean(b), 1e-9)
    assert rmse < 0.02, rmse
    assert abs(gpu.mean() / cpu.mean() - 1.0) < 0.005


def test_three_builders_one_image():
    """Mill through the reference's builder, the uncapped sweep and the GPU builder: the closest hit does
    not depend on the tree, so the three images are equal (a difference could only come from two
    triangles hit at exactly the same distance)."""
    s = gltf.GltfScene.from_npz(os.path.join(SCENES, "mill.npz"))
    mats, textures = s.materials()
    hdri, cache = environment(256, 128)
    p = R.default_params(200, 120)
    R.set_camera(p, (14.0, 9.0, 16.0), (0.0, 5.0, 0.0), 200, 120, fovy=1.0)
    p.max_bounces = 6
    images = []
    for how in ("reference", "full-sweep", "device"):
        g = R.Renderer(200, 120)
        for slot, texels, srgb in textures:
            g.upload_texture(slot, texels, srgb)
        for (v, i), m in zip(s.submeshes, mats):
            g.AddPathTracingModel(v, i, m)
        if how == "device":
            g.ProcessBuffersDevice()
        else:
            g.ProcessBuffers(full_sweep=how == "full-sweep")
        g.upload_environment(hdri, cache)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 4, 1)
        images.append([g.read_image(k) for k in range(3)])
    for other in images[1:]:
        for a, b in zip(images[0], other):
            differing = int(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1).sum())
            assert differing <= 1e-4 * 200 * 120, differing


def _interior():
    from pathtracer_amd import recipes
    meshes, mats, textures, _ = recipes.load_scene("Interior")
    tex = {slot: (texels, srgb) for slot, texels, srgb in textures}
    return built_meshes(meshes, mats), tex, recipes.INTERIOR_CAMERA


@PIPELINES
def test_interior_stand_in_bit_exact(pipeline):
    """The stand-in for the reference's Fireplace scene (closed textured room, one window, emissive
    panel, transmissive solids): 160 x 120 x 2 spp, 8 bounces, bit-exact against the oracle."""
    scene, tex, cam = _interior()
    gpu, st = _render_both(scene, tex, 160, 120, cam["eye"], cam["look_at"], 8, 2, pipeline, "interior 160x120 2spp")
    assert st.rays_any > 0 and gpu[1][..., :3].std() > 0.02  # shadow rays were traced, the textures show
    # a closed room seen from inside: nearly every primary ray hits a surface and leaves a primary normal
    # (the exceptions look out of the window, or end at their first bounce)
    n = gpu[2][..., :3].astype(np.float64)
    assert ((n * n).sum(-1) > 0.25).mean() > 0.5


def test_interior_full_hd_properties():
    """The interior at the benchmark's 1920 x 1080, checked through properties the oracle does not have
    to render in full: one 8-frame dispatch equals eight one-frame dispatches bit for bit, two disjoint
    FrameCount shards in sum mode add up to their union (the multi-GPU identity), and three 8-row bands
    equal the oracle."""
    scene, tex, cam = _interior()
    width, height = 1920, 1080
    o, g = _pair(scene, width, height, textures=tex)
    p = R.default_params(width, height)
    p.max_bounces = 8
    R.set_camera(p, cam["eye"], cam["look_at"], width, height)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 8, 1)
    whole = [g.read_image(k) for k in range(3)]
    g.clear()
    for f in range(1, 9):
        g.dispatch(f, 1, 1)
    for k in range(3):
        assert g.read_image(k).tobytes() == whole[k].tobytes()
    g.clear()
    g.dispatch(1, 2, 1)
    gpu = [g.read_image(k) for k in range(3)]
    for y0, y1 in ((0, 8), (536, 544), (1072, 1080)):
        cpu = o.render(p, width, height, 1, 2, 1, rows=(y0, y1))
        assert cpu[4].stack_overflows == 0
        for k in range(3):
            assert gpu[k][y0:y1].tobytes() == cpu[k][y0:y1].tobytes(), (y0, k)
    p.accumulate_mode = 1
    g.set_params(p)
    parts = []
    for rank in range(2):
        g.clear()
        g.dispatch(1 + rank, 4, 2)
        parts.append(g.read_image(0).astype(np.float64))
    g.clear()
    g.dispatch(1, 8, 1)
    union = g.read_image(0).astype(np.float64)
    assert np.all(union[..., 3] == 8.0)
    assert np.all(np.abs(parts[0] + parts[1] - union) <= 1e-5 * np.maximum(1.0, np.abs(union)))


def test_twenty_million_triangles_uhd():
    """The largest benchmark configuration at full size: 139 bunnies flattened on a grid (20 M triangles,
    a 3 GB scene that no cache holds), tree built on the GPU, 3840 x 2160.  Two frames in one dispatch
    equal two one-frame dispatches bit for bit, and three 4-row bands equal the oracle traversing the
    downloaded tree."""
    from oracle import binding as ob
    meshes, mats, side = scenes.instanced_bunnies(139)
    soup112 = R.align_primitives(np.concatenate([R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)]))
    assert soup112.shape[0] > 20_000_000
    del meshes
    width, height = 3840, 2160
    g = R.Renderer(width, height)
    g.build_bvh_device(soup112)
    del soup112
    g.upload_materials(mats)
    hdri, cache = environment(256, 128)
    g.upload_environment(hdri, cache)
    p = R.default_params(width, height)
    p.max_bounces = 8
    R.set_camera(p, (-2.5 - 1.1 * side, 1.0 + 0.8 * side, 0.0), (0.0, 0.0, 0.0), width, height)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 2, 1, counted=True)
    st = g.stats()
    whole = [g.read_image(k) for k in range(3)]
    assert st.samples == 2 * width * height and 0 < st.max_stack_depth <= 64
    print(f"20 M triangles: max traversal-stack occupancy {st.max_stack_depth}, "
          f"{st.node_visits / (st.rays_closest + st.rays_any):.1f} node visits per ray")
    g.clear()
    g.dispatch(1, 1, 1)
    g.dispatch(2, 1, 1)
    for k in range(3):
        assert g.read_image(k).tobytes() == whole[k].tobytes()
    nodes, prims = g.download_bvh()
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    for y0, y1 in ((300, 304), (1078, 1082), (1900, 1904)):
        cpu = o.render(p, width, height, 1, 2, 1, rows=(y0, y1))
        for k in range(3):
            bad = _diff_rows(whole[k][y0:y1], cpu[k][y0:y1], cpu[3][y0:y1])
            assert bad == 0, (y0, k, bad)


def _diff_rows(a, b, overflow_mask):
    """Differing pixels of two row bands, not counting pixels where the oracle's 16-entry traversal stack
    overflowed (the shader's behaviour there is undefined; the kernel's stack cannot overflow)."""
    ne = np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1)
    return int((ne & (overflow_mask == 0)).sum())
