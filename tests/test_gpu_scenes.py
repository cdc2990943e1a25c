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
