"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar: BIT-EXACT float32 images.  Both sides evaluate the same IEEE binary32 operation sequences
(no FMA contraction, correctly rounded divide/sqrt, the same polynomial transcendentals), so the
tolerance written into these tests is zero ulp per pixel; a small budget of differing pixels
(MAX_DIFF_FRACTION) is allowed only for traversal pruning near exact distance ties and is
reported when used.
"""
import ctypes

import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import lib as L
from pathtracer_amd import renderer as R
from pathtracer_amd import scenes
from tests.common import built_scene, environment

pytestmark = pytest.mark.gpu

MAX_DIFF_FRACTION = 1e-4
# both schedulers drive the same device code and must give the same bits
PIPELINES = pytest.mark.parametrize("pipeline", [0, 1], ids=["megakernel", "wavefront"])


def _pair(recipe, width, height, env=(256, 128), textures=None, pipeline=0):
    nodes, prims, mats = built_scene(recipe) if isinstance(recipe, str) else recipe
    hdri, cache = environment(*env)
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    g = R.Renderer(width, height)
    g.set_tuning(pipeline=pipeline)
    g.upload_nodes(nodes)
    g.upload_primitives(prims)
    g.upload_materials(mats)
    g.upload_environment(hdri, cache)
    for slot, (texels, srgb) in (textures or {}).items():
        o.set_texture(slot, texels, srgb)
        g.upload_texture(slot, texels, srgb)
    return o, g


def _diff_pixels(a, b, mask=None):
    ne = np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1)
    both_nan = np.all((a != a) == (b != b), axis=-1) & np.any(a != a, axis=-1)
    ne &= ~(both_nan & np.all((a.view(np.uint32) == b.view(np.uint32)) | ((a != a) & (b != b)), axis=-1))
    if mask is not None:
        ne &= mask == 0
    return ne


def _assert_images_equal(gpu_imgs, cpu_imgs, mask, what):
    total = gpu_imgs[0].shape[0] * gpu_imgs[0].shape[1]
    for name, a, b in zip(("radiance", "albedo", "normal"), gpu_imgs, cpu_imgs):
        bad = _diff_pixels(a, b, mask)
        n = int(bad.sum())
        if n:
            ys, xs = np.nonzero(bad)
            print(f"{what}/{name}: {n} of {total} pixels differ, first at (x={xs[0]}, y={ys[0]}): "
                  f"gpu={a[ys[0], xs[0]]} cpu={b[ys[0], xs[0]]}")
        assert n <= MAX_DIFF_FRACTION * total, f"{what}/{name}: {n} of {total} pixels differ"


@pytest.mark.parametrize("fn,name", list(enumerate(["sin", "cos", "asin", "atan2", "pow", "min", "max", "div", "sqrt", "rsqrt"])))
def test_math_building_blocks_bit_exact(fn, name):
    rng = np.random.default_rng(100 + fn)
    n = 1 << 20
    if name in ("sin", "cos"):
        a = rng.uniform(-7.0, 7.0, n).astype(np.float32)
        b = np.zeros(n, np.float32)
    elif name == "asin":
        a = rng.uniform(-1.0, 1.0, n).astype(np.float32)
        a[:8] = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.0000001, 1e-5]
        b = np.zeros(n, np.float32)
    elif name == "atan2":
        a = rng.uniform(-2.0, 2.0, n).astype(np.float32)
        b = rng.uniform(-2.0, 2.0, n).astype(np.float32)
        a[:6] = [0.0, 1.0, -1.0, 0.0, 0.5, -0.0]
        b[:6] = [0.0, 0.0, 0.0, -1.0, 0.0, 1.0]
    elif name == "pow":
        a = rng.uniform(0.0, 1.0, n).astype(np.float32)
        b = rng.uniform(0.05, 30.0, n).astype(np.float32)
        a[:4] = [0.0, 1.0, 1e-40, 0.64]
        b[:4] = [0.5, 0.5, 0.5, 0.5]
    elif name in ("min", "max"):
        a = rng.uniform(-2.0, 2.0, n).astype(np.float32)
        b = rng.uniform(-2.0, 2.0, n).astype(np.float32)
        special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0], np.float32)
        a[:36] = np.repeat(special, 6)
        b[:36] = np.tile(special, 6)
    elif name == "div":
        a = rng.uniform(-1e3, 1e3, n).astype(np.float32)
        b = rng.uniform(-1e3, 1e3, n).astype(np.float32)
        a[:5] = [1.0, 1.0, 0.0, 1e-38, 3.0]
        b[:5] = [0.0, -0.0, 0.0, 1e3, 1e-39]
    else:
        a = np.abs(rng.normal(0, 10, n)).astype(np.float32)
        a[:4] = [0.0, 1e-40, 1.0, 3.4e38]
        b = np.zeros(n, np.float32)
    gpu = R.eval_math(fn, a, b)
    cpu = ob.eval_math(fn, a, b)
    same = (gpu.view(np.uint32) == cpu.view(np.uint32)) | ((gpu != gpu) & (cpu != cpu))
    bad = np.nonzero(~same)[0]
    assert bad.size == 0, f"{name}: {bad.size} mismatches, first a={a[bad[0]]!r} b={b[bad[0]]!r} gpu={gpu[bad[0]]!r} cpu={cpu[bad[0]]!r}"


@PIPELINES
def test_small_scene_all_lobes_bit_exact(pipeline):
    o, g = _pair("TorusMixed", 96, 80, pipeline=pipeline)
    p = R.default_params(96, 80)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 4, 1, counted=True)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 96, 80, 1, 4, 1)
    assert cpu[4].stack_overflows == 0
    _assert_images_equal(gpu, cpu[:3], cpu[3], "TorusMixed 96x80 4spp")
    st = g.stats()
    assert st.samples == 96 * 80 * 4
    assert (st.rays_closest, st.closest_hits) == (cpu[4].rays_closest, cpu[4].closest_hits)
    # shadow queries whose contribution is exactly zero are not issued
    assert 0 < st.rays_any <= cpu[4].rays_any
    assert st.node_visits <= cpu[4].pair_tests and st.triangle_tests <= cpu[4].triangle_tests
    if pipeline == 0:  # the megakernel's counting build also sorts the closest-hit queries by path depth
        depth = g.depth_histogram()
        assert sum(depth) == st.rays_closest and depth[0] == st.samples
        assert all(a >= b for a, b in zip(depth, depth[1:])) and sum(depth[p.max_bounces:]) == 0
    assert np.isfinite(gpu[0]).all() and gpu[0][..., :3].mean() > 0.01


@PIPELINES
@pytest.mark.parametrize("recipe", ["BunnyClearcoat", "BunnyGlass", "BunnyDielectric", "BunnyMetallic"])
def test_bunny_recipes_one_spp_bit_exact(recipe, pipeline):
    o, g = _pair(recipe, 256, 256, pipeline=pipeline)
    p = R.default_params(256, 256)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 1, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 256, 256, 1, 1, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], f"{recipe} 256x256 1spp")


@PIPELINES
def test_c1_bunny_clearcoat_512_first_and_later_frames(pipeline):
    # the CPU-runnable reference case: BunnyClearcoat 512x512, fixed seed pixelIndex + FrameCount*719393
    o, g = _pair("BunnyClearcoat", 512, 512, env=(1024, 512), pipeline=pipeline)
    p = R.default_params(512, 512)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 1, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 512, 512, 1, 1, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "C1 frame 1")
    # frames 2..4 accumulate with the running mean; one launch for three FrameCount values
    g.dispatch(2, 3, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu2 = o.render(p, 512, 512, 2, 3, 1, images=[x.copy() for x in cpu[:3]])
    _assert_images_equal(gpu, cpu2[:3], cpu[3] | cpu2[3], "C1 frames 1-4")


@PIPELINES
def test_deep_paths_eight_bounces(pipeline):
    o, g = _pair("BunnyGlass", 192, 160, pipeline=pipeline)
    p = R.default_params(192, 160)
    p.max_bounces = 8
    g.set_params(p)
    g.clear()
    g.dispatch(3, 2, 5)  # FrameCount 3 and 8: strided like a sample-sharded rank
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 192, 160, 3, 2, 5)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "BunnyGlass 8 bounces")


@PIPELINES
def test_thin_lens_no_antialiasing_and_modifiers(pipeline):
    o, g = _pair("TorusMixed", 64, 64, pipeline=pipeline)
    p = R.default_params(64, 64)
    p.aperture_size = 0.08
    p.aperture_shape[0], p.aperture_shape[1] = 1.0, 0.5
    p.focal_length = 2.5
    p.anti_aliasing_enabled = 0
    p.roughness_modifier = -0.2
    p.metallic_modifier = 0.3
    p.refraction_modifier = 0.4
    p.transmission_modifier = 0.2
    g.set_params(p)
    g.clear()
    g.dispatch(1, 2, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 64, 64, 1, 2, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "thin lens")


@PIPELINES
def test_ragged_image_size_renders_every_row(pipeline):
    # 100 x 70 is not a multiple of the 16x16 tile; the reference would drop rows 64..69
    o, g = _pair("TorusMixed", 100, 70, pipeline=pipeline)
    p = R.default_params(100, 70)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 1, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 100, 70, 1, 1, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "100x70")
    assert np.all(gpu[0][..., 3] == 1.0)


@PIPELINES
def test_multi_frame_launch_equals_frame_by_frame(pipeline):
    _, g = _pair("TorusMixed", 64, 48, pipeline=pipeline)
    p = R.default_params(64, 48)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 5, 1)
    one = [g.read_image(k) for k in range(3)]
    g.clear()
    for f in range(1, 6):
        g.dispatch(f, 1, 1)
    seq = [g.read_image(k) for k in range(3)]
    for a, b in zip(one, seq):
        assert a.tobytes() == b.tobytes()


def test_launch_splitting_does_not_change_a_bit():
    """The megakernel's sample buffer has a budget: a dispatch that exceeds it runs as several
    launches.  An expensive tile's samples are also spread over several blocks.  Neither may change the
    images, and the oracle's frame-by-frame running mean is the reference for all of them."""
    o, g = _pair("TorusMixed", 96, 80)
    p = R.default_params(96, 80)
    g.set_params(p)
    cpu = o.render(p, 96, 80, 3, 11, 2)  # FrameCount 3, 5, ..., 23
    frame_bytes_mib = 1  # 96*80*36 B = 0.26 MiB per frame: a 1 MiB budget holds 3 frames per launch
    for tuning in (dict(), dict(sample_budget_mib=frame_bytes_mib), dict(tile_split=16),
                   dict(sample_budget_mib=frame_bytes_mib, tile_split=0), dict(tile_order=0)):
        g.set_tuning(**dict(dict(sample_budget_mib=8192, tile_split=-1, tile_order=1), **tuning))
        g.clear()
        g.dispatch(3, 11, 2)
        for k in range(3):
            assert g.read_image(k).tobytes() == cpu[k].tobytes(), (tuning, k)


def test_uhd_launch_equals_frame_by_frame_and_oracle_bands():
    """3840 x 2160 (the largest benchmark configuration's frame): one 6-frame dispatch equals six
    one-frame dispatches bit for bit, and bands of rows equal the oracle."""
    o, g = _pair("BunnyClearcoat", 3840, 2160)
    p = R.default_params(3840, 2160)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 6, 1)
    whole = [g.read_image(k) for k in range(3)]
    g.clear()
    for f in range(1, 7):
        g.dispatch(f, 1, 1)
    for k in range(3):
        assert g.read_image(k).tobytes() == whole[k].tobytes()
    g.clear()
    g.dispatch(1, 2, 1)
    gpu = [g.read_image(k) for k in range(3)]
    for y0, y1 in ((0, 4), (1078, 1082), (2156, 2160)):
        cpu = o.render(p, 3840, 2160, 1, 2, 1, rows=(y0, y1))
        for k in range(3):
            assert gpu[k][y0:y1].tobytes() == cpu[k][y0:y1].tobytes(), (y0, k)


def test_full_hd_launch_properties():
    """At the benchmark's image size the oracle is too slow to render every pixel, so the full-size
    launch is checked through properties: one 12-frame dispatch equals 12 one-frame dispatches bit for
    bit, sum mode over disjoint FrameCount sets adds up to the sum over their union (the multi-GPU
    sharding identity, exact up to float summation order), and three bands of rows equal the oracle."""
    o, g = _pair("BunnyGlass", 1920, 1080)
    p = R.default_params(1920, 1080)
    p.max_bounces = 8
    g.set_params(p)
    g.clear()
    g.dispatch(1, 12, 1)
    whole = [g.read_image(k) for k in range(3)]
    g.clear()
    for f in range(1, 13):
        g.dispatch(f, 1, 1)
    for k in range(3):
        assert g.read_image(k).tobytes() == whole[k].tobytes()
    # three 8-row bands against the oracle (FrameCount 1..3 keeps the CPU side to seconds)
    g.clear()
    g.dispatch(1, 3, 1)
    gpu = [g.read_image(k) for k in range(3)]
    for y0, y1 in ((0, 8), (536, 544), (1072, 1080)):
        cpu = o.render(p, 1920, 1080, 1, 3, 1, rows=(y0, y1))
        for k in range(3):
            assert gpu[k][y0:y1].tobytes() == cpu[k][y0:y1].tobytes(), (y0, k)
    # sharding identity in sum mode
    p.accumulate_mode = 1
    g.set_params(p)
    parts = []
    for rank in range(2):
        g.clear()
        g.dispatch(1 + rank, 6, 2)
        parts.append(g.read_image(0).astype(np.float64))
    g.clear()
    g.dispatch(1, 12, 1)
    union = g.read_image(0).astype(np.float64)
    assert np.all(union[..., 3] == 12.0)
    err = np.abs(parts[0] + parts[1] - union)
    assert np.all(err <= 1e-5 * np.maximum(1.0, np.abs(union)))


@PIPELINES
def test_sum_mode_and_scale(pipeline):
    o, g = _pair("TorusMixed", 64, 48, pipeline=pipeline)
    p = R.default_params(64, 48)
    p.accumulate_mode = 1
    g.set_params(p)
    g.clear()
    g.dispatch(1, 4, 1)
    total = g.read_image(0)
    cpu = o.render(p, 64, 48, 1, 4, 1)
    assert total.tobytes() == cpu[0].tobytes()
    assert np.all(total[..., 3] == 4.0)
    g.scale_images(0.25)
    assert np.array_equal(g.read_image(0), total * np.float32(0.25))


def _textured_quad_scene():
    # two triangles with uvs running beyond [0,1] (REPEAT), lit from above, all 11 texture slots bound
    v = np.array([[-1.5, 0.2, -1.5, 0, 1, 0, -0.5, -0.5], [1.5, 0.2, -1.5, 0, 1, 0, 2.5, -0.5],
                  [1.5, 0.6, 1.5, 0, 1, 0, 2.5, 2.5], [-1.5, 0.6, 1.5, 0, 1, 0, -0.5, 2.5]], np.float32)
    idx = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    rng = np.random.default_rng(5)
    tex = {}
    slots = {}
    for k, name in enumerate(scenes.TEXTURE_SLOTS):
        slot = k + 1
        ch = 4 if name in ("specular", "sheenRoughness") else 3
        if name == "emission":
            texels = rng.uniform(0, 0.2, (8, 8, 3)).astype(np.float32)  # float texture
        else:
            texels = rng.integers(64, 256, (7 + k, 5 + k, ch), dtype=np.uint8)
        tex[slot] = (texels, name == "albedo")
        slots[name] = slot
    mat = scenes.material(albedo=(0.9, 0.8, 0.7), emission=(1.0, 1.0, 1.0), roughness=0.9, metallic=0.8,
                          sheenRoughness=0.9, sheenTint=0.5, clearcoat=0.9, clearcoatRoughness=0.6,
                          transmission=0.5, specularTint=0.5, textures=slots)
    prims = R.assemble_primitives(v, idx, 0)
    fv, fi = scenes.load_ptm("floor")
    prims = np.concatenate([prims, R.assemble_primitives(fv, fi, 1)])
    sp, nodes = R.build_bvh_sah(prims, 4)
    mats = np.array([mat, scenes.floor_material()], dtype=scenes.MATERIAL_DTYPE)
    return (R.align_nodes(nodes), R.align_primitives(sp), mats), tex


@PIPELINES
@pytest.mark.parametrize("decode_mib", [None, 0], ids=["texels-decoded-at-upload", "texels-8-bit"])
def test_textured_materials_bit_exact(pipeline, decode_mib):
    """All eleven slots bound, 8-bit (sRGB and linear, three and four channels) and float texels; the library keeps 8-bit
    textures as decoded floats up to a byte budget (tuning `texture_decode_mib`) and as 8-bit words beyond it: both
    forms must give the oracle's bits."""
    scene, tex = _textured_quad_scene()
    o, g = _pair(scene, 80, 80, textures=tex, pipeline=pipeline)
    if decode_mib is not None:
        g.set_tuning(texture_decode_mib=decode_mib)
    p = R.default_params(80, 80)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 3, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 80, 80, 1, 3, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "textured quad")
    assert np.isfinite(gpu[0]).all()
    # the other storage form, switched on the same context, renders the same bits again
    g.set_tuning(texture_decode_mib=64 if decode_mib == 0 else 0)
    g.clear()
    g.dispatch(1, 3, 1)
    for k in range(3):
        assert g.read_image(k).tobytes() == gpu[k].tobytes()


@pytest.mark.parametrize("decode_mib", [64, 0], ids=["texels-decoded-at-upload", "texels-8-bit"])
def test_textures_of_one_and_two_channels_read_as_r001_and_rg01(decode_mib):
    """A sampler returns (r, 0, 0, 1) for a one-channel texture and (r, g, 0, 1) for a two-channel one: the metallic-
    roughness slot reads .x and .y, the specular slot .w (alpha: 1 without an alpha channel), the albedo slot rgb."""
    scene, tex = _textured_quad_scene()
    rng = np.random.default_rng(11)
    names = list(scenes.TEXTURE_SLOTS)
    for name, ch in (("metallicRoughness", 2), ("specular", 1), ("albedo", 1), ("clearcoatRoughness", 2)):
        slot = names.index(name) + 1
        tex[slot] = (rng.integers(32, 256, (9, 6, ch), dtype=np.uint8), name == "albedo")
    tex[names.index("transmission") + 1] = (rng.uniform(0.1, 0.9, (5, 4, 1)).astype(np.float32), False)
    o, g = _pair(scene, 64, 64, textures=tex)
    g.set_tuning(texture_decode_mib=decode_mib)
    p = R.default_params(64, 64)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 3, 1)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 64, 64, 1, 3, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "one- and two-channel textures")


def test_process_buffers_path_matches_manual_upload():
    meshes, mats = scenes.flatten(scenes.recipe("TorusMixed"))
    hdri, cache = environment(256, 128)
    g = R.Renderer(64, 64)
    for (v, i), m in zip(meshes, mats):
        g.AddPathTracingModel(v, i.astype(np.uint16), m)
    g.ProcessBuffers()
    g.upload_environment(hdri)  # cache built by the library this time
    p = R.default_params(64, 64)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 2, 1)
    a = g.read_image(0)
    _, g2 = _pair("TorusMixed", 64, 64)
    g2.set_params(p)
    g2.clear()
    g2.dispatch(1, 2, 1)
    assert a.tobytes() == g2.read_image(0).tobytes()


def test_bind_images_renders_into_caller_memory():
    import torch
    _, g = _pair("TorusMixed", 64, 48)
    p = R.default_params(64, 48)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 2, 1)
    want = g.read_image(0)
    bufs = [torch.zeros(48, 64, 4, device="cuda", dtype=torch.float32) for _ in range(3)]
    torch.cuda.synchronize()
    g.bind_images(*[b.data_ptr() for b in bufs])
    g.dispatch(1, 2, 1)
    g.synchronize()
    assert np.array_equal(bufs[0].cpu().numpy(), want)
    g.bind_images(0, 0, 0)
    with pytest.raises(L.PtError):
        g.bind_images(bufs[0].data_ptr(), 0, 0)


def test_read_image_rgb_is_the_rgba_image_without_alpha():
    # 37 x 29 = 1073 texels: not a multiple of four, so the packed copy's tail path runs too
    for width, height in ((37, 29), (64, 32)):
        o, g = _pair("TorusMixed", width, height)
        g.set_params(R.default_params(width, height))
        g.clear()
        g.dispatch(1, 2, 1)
        for which in range(3):
            rgba = g.read_image(which)
            rgb = g.read_image_rgb(which)
            assert rgb.shape == (height, width, 3)
            assert np.array_equal(rgb.view(np.uint32), rgba[..., :3].view(np.uint32).reshape(height, width, 3))


def test_read_display_is_the_display_pass_of_the_radiance_image():
    """pt_read_display (tone mapping + sRGB encoding on the device) against the oracle's restatement of
    tonemapping.frag:36-71 applied to the same radiance image.  Tolerance: 0 of 255 -- both sides run the
    same float32 operation sequence, pow included (it is built from pinned polynomials on both sides)."""
    for width, height in ((37, 29), (128, 64)):  # 1073 texels: the kernel's tail path runs too
        o, g = _pair("TorusMixed", width, height)
        g.set_params(R.default_params(width, height))
        g.clear()
        g.dispatch(1, 4, 1)
        hdr = g.read_image(0)
        for exposure in (1.0, 0.2, 6.0):
            got = g.read_display(exposure)
            assert got.shape == (height, width, 3) and got.dtype == np.uint8
            assert np.array_equal(got, ob.display(hdr, exposure))
        assert 0 < g.read_display(1.0).mean() < 255
    # synthetic radiance through pt_write_image: zeros, huge values, tiny values, a NaN-free ramp
    g = R.Renderer(64, 16)
    ramp = np.zeros((16, 64, 4), np.float32)
    ramp[..., 0] = np.geomspace(1e-6, 1e4, 64, dtype=np.float32)[None, :]
    ramp[..., 1] = np.linspace(0.0, 2.0, 16, dtype=np.float32)[:, None]
    ramp[..., 2] = 1e30
    ramp[0, 0] = 0.0
    g.write_image(0, ramp)
    assert np.array_equal(g.read_display(1.0), ob.display(ramp, 1.0))


def test_debug_views_are_the_shaders_debug_output():
    """pt_debug_view against the oracle's restatement of Library/debug.glsl, bit for bit: the BVH heat map
    (on an uploaded tree, on a device-built tree traced by the oracle from its download, and on a chain
    tree deep enough to overflow the shader's 16-entry stack: purple) and the HDRI-cache view."""
    w, h = 160, 96
    o, g = _pair("BunnyClearcoat", w, h)
    g.set_bvh_rebuild(False)  # the view shows the tree in use: here the uploaded one, which the oracle walks too
    p = R.default_params(w, h)
    g.set_params(p)
    heat = g.debug_view(1)
    assert heat.tobytes() == o.debug_view(p, w, h, 1).tobytes()
    assert len(np.unique(heat.reshape(-1, 4), axis=0)) > 20  # a picture, not a constant
    assert np.array_equal(heat[..., 3], np.ones((h, w), np.float32))
    cache = g.debug_view(2)
    assert cache.tobytes() == o.debug_view(p, w, h, 2).tobytes()
    assert np.any(np.all(cache[..., :3] == np.array([1, 0, 0], np.float32), axis=-1))  # red dots
    # a frame dispatched afterwards starts from the debug image only if the caller does not clear: clear works
    g.clear()
    g.dispatch(1, 1, 1)
    cpu = o.render(p, w, h, 1, 1, 1)
    _assert_images_equal([g.read_image(k) for k in range(3)], cpu[:3], cpu[3], "frame after a debug view")
    # device-built tree: the oracle walks the downloaded reference-layout tree
    nodes, prims, mats = built_scene("BunnyClearcoat")
    g.build_bvh_device(prims)
    dn, dp = g.download_bvh()
    o.set_scene(dn, dp, mats)
    assert g.debug_view(1).tobytes() == o.debug_view(p, w, h, 1).tobytes()
    # chain tree, 40 levels: the far child of every level waits on the stack
    nodes40, cprims = _chain_tree(40)
    cmats = np.array([scenes.material(albedo=(0.7, 0.6, 0.5), roughness=0.6)], dtype=scenes.MATERIAL_DTYPE)
    o.set_scene(R.align_nodes(nodes40), R.align_primitives(cprims), cmats)
    g.upload_nodes(R.align_nodes(nodes40))
    g.upload_primitives(R.align_primitives(cprims))
    g.upload_materials(cmats)
    deep = g.debug_view(1)
    assert deep.tobytes() == o.debug_view(p, w, h, 1).tobytes()
    assert np.any(np.all(deep == np.array([1, 0, 1, 1], np.float32), axis=-1))  # overflow: purple


def test_error_behaviour():
    g = R.Renderer(32, 32)
    with pytest.raises(L.PtError) as e:
        g.dispatch(1)
    assert e.value.code == -2
    nodes, prims, mats = built_scene("TorusMixed")
    hdri, cache = environment(256, 128)
    g.upload_primitives(prims)
    g.upload_materials(mats)
    g.upload_environment(hdri, cache)
    g.set_params(R.default_params(32, 32))
    bad = nodes.copy()
    bad.view(np.int32)[1, 0] = nodes.shape[0] + 5  # root's left child out of range
    g.upload_nodes(bad)
    with pytest.raises(L.PtError) as e:
        g.dispatch(1)
    assert e.value.code == -5 and "child index" in e.value.message
    cyc = nodes.copy()
    cyc.view(np.int32)[2, 0] = 1  # a cycle back to the root
    g.upload_nodes(cyc)
    with pytest.raises(L.PtError) as e:
        g.dispatch(1)
    assert e.value.code == -5
    g.upload_nodes(nodes)
    g.upload_materials(mats[:1])  # floor's meshIndex 1 now has no material
    with pytest.raises(L.PtError) as e:
        g.dispatch(1)
    assert e.value.code == -5 and "meshIndex" in e.value.message
    g.upload_materials(mats)
    with pytest.raises(L.PtError):
        g.dispatch(0)  # FrameCount 0 would divide by zero
    g.dispatch(1)
    g.synchronize()
    pp = R.default_params(64, 64)
    with pytest.raises(L.PtError):
        g.set_params(pp)  # FrameDimensions mismatch


def test_checkpoint_resumes_bit_identically(tmp_path):
    """Images saved after FrameCount 1..4 and restored into a fresh context, then FrameCount 5..8:
    the same bits as FrameCount 1..8 in one go."""
    _, g = _pair("TorusMixed", 72, 40)
    p = R.default_params(72, 40)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 8, 1)
    whole = [g.read_image(k) for k in range(3)]
    g.clear()
    g.dispatch(1, 4, 1)
    g.save_checkpoint(tmp_path / "half", 5)
    _, h = _pair("TorusMixed", 72, 40)
    h.set_params(p)
    nxt = h.load_checkpoint(tmp_path / "half")
    assert nxt == 5
    h.dispatch(nxt, 4, 1)
    for k in range(3):
        assert h.read_image(k).tobytes() == whole[k].tobytes()
    with pytest.raises(ValueError):
        R.Renderer(64, 40).load_checkpoint(tmp_path / "half")


def _chain_tree(levels):
    """levels + 1 big triangles facing -x at x = 0, 0.1, ..., each in a leaf of its own, under a tree
    that is one long chain: interior node j holds the farthest remaining triangle as its left leaf
    and everything nearer below its right child.  A ray travelling +x descends to the right and
    postpones a leaf at every level, so its traversal stack grows to `levels` entries."""
    verts, idx = [], []
    for k in range(levels + 1):
        x = 0.1 * k
        verts += [[x, -3, -3, -1, 0, 0, 0, 0], [x, -3, 3, -1, 0, 0, 0, 1], [x, 3, -3, -1, 0, 0, 1, 0]]
        idx += [3 * k, 3 * k + 1, 3 * k + 2]
    prims = R.assemble_primitives(np.array(verts, np.float32), np.array(idx, np.uint32), 0)
    nodes = np.zeros((2 * levels + 2, 10), np.int32)
    boxes = nodes.view(np.float32)

    def leaf(node, tri):
        nodes[node, 2], nodes[node, 3] = 1, tri
        boxes[node, 4:7] = (0.1 * tri, -3, -3)
        boxes[node, 7:10] = (0.1 * tri, 3, 3)

    for j in range(levels):
        me = 1 + 2 * j
        nodes[me, 0] = me + 1                                       # left: leaf of triangle levels - j
        nodes[me, 1] = me + 2 if j + 1 < levels else 1 + 2 * levels  # right: the rest, or the last leaf
        boxes[me, 4:7] = (0.0, -3, -3)
        boxes[me, 7:10] = (np.float32(0.1 * (levels - j)), 3, 3)
        leaf(me + 1, levels - j)
    leaf(1 + 2 * levels, 0)
    return nodes.view(np.float32), prims


def _caterpillar_tree(levels):
    """A path of `levels` interior nodes whose every node has, beside the next path node, a sibling sub-tree
    (two interior nodes over four one-triangle leaves) with larger boxes than the path's.  Folding by surface
    area opens the sibling and its two children at every step and never the path node, so the four-wide
    tree is as deep as the path: deeper than the traversal stack allows, which makes the library fold the
    tree again with both children of every node opened."""
    verts, idx = [], []

    def tri(x):
        k = len(idx) // 3
        verts.extend([[x, -1, -1, -1, 0, 0, 0, 0], [x, -1, 1, -1, 0, 0, 0, 1], [x, 1, -1, -1, 0, 0, 1, 0]])
        idx.extend([3 * k, 3 * k + 1, 3 * k + 2])
        return k

    count = 8 * levels + 2
    nodes = np.zeros((count, 10), np.int32)
    boxes = nodes.view(np.float32)

    def box(node, lo, hi):
        boxes[node, 4:7] = lo
        boxes[node, 7:10] = hi

    def leaf(node, x):
        nodes[node, 2], nodes[node, 3] = 1, tri(x)
        box(node, (x, -1, -1), (x, 1, 1))

    for j in range(levels):
        me = 1 + 8 * j
        sib = me + 1
        nodes[me, 0], nodes[me, 1] = sib, me + 8  # left: the sibling sub-tree, right: the next path node (or the last leaf)
        box(me, (-20, -20, -20), (20, 20, 20))
        box(sib, (-19, -19, -19), (19, 19, 19))
        nodes[sib, 0], nodes[sib, 1] = sib + 1, sib + 4
        for half in (sib + 1, sib + 4):
            box(half, (-18, -18, -18), (18, 18, 18))
            nodes[half, 0], nodes[half, 1] = half + 1, half + 2
            leaf(half + 1, 0.01 * (4 * j + (0 if half == sib + 1 else 2)))
            leaf(half + 2, 0.01 * (4 * j + (1 if half == sib + 1 else 3)))
    # path boxes: smaller than every sibling box (which is what keeps them closed), still around all triangles
    for j in range(1, levels):
        box(1 + 8 * j, (-2, -2, -2), (0.01 * 4 * levels + 2, 2, 2))
    leaf(1 + 8 * levels, 0.01 * 4 * levels)
    prims = R.assemble_primitives(np.array(verts, np.float32), np.array(idx, np.uint32), 0)
    return nodes.view(np.float32), prims


def test_tree_whose_fold_by_area_is_too_deep_is_folded_level_by_level():
    levels = 60
    nodes40, prims = _caterpillar_tree(levels)
    mats = np.array([scenes.material(albedo=(0.7, 0.6, 0.5), roughness=0.6)], dtype=scenes.MATERIAL_DTYPE)
    hdri, cache = environment(256, 128)
    sorted_prims, sah_nodes = R.build_bvh_sah(prims.copy(), 4)
    images = {}
    for name, (n40, p100) in {"caterpillar": (nodes40, prims), "sah": (sah_nodes, sorted_prims)}.items():
        g = R.Renderer(96, 64)
        g.set_bvh_rebuild(False)  # this test is about the uploaded tree itself
        g.upload_nodes(R.align_nodes(n40))
        g.upload_primitives(R.align_primitives(p100))
        g.upload_materials(mats)
        g.upload_environment(hdri, cache)
        g.set_params(R.default_params(96, 64))
        g.clear()
        g.dispatch(1, 3, 1, counted=True)
        images[name] = [g.read_image(k) for k in range(3)]
        levels_wide, level_by_level = g.tree_info()
        if name == "caterpillar":
            assert level_by_level and 16 < levels_wide <= 32, (levels_wide, level_by_level)
            # (how deep the stack gets depends on the order in which overlapping boxes are visited; the chain tree of
            # test_traversal_stack_beyond_32_entries is the one that drives it deep by construction)
            assert 0 < g.stats().max_stack_depth <= 96
        else:
            assert not level_by_level and levels_wide <= 16
    for k in range(3):
        assert images["caterpillar"][k].tobytes() == images["sah"][k].tobytes()
    assert np.any(images["caterpillar"][0][..., :3] > 0)


def test_four_wave_kernel_and_scheduler_thresholds_give_the_same_image():
    """The 4-waves-per-SIMD build of the megakernel (A/B variant), one block per work item instead of waves
    that pull the items, and other phase thresholds change the schedule, not a bit of the result."""
    o, g = _pair("TorusMixed", 120, 72)
    p = R.default_params(120, 72)
    p.max_bounces = 6
    images = {}
    variants = {"default": {}, "four-waves": dict(waves_per_simd=4), "thresholds": dict(trav_exit=8, hit_min=48),
                "eager": dict(trav_exit=64, hit_min=1), "block-per-item": dict(pull_items=0),
                "four-waves-block-per-item": dict(waves_per_simd=4, pull_items=0), "unsplit": dict(tile_split=0),
                "fine-split": dict(tile_split=64)}
    for name, tuning in variants.items():
        g.set_tuning(waves_per_simd=0, trav_exit=32, hit_min=16, pull_items=1, tile_split=-1)
        g.set_tuning(**tuning)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 2, 1)  # (the second dispatch runs from the plan made of the first one's tile costs)
        g.dispatch(3, 3, 1)
        images[name] = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, 120, 72, 1, 5, 1)
    _assert_images_equal(images["default"], cpu[:3], cpu[3], "TorusMixed 120x72 5spp 6 bounces")
    for name in variants:
        for k in range(3):
            assert images[name][k].tobytes() == images["default"][k].tobytes(), (name, k)


@pytest.mark.parametrize("tuning", [dict(), dict(plan_every=1, overlap_share=0), dict(plan_every=3, overlap_share=3),
                                    dict(plan_every=1024, overlap_share=64), dict(plan_every=8, overlap_share=2, tile_split=4)],
                         ids=["defaults", "plan-every-call-all-slots", "every-third-third-of-slots", "one-plan-few-waves", "split-tiles"])
def test_short_calls_reusing_a_plan_on_a_share_of_the_waves(tuning):
    """Calls of one or two frames reuse their launch lane's item list for some launches and start a share of the
    resident waves each: neither may move a bit.  40 frames in one- and two-frame calls, with a change of frames per
    call (a list made for one frame per call must not serve a two-frame call), against the oracle's 40 frames."""
    w, h = 136, 88
    o, g = _pair("TorusMixed", w, h)
    p = R.default_params(w, h)
    p.max_bounces = 4
    g.set_params(p)
    g.set_tuning(**tuning)
    g.clear()
    f = 1
    for n in [1] * 14 + [2] * 7 + [1] * 6 + [2] * 3:
        g.dispatch(f, n, 1)
        f += n
    assert f == 41
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, w, h, 1, 40, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], "40 frames in calls of one and two")
    assert g.stats().samples == w * h * 2  # the last call's


def test_overlapping_calls_keep_frame_order_and_the_last_calls_counters():
    """Dispatch calls of few frames trace on streams of their own, two or three calls in flight at a time,
    and only their resolve passes are ordered.  A run of one-frame calls, interleaved with the calls that
    have to wait for them (a read, a clear, a scale, new parameters, the static dispatch, a long call),
    must leave the bits the oracle computes frame by frame; the counters are those of the last call alone;
    on a caller's stream nothing overlaps and the bits are the same."""
    import torch
    w, h = 104, 72
    o, g = _pair("TorusMixed", w, h)
    p = R.default_params(w, h)
    p.max_bounces = 5
    g.set_params(p)
    cpu = o.render(p, w, h, 1, 24, 1)
    runs = {}
    for name, tuning in {"never": dict(overlap_calls=0), "few-frames": dict(overlap_calls=-1), "always": dict(overlap_calls=1)}.items():
        g.set_tuning(**tuning)
        g.clear()
        for f in range(1, 13):  # one frame per call: three calls in flight
            g.dispatch(f, 1, 1)
        assert g.stats().samples == w * h  # the last call's, not the sum of those still running
        mid = g.read_image(0)  # waits for all twelve
        g.dispatch(13, 3, 1)  # two calls in flight
        g.dispatch(16, 2, 1)
        g.dispatch_static(18, 1, 1)  # on the context's stream, behind every resolve pass
        assert g.stats().samples == w * h
        g.dispatch(19, 6, 1)
        assert g.stats().samples == w * h * 6
        runs[name] = [g.read_image(k) for k in range(3)] + [mid]
    _assert_images_equal(runs["never"][:3], cpu[:3], cpu[3], "24 frames in calls of 1, 3, 2, 1 and 6")
    for name in ("few-frames", "always"):
        for k in range(4):
            assert runs[name][k].tobytes() == runs["never"][k].tobytes(), (name, k)
    # work between the calls that the next trace must see (parameters) or that must see the traces (clear, scale)
    g.set_tuning(overlap_calls=1)
    q = R.default_params(w, h)
    q.max_bounces = 2
    want = []
    for setting in (0, 1):
        g.set_tuning(overlap_calls=setting)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 1, 1)
        g.dispatch(2, 1, 1)
        g.clear()
        g.set_params(q)
        g.dispatch(1, 1, 1)
        g.dispatch(2, 1, 1)
        g.scale_images(0.5)
        g.dispatch(3, 1, 1)
        want.append([g.read_image(k) for k in range(3)])
    for k in range(3):
        assert want[0][k].tobytes() == want[1][k].tobytes(), k
    # a caller's stream: everything on that one stream
    stream = torch.cuda.Stream()
    g.set_stream(stream.cuda_stream)
    g.set_params(p)
    g.clear()
    for f in range(1, 13):
        g.dispatch(f, 1, 1)
    stream.synchronize()
    assert g.read_image(0).tobytes() == runs["never"][3].tobytes()
    g.set_stream(0)


@PIPELINES
def test_traversal_stack_beyond_32_entries(pipeline):
    """The megakernel keeps 48 stack entries in LDS and, for trees whose four-wide form is more than 16 levels
    deep, the rest in private memory.  A chain tree of 60 levels (20 four-wide ones, three leaves postponed at
    each) drives the stack beyond the LDS part; the image must equal the one traced through the sweep-SAH
    tree of the same triangles, with one block per work item and with waves that pull the items."""
    levels = 60
    nodes40, prims = _chain_tree(levels)
    mats = np.array([scenes.material(albedo=(0.7, 0.6, 0.5), roughness=0.6)], dtype=scenes.MATERIAL_DTYPE)
    hdri, cache = environment(256, 128)
    images = {}
    sorted_prims, sah_nodes = R.build_bvh_sah(prims, 4)
    for name, (n40, p100, tuning) in {
        "chain": (nodes40, prims, dict(pull_items=0)),
        "chain-pull": (nodes40, prims, dict(pull_items=1)),
        "sah": (sah_nodes, sorted_prims, dict()),
    }.items():
        g = R.Renderer(96, 64)
        g.set_bvh_rebuild(False)  # the uploaded chain is what drives the stack deep
        g.set_tuning(pipeline=pipeline, **tuning)
        g.upload_nodes(R.align_nodes(n40))
        g.upload_primitives(R.align_primitives(p100))
        g.upload_materials(mats)
        g.upload_environment(hdri, cache)
        p = R.default_params(96, 64)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 3, 1, counted=True)
        images[name] = [g.read_image(k) for k in range(3)]
        if name.startswith("chain"):
            assert 48 < g.stats().max_stack_depth <= levels, g.stats().max_stack_depth
    for k in range(3):
        assert images["chain"][k].tobytes() == images["chain-pull"][k].tobytes()
        assert images["chain"][k].tobytes() == images["sah"][k].tobytes()
    assert np.any(images["chain"][0][..., :3] > 0)


def _transformed_scene(recipe, scale, offset):
    """The recipe's meshes scaled and moved (positions only), the reference builder's tree over them."""
    from tests.common import built_meshes
    meshes, mats = scenes.flatten(scenes.recipe(recipe))
    moved = []
    for v, i in meshes:
        v = v.copy()
        v[:, 0:3] = v[:, 0:3] * np.float32(scale) + np.asarray(offset, np.float32)
        moved.append((v, i))
    return built_meshes(moved, mats)


@pytest.mark.parametrize("rebuild", [True, False], ids=["library-tree", "uploaded-tree"])
@pytest.mark.parametrize("scale,offset", [(1.0, (1000.0, 0.0, -2000.0)), (0.1, (0.0, 0.0, 0.0)), (500.0, (3.0e4, 0.0, 0.0))],
                         ids=["far-from-origin", "tenth-scale", "large-and-far"])
def test_grid_boxes_hold_far_from_the_origin_and_at_other_scales(scale, offset, rebuild):
    """The traversal's boxes are grid planes around the exact ones, and its slab distances are multiply-adds in
    the node's grid; both lose precision where coordinates are large against the node (a scene far from the
    origin) and where cells get tiny.  The images must still be the oracle's, bit for bit: a box test that
    wrongly missed would drop a hit."""
    scene = _transformed_scene("TorusMixed", scale, offset)
    w, h = 128, 96
    o, g = _pair(scene, w, h)
    g.set_bvh_rebuild(rebuild)
    p = R.default_params(w, h)
    p.max_bounces = 4
    eye = tuple(np.float32(c) * np.float32(scale) + np.float32(d) for c, d in zip((-2.5, 1.0, 0.3), offset))
    R.set_camera(p, eye, tuple(float(x) for x in offset), w, h)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 3, 1, counted=True)
    gpu = [g.read_image(k) for k in range(3)]
    cpu = o.render(p, w, h, 1, 3, 1)
    _assert_images_equal(gpu, cpu[:3], cpu[3], f"TorusMixed x{scale} at {offset}")
    st = g.stats()
    assert st.closest_hits == cpu[4].closest_hits and st.closest_hits > 0.2 * w * h


def test_axis_parallel_rays_through_grid_boxes():
    """Rays with direction components of exactly zero (an axis-aligned camera, no anti-aliasing, odd image size:
    the centre row and column) take the clamped-reciprocal path of the box test, where the shader has
    infinities; boxes of an axis-aligned cube and floor have zero extent on an axis at some nodes."""
    from tests.common import built_meshes
    cube_v, cube_i = scenes.load_ptm("cube")
    floor_v, floor_i = scenes.load_ptm("floor")
    mats = np.array([scenes.material(albedo=(0.8, 0.3, 0.2), roughness=0.4, metallic=0.3), scenes.floor_material()],
                    dtype=scenes.MATERIAL_DTYPE)
    scene = built_meshes([(cube_v, cube_i), (floor_v, floor_i)], mats)
    w, h = 65, 49
    for rebuild in (True, False):
        o, g = _pair(scene, w, h)
        g.set_bvh_rebuild(rebuild)
        p = R.default_params(w, h)
        p.anti_aliasing_enabled = 0
        R.set_camera(p, (0.0, 0.5, 6.0), (0.0, 0.5, 0.0), w, h)  # looking straight down -z
        g.set_params(p)
        g.clear()
        g.dispatch(1, 2, 1)
        gpu = [g.read_image(k) for k in range(3)]
        cpu = o.render(p, w, h, 1, 2, 1)
        _assert_images_equal(gpu, cpu[:3], cpu[3], f"axis-aligned camera, rebuild={rebuild}")
        assert np.any(gpu[1][h // 2, w // 2, :3] > 0)  # the centre ray hits the cube
