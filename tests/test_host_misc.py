"""CPU-only checks: HDRI cache, camera matrices, C-ABI surface and error behaviour, oracle sanity."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import lib as L
from pathtracer_amd import renderer as R
from pathtracer_amd import scenes
from tests.common import built_scene, environment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    declared -= {"pt_ctx"}
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)


def test_params_layout_matches_oracle():
    assert ctypes.sizeof(L.Params) == ob.oracle_lib().oracle_sizeof_params() == 288


def test_hdri_cache_matches_oracle_bitwise():
    for (w, h) in [(64, 32), (256, 128), (37, 19)]:
        hdri = scenes.sky_hdri(w, h)
        ours = R.calculate_hdri_cache(hdri)
        ref = ob.calculate_hdri_cache(hdri)
        assert ours.tobytes() == ref.tobytes()
        # R = x / W, G = y / H are grid positions, B sums to 1 over the image
        assert np.all(ours[..., 0] <= 1.0) and np.all(ours[..., 1] <= 1.0)
        assert abs(float(ours[..., 2].astype(np.float64).sum()) - 1.0) < 1e-3


@pytest.mark.skipif(ob.ref_hdri_lib() is None, reason="oracle/_ref/libref_hdri.so not built (reference absent)")
def test_hdri_cache_matches_the_reference_code_bitwise():
    """pt_calculate_hdri_cache and the oracle's restatement against the reference's own statements
    (PathTracingRenderer.cpp:803-901 compiled in place): every float of the cache, bit for bit."""
    rng = np.random.default_rng(7)
    cases = [scenes.sky_hdri(w, h) for (w, h) in [(64, 32), (256, 128), (37, 19), (1024, 512)]]
    cases.append(rng.uniform(0.0, 4.0, (48, 96, 3)).astype(np.float32))          # noise
    spiky = rng.uniform(0.0, 0.01, (33, 70, 3)).astype(np.float32)               # a few very bright texels
    spiky[5, 7] = 5000.0
    spiky[20, 60] = (900.0, 0.0, 30.0)
    cases.append(spiky)
    cases.append(np.full((16, 16, 3), 0.25, np.float32))                          # constant: every CDF step equal
    for hdri in cases:
        want = ob.ref_calculate_hdri_cache(hdri)
        assert R.calculate_hdri_cache(hdri).tobytes() == want.tobytes(), hdri.shape
        assert ob.calculate_hdri_cache(hdri).tobytes() == want.tobytes(), hdri.shape


def test_hdri_cache_follows_the_sun():
    hdri = scenes.sky_hdri(128, 64)
    cache = R.calculate_hdri_cache(hdri)
    # most table cells should point at bright texels: mean luminance at the sampled positions
    # must exceed the image's mean luminance by a wide margin
    x = np.minimum((cache[..., 0] * 128).astype(int), 127)
    y = np.minimum((cache[..., 1] * 64).astype(int), 63)
    lum = hdri @ np.array([0.212671, 0.715160, 0.072169], np.float32)
    assert lum[y, x].mean() > 2.0 * lum.mean()


def test_black_hdri_gives_nan_pdf_like_reference():
    # Content/HDRI/Black.hdr is 1x1 black: lumSum = 0 -> 0/0 (PathTracingRenderer.cpp:803-825)
    cache = R.calculate_hdri_cache(np.zeros((1, 1, 3), np.float32))
    assert np.isnan(cache[0, 0, 2])


def _f16(a):
    return (ctypes.c_float * 16)(*a)


@pytest.mark.skipif(ob.ref_lib() is None, reason="oracle/_ref not built (reference absent)")
def test_camera_matrices_match_glm_bitwise():
    lib, ref = L.load(), ob.ref_lib()
    f3 = lambda *v: (ctypes.c_float * 3)(*v)
    rng = np.random.default_rng(3)
    cases = [((-2.5, 1.0, 0.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))]
    cases += [tuple(tuple(rng.uniform(-5, 5, 3)) for _ in range(3)) for _ in range(20)]
    for eye, center, up in cases:
        a, b = (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
        lib.pt_look_at(f3(*eye), f3(*center), f3(*up), a)
        ref.ref_glm_look_at(f3(*eye), f3(*center), f3(*up), b)
        assert bytes(a) == bytes(b)
        ia, ib = (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
        lib.pt_inverse4(a, ia)
        ref.ref_glm_inverse(b, ib)
        assert bytes(ia) == bytes(ib)
    for fovy, aspect in [(1.57, 1.0), (1.57, 1920 / 1080), (0.6, 2.0)]:
        a, b = (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
        lib.pt_perspective(fovy, aspect, 0.01, 1000.0, a)
        ref.ref_glm_perspective(fovy, aspect, 0.01, 1000.0, b)
        assert bytes(a) == bytes(b)
        ia, ib = (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
        lib.pt_inverse4(a, ia)
        ref.ref_glm_inverse(b, ib)
        assert bytes(ia) == bytes(ib)


def test_default_params_are_the_reference_defaults():
    p = R.default_params(1024, 1024)
    assert p.anti_aliasing_enabled == 1 and p.max_bounces == 3 and p.accumulate_mode == 0
    assert p.focal_length == np.float32(3.5) and p.aperture_size == 0.0
    assert tuple(p.aperture_shape) == (1.0, 1.0)
    view = np.array(p.view_matrix, np.float32).reshape(4, 4).T  # column-major -> rows
    eye = -view[:3, :3].T @ view[:3, 3]
    assert np.allclose(eye, (-2.5, 1.0, 0.0), atol=1e-6)
    proj = np.array(p.proj_matrix, np.float32).reshape(4, 4).T
    assert abs(proj[1, 1] - 1.0 / np.tan(1.57 / 2)) < 1e-6 and proj[3, 2] == -1.0


def test_host_entry_points_reject_bad_arguments():
    lib = L.load()
    cnt = ctypes.c_size_t()
    buf = np.zeros((4, 25), np.float32)
    nodes = np.zeros((2, 10), np.float32)
    rc = lib.pt_build_bvh_sah(buf.ctypes.data, 4, 0, nodes.ctypes.data, 2, ctypes.byref(cnt))
    assert rc == -1 and b"leaf_size" in lib.pt_last_error()
    rc = lib.pt_build_bvh_sah(buf.ctypes.data, 4, 4, nodes.ctypes.data, 1, ctypes.byref(cnt))
    assert rc == -1 and b"node_capacity" in lib.pt_last_error()
    rc = lib.pt_build_bvh_sah(None, 4, 4, nodes.ctypes.data, 2, ctypes.byref(cnt))
    assert rc == -1
    v = np.zeros((3, 8), np.float32)
    bad = np.array([0, 1, 5], np.uint16)
    out = np.zeros((1, 25), np.float32)
    rc = lib.pt_assemble_primitives(v.ctypes.data, 3, bad.ctypes.data, 3, 2, 0, out.ctypes.data)
    assert rc == -1 and b"exceeds" in lib.pt_last_error()
    rc = lib.pt_assemble_primitives(v.ctypes.data, 3, bad.ctypes.data, 3, 3, 0, out.ctypes.data)
    assert rc == -1 and b"EBO Type not supported" in lib.pt_last_error()


def test_index_widths_give_identical_soup():
    v, i = scenes.load_ptm("cube")
    a = R.assemble_primitives(v, i.astype(np.uint8), 3)
    b = R.assemble_primitives(v, i.astype(np.uint16), 3)
    c = R.assemble_primitives(v, i.astype(np.uint32), 3)
    assert a.tobytes() == b.tobytes() == c.tobytes()
    assert np.all(a[:, 24].view(np.uint32) == 3)
    t0 = i[:3]
    assert np.array_equal(a[0, 0:3], v[t0[0], 0:3]) and np.array_equal(a[0, 3:6], v[t0[1], 0:3])
    assert np.array_equal(a[0, 9:12], v[t0[0], 3:6]) and np.array_equal(a[0, 18:20], v[t0[0], 6:8])


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(L.PtError) as e:
        R.Renderer(64, 64)
    assert e.value.code == -3 and "no CPU rendering path" in e.value.message


# ------------------------------------------------------------------------------- oracle sanity
def _oracle_scene(name, env=(64, 32)):
    nodes, prims, mats = built_scene(name)
    hdri, cache = environment(*env)
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    return o


def test_oracle_renders_plausible_image_and_is_thread_invariant():
    o = _oracle_scene("TorusMixed")
    p = R.default_params(48, 48)
    a = o.render(p, 48, 48, 1, 2, 1, threads=1)
    b = o.render(p, 48, 48, 1, 2, 1, threads=4)
    for x, y in zip(a[:3], b[:3]):
        assert x.tobytes() == y.tobytes()
    rad = a[0]
    assert np.isfinite(rad).all() and rad[..., :3].mean() > 0.01 and np.all(rad[..., 3] == 1.0)
    ctr = a[4]
    assert ctr.samples == 48 * 48 * 2 and ctr.rays_closest >= ctr.samples
    assert ctr.stack_overflows == 0 and ctr.max_stack <= 16
    # sky above the horizon is brighter than the floor far away: top rows vs. bottom rows
    assert rad[-4:, :, :3].mean() > 0.1


def test_oracle_debug_views_known_answers():
    """Values that follow from Library/debug.glsl by hand (the reference holds no images of these views:
    beyond these the views are parity-unpinned).  A tree whose root is a leaf: every ray counts exactly one
    leaf (:46-49, no box test on the root), l = 1/100 * 5, mix(black, blue, 0.05).  A real tree: the count
    over the heat map's ramp is monotone (black -> blue -> cyan -> green), never purple with 16 entries to
    spare, and zero where the ray misses the root's children.  The HDRI view: every texel is the environment's
    colour or pure red."""
    o = _oracle_scene("TorusMixed")
    p = R.default_params(40, 24)
    # root leaf
    nodes = np.zeros((2, 12), np.float32)
    ni = nodes.view(np.int32)
    ni[1, 2], ni[1, 3] = 1, 0
    _, prims, mats = built_scene("TorusMixed")
    leaf = ob.Oracle()
    leaf.set_scene(nodes, prims[:1], mats)
    img = leaf.debug_view(p, 40, 24, 1)
    want = np.float32(0.0) * (np.float32(1.0) - np.float32(0.05)) + np.float32(1.0) * np.float32(np.float32(np.float32(1.0) / np.float32(100.0)) * np.float32(5.0))
    assert np.all(img[..., 2] == want) and np.all(img[..., :2] == 0) and np.all(img[..., 3] == 1)
    # a real tree
    heat = o.debug_view(p, 40, 24, 1, threads=1)
    assert heat.tobytes() == o.debug_view(p, 40, 24, 1, threads=4).tobytes()
    rgb = heat[..., :3].reshape(-1, 3)
    assert not np.any(np.all(rgb == np.array([1, 0, 1], np.float32), axis=1))
    assert np.any(np.all(rgb == 0, axis=1)) and np.any(rgb[:, 2] > 0)
    assert np.all((rgb[:, 0] == 0) | (rgb[:, 2] == 0))  # red only past green: the ramp never mixes red with blue
    view = o.debug_view(p, 40, 24, 2)
    red = np.all(view[..., :3] == np.array([1, 0, 0], np.float32), axis=-1)
    assert 0 < red.sum() < red.size
    assert np.isfinite(view).all() and np.all(view[..., 3] == 1)


def test_oracle_frame_sequence_equals_single_call():
    o = _oracle_scene("TorusMixed")
    p = R.default_params(32, 32)
    one = o.render(p, 32, 32, 1, 3, 1)
    imgs = None
    for f in (1, 2, 3):
        r = o.render(p, 32, 32, f, 1, 1, images=imgs)
        imgs = list(r[:3])
    for x, y in zip(one[:3], imgs):
        assert x.tobytes() == y.tobytes()


def test_oracle_running_mean_vs_sum_mode():
    o = _oracle_scene("TorusMixed")
    p = R.default_params(32, 32)
    mean = o.render(p, 32, 32, 1, 4, 1)[0]
    p.accumulate_mode = 1
    total = o.render(p, 32, 32, 1, 4, 1)[0]
    assert np.all(total[..., 3] == 4.0)
    assert np.allclose(total[..., :3] / 4.0, mean[..., :3], rtol=2e-5, atol=1e-6)


def test_oracle_math_close_to_libm():
    rng = np.random.default_rng(0)
    x = rng.uniform(-7, 7, 200000).astype(np.float32)
    ulp = lambda got, want: np.abs(got.astype(np.float64) - want) / np.maximum(np.spacing(np.abs(want).astype(np.float32)), 1e-45)
    assert np.max(np.abs(ob.eval_math(0, x).astype(np.float64) - np.sin(x.astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(ob.eval_math(1, x).astype(np.float64) - np.cos(x.astype(np.float64)))) < 2.5e-7
    u = rng.uniform(-1, 1, 200000).astype(np.float32)
    assert ulp(ob.eval_math(2, u), np.arcsin(u.astype(np.float64))).max() < 4
    y = rng.uniform(-3, 3, 200000).astype(np.float32)
    assert np.max(np.abs(ob.eval_math(3, y, x).astype(np.float64) - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 6e-7
    b = rng.uniform(0, 1, 200000).astype(np.float32)
    e = rng.uniform(0.2, 12, 200000).astype(np.float32)
    got = ob.eval_math(4, b, e).astype(np.float64)
    want = np.power(b.astype(np.float64), e.astype(np.float64))
    assert np.max(np.abs(got - want) / np.maximum(want, 1e-30)) < 1e-5
    assert ob.eval_math(4, np.array([0.0], np.float32), np.array([0.5], np.float32))[0] == 0.0
    assert ob.eval_math(4, np.array([0.25], np.float32), np.array([0.5], np.float32))[0] == np.float32(0.5)


def test_bvh_cache_round_trip_and_rejections(tmp_path):
    from pathtracer_amd.lib import PtError
    from tests.common import soup
    from pathtracer_amd import scenes
    prims100, _ = soup(scenes.recipe("TorusMixed"))
    key = R.bvh_cache_key(prims100)
    assert key == R.bvh_cache_key(prims100.copy()) and key != R.bvh_cache_key(prims100[::-1].copy())
    sorted100, nodes40 = R.build_bvh_sah(prims100, 4)
    nodes48, prims112 = R.align_nodes(nodes40), R.align_primitives(sorted100)
    path = tmp_path / "torus.ptbvh"
    R.bvh_cache_save(path, key, nodes48, prims112)
    n2, p2 = R.bvh_cache_load(path, key)
    assert np.array_equal(n2.view(np.uint32), np.ascontiguousarray(nodes48).view(np.uint32).reshape(n2.shape))
    assert np.array_equal(p2.view(np.uint32), np.ascontiguousarray(prims112).view(np.uint32).reshape(p2.shape))
    with pytest.raises(PtError, match="stale"):
        R.bvh_cache_load(path, key ^ 1)
    with pytest.raises(PtError, match="no cache file"):
        R.bvh_cache_load(tmp_path / "absent.ptbvh", key)
    blob = path.read_bytes()
    (tmp_path / "cut.ptbvh").write_bytes(blob[:-100])
    with pytest.raises(PtError, match="truncated"):
        R.bvh_cache_load(tmp_path / "cut.ptbvh", key)
    flipped = bytearray(blob)
    flipped[len(blob) // 2] ^= 0x40
    (tmp_path / "flip.ptbvh").write_bytes(bytes(flipped))
    with pytest.raises(PtError, match="corrupt"):
        R.bvh_cache_load(tmp_path / "flip.ptbvh", key)
    (tmp_path / "junk.ptbvh").write_bytes(b"not a cache")
    with pytest.raises(PtError, match="not a BVH cache"):
        R.bvh_cache_load(tmp_path / "junk.ptbvh", key)
