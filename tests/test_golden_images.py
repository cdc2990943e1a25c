"""The committed image fixtures of the CPU-runnable reference case (tests/golden/images/c1.json, written by
tests/golden/make_image_fixtures.py in the build container): BunnyClearcoat 512 x 512, FrameCount 1 and 1..64.

CPU test: the oracle, rebuilt and run on this host, must reproduce the hashes and the crops bit for bit (an
oracle that drifts between hosts or compilers shows here, not as a GPU mismatch).  GPU test: the HIP images
against the same numbers, without loading the oracle library at all.  What the fixtures pin is the oracle's
output; the reference itself holds no rendered vectors for this path (shading parity stays unpinned)."""
import json
import os

import numpy as np
import pytest

from pathtracer_amd import renderer as R
from tests.common import built_scene, environment, image_hash

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "images", "c1.json")))
CROPS = np.load(os.path.join(HERE, "golden", "images", "c1_crops.npz"))
NAMES = ("radiance", "albedo", "normal")
W = H = 512


def _inputs():
    import ctypes
    nodes, prims, mats = built_scene("BunnyClearcoat")
    hdri, cache = environment(1024, 512)
    p = R.default_params(W, H)
    raw = np.frombuffer(ctypes.string_at(ctypes.addressof(p), ctypes.sizeof(p)), np.uint8)
    got = {"nodes": image_hash(nodes), "primitives": image_hash(prims), "materials": image_hash(np.ascontiguousarray(mats).view(np.uint8)),
           "hdri": image_hash(hdri), "cache": image_hash(cache), "params": image_hash(raw)}
    # a mismatch here means the inputs (host builder, sky formula, numpy) moved, not the renderer under test
    assert got == GOLD["inputs"], f"fixture inputs differ on this host: {got} != {GOLD['inputs']}"
    return nodes, prims, mats, hdri, cache, p


def _check(images, key, tag):
    for name, img in zip(NAMES, images):
        for cname, (x0, y0) in GOLD["crops"].items():
            want = CROPS[f"{tag}_{name}_{cname}"]
            got = img[y0:y0 + 32, x0:x0 + 32]
            bad = int(np.any(got.view(np.uint32) != want.view(np.uint32), axis=-1).sum())
            assert bad == 0, f"{key}/{name}: {bad} of 1024 pixels of crop '{cname}' differ from the fixture"
        assert image_hash(img) == GOLD[key][name]["hash"], f"{key}/{name}: image hash differs from the fixture"


def test_image_hash_known_answers():
    # (offset basis 1469598103934665603, the one the BVH fixtures' hashes were taken with; with the standard basis
    # 14695981039346656037 the same loop gives the published FNV-1a-64("a") = af63dc4c8601ec8c)
    def fnv(data, h=1469598103934665603):
        for b in data:
            h = ((h ^ b) * 1099511628211) & ((1 << 64) - 1)
        return h
    assert fnv(b"a", 14695981039346656037) == 0xAF63DC4C8601EC8C
    row = fnv(b"a")
    assert image_hash(np.frombuffer(b"a", np.uint8)) == f"{fnv(row.to_bytes(8, 'little')):016x}"
    two = np.arange(24, dtype=np.uint8).reshape(2, 12)
    rows = b"".join(fnv(bytes(r)).to_bytes(8, "little") for r in two)
    assert image_hash(two) == f"{fnv(rows):016x}"
    assert image_hash(two.view(np.float32)) == image_hash(two)  # bytes, not values


def test_fixture_crops_show_the_scene():
    assert GOLD["stack_overflows"] == 0
    for key in CROPS.files:
        assert np.isfinite(CROPS[key]).all()
    assert CROPS["f64_radiance_bunny"][..., :3].std() > 0.01 and CROPS["f64_albedo_bunny"][..., 2].mean() > 0.3


def test_oracle_reproduces_the_committed_c1_images():
    from oracle import binding as ob
    nodes, prims, mats, hdri, cache, p = _inputs()
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    one = o.render(p, W, H, 1, 1, 1, threads=os.cpu_count() or 1)
    _check(one[:3], "frame_1", "f1")
    assert (int(one[4].rays_closest), int(one[4].rays_any), int(one[4].closest_hits)) == \
        (GOLD["rays_frame_1"]["closest"], GOLD["rays_frame_1"]["shadow"], GOLD["rays_frame_1"]["hits"])
    full = o.render(p, W, H, 2, 63, 1, images=[x.copy() for x in one[:3]], threads=os.cpu_count() or 1)
    _check(full[:3], "frames_1_to_64", "f64")


@pytest.mark.gpu
def test_hip_images_match_the_committed_c1_images():
    nodes, prims, mats, hdri, cache, p = _inputs()
    g = R.Renderer(W, H)
    g.upload_nodes(nodes)
    g.upload_primitives(prims)
    g.upload_materials(mats)
    g.upload_environment(hdri, cache)
    g.set_params(p)
    g.clear()
    g.dispatch(1, 1, 1, counted=True)
    _check([g.read_image(k) for k in range(3)], "frame_1", "f1")
    st = g.stats()
    assert (st.rays_closest, st.closest_hits) == (GOLD["rays_frame_1"]["closest"], GOLD["rays_frame_1"]["hits"])
    assert 0 < st.rays_any <= GOLD["rays_frame_1"]["shadow"]  # shadow queries of exactly zero weight are not issued
    g.dispatch(2, 63, 1)
    _check([g.read_image(k) for k in range(3)], "frames_1_to_64", "f64")
