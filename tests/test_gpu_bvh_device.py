"""The GPU BVH builder (Morton-code linear BVH): structural validity of the tree it returns in the
reference's layouts, quality against the sweep-SAH builder, and image parity -- the oracle
traverses the downloaded tree, so GPU and oracle images must again agree bit for bit."""
import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import renderer as R
from pathtracer_amd import scenes
from pathtracer_amd.lib import PtError
from tests.common import check_tree as _check_tree, environment, sah_cost as _sah_cost, soup
from tests.test_gpu_parity import _assert_images_equal

pytestmark = pytest.mark.gpu


def _render_pair(nodes, prims, mats, width, height, bounces, device_built):
    hdri, cache = environment(256, 128)
    o = ob.Oracle()
    o.set_scene(nodes, prims, mats)
    o.set_environment(hdri, cache)
    p = R.default_params(width, height)
    p.max_bounces = bounces
    device_built.upload_materials(mats)
    device_built.upload_environment(hdri, cache)
    device_built.set_params(p)
    device_built.clear()
    device_built.dispatch(1, 2, 1)
    gpu = [device_built.read_image(k) for k in range(3)]
    cpu = o.render(p, width, height, 1, 2, 1)
    return gpu, cpu


# the topologies of the device builder: locally-ordered clustering followed by reinsertion passes (the default), the
# clustered tree as built, and the Morton radix tree
BUILDERS = pytest.mark.parametrize("radius,passes", [(32, -1), (32, 64), (32, 0), (0, 0)],
                                   ids=["clustering+reinsertion", "clustering+64-passes", "clustering", "radix-tree"])


@BUILDERS
@pytest.mark.parametrize("recipe", ["TorusMixed", "BunnyGlass"])
def test_device_built_tree_is_valid_and_renders_like_the_oracle(recipe, radius, passes):
    prims100, mats = soup(scenes.recipe(recipe))
    source = R.align_primitives(prims100)
    width, height = 160, 128
    g = R.Renderer(width, height)
    g.set_tuning(ploc_radius=radius, bvh_optimise=passes, bvh_optimise_mod=1 if passes > 16 else 2)
    g.build_bvh_device(source)
    nodes, prims = g.download_bvh()
    depth, interior = _check_tree(nodes, prims, source)
    # quality against the sweep-SAH tree of the same soup
    _, nodes40 = R.build_bvh_sah(prims100, 4)
    ratio = _sah_cost(nodes) / _sah_cost(R.align_nodes(nodes40))
    print(f"{recipe}: {prims.shape[0]} triangles, {nodes.shape[0]} nodes, depth {depth}, SAH cost {ratio:.2f}x the sweep builder's")
    assert ratio < (1.0 if radius else 2.0)  # the clustered tree beats the reference's capped sweep
    gpu, cpu = _render_pair(nodes, prims, mats, width, height, 8 if recipe == "BunnyGlass" else 3, g)
    # the oracle aborts a query when the reference shader's 16-entry stack overflows and masks the
    # pixel; deeper linear-BVH trees make that more likely than with the SAH tree
    assert cpu[3].mean() < 0.02
    _assert_images_equal(gpu, cpu[:3], cpu[3], f"{recipe} on the device-built BVH")
    # the same tree through the host packer (pt_upload_nodes / pt_upload_primitives): same images
    h = R.Renderer(width, height)
    h.set_bvh_rebuild(False)  # walk the uploaded tree itself
    h.upload_nodes(nodes)
    h.upload_primitives(prims)
    gpu2, _ = _render_pair(nodes, prims, mats, width, height, 8 if recipe == "BunnyGlass" else 3, h)
    for a, b in zip(gpu, gpu2):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@BUILDERS
@pytest.mark.parametrize("count", [1, 2, 4, 5, 9, 200])
def test_tiny_and_degenerate_inputs(count, radius, passes):
    # `count` copies of ONE triangle: all Morton codes equal, the builder must split by position
    v = np.array([[0, 0, 0, 0, 0, 1, 0, 0], [1, 0, 0, 0, 0, 1, 1, 0], [0, 1, 0, 0, 0, 1, 0, 1]], np.float32)
    idx = np.tile(np.array([0, 1, 2], np.uint32), count)
    source = R.align_primitives(R.assemble_primitives(v, idx, 0))
    g = R.Renderer(32, 32)
    g.set_tuning(ploc_radius=radius, bvh_optimise=passes, bvh_optimise_mod=1 if passes > 16 else 2)
    g.build_bvh_device(source)
    nodes, prims = g.download_bvh()
    assert prims.shape[0] == count
    _check_tree(nodes, prims, source)
    mats = np.array([scenes.material()], dtype=scenes.MATERIAL_DTYPE)
    gpu, cpu = _render_pair(nodes, prims, mats, 32, 32, 3, g)
    _assert_images_equal(gpu, cpu[:3], cpu[3], f"{count} coincident triangles")


def test_process_buffers_device_and_errors():
    meshes, mats = scenes.flatten(scenes.recipe("TorusMixed"))
    g = R.Renderer(64, 64)
    with pytest.raises(PtError):
        g.download_bvh()  # nothing built on the device yet
    for (v, i), m in zip(meshes, mats):
        g.AddPathTracingModel(v, i, m)
    g.ProcessBuffersDevice()
    nodes, prims = g.download_bvh()
    assert prims.shape[0] == sum(i.size // 3 for _, i in meshes)
    hdri, cache = environment(256, 128)
    g.upload_environment(hdri, cache)
    g.set_params(R.default_params(64, 64))
    g.clear()
    g.dispatch(1, 1, 1)
    assert np.isfinite(g.read_image(0)).all()
    # fewer materials than the soup's mesh indices need: rejected at dispatch, nothing launched
    g.upload_materials(mats[:1])
    with pytest.raises(PtError, match="meshIndex") as e:
        g.dispatch(2, 1, 1)
    assert e.value.code == -5


def _morton_chain_soup():
    """Tiny triangles whose centroids quantise to Morton codes with one bit set each (63 of them), plus the
    all-ones corner and a pile at the origin: the Morton radix tree over them is a chain more than 64 levels
    deep, which the device builder refuses (PT_ERR_SCENE)."""
    pts = [(2097151.0, 2097151.0, 2097151.0)] + [(0.0, 0.0, 0.0)] * 40
    for j in range(21):
        c = float(2 ** j) + 0.5
        pts += [(c, 0.0, 0.0), (0.0, c, 0.0), (0.0, 0.0, c)]
    e = 0.01
    v, idx = [], []
    for k, (x, y, z) in enumerate(pts):
        v += [[x, y, z, 0, 0, 1, 0, 0], [x + e, y, z, 0, 0, 1, 1, 0], [x, y + e, z, 0, 0, 1, 0, 1]]
        idx += [3 * k, 3 * k + 1, 3 * k + 2]
    return np.array(v, np.float32), np.array(idx, np.uint32)


def test_failed_device_build_leaves_no_stale_geometry():
    """A device build that fails after it has started writing must not leave the previous tree in place
    over new triangles: the next dispatch reports an error until a build succeeds."""
    meshes, mats = scenes.flatten(scenes.recipe("TorusMixed"))
    hdri, cache = environment(256, 128)
    g = R.Renderer(64, 64)
    for (v, i), m in zip(meshes, mats):
        g.AddPathTracingModel(v, i, m)
    g.ProcessBuffersDevice()
    g.upload_environment(hdri, cache)
    g.set_params(R.default_params(64, 64))
    g.clear()
    g.dispatch(1, 1, 1)  # a good scene is live
    cv, ci = _morton_chain_soup()
    chain = R.align_primitives(R.assemble_primitives(cv, ci, 0))
    g.set_tuning(ploc_radius=0)  # the Morton radix tree: its depth follows the key bits
    with pytest.raises(PtError, match="levels deep") as e:
        g.build_bvh_device(chain)
    assert e.value.code == -5
    with pytest.raises(PtError) as e:  # no geometry now: neither the old tree nor a half-built one
        g.dispatch(2, 1, 1)
    assert e.value.code == -2
    with pytest.raises(PtError):
        g.download_bvh()
    # the rebuild-on-upload path: the uploaded tree is valid, the device build over its primitives fails;
    # every dispatch says so, and the uploaded buffers survive for a later, successful build
    sorted_prims, nodes40 = R.build_bvh_sah(R.assemble_primitives(cv, ci, 0), 4)
    g.set_bvh_rebuild(True)
    g.upload_nodes(R.align_nodes(nodes40))
    g.upload_primitives(R.align_primitives(sorted_prims))
    for frame in (2, 3):
        with pytest.raises(PtError, match="levels deep") as e:
            g.dispatch(frame, 1, 1)
        assert e.value.code == -5
    g.set_tuning(ploc_radius=32)  # the clustering builder copes with the same soup
    g.dispatch(2, 1, 1)
    nodes, prims = g.download_bvh()
    _check_tree(nodes, prims, chain)
    assert np.isfinite(g.read_image(0)).all()


def test_full_sweep_tree_renders_like_the_oracle_and_like_the_reference_tree():
    """The sweep builder without the reference's cost cap gives another tree over the same triangles:
    the GPU image equals the oracle's on that tree bit for bit, and equals the image traced through
    the reference's tree (the closest hit does not depend on the tree unless distances tie exactly)."""
    prims100, mats = soup(scenes.recipe("BunnyGlass"))
    width, height = 160, 128
    images = {}
    for full in (False, True):
        sorted_prims, nodes40 = R.build_bvh_sah(prims100, 4, full_sweep=full)
        nodes, prims = R.align_nodes(nodes40), R.align_primitives(sorted_prims)
        g = R.Renderer(width, height)
        g.set_bvh_rebuild(False)  # the uploaded trees themselves are what is compared
        g.upload_nodes(nodes)
        g.upload_primitives(prims)
        gpu, cpu = _render_pair(nodes, prims, mats, width, height, 8, g)
        _assert_images_equal(gpu, cpu[:3], cpu[3], f"BunnyGlass, full_sweep={full}")
        images[full] = gpu
    for a, b in zip(images[False], images[True]):
        differing = int(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1).sum())
        assert differing <= 1e-4 * width * height, differing


def test_rebuild_on_upload_keeps_the_image_and_cuts_the_visits():
    """pt_set_bvh_rebuild (on unless switched off): the reference's tree is uploaded as usual, the device
    builder's tree over the same primitives is traversed.  Same image (up to exact distance ties), fewer node
    visits per ray."""
    from tests.common import built_scene
    nodes, prims, mats = built_scene("BunnyGlass")
    hdri, cache = environment(256, 128)
    width, height = 160, 128
    p = R.default_params(width, height)
    p.max_bounces = 8
    images, visits = {}, {}
    for rebuild in (False, True, None):  # None: the library's default
        g = R.Renderer(width, height)
        if rebuild is not None:
            g.set_bvh_rebuild(rebuild)
        g.upload_nodes(nodes)
        g.upload_primitives(prims)
        g.upload_materials(mats)
        g.upload_environment(hdri, cache)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 2, 1, counted=True)
        images[rebuild] = [g.read_image(k) for k in range(3)]
        st = g.stats()
        visits[rebuild] = st.node_visits / (st.rays_closest + st.rays_any)
        if rebuild:  # the tree in use can be read back, and new materials do not undo it
            got_nodes, got_prims = g.download_bvh()
            assert got_prims.shape[0] == prims.shape[0] and got_nodes.shape[0] != nodes.shape[0]
            g.upload_materials(mats)
            g.clear()
            g.dispatch(1, 2, 1)
            assert g.read_image(0).tobytes() == images[True][0].tobytes()
    for a, b in zip(images[False], images[True]):
        differing = int(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1).sum())
        assert differing <= 1e-4 * width * height, differing
    assert visits[None] == visits[True] and images[None][0].tobytes() == images[True][0].tobytes()  # the default is to rebuild
    assert visits[True] < 0.8 * visits[False], visits  # (four-wide nodes folded by surface area hide much of the capped sweep's cost)


def test_reinsertion_passes_lower_the_cost_and_the_visits_and_repeat_exactly():
    """The reinsertion passes leave a valid tree of lower surface-area cost that rays visit fewer nodes of, and the same
    tree every time (they take no decision that depends on scheduling)."""
    prims100, mats = soup(scenes.recipe("BunnyGlass"))
    source = R.align_primitives(prims100)
    hdri, cache = environment(256, 128)
    width, height = 160, 128
    p = R.default_params(width, height)
    p.max_bounces = 8
    cost, visits, trees, images = {}, {}, {}, {}
    for passes in (0, 16, 16):
        g = R.Renderer(width, height)
        g.set_tuning(bvh_optimise=passes)
        g.build_bvh_device(source)
        nodes, prims = g.download_bvh()
        _check_tree(nodes, prims, source)
        if passes in trees:
            assert trees[passes][0].tobytes() == nodes.tobytes() and trees[passes][1].tobytes() == prims.tobytes()
        trees[passes] = (nodes, prims)
        cost[passes] = _sah_cost(nodes)
        g.upload_materials(mats)
        g.upload_environment(hdri, cache)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 2, 1, counted=True)
        st = g.stats()
        visits[passes] = st.node_visits / (st.rays_closest + st.rays_any)
        images[passes] = g.read_image(0)
    print(f"cost {cost[0]:.3f} -> {cost[16]:.3f}, node visits per ray {visits[0]:.2f} -> {visits[16]:.2f}")
    assert cost[16] < cost[0] and visits[16] < visits[0]
    differing = int(np.any(images[0].view(np.uint32) != images[16].view(np.uint32), axis=-1).sum())
    assert differing <= 1e-4 * width * height, differing  # (the closest hit does not depend on the tree unless distances tie)


def test_a_tree_the_passes_left_too_deep_is_put_back_as_built():
    """With the height limit of the optimised tree below any real tree's height the builder falls back to the clustered
    tree as it was before the passes: byte for byte the tree of a build without passes."""
    prims100, _ = soup(scenes.recipe("TorusMixed"))
    source = R.align_primitives(prims100)
    trees = []
    for tuning in (dict(bvh_optimise=0), dict(bvh_optimise=16, bvh_optimise_max_height=1)):
        g = R.Renderer(32, 32)
        g.set_tuning(**tuning)
        g.build_bvh_device(source)
        nodes, prims = g.download_bvh()
        _check_tree(nodes, prims, source)
        trees.append((nodes.tobytes(), prims.tobytes()))
    assert trees[0] == trees[1]


@pytest.mark.parametrize("cut", [1, 2, 3])
def test_uploaded_tree_with_its_leaves_cut_renders_the_same(cut):
    """leaf_recut: the caller's tree traversed as uploaded, its leaves of up to four triangles cut into leaves of at most
    `cut` under new interior nodes.  Same image as with the leaves as they are (up to exact distance ties), fewer triangle
    tests per ray."""
    from tests.common import built_scene
    nodes, prims, mats = built_scene("BunnyGlass")
    hdri, cache = environment(256, 128)
    width, height = 160, 128
    p = R.default_params(width, height)
    p.max_bounces = 8
    images, tests_per_ray = {}, {}
    for setting in (0, cut):
        g = R.Renderer(width, height)
        g.set_bvh_rebuild(False)
        g.set_tuning(leaf_recut=setting)
        g.upload_nodes(nodes)
        g.upload_primitives(prims)
        g.upload_materials(mats)
        g.upload_environment(hdri, cache)
        g.set_params(p)
        g.clear()
        g.dispatch(1, 2, 1, counted=True)
        st = g.stats()
        tests_per_ray[setting] = st.triangle_tests / (st.rays_closest + st.rays_any)
        images[setting] = [g.read_image(k) for k in range(3)]
    for a, b in zip(images[0], images[cut]):
        differing = int(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1).sum())
        assert differing <= 1e-4 * width * height, differing
    assert tests_per_ray[cut] < tests_per_ray[0], tests_per_ray


def test_random_soups_and_builder_settings_give_valid_trees():
    """A few seconds of tools/bvh_fuzz.py: random soups (spheres, clumps far apart, coincident triangles, needles, a plane
    under one huge triangle) through the builder with random radius, passes, candidate spacing and leaf size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "bvh_fuzz.py"), "6", "11"], capture_output=True, text=True,
                         timeout=110, cwd=root)
    assert run.returncode == 0 and "all trees valid" in run.stdout, run.stdout[-1500:] + run.stderr[-1500:]
