"""pt_dispatch_static (the entry point that allocates nothing) against pt_dispatch, through the C ABI:
bit-identical images on two calls in a row, the second on a smaller input; sizes beyond the static
scratch array and a scene that was never uploaded to the device are refused."""
import numpy as np
import pytest

from pathtracer_amd import renderer as R
from pathtracer_amd.lib import PtError
from tests.common import built_scene, environment

pytestmark = pytest.mark.gpu



def _free_device_memory():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]  # hipMemGetInfo: free bytes on the device


def _context(width, height, recipe, bounces):
    nodes, prims, mats = built_scene(recipe)
    hdri, cache = environment(256, 128)
    g = R.Renderer(width, height, device=0)
    g.upload_nodes(nodes)
    g.upload_primitives(prims)
    g.upload_materials(mats)
    g.upload_environment(hdri, cache)
    p = R.default_params(width, height)
    p.max_bounces = bounces
    g.set_params(p)
    return g


def _images(g):
    return [g.read_image(k) for k in range(3)]


def _same_bits(a, b):
    return all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


def test_static_dispatch_matches_dispatch_twice_second_smaller():
    # first call: ragged 200 x 120 frame (tiles cut by both image edges), 5 frames, 8 bounces of glass
    big = _context(200, 120, "BunnyGlass", 8)
    big.clear()
    big.dispatch(1, 5, 1)  # also uploads the scene; the static entry point uploads nothing
    want_big = _images(big)
    big.clear()
    big.synchronize()
    free_before = _free_device_memory()
    big.dispatch_static(1, 5, 1)
    big.dispatch_static(6, 1, 1)
    big.synchronize()
    assert _free_device_memory() == free_before  # the entry point's defining property: nothing was allocated
    big.clear()
    big.dispatch_static(1, 5, 1)
    got_big = _images(big)
    assert _same_bits(got_big, want_big)
    assert big.stats().samples == 200 * 120 * 5
    # second call in a row, smaller input (fewer pixels, fewer frames, another scene): the array still
    # holds the first call's records and tile costs
    small = _context(72, 40, "TorusMixed", 3)
    small.clear()
    small.dispatch(3, 2, 2)
    want_small = _images(small)
    small.clear()
    small.dispatch_static(3, 2, 2)
    got_small = _images(small)
    assert _same_bits(got_small, want_small)
    # and the first context again, continuing its accumulation (frames 6..7 on top of 1..5)
    big.dispatch_static(6, 2, 1)
    cont = _images(big)
    big.clear()
    big.dispatch(1, 7, 1)
    assert _same_bits(cont, _images(big))
    assert np.isfinite(cont[0]).all() and cont[0][..., :3].mean() > 0.01


def test_static_dispatch_limits_and_state():
    g = _context(256, 256, "TorusMixed", 3)
    with pytest.raises(PtError) as e:  # nothing has been uploaded to the device yet
        g.dispatch_static(1, 1, 1)
    assert e.value.code == -2
    g.dispatch(1, 1, 1)
    scratch_bytes, record_bytes = R.static_scratch_info()
    assert scratch_bytes >= (64 << 20) and record_bytes == 36
    frames_fit = (scratch_bytes - 4096) // (256 * 256 * record_bytes)
    g.dispatch_static(2, frames_fit, 1)
    with pytest.raises(PtError) as e:
        g.dispatch_static(2, frames_fit + 1, 1)
    assert e.value.code == -1 and "scratch" in e.value.message
    g.synchronize()
