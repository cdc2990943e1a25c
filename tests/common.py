"""Shared scene construction for the tests: the same reference-layout buffers go to the product
(through the C ABI) and to the oracle."""
import functools

import numpy as np

from pathtracer_amd import renderer as R
from pathtracer_amd import scenes


def soup(recipe_list):
    meshes, mats = scenes.flatten(recipe_list)
    parts = [R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)]
    return np.concatenate(parts), mats


@functools.lru_cache(maxsize=8)
def built_scene(recipe_name):
    """-> (nodes48 (n,12) f32, prims112 (m,28) f32, materials) built with the product's host code."""
    prims100, mats = soup(scenes.recipe(recipe_name))
    prims_sorted, nodes40 = R.build_bvh_sah(prims100, 4)
    return R.align_nodes(nodes40), R.align_primitives(prims_sorted), mats


def built_meshes(meshes, mats):
    """Same as built_scene for an explicit list of (vertices, indices) sub-meshes."""
    prims100 = np.concatenate([R.assemble_primitives(v, i, k) for k, (v, i) in enumerate(meshes)])
    prims_sorted, nodes40 = R.build_bvh_sah(prims100, 4)
    return R.align_nodes(nodes40), R.align_primitives(prims_sorted), mats


@functools.lru_cache(maxsize=4)
def environment(width=256, height=128):
    hdri = scenes.sky_hdri(width, height)
    cache = R.calculate_hdri_cache(hdri)
    return hdri, cache


def nodes_view(nodes48):
    raw = np.ascontiguousarray(nodes48).view(np.int32).reshape(-1, 12)
    f = np.ascontiguousarray(nodes48).view(np.float32).reshape(-1, 12)
    return raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3], f[:, 4:7], f[:, 8:11]


def check_tree(nodes48, prims112, source112):
    """Every primitive in exactly one leaf, leaves of 1..4, boxes tight around their contents."""
    left, right, n, index, lo, hi = nodes_view(nodes48)
    p = np.ascontiguousarray(prims112).view(np.float32).reshape(-1, 28)
    src = np.ascontiguousarray(source112).view(np.float32).reshape(-1, 28)
    # same multiset of primitive records (compare as raw bytes)
    a = np.sort(np.ascontiguousarray(p).view("V112").ravel())
    b = np.sort(np.ascontiguousarray(src).view("V112").ravel())
    assert a.tobytes() == b.tobytes()
    pmin = np.minimum(np.minimum(p[:, 0:3], p[:, 8:11]), p[:, 16:19])
    pmax = np.maximum(np.maximum(p[:, 0:3], p[:, 8:11]), p[:, 16:19])
    covered = np.zeros(p.shape[0], np.int32)
    seen = np.zeros(left.shape[0], np.int32)
    stack, depth_max, interior = [(1, 1)], 0, 0
    while stack:
        k, depth = stack.pop()
        seen[k] += 1
        depth_max = max(depth_max, depth)
        if n[k] > 0:
            assert 1 <= n[k] <= 4
            sl = slice(index[k], index[k] + n[k])
            covered[sl] += 1
            assert np.array_equal(lo[k], pmin[sl].min(0)) and np.array_equal(hi[k], pmax[sl].max(0))
        else:
            interior += 1
            for c in (left[k], right[k]):
                assert 1 < c < left.shape[0]
                assert (lo[c] >= lo[k]).all() and (hi[c] <= hi[k]).all()
                stack.append((c, depth + 1))
            assert np.array_equal(lo[k], np.minimum(lo[left[k]], lo[right[k]]))
            assert np.array_equal(hi[k], np.maximum(hi[left[k]], hi[right[k]]))
    assert (covered == 1).all()
    assert (seen[1:] == 1).all() and seen[0] == 0
    return depth_max, interior


def sah_cost(nodes48):
    left, right, n, index, lo, hi = nodes_view(nodes48)
    d = (hi - lo).astype(np.float64)
    area = 2.0 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
    used = np.zeros(left.shape[0], bool)
    stack = [1]
    while stack:
        k = stack.pop()
        used[k] = True
        if n[k] == 0:
            stack += [left[k], right[k]]
    leaf = used & (n > 0)
    inner = used & (n == 0)
    return (area[inner].sum() * 1.0 + (area[leaf] * n[leaf]).sum() * 1.0) / area[1]


def image_hash(array):
    """A checksum of checksums of an array's bytes, in numpy alone (so that it needs neither library):
    FNV-1a-64 of every row (the first axis; a 1-D array is cut into rows of 4096 bytes, the last one
    short), all rows side by side, then FNV-1a-64 of the rows' hashes as little-endian bytes."""
    a = np.ascontiguousarray(array)
    raw = a.view(np.uint8).reshape(a.shape[0], -1) if a.ndim > 1 else a.view(np.uint8).ravel()
    offset, prime = np.uint64(1469598103934665603), np.uint64(1099511628211)

    def rows_fnv(rows):  # rows: (n, m) uint8 -> (n,) uint64
        h = np.full(rows.shape[0], offset, np.uint64)
        with np.errstate(over="ignore"):
            for k in range(rows.shape[1]):
                h = (h ^ rows[:, k].astype(np.uint64)) * prime
        return h

    if raw.ndim == 1:
        full = raw.size // 4096 * 4096
        hashes = rows_fnv(raw[:full].reshape(-1, 4096)) if full else np.zeros(0, np.uint64)
        if raw.size > full:
            hashes = np.concatenate([hashes, rows_fnv(raw[full:].reshape(1, -1))])
    else:
        hashes = rows_fnv(raw)
    return f"{int(rows_fnv(hashes.astype('<u8').view(np.uint8).reshape(1, -1))[0]):016x}"
