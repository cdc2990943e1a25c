"""Thin object wrapper over the C ABI: one Renderer = one pt_ctx (one GPU, one stream).

Method names follow the reference host class they stand in for
(Source/PathTracer/PathTracingRenderer.{h,cpp}: AddPathTracingModel, ClearPathTracingModels,
ProcessBuffers; PathTracingRenderPass::Render -> dispatch).
"""
import ctypes

import numpy as np

from . import lib as _l
from .lib import Params, Stats, check

RADIANCE, ALBEDO, NORMAL = 0, 1, 2


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Renderer:
    def __init__(self, width, height, device=0):
        self._lib = _l.load()
        self.width, self.height = int(width), int(height)
        h = ctypes.c_void_p()
        check(self._lib.pt_create(int(device), self.width, self.height, ctypes.byref(h)))
        self._h = h
        self._keep = []

    # ------------------------------------------------------------------ life cycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        check(self._lib.pt_set_stream(self._h, ctypes.c_void_p(hip_stream)))

    def bind_images(self, d_radiance, d_albedo, d_normal):
        """Device pointers (ints) of three width*height*4 float buffers, or three zeros."""
        check(self._lib.pt_bind_images(self._h, ctypes.c_void_p(d_radiance), ctypes.c_void_p(d_albedo),
                                       ctypes.c_void_p(d_normal)))

    # ---------------------------------------------------------------- scene upload
    def upload_nodes(self, nodes48):
        a = np.ascontiguousarray(nodes48)
        assert a.nbytes % 48 == 0
        check(self._lib.pt_upload_nodes(self._h, _ptr(a), a.nbytes // 48))

    def upload_primitives(self, prims112):
        a = np.ascontiguousarray(prims112)
        assert a.nbytes % 112 == 0
        check(self._lib.pt_upload_primitives(self._h, _ptr(a), a.nbytes // 112))

    def upload_materials(self, mats176):
        a = np.ascontiguousarray(mats176)
        assert a.nbytes % 176 == 0
        check(self._lib.pt_upload_materials(self._h, _ptr(a), a.nbytes // 176))

    def upload_texture(self, slot, texels, is_srgb=False):
        a = np.ascontiguousarray(texels)
        h, w = a.shape[:2]
        ch = 1 if a.ndim == 2 else a.shape[2]
        is_float = a.dtype == np.float32
        assert is_float or a.dtype == np.uint8
        check(self._lib.pt_upload_texture(self._h, int(slot), w, h, ch, int(is_srgb), int(is_float), _ptr(a)))

    def upload_environment(self, hdri_rgb, cache_rgb=None):
        a = np.ascontiguousarray(hdri_rgb, dtype=np.float32)
        h, w = a.shape[:2]
        if cache_rgb is None:
            check(self._lib.pt_upload_environment(self._h, _ptr(a), w, h))
        else:
            c = np.ascontiguousarray(cache_rgb, dtype=np.float32)
            assert c.shape == a.shape
            check(self._lib.pt_upload_environment_with_cache(self._h, _ptr(a), _ptr(c), w, h))

    # PathTracingRenderer::AddPathTracingModel / ClearPathTracingModels / ProcessBuffers
    def AddPathTracingModel(self, vertices, indices, material176):
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        i = np.ascontiguousarray(indices)
        m = np.ascontiguousarray(material176)
        assert m.nbytes == 176 and v.nbytes % 32 == 0
        check(self._lib.pt_add_model(self._h, _ptr(v), v.nbytes // 32, _ptr(i), i.size, i.dtype.itemsize, _ptr(m)))

    def ClearPathTracingModels(self):
        check(self._lib.pt_clear_models(self._h))

    def set_bvh_rebuild(self, enable=True):
        """Uploaded trees are replaced by the GPU builder's over the same primitives (see pt_set_bvh_rebuild)."""
        check(self._lib.pt_set_bvh_rebuild(self._h, int(bool(enable))))

    def ProcessBuffers(self, full_sweep=False):
        """full_sweep: the sweep-SAH builder without the reference's cost cap (a different, better tree)."""
        check((self._lib.pt_process_buffers_full_sweep if full_sweep else self._lib.pt_process_buffers)(self._h))

    def ProcessBuffersDevice(self):
        """ProcessBuffers with the BVH built on the GPU (Morton-code linear BVH) instead of the sweep SAH."""
        check(self._lib.pt_process_buffers_device(self._h))

    def build_bvh_device(self, prims112):
        a = np.ascontiguousarray(prims112)
        assert a.nbytes % 112 == 0
        check(self._lib.pt_build_bvh_device(self._h, _ptr(a), a.nbytes // 112))

    def download_bvh(self):
        """-> (nodes48 (n, 12) float32 view, prims112 (m, 28) float32 view) of the device-built tree."""
        nn, npr = ctypes.c_size_t(), ctypes.c_size_t()
        check(self._lib.pt_download_bvh(self._h, None, 0, ctypes.byref(nn), None, 0, ctypes.byref(npr)))
        nodes = np.zeros((nn.value, 12), np.float32)
        prims = np.zeros((npr.value, 28), np.float32)
        check(self._lib.pt_download_bvh(self._h, _ptr(nodes), nn.value, ctypes.byref(nn), _ptr(prims), npr.value,
                                        ctypes.byref(npr)))
        return nodes, prims

    # ------------------------------------------------------------------- per frame
    def set_params(self, params):
        check(self._lib.pt_set_params(self._h, ctypes.byref(params)))

    def clear(self):
        check(self._lib.pt_clear(self._h))

    def dispatch(self, frame_first, frame_count=1, frame_stride=1, counted=False):
        fn = self._lib.pt_dispatch_counted if counted else self._lib.pt_dispatch
        check(fn(self._h, int(frame_first), int(frame_count), int(frame_stride)))

    def dispatch_static(self, frame_first, frame_count=1, frame_stride=1):
        """pt_dispatch_static: the dispatch that allocates nothing (scratch in a static device array)."""
        check(self._lib.pt_dispatch_static(self._h, int(frame_first), int(frame_count), int(frame_stride)))

    TUNING_KEYS = {"pipeline": 0, "trav_exit": 1, "refill_min": 2, "check_every": 3, "traverse_blocks": 4, "hit_min": 5, "tile_order": 6, "tile_split": 7, "sample_budget_mib": 8, "pull_items": 9, "ploc_radius": 10, "waves_per_simd": 11, "overlap_calls": 12, "bvh_leaf_max": 13, "texture_decode_mib": 14, "plan_every": 15, "overlap_share": 16, "bvh_optimise": 17, "bvh_optimise_mod": 18, "bvh_optimise_max_height": 19, "leaf_recut": 20}

    def set_tuning(self, **kw):
        """pipeline=0|1 (megakernel|wavefront), trav_exit, refill_min, check_every, traverse_blocks, hit_min, tile_order, tile_split, sample_budget_mib, pull_items, ploc_radius, waves_per_simd, overlap_calls, bvh_leaf_max, texture_decode_mib, plan_every, overlap_share, bvh_optimise, bvh_optimise_mod, bvh_optimise_max_height, leaf_recut."""
        for k, v in kw.items():
            check(self._lib.pt_debug_set_tuning(self._h, self.TUNING_KEYS[k], int(v)))

    def profile(self):
        out = (ctypes.c_uint64 * 16)()
        check(self._lib.pt_debug_get_profile(self._h, out))
        return list(out)

    def depth_histogram(self):
        """Closest-hit queries by path depth of the last counted megakernel dispatch (primary rays first)."""
        out = (ctypes.c_uint64 * 32)()
        check(self._lib.pt_debug_get_depth_histogram(self._h, out))
        return list(out)

    def wave_times(self):
        """-> ((n, 2) uint64 start/end ticks per block of the grid (start 0: the block found no work), tick rate
        in kHz) of the last counted dispatch."""
        n, khz = ctypes.c_size_t(), ctypes.c_int()
        check(self._lib.pt_debug_get_wave_times(self._h, None, 0, ctypes.byref(n), ctypes.byref(khz)))
        out = np.zeros((n.value, 2), np.uint64)
        if n.value:
            check(self._lib.pt_debug_get_wave_times(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value,
                                                    ctypes.byref(n), ctypes.byref(khz)))
        return out, khz.value

    def scale_images(self, scale):
        check(self._lib.pt_scale_images(self._h, float(scale)))

    def synchronize(self):
        check(self._lib.pt_synchronize(self._h))

    def read_image(self, which):
        out = np.empty((self.height, self.width, 4), np.float32)
        check(self._lib.pt_read_image(self._h, int(which), _ptr(out)))
        return out

    def write_image(self, which, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.float32)
        if a.shape != (self.height, self.width, 4):
            raise ValueError(f"image must be ({self.height}, {self.width}, 4)")
        check(self._lib.pt_write_image(self._h, int(which), _ptr(a)))

    def save_checkpoint(self, path, next_frame):
        """Accumulation images + the FrameCount to go on with, as an .npz (a progressive render
        continues bit-identically from it: see load_checkpoint)."""
        np.savez(path, radiance=self.read_image(0), albedo=self.read_image(1), normal=self.read_image(2),
                 next_frame=np.uint32(next_frame), size=np.array([self.width, self.height], np.int32))

    def load_checkpoint(self, path):
        """-> the FrameCount to dispatch next.  The scene and parameters are the caller's to restore."""
        with np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False) as z:
            if tuple(z["size"]) != (self.width, self.height):
                raise ValueError(f"checkpoint is {tuple(z['size'])}, the renderer {self.width}x{self.height}")
            for which, name in enumerate(("radiance", "albedo", "normal")):
                self.write_image(which, z[name])
            return int(z["next_frame"])

    def read_image_rgb(self, which):
        """(height, width, 3) float32: the layout the reference reads back for its denoiser."""
        out = np.empty((self.height, self.width, 3), np.float32)
        check(self._lib.pt_read_image_rgb(self._h, int(which), _ptr(out)))
        return out

    def read_display(self, exposure=1.0):
        """(height, width, 3) uint8: the radiance image through the reference's display pass
        (exposure, ACES fit, sRGB encoding), row 0 = bottom row."""
        out = np.empty((self.height, self.width, 3), np.uint8)
        check(self._lib.pt_read_display(self._h, float(exposure), _ptr(out)))
        return out

    def tree_info(self):
        """(levels of the four-wide tree in use, folded level by level instead of by surface area)."""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        check(self._lib.pt_debug_get_tree_info(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, bool(b.value)

    def resident_waves(self):
        """Waves the most recent megakernel launch's kernel variant keeps resident on the GPU."""
        n = ctypes.c_uint32()
        check(self._lib.pt_debug_get_resident_waves(self._h, ctypes.byref(n)))
        return n.value

    def debug_view(self, view):
        """The reference shader's DEBUG_ENABLED output into the radiance image: 1 = BVH heat map, 2 = HDRI
        sampling cache.  Returns the (height, width, 4) float32 image."""
        check(self._lib.pt_debug_view(self._h, int(view)))
        return self.read_image(0)

    def stats(self):
        s = Stats()
        check(self._lib.pt_get_stats(self._h, ctypes.byref(s)))
        return s

    def last_dispatch_ms(self):
        ms = ctypes.c_float()
        check(self._lib.pt_last_dispatch_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def last_kernel_ms(self):
        """(trace kernel ms, resolve kernel ms) of the most recent megakernel launch."""
        a, b = ctypes.c_float(), ctypes.c_float()
        check(self._lib.pt_last_kernel_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def scene_bytes(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        check(self._lib.pt_scene_bytes(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value


# ----------------------------------------------------------- host-side helpers (no GPU)
def default_params(width, height):
    p = Params()
    _l.load().pt_default_params(int(width), int(height), ctypes.byref(p))
    return p


def set_camera(params, eye, center, width, height, fovy=1.57, up=(0.0, 1.0, 0.0), z_near=0.01, z_far=1000.0):
    """Fill the three camera matrices of `params` the way the reference's camera does
    (Runtime/Camera/Camera.cpp:10-18: glm::lookAt, glm::perspective; PathTracingApplication.cpp:250:
    glm::inverse of the projection)."""
    lib = _l.load()
    f = lambda v, n: (ctypes.c_float * n)(*[float(x) for x in v])
    view, proj, inv = (ctypes.c_float * 16)(), (ctypes.c_float * 16)(), (ctypes.c_float * 16)()
    lib.pt_look_at(f(eye, 3), f(center, 3), f(up, 3), view)
    lib.pt_perspective(float(fovy), float(width) / float(height), float(z_near), float(z_far), proj)
    lib.pt_inverse4(proj, inv)
    params.view_matrix[:] = list(view)
    params.proj_matrix[:] = list(proj)
    params.inv_proj_matrix[:] = list(inv)
    return params


def static_scratch_info():
    """(bytes of pt_dispatch_static's static device array, bytes of one sample record) of the library as built."""
    size, rec = ctypes.c_uint64(), ctypes.c_uint32()
    check(_l.load().pt_static_scratch_info(ctypes.byref(size), ctypes.byref(rec)))
    return size.value, rec.value


def build_bvh_sah(prims100, leaf_size=4, full_sweep=False):
    """prims100: (n, 25) float32 view of 100-B records. Returns (reordered prims, nodes (m, 10)).
    full_sweep: without the reference's cost cap (a different, better tree)."""
    p = np.ascontiguousarray(prims100, dtype=np.float32).copy()
    assert p.nbytes % 100 == 0
    n = p.nbytes // 100
    nodes = np.zeros((2 * n + 2, 10), np.float32)
    cnt = ctypes.c_size_t()
    build = _l.load().pt_build_bvh_full_sweep if full_sweep else _l.load().pt_build_bvh_sah
    check(build(_ptr(p), n, int(leaf_size), _ptr(nodes), nodes.shape[0], ctypes.byref(cnt)))
    return p, nodes[:cnt.value].copy()


def assemble_primitives(vertices, indices, mesh_index):
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    i = np.ascontiguousarray(indices)
    out = np.zeros((i.size // 3, 25), np.float32)
    check(_l.load().pt_assemble_primitives(_ptr(v), v.nbytes // 32, _ptr(i), i.size, i.dtype.itemsize, int(mesh_index), _ptr(out)))
    return out


def align_nodes(nodes40):
    a = np.ascontiguousarray(nodes40, dtype=np.float32)
    n = a.nbytes // 40
    out = np.zeros((n, 12), np.float32)
    check(_l.load().pt_align_nodes(_ptr(a), n, _ptr(out)))
    return out


def align_primitives(prims100):
    a = np.ascontiguousarray(prims100, dtype=np.float32)
    n = a.nbytes // 100
    out = np.zeros((n, 28), np.float32)
    check(_l.load().pt_align_primitives(_ptr(a), n, _ptr(out)))
    return out


def bvh_cache_key(primitives):
    """Key of the on-disk BVH cache: hash of the unsorted primitive soup the tree is built from."""
    a = np.ascontiguousarray(primitives)
    key = ctypes.c_uint64()
    check(_l.load().pt_bvh_cache_key(_ptr(a), a.nbytes, ctypes.byref(key)))
    return key.value


def bvh_cache_save(path, key, nodes48, prims112):
    n, p = np.ascontiguousarray(nodes48), np.ascontiguousarray(prims112)
    assert n.nbytes % 48 == 0 and p.nbytes % 112 == 0
    check(_l.load().pt_bvh_cache_save(str(path).encode(), key, _ptr(n), n.nbytes // 48, _ptr(p), p.nbytes // 112))


def bvh_cache_load(path, key):
    """-> (nodes48 (n, 12) float32 view, prims112 (m, 28) float32 view); PtError if the file is missing,
    stale (other key), truncated or corrupt."""
    lib = _l.load()
    nn, npr = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.pt_bvh_cache_load(str(path).encode(), key, None, 0, ctypes.byref(nn), None, 0, ctypes.byref(npr)))
    nodes = np.zeros((nn.value, 12), np.float32)
    prims = np.zeros((npr.value, 28), np.float32)
    check(lib.pt_bvh_cache_load(str(path).encode(), key, _ptr(nodes), nn.value, ctypes.byref(nn), _ptr(prims), npr.value,
                                ctypes.byref(npr)))
    return nodes, prims


def calculate_hdri_cache(hdri_rgb):
    a = np.ascontiguousarray(hdri_rgb, dtype=np.float32)
    h, w = a.shape[:2]
    out = np.zeros_like(a)
    check(_l.load().pt_calculate_hdri_cache(_ptr(a), w, h, _ptr(out)))
    return out


def eval_math(fn, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    check(_l.load().pt_debug_eval_math(int(device), int(fn), _ptr(a), _ptr(b), _ptr(out), a.size))
    return out
