"""TEST INFRASTRUCTURE ONLY: ctypes access to the CPU oracle (oracle/liboracle.so) and to the
reference's own BVH builder / glm matrices (oracle/_ref/libref_shim.so, built from the reference
sources in place).  Import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref_shim.so")


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleCounters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("samples", "rays_closest", "rays_any", "node_fetches", "pair_tests",
                                               "triangle_tests", "closest_hits", "stack_overflows")] + \
               [("max_stack", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_PATH):
            raise ImportError(f"{ORACLE_PATH} missing: run `make -C oracle`")
        lib = ctypes.CDLL(ORACLE_PATH)
        lib.oracle_create.restype = ctypes.c_void_p
        lib.oracle_destroy.argtypes = [ctypes.c_void_p]
        lib.oracle_set_scene.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                         ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
        lib.oracle_set_texture.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        lib.oracle_calculate_hdri_cache.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_set_environment.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.oracle_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.oracle_debug_view.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        lib.oracle_debug_view.restype = None
        lib.oracle_eval_math.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        lib.oracle_display.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_float, ctypes.c_void_p]
        lib.oracle_display.restype = None
        lib.oracle_fnv1a64.restype = ctypes.c_uint64
        lib.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_uint64]
        assert lib.oracle_sizeof_counters() == ctypes.sizeof(OracleCounters)
        _oracle = lib
    return _oracle


def ref_lib():
    """The reference's own code (BVH.cpp, glm) compiled in place; None when not built."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        lib = ctypes.CDLL(REF_PATH)
        lib.ref_build_bvh_sah.restype = ctypes.c_long
        lib.ref_build_bvh_sah.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        fp = ctypes.POINTER(ctypes.c_float)
        lib.ref_glm_look_at.argtypes = [fp, fp, fp, fp]
        lib.ref_glm_perspective.argtypes = [ctypes.c_float] * 4 + [fp]
        lib.ref_glm_inverse.argtypes = [fp, fp]
        _ref = lib
    return _ref


REF_HDRI_PATH = os.path.join(_HERE, "_ref", "libref_hdri.so")
_ref_hdri = None


def ref_hdri_lib():
    """Lines 803-901 of the reference's PathTracingRenderer.cpp (CalculateHdriCache's arithmetic)
    compiled in place; None when not built."""
    global _ref_hdri
    if _ref_hdri is None and os.path.exists(REF_HDRI_PATH):
        lib = ctypes.CDLL(REF_HDRI_PATH)
        lib.ref_calculate_hdri_cache.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.ref_calculate_hdri_cache.restype = None
        _ref_hdri = lib
    return _ref_hdri


def ref_calculate_hdri_cache(hdri_rgb):
    a = np.ascontiguousarray(hdri_rgb, dtype=np.float32)
    h, w = a.shape[:2]
    out = np.zeros_like(a)
    ref_hdri_lib().ref_calculate_hdri_cache(_ptr(a), w, h, _ptr(out))
    return out


def ref_build_bvh_sah(prims100, leaf_size=4):
    lib = ref_lib()
    p = np.ascontiguousarray(prims100, dtype=np.float32).copy()
    n = p.nbytes // 100
    nodes = np.zeros((2 * n + 2, 10), np.float32)
    cnt = lib.ref_build_bvh_sah(_ptr(p), n, leaf_size, _ptr(nodes), nodes.shape[0])
    assert cnt > 0
    return p, nodes[:cnt].copy()


def calculate_hdri_cache(hdri_rgb):
    a = np.ascontiguousarray(hdri_rgb, dtype=np.float32)
    h, w = a.shape[:2]
    out = np.zeros_like(a)
    oracle_lib().oracle_calculate_hdri_cache(_ptr(a), w, h, _ptr(out))
    return out


def display(rgba, exposure=1.0):
    """(h, w, 4) float32 radiance -> (h, w, 3) uint8 through the reference's display pass."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(a.shape[:-1] + (3,), np.uint8)
    oracle_lib().oracle_display(_ptr(a), a.size // 4, float(exposure), _ptr(out))
    return out


def fnv1a64(array, h=1469598103934665603):
    a = np.ascontiguousarray(array)
    return int(oracle_lib().oracle_fnv1a64(_ptr(a), a.nbytes, h))


def eval_math(fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    oracle_lib().oracle_eval_math(int(fn), _ptr(a), _ptr(b), _ptr(out), a.size)
    return out


class Oracle:
    """CPU restatement of the shader: same inputs as the product's C ABI (reference layouts)."""

    def __init__(self):
        self._lib = oracle_lib()
        self._h = ctypes.c_void_p(self._lib.oracle_create())

    def close(self):
        if self._h:
            self._lib.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, nodes48, prims112, mats176):
        n, p, m = (np.ascontiguousarray(x) for x in (nodes48, prims112, mats176))
        self._lib.oracle_set_scene(self._h, _ptr(n), n.nbytes // 48, _ptr(p), p.nbytes // 112, _ptr(m), m.nbytes // 176)

    def set_texture(self, slot, texels, is_srgb=False):
        a = np.ascontiguousarray(texels)
        h, w = a.shape[:2]
        ch = 1 if a.ndim == 2 else a.shape[2]
        self._lib.oracle_set_texture(self._h, slot, w, h, ch, int(is_srgb), int(a.dtype == np.float32), _ptr(a))

    def set_environment(self, hdri_rgb, cache_rgb):
        a = np.ascontiguousarray(hdri_rgb, dtype=np.float32)
        c = np.ascontiguousarray(cache_rgb, dtype=np.float32)
        h, w = a.shape[:2]
        self._lib.oracle_set_environment(self._h, _ptr(a), _ptr(c), w, h)

    def debug_view(self, params, width, height, view, threads=0):
        """The shader's DEBUG_ENABLED output (1 = DebugBvh heat map, 2 = DebugHdriCache): (h, w, 4) float32."""
        out = np.zeros((height, width, 4), np.float32)
        assert ctypes.sizeof(params) == self._lib.oracle_sizeof_params()
        self._lib.oracle_debug_view(self._h, ctypes.byref(params), width, height, int(view), _ptr(out), threads)
        return out

    def render(self, params, width, height, frame_first=1, frame_count=1, frame_stride=1, images=None, rows=None,
               threads=0):
        """params: a ctypes struct with pt_params' layout.  Returns (radiance, albedo, normal, overflow_mask,
        counters); images (three (h, w, 4) float32 arrays) are accumulated into when given."""
        if images is None:
            images = [np.zeros((height, width, 4), np.float32) for _ in range(3)]
        mask = np.zeros((height, width), np.uint8)
        ctr = OracleCounters()
        y0, y1 = rows if rows else (0, height)
        assert ctypes.sizeof(params) == self._lib.oracle_sizeof_params()
        self._lib.oracle_render(self._h, ctypes.byref(params), width, height, frame_first, frame_count, frame_stride,
                                y0, y1, _ptr(images[0]), _ptr(images[1]), _ptr(images[2]), _ptr(mask),
                                ctypes.byref(ctr), threads)
        return images[0], images[1], images[2], mask, ctr
