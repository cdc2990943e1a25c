"""ctypes binding of include/ptcore.h (one-to-one; no logic)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_LIBPTCORE: another build of the library (A/B measurements of kernel variants: tools/ab_libs.py)
LIB_PATH = os.environ.get("PT_LIBPTCORE") or os.path.join(_HERE, "libptcore.so")

c_float_p = ctypes.POINTER(ctypes.c_float)


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ptcore error {code}: {message}")
        self.code = code
        self.message = message


class Params(ctypes.Structure):
    """pt_params: the shader's uniform block (Shaders/Library/resources.glsl:6-31) + 2 knobs."""
    _fields_ = [
        ("view_matrix", ctypes.c_float * 16),
        ("proj_matrix", ctypes.c_float * 16),
        ("inv_proj_matrix", ctypes.c_float * 16),
        ("frame_count", ctypes.c_uint32),
        ("frame_dimensions", ctypes.c_float * 2),
        ("anti_aliasing_enabled", ctypes.c_uint32),
        ("focal_length", ctypes.c_float),
        ("aperture_size", ctypes.c_float),
        ("aperture_shape", ctypes.c_float * 2),
        ("specular_modifier", ctypes.c_float),
        ("specular_tint_modifier", ctypes.c_float),
        ("metallic_modifier", ctypes.c_float),
        ("roughness_modifier", ctypes.c_float),
        ("subsurface_modifier", ctypes.c_float),
        ("anisotropy_modifier", ctypes.c_float),
        ("sheen_roughness_modifier", ctypes.c_float),
        ("sheen_tint_modifier", ctypes.c_float),
        ("clearcoat_modifier", ctypes.c_float),
        ("clearcoat_roughness_modifier", ctypes.c_float),
        ("refraction_modifier", ctypes.c_float),
        ("transmission_modifier", ctypes.c_float),
        ("debug_value_a", ctypes.c_float),
        ("debug_value_b", ctypes.c_float),
        ("max_bounces", ctypes.c_uint32),
        ("accumulate_mode", ctypes.c_uint32),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("samples", ctypes.c_uint64),
        ("rays_closest", ctypes.c_uint64),
        ("rays_any", ctypes.c_uint64),
        ("node_visits", ctypes.c_uint64),
        ("triangle_tests", ctypes.c_uint64),
        ("closest_hits", ctypes.c_uint64),
        ("max_stack_depth", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


# name -> (restype, argtypes); every symbol include/ptcore.h declares
SYMBOLS = {
    "pt_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "pt_destroy": (None, [ctypes.c_void_p]),
    "pt_last_error": (ctypes.c_char_p, []),
    "pt_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "pt_bind_images": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pt_upload_nodes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "pt_upload_primitives": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "pt_upload_materials": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "pt_upload_texture": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "pt_upload_environment": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "pt_upload_environment_with_cache": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                        ctypes.c_int, ctypes.c_int]),
    "pt_add_model": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "pt_clear_models": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_process_buffers": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_set_bvh_rebuild": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pt_process_buffers_full_sweep": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_build_bvh_full_sweep": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "pt_build_bvh_sah": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "pt_assemble_primitives": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]),
    "pt_align_nodes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "pt_align_primitives": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "pt_calculate_hdri_cache": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "pt_build_bvh_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "pt_process_buffers_device": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_download_bvh": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "pt_bvh_cache_key": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]),
    "pt_bvh_cache_save": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p, ctypes.c_size_t]),
    "pt_bvh_cache_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t)]),
    "pt_gltf_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "pt_gltf_free": (None, [ctypes.c_void_p]),
    "pt_gltf_submesh_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]),
    "pt_gltf_submesh_size": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.POINTER(ctypes.c_uint64)]),
    "pt_gltf_submesh_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_uint32)]),
    "pt_gltf_submesh_material": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, c_float_p,
                                                ctypes.POINTER(ctypes.c_int32)]),
    "pt_gltf_image_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]),
    "pt_gltf_image_source": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_char_p),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]),
    "pt_look_at": (None, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "pt_perspective": (None, [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_float_p]),
    "pt_inverse4": (None, [c_float_p, c_float_p]),
    "pt_default_params": (None, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(Params)]),
    "pt_set_params": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Params)]),
    "pt_clear": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_dispatch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "pt_dispatch_counted": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "pt_dispatch_static": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "pt_static_scratch_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]),
    "pt_scale_images": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float]),
    "pt_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "pt_read_image": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "pt_write_image": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "pt_read_image_rgb": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "pt_read_display": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "pt_debug_view": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pt_debug_get_resident_waves": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]),
    "pt_debug_get_tree_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "pt_get_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Stats)]),
    "pt_last_dispatch_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "pt_last_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "pt_scene_bytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "pt_debug_set_tuning": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "pt_debug_get_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]),
    "pt_debug_get_depth_histogram": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]),
    "pt_debug_get_wave_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]),
    "pt_debug_eval_math": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t]),
}

_lib = None


def load():
    """Load libptcore.so (built in-tree by `make -C pathtracer_amd/csrc`).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 / HSA runtime,
    # and two copies in one process cannot both open the GPU.  Importing torch first makes the
    # dynamic loader satisfy libptcore's NEEDED libamdhip64.so.7 with the copy torch already
    # loaded (same SONAME), so the library shares torch's runtime, streams and device memory.
    # Without torch the system runtime under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise PtError(code, load().pt_last_error().decode("utf-8", "replace"))
