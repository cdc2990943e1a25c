"""glTF 2.0 / GLB scenes: ctypes wrapper of the pt_gltf_* reader plus the image decoding the
reference leaves to stb_image (here: Pillow).

The reader mirrors Source/Runtime/Asset/ModelLoader.cpp:83-121 (one sub-mesh per primitive, node
transforms ignored) and :150-287 (material attributes and texture slots).  Images are decoded to
8-bit RGB or RGBA as the reference requests per slot (ModelLoader.cpp:201-231) and flipped
vertically (PathTracingApplication.cpp:212).  JPEG decoders differ in their last bit, so texels of
.jpg images are not guaranteed to equal stb_image's.
"""
import ctypes
import io

import numpy as np

from . import lib as _l
from .lib import check
from .scenes import MATERIAL_DTYPE, TEXTURE_SLOTS

ATTRIBUTE_NAMES = ["specular", "specularTint", "metallic", "roughness", "subsurface", "anisotropy", "sheenRoughness",
                   "sheenTint", "clearcoat", "clearcoatRoughness", "refraction", "transmission"]  # Material.h:47-58
# channels the reference asks its texture loader for, per Material::MaterialTextureSlot
SLOT_CHANNELS = [3, 3, 3, 4, 3, 3, 4, 3, 3, 3, 3]
SLOT_SRGB = [False, True] + [False] * 9


class GltfScene:
    """submeshes: list of (vertices (n, 8) float32, indices uint32); attributes: (n, 18) float32;
    slot_images: (n, 11) int32 image index per texture slot or -1; image_sources: list of
    ("path", str) or ("bytes", bytes)."""

    def __init__(self, path):
        lib = _l.load()
        h = ctypes.c_void_p()
        check(lib.pt_gltf_load(str(path).encode(), ctypes.byref(h)))
        try:
            n = ctypes.c_uint32()
            check(lib.pt_gltf_submesh_count(h, ctypes.byref(n)))
            self.submeshes = []
            self.attributes = np.zeros((n.value, 18), np.float32)
            self.slot_images = np.full((n.value, 11), -1, np.int32)
            for s in range(n.value):
                nv, ni = ctypes.c_uint64(), ctypes.c_uint64()
                check(lib.pt_gltf_submesh_size(h, s, ctypes.byref(nv), ctypes.byref(ni)))
                v = np.zeros((nv.value, 8), np.float32)
                i = np.zeros(ni.value, np.uint32)
                check(lib.pt_gltf_submesh_read(h, s, v.ctypes.data_as(ctypes.c_void_p),
                                               i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
                self.submeshes.append((v, i))
                check(lib.pt_gltf_submesh_material(h, s, self.attributes[s].ctypes.data_as(_l.c_float_p),
                                                   self.slot_images[s].ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
            check(lib.pt_gltf_image_count(h, ctypes.byref(n)))
            self.image_sources = []
            for k in range(n.value):
                p, b, nb = ctypes.c_char_p(), ctypes.c_void_p(), ctypes.c_uint64()
                check(lib.pt_gltf_image_source(h, k, ctypes.byref(p), ctypes.byref(b), ctypes.byref(nb)))
                if p.value is not None:
                    self.image_sources.append(("path", p.value.decode()))
                else:
                    self.image_sources.append(("bytes", ctypes.string_at(b.value, nb.value) if nb.value else b""))
        finally:
            lib.pt_gltf_free(h)

    @classmethod
    def from_npz(cls, path):
        """A scene stored by tests/golden/make_scene_fixtures.py: the reader's output as arrays, images
        already decoded (RGBA uint8, bottom row first)."""
        z = np.load(path, allow_pickle=False)
        self = cls.__new__(cls)
        self.attributes = z["attributes"].astype(np.float32)
        self.slot_images = z["slot_images"].astype(np.int32)
        self.submeshes = [(z[f"v{k}"], z[f"i{k}"]) for k in range(self.attributes.shape[0])]
        self.image_sources = []
        k = 0
        while f"image{k}" in z.files:
            self.image_sources.append(("texels", z[f"image{k}"]))
            k += 1
        return self

    def triangle_count(self):
        return sum(i.size // 3 for _, i in self.submeshes)

    def decode_image(self, image, channels, max_size=None):
        """-> uint8 (h, w, channels), first row = bottom of the picture (flipped like the reference)."""
        kind, src = self.image_sources[image]
        if kind == "texels":
            return np.ascontiguousarray(src[..., :channels])
        from PIL import Image  # the reference decodes with stb_image; any decoder will do
        im = Image.open(src if kind == "path" else io.BytesIO(src))
        im = im.convert("RGBA" if channels == 4 else "RGB")
        if max_size and max(im.size) > max_size:
            s = max_size / max(im.size)
            im = im.resize((max(1, round(im.size[0] * s)), max(1, round(im.size[1] * s))), Image.BOX)
        return np.ascontiguousarray(np.asarray(im)[::-1])

    def materials(self, max_texture_size=None):
        """-> (MaterialAlign records with 1-based texture-table indices in the handle fields,
        textures: list of (slot index, texels uint8, is_srgb)), ready for upload_materials /
        upload_texture.  One table entry per distinct (image, channels, sRGB) combination."""
        mats = np.zeros(len(self.submeshes), MATERIAL_DTYPE)
        table, textures = {}, []
        for s in range(len(self.submeshes)):
            a = self.attributes[s]
            mats[s]["emission"] = a[0:3]
            mats[s]["albedo"] = a[3:6]
            for k, name in enumerate(ATTRIBUTE_NAMES):
                mats[s][name] = a[6 + k]
            for slot in range(len(TEXTURE_SLOTS)):
                image = int(self.slot_images[s, slot])
                if image < 0:
                    continue
                key = (image, SLOT_CHANNELS[slot], SLOT_SRGB[slot])
                if key not in table:
                    table[key] = len(table) + 1
                    textures.append((table[key], self.decode_image(image, SLOT_CHANNELS[slot], max_texture_size),
                                     SLOT_SRGB[slot]))
                mats[s]["handles"][slot] = table[key]
        return mats, textures


def load(path):
    return GltfScene(path)


def add_to_renderer(renderer, scene, max_texture_size=None):
    """Scene-visitor order of the reference (PathTracingRenderer.cpp:184-446): every sub-mesh becomes
    one path-tracing model with its own material row; textures go to the renderer's texture table."""
    mats, textures = scene.materials(max_texture_size)
    for slot, texels, is_srgb in textures:
        renderer.upload_texture(slot, texels, is_srgb)
    for (v, i), m in zip(scene.submeshes, mats):
        renderer.AddPathTracingModel(v, i, m)
    return mats, textures
