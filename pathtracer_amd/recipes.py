"""The reference application's thirteen scenes (PathTracingApplication.cpp:442-491, 505-739) as
loadable recipes.

Five are plain glTF models (AreaLight, Fireplace, Mill, Sponza, SponzaReduced: :505-553); eight put
Bunny.glb or Dragon.glb on Floor.glb and replace the first material's attributes by a fresh
Material::MaterialAttributes with a few fields set (:555-739; the texture handles of that material
stay, Material.h:43-59 gives the defaults of every other field).

Models are read from `content_dir` (the application's Content directory) when it holds them.
Without it the meshes committed with this package stand in: bunny/floor as .ptm vertex dumps, the
AreaLight and Mill models as array dumps of the glTF reader's output; the dragon scenes then use
the bunny (Dragon.glb is not redistributed here), and Fireplace / Sponza need a content directory
that has their .bin files.
"""
import os

import numpy as np

from . import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden", "scenes")

GLTF_SCENES = {  # name -> path below Content/Models
    "AreaLight": "AreaLight/AreaLight.gltf",
    "Fireplace": "Fireplace/Fireplace.gltf",
    "Mill": "Mill/Mill.gltf",
    "Sponza": "Sponza/Sponza.gltf",
    "SponzaReduced": "Sponza/SponzaReduced.gltf",
}
_f32 = np.float32
_scaled = lambda r, g, b: tuple(_f32(x) * _f32(0.8) for x in (r, g, b))  # glm::vec3(r, g, b) * 0.8f
OVERRIDES = {  # fields set on a default MaterialAttributes, per look
    "Dielectric": dict(albedo=_scaled(1.0, 0.73, 0.05), roughness=0.05),  # :559-563, :653-656
    "Metallic": dict(albedo=_scaled(1.0, 0.73, 0.05), metallic=1.0, roughness=0.05),  # :581-586, :675-679
    "Glass": dict(albedo=_scaled(1.0, 0.15, 0.0), roughness=0.05, transmission=1.0),  # :604-609, :698-702
    "Clearcoat": dict(albedo=_scaled(0.0, 0.4, 1.0), metallic=1.0, roughness=0.8, clearcoat=1.0,
                      clearcoatRoughness=0.01),  # :627-634, :721-727
}
STAND_IN_SCENES = ["Interior"]  # not one of the application's scenes: see interior_room
REFERENCE_SCENES = list(GLTF_SCENES) + [m + look for m in ("Bunny", "Dragon") for look in OVERRIDES]


def _find(content_dir, relative):
    """Path of Content/Models/<relative>, matching the file name without regard to case (the
    reference runs on a case-insensitive file system: it asks for Bunny.glb, the file is bunny.glb)."""
    if not content_dir:
        return None
    folder = os.path.join(content_dir, "Models", os.path.dirname(relative))
    if not os.path.isdir(folder):
        return None
    want = os.path.basename(relative).lower()
    for entry in os.listdir(folder):
        if entry.lower() == want:
            return os.path.join(folder, entry)
    return None


def _gltf_model(path, max_texture_size, textures):
    """-> (submeshes, MaterialAlign records); the model's textures are appended to `textures` with
    table slots that continue after the entries already there."""
    from . import gltf
    model = gltf.GltfScene.from_npz(path) if path.endswith(".npz") else gltf.load(path)
    mats, tex = model.materials(max_texture_size)
    shift = len(textures)
    for rec in mats:
        rec["handles"][rec["handles"] > 0] += shift
    textures += [(slot + shift, texels, srgb) for slot, texels, srgb in tex]
    return model.submeshes, list(mats)


def _quad(corners, normal, uv_scale):
    """One inward-facing wall piece: 4 corners -> 2 triangles whose AB x AC points along `normal` (the
    shader hits front faces only: intersection.glsl:26-29), uv = the two in-plane coordinates x uv_scale."""
    c = np.asarray(corners, np.float64)
    n = np.asarray(normal, np.float64)
    if np.dot(np.cross(c[1] - c[0], c[2] - c[0]), n) < 0.0:
        c = c[::-1].copy()
    axes = [k for k in range(3) if n[k] == 0.0]
    v = np.zeros((4, 8), np.float32)
    v[:, 0:3] = c
    v[:, 3:6] = n
    v[:, 6] = c[:, axes[0]] * uv_scale
    v[:, 7] = c[:, axes[1]] * uv_scale
    return v, np.array([0, 1, 2, 0, 2, 3], np.uint32)


def _merge(parts):
    vs, idx, base = [], [], 0
    for v, i in parts:
        vs.append(v)
        idx.append(i + np.uint32(base))
        base += v.shape[0]
    return np.concatenate(vs), np.concatenate(idx)


# camera of the interior stand-in: from a corner of the room towards the lit objects and the window wall
INTERIOR_CAMERA = dict(eye=(5.6, 2.6, -4.9), look_at=(-2.0, 1.3, 3.0))


def interior_room(max_texture_size=None):
    """Stand-in for the application's Fireplace scene (PathTracingApplication.cpp:515-523; its .bin is not
    available): a closed 14 x 5 x 15 room whose only opening is a window in the wall the sky's sun shines
    on, so the HDRI reaches the inside through that window alone (almost every shadow ray is occluded);
    inside, the AreaLight model's emissive panel and its three transmissive solids, and the Mill model's
    building at model-railway scale; floor, walls, ceiling and the building wear the Mill model's three
    sRGB base-colour textures, tiled.  Same return value as load_scene."""
    textures, notes = [], ["Interior: stand-in for Fireplace (room + window + AreaLight and Mill content)"]
    mill_meshes, mill_mats = _gltf_model(os.path.join(FIXTURE_DIR, "mill.npz"), max_texture_size, textures)
    light_meshes, light_mats = _gltf_model(os.path.join(FIXTURE_DIR, "arealight.npz"), max_texture_size, textures)
    x0, x1, y0, y1, z0, z1 = -7.0, 7.0, 0.0, 5.0, -6.0, 9.0
    wy0, wy1, wz0, wz1 = 1.2, 3.8, -2.0, 3.5  # the window, in the wall x = x0
    floor = [_quad([(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], (0, 1, 0), 0.25)]
    ceiling = [_quad([(x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)], (0, -1, 0), 0.25)]
    walls = [
        _quad([(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)], (-1, 0, 0), 0.25),
        _quad([(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0)], (0, 0, 1), 0.25),
        _quad([(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], (0, 0, -1), 0.25),
        # the window wall: four pieces around the opening
        _quad([(x0, y0, z0), (x0, y1, z0), (x0, y1, wz0), (x0, y0, wz0)], (1, 0, 0), 0.25),
        _quad([(x0, y0, wz1), (x0, y1, wz1), (x0, y1, z1), (x0, y0, z1)], (1, 0, 0), 0.25),
        _quad([(x0, y0, wz0), (x0, wy0, wz0), (x0, wy0, wz1), (x0, y0, wz1)], (1, 0, 0), 0.25),
        _quad([(x0, wy1, wz0), (x0, y1, wz0), (x0, y1, wz1), (x0, wy1, wz1)], (1, 0, 0), 0.25),
    ]
    tex_of = [int(m["handles"][scenes.TEXTURE_SLOTS.index("albedo")]) for m in mill_mats]  # ground, terrain, building
    surface = lambda slot: scenes.material(albedo=(1.0, 1.0, 1.0), specular=0.0, specularTint=1.0, roughness=1.0,
                                           textures={"albedo": slot})
    building_v, building_i = mill_meshes[2]
    building_v = building_v.copy()
    building_v[:, 0:3] = building_v[:, 0:3] * np.float32(0.15) + np.array([0.5, 0.02, -3.6], np.float32)
    meshes = [_merge(floor), _merge(walls), _merge(ceiling), (building_v, building_i)] + list(light_meshes[1:])
    mats = [surface(tex_of[0]), surface(tex_of[1]), surface(tex_of[2]), mill_mats[2]] + list(light_mats[1:])
    return meshes, np.array(mats, dtype=scenes.MATERIAL_DTYPE), textures, notes


def load_scene(name, content_dir=None, max_texture_size=None):
    """-> (meshes [(vertices (n, 8) f32, indices u32)], materials (MaterialAlign array, one row per
    sub-mesh in scene-node order), textures [(table slot, uint8 texels, is_srgb)], notes [str])."""
    notes, textures = [], []
    if name == "Interior":
        return interior_room(max_texture_size)
    if name in GLTF_SCENES:
        path = _find(content_dir, GLTF_SCENES[name])
        if path is None:
            path = os.path.join(FIXTURE_DIR, name.lower() + ".npz")
            if not os.path.exists(path):
                raise FileNotFoundError(f"scene {name}: needs Content/Models/{GLTF_SCENES[name]} with its .bin and "
                                        f"images; pass the application's Content directory")
            notes.append(f"{name}: using the array dump committed with this package")
        meshes, mats = _gltf_model(path, max_texture_size, textures)
        return meshes, np.array(mats, dtype=scenes.MATERIAL_DTYPE), textures, notes
    for model_name in ("Bunny", "Dragon"):
        if name.startswith(model_name) and name[len(model_name):] in OVERRIDES:
            look = OVERRIDES[name[len(model_name):]]
            break
    else:
        raise KeyError(f"no such scene: {name} (known: {', '.join(REFERENCE_SCENES)})")
    meshes, mats = [], []
    for part in (model_name, "Floor"):
        path = _find(content_dir, part + ".glb")
        if path is not None:
            sub, m = _gltf_model(path, max_texture_size, textures)
        else:
            stand_in = {"Bunny": "bunny", "Dragon": "bunny", "Floor": "floor"}[part]
            if part == "Dragon":
                notes.append("Dragon.glb not found: the bunny stands in")
            sub = [scenes.load_ptm(stand_in)]
            m = [scenes.floor_material() if part == "Floor" else scenes.material()]
        if part != "Floor":  # SetMaterial(0, ...): only the first material is replaced, its textures stay
            handles = np.array(m[0]["handles"])
            m[0] = scenes.material(**look)
            m[0]["handles"] = handles
        meshes += sub
        mats += m
    return meshes, np.array(mats, dtype=scenes.MATERIAL_DTYPE), textures, notes
