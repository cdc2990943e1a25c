"""Scene recipes and synthetic inputs used by the tests, the benchmark and the headless driver.

Geometry comes from PTM1 files (tests/golden/meshes/*.ptm: the reference's .glb meshes as raw
{pos,nor,uv} vertices + u32 indices, node transforms ignored exactly like the reference does,
Source/Runtime/Asset/ModelLoader.cpp:87-88).  Materials follow
Source/Runtime/Shader/Material.h:43-59 (defaults) and the scene recipes in
Source/PathTracer/PathTracingApplication.cpp:555-739.  All of the reference's HDRIs except a 1x1
black one are missing, so the environment is an analytic sky generated from a fixed formula.
"""
import os
import struct

import numpy as np

MESH_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "meshes")

# MaterialAlign, 176 B (PathTracingRenderer.h:163-193)
MATERIAL_DTYPE = np.dtype({
    "names": ["handles", "emission", "albedo", "specular", "specularTint", "metallic", "roughness", "subsurface",
              "anisotropy", "sheenRoughness", "sheenTint", "clearcoat", "clearcoatRoughness", "refraction",
              "transmission"],
    "formats": [("<u8", 11), ("<f4", 3), ("<f4", 3)] + ["<f4"] * 12,
    "offsets": [0, 96, 112] + [124 + 4 * i for i in range(12)],
    "itemsize": 176,
})

TEXTURE_SLOTS = ["emission", "albedo", "normal", "specular", "specularColor", "metallicRoughness", "sheenRoughness",
                 "sheenColor", "clearcoat", "clearcoatRoughness", "transmission"]  # Material.h:26-40


def material(**kw):
    """One MaterialAlign record with Material::MaterialAttributes defaults (Material.h:43-59)."""
    m = np.zeros((), MATERIAL_DTYPE)
    m["emission"] = (0, 0, 0)
    m["albedo"] = (1, 1, 1)
    m["specular"] = 1.0
    m["roughness"] = 1.0
    m["refraction"] = 1.5
    textures = kw.pop("textures", {})
    for k, v in kw.items():
        m[k] = v
    for name, slot in textures.items():
        m["handles"][TEXTURE_SLOTS.index(name)] = slot
    return m


def load_ptm(name):
    path = name if os.path.isabs(name) else os.path.join(MESH_DIR, name + ".ptm")
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"PTM1", path
    nv, ni = struct.unpack("<II", blob[4:12])
    v = np.frombuffer(blob, "<f4", nv * 8, 12).reshape(nv, 8).copy()
    i = np.frombuffer(blob, "<u4", ni, 12 + nv * 32).copy()
    return v, i


# --- scene recipes: list of (mesh name, material) in scene-node order -----------------------
def floor_material():
    # Floor.glb: baseColorFactor 0.8 grey, metallicFactor 0, roughness absent (1.0),
    # KHR_materials_specular.specularFactor 0 -> through ModelLoader.cpp:235-284.
    return material(albedo=(0.8, 0.8, 0.8), specular=0.0, specularTint=1.0, metallic=0.0, roughness=1.0)


def recipe(name):
    f32 = np.float32
    a = lambda r, g, b: tuple(f32(x) * f32(0.8) for x in (r, g, b))  # glm::vec3(...) * 0.8f
    if name == "BunnyClearcoat":  # PathTracingApplication.cpp:623-646
        return [("bunny", material(albedo=a(0.0, 0.4, 1.0), metallic=1.0, roughness=0.8, clearcoat=1.0,
                                   clearcoatRoughness=0.01)), ("floor", floor_material())]
    if name == "BunnyDielectric":  # :555-575
        return [("bunny", material(albedo=a(1.0, 0.73, 0.05), roughness=0.05)), ("floor", floor_material())]
    if name == "BunnyMetallic":  # :577-598
        return [("bunny", material(albedo=a(1.0, 0.73, 0.05), metallic=1.0, roughness=0.05)),
                ("floor", floor_material())]
    if name in ("BunnyGlass", "DragonGlass"):  # :600-621 and :693-714 (same material; Dragon.glb is absent)
        return [("bunny", material(albedo=a(1.0, 0.15, 0.0), roughness=0.05, transmission=1.0)),
                ("floor", floor_material())]
    if name == "TorusMixed":  # small test scene, not a reference recipe
        return [("torus", material(albedo=(0.9, 0.5, 0.2), roughness=0.3, metallic=0.5, sheenRoughness=0.4,
                                   sheenTint=0.5, clearcoat=0.5, clearcoatRoughness=0.1, transmission=0.3,
                                   anisotropy=0.6, subsurface=0.3, specularTint=0.4, emission=(0.05, 0.0, 0.0))),
                ("floor", floor_material())]
    raise KeyError(name)


def sky_hdri(width=1024, height=512):
    """Analytic equirectangular sky: horizon-to-zenith gradient, grey ground, Gaussian sun lobe.
    Row 0 is the texture's v = 0 (which EvaluateHdri maps to straight up: v = 0.5 - asin(y)/pi)."""
    j = (np.arange(width, dtype=np.float64) + 0.5) / width
    i = (np.arange(height, dtype=np.float64) + 0.5) / height
    u, v = np.meshgrid(j, i)
    phi = 2.0 * np.pi * (u - 0.5)
    elev = np.pi * (0.5 - v)  # +pi/2 at row 0
    d = np.stack([np.cos(elev) * np.cos(phi), np.sin(elev), np.cos(elev) * np.sin(phi)], -1)
    up = np.clip(d[..., 1], 0.0, 1.0)[..., None]
    sky = (1.0 - up) * np.array([0.9, 0.9, 0.95]) + up * np.array([0.25, 0.45, 0.9])
    ground = np.array([0.18, 0.16, 0.14])
    base = np.where(d[..., 1:2] >= 0.0, sky, ground)
    sun_dir = np.array([-0.45, 0.75, 0.48])
    sun_dir /= np.linalg.norm(sun_dir)
    c = np.clip(d @ sun_dir, -1.0, 1.0)
    sun = 60.0 * np.exp((c - 1.0) / 0.004)[..., None] * np.array([1.0, 0.93, 0.82])
    return (base + sun).astype(np.float32)


def flatten(recipe_list):
    """-> (list of (vertices, indices)), materials array in sub-mesh order."""
    meshes = [load_ptm(name) for name, _ in recipe_list]
    mats = np.array([m for _, m in recipe_list], dtype=MATERIAL_DTYPE)
    return meshes, mats


def sphere_field(grid=8, spacing=2.5, radius=1.0, seed=1, levels=1):
    """Synthetic divergence workload: grid x grid spheres (the reference's Icosphere mesh, midpoint
    subdivided `levels` times, vertices pushed out to the sphere) resting on the floor, one material
    per sphere drawn from a seeded PCG64 stream: roughness U[0.05, 1], metallic 0 or 1, albedo
    U[0.1, 0.9]^3.  The reference has no instancing, so every sphere is its own sub-mesh.
    -> (list of (vertices, indices)), materials array; the floor is the last sub-mesh."""
    v0, i0 = load_ptm("icosphere")
    if levels:
        v0, i0 = subdivide(v0, i0, levels)
    p = v0[:, :3].astype(np.float64)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    v0 = v0.copy()
    v0[:, :3] = p
    v0[:, 3:6] = p
    rng = np.random.default_rng(seed)
    meshes, mats = [], []
    for gz in range(grid):
        for gx in range(grid):
            v = v0.copy()
            centre = np.array([(gx - 0.5 * (grid - 1)) * spacing, radius, (gz - 0.5 * (grid - 1)) * spacing])
            v[:, :3] = (p * radius + centre).astype(np.float32)
            meshes.append((v, i0))
            mats.append(material(albedo=tuple(rng.uniform(0.1, 0.9, 3).astype(np.float32)),
                                 roughness=np.float32(rng.uniform(0.05, 1.0)), metallic=float(rng.integers(0, 2))))
    fv, fi = load_ptm("floor")
    if grid * spacing > 18.0:  # Floor.glb spans +-10
        fv = fv.copy()
        fv[:, [0, 2]] *= np.float32(grid * spacing / 18.0)
    meshes.append((fv, fi))
    mats.append(floor_material())
    return meshes, np.array(mats, dtype=MATERIAL_DTYPE)


def instanced_bunnies(copies, seed=1):
    """`copies` bunnies on a jittered square grid (offsets from a 64-bit LCG seeded with `seed`)
    over one floor scaled to fit, every copy flattened into the soup: the reference has no
    instancing.  139 copies are 20 M triangles.  -> (list of (vertices, indices)), materials
    (DragonGlass recipe: all bunnies share material row 0, the floor is row 1), grid side."""
    bv, bi = load_ptm("bunny")
    fv, fi = load_ptm("floor")
    side = int(np.ceil(np.sqrt(copies)))
    state = int(seed)
    vs, is_, k = [], [], 0
    for gz in range(side):
        for gx in range(side):
            if k == copies:
                break
            state = (state * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
            jx = (((state >> 40) & 0xFFFF) / 65535.0 - 0.5) * 0.6
            jz = (((state >> 20) & 0xFFFF) / 65535.0 - 0.5) * 0.6
            v = bv.copy()
            v[:, 0] += np.float32((gx - 0.5 * (side - 1)) * 2.2 + jx)
            v[:, 2] += np.float32((gz - 0.5 * (side - 1)) * 2.2 + jz)
            vs.append(v)
            is_.append(bi + np.uint32(k * bv.shape[0]))
            k += 1
    f = fv.copy()
    f[:, [0, 2]] *= np.float32(max(1.0, side * 2.2 / 18.0))
    mats = np.array([m for _, m in recipe("DragonGlass")], dtype=MATERIAL_DTYPE)
    return [(np.concatenate(vs), np.concatenate(is_)), (f, fi)], mats, side


def subdivide(vertices, indices, levels=1):
    """Midpoint subdivision (each triangle -> 4), attributes interpolated, normals renormalised.
    Used to grow the bunny towards the triangle count of the reference's missing Dragon mesh."""
    v, idx = vertices.astype(np.float64), indices.reshape(-1, 3).astype(np.int64)
    for _ in range(levels):
        e = np.concatenate([idx[:, [0, 1]], idx[:, [1, 2]], idx[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * (v.shape[0] + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // (v.shape[0] + 1), uniq % (v.shape[0] + 1)
        mid = 0.5 * (v[a] + v[b])
        n = mid[:, 3:6]
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        mid[:, 3:6] = np.where(ln > 0, n / np.maximum(ln, 1e-30), n)
        base = v.shape[0]
        v = np.concatenate([v, mid])
        t = idx.shape[0]
        m01, m12, m20 = base + inv[:t], base + inv[t:2 * t], base + inv[2 * t:]
        idx = np.concatenate([np.stack([idx[:, 0], m01, m20], 1), np.stack([m01, idx[:, 1], m12], 1),
                              np.stack([m20, m12, idx[:, 2]], 1), np.stack([m01, m12, m20], 1)])
    return v.astype(np.float32), idx.reshape(-1).astype(np.uint32)
