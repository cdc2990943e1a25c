"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (g++), natively, without Python in the
process: the product's host code (glTF/GLB reader on well-formed and damaged files, sweep-SAH builder,
packers, HDRI cache, camera, BVH disk cache: `make -C pathtracer_amd/csrc host-asan`) and the test oracle
(`make -C oracle asan`).  The drivers are tools/host_sanitize.cpp and tools/oracle_sanitize.cpp."""
import os
import shutil
import subprocess

import pytest

from tests import test_gltf as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")


def _make(directory, target, env=None, *extra):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), target, *extra], capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-6000:])
    return r.stdout


def test_host_code_under_asan_and_ubsan(tmp_path):
    # files of the test's own making, in the scratch directory the damaged copies go to (so that a .gltf's
    # separate buffer file is found next to them)
    mats = [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0], "baseColorTexture": {"index": 0}},
             "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.5}, "KHR_materials_ior": {"ior": 1.33}}}]
    images = [{"uri": "albedo.png"}]
    textures = [{"source": 0}]
    files = []
    for k, index_type in enumerate((5121, 5123, 5125)):
        doc, blob, _ = T._quad_document(index_type=index_type, stride=16 if k else 0, materials=mats, material=0, images=images,
                                        textures=textures, with_normals=k != 1)
        files.append(T._write_glb(tmp_path, doc, blob, name=f"quad{k}"))
        files.append(T._write_gltf(tmp_path, doc, blob, name=f"sep{k}"))
        files.append(T._write_gltf(tmp_path, doc, blob, name=f"emb{k}", embed=True))
    content = "/root/reference/Source/PathTracer/Content/Models"  # the reference's own files, where mounted (read only)
    for rel in ("Torus.glb", "Cube.glb", "AreaLight/AreaLight.gltf", "Mill/Mill.gltf"):
        if os.path.exists(os.path.join(content, rel)):
            files.append(os.path.join(content, rel))
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = _make("pathtracer_amd/csrc", "host-asan", env, "SANITIZE_FILES=" + " ".join(files))
    assert "host_sanitize ok" in out, out[-2000:]


def test_oracle_under_asan_and_ubsan():
    out = _make("oracle", "asan")
    assert "oracle_sanitize ok" in out, out[-2000:]
