import numpy as np

from pathtracer_amd import image_io


def test_tonemap_matches_shader_formula_and_is_monotone():
    x = np.linspace(0, 8, 64, dtype=np.float32)
    grey = np.stack([x, x, x], -1)
    out = image_io.tonemap(grey, exposure=1.0, srgb=False)
    assert out.min() >= 0.0 and out.max() <= 1.0
    assert np.all(np.diff(out[:, 1]) >= -1e-6)
    # hand evaluation of Shaders/tonemapping.frag for one colour
    c = np.array([0.5, 0.25, 1.0], np.float64)
    a_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    a_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    v = a_in @ c
    v = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
    want = np.clip(a_out @ v, 0, 1)
    got = image_io.tonemap(c.astype(np.float32)[None], srgb=False)[0]
    assert np.allclose(got, want, atol=1e-5)
    # sRGB encoding on top
    enc = image_io.tonemap(c.astype(np.float32)[None], srgb=True)[0]
    assert np.allclose(enc, np.where(want <= 0.0031308, want * 12.92, 1.055 * want ** (1 / 2.4) - 0.055), atol=1e-5)


def test_oracle_display_pass_agrees_with_the_numpy_transform():
    """The oracle's restatement of the display pass (tonemapping.frag:36-71 + sRGB encoding + 8-bit
    conversion, built on its fixed-polynomial pow) against the NumPy transform of image_io, whose pow is
    libm's: at most 1 of 255 apart, almost always equal."""
    from oracle import binding as ob
    rng = np.random.default_rng(3)
    hdr = np.concatenate([rng.uniform(0, 4, (64, 64, 4)), rng.uniform(0, 0.02, (64, 64, 4)), rng.uniform(0, 300, (64, 64, 4))]).astype(np.float32)
    hdr[0, 0] = 0.0
    hdr[0, 1] = (1e15, 0.0, 1e-30, 1.0)
    for exposure in (1.0, 0.25, 3.0):
        got = ob.display(hdr, exposure).astype(np.int32)
        want = (np.clip(image_io.tonemap(hdr[..., :3], exposure), 0.0, 1.0) * 255.0 + 0.5).astype(np.int32)
        assert np.abs(got - want).max() <= 1
        assert (got != want).mean() < 1e-3
    assert tuple(ob.display(np.zeros((1, 1, 4), np.float32))[0, 0]) == (0, 0, 0)
    assert tuple(ob.display(np.full((1, 1, 4), 1e6, np.float32))[0, 0]) == (255, 255, 255)


def test_png_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (7, 5, 3)).astype(np.float32)
    p = tmp_path / "a.png"
    image_io.write_png(str(p), img)
    back = np.asarray(Image.open(p), np.float32) / 255.0
    assert back.shape == (7, 5, 3)
    assert np.max(np.abs(back[::-1] - img)) <= 0.5 / 255 + 1e-6  # row 0 of ours is the bottom row


def test_pfm_and_exr_roundtrip_bit_exact(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.normal(0, 10, (6, 9, 3)).astype(np.float32)
    image_io.write_pfm(str(tmp_path / "a.pfm"), img)
    assert image_io.read_pfm(str(tmp_path / "a.pfm")).tobytes() == img.tobytes()
    image_io.write_exr(str(tmp_path / "a.exr"), img)
    assert image_io.read_exr(str(tmp_path / "a.exr")).tobytes() == img.tobytes()
    rgba = rng.normal(0, 1, (4, 3, 4)).astype(np.float32)
    image_io.write_exr(str(tmp_path / "b.exr"), rgba)
    assert image_io.read_exr(str(tmp_path / "b.exr")).tobytes() == rgba.tobytes()
    blob = open(tmp_path / "a.exr", "rb").read()
    assert blob[:4] == bytes([0x76, 0x2F, 0x31, 0x01])  # OpenEXR magic
