"""Display transform and image writers for the headless driver.

tonemap(): the reference's post-processing pass, Shaders/tonemapping.frag:21-71 (ACES fitted curve
after exposure) followed by the sRGB encoding the reference gets from GL_FRAMEBUFFER_SRGB
(Runtime/Renderer/Renderer.cpp:25).  Writers: 8-bit PNG (zlib), 32-bit float PFM, and an
uncompressed scan-line OpenEXR (float RGB or RGBA), all from first principles (no imaging
libraries).  Images are (height, width, channels) arrays whose row 0 is the BOTTOM row, the
orientation of the accumulation images (GL image origin).
"""
import struct
import zlib

import numpy as np

_ACES_IN = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]], np.float32)
_ACES_OUT = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]],
                     np.float32)


def tonemap(hdr_rgb, exposure=1.0, srgb=True):
    """hdr_rgb: (..., 3) linear radiance -> display-referred values in [0, 1]."""
    c = np.asarray(hdr_rgb, np.float32)[..., :3] * np.float32(exposure)
    # `color * mat3(a, b, c)` is a row vector times a matrix whose COLUMNS are a, b, c
    c = np.stack([c @ _ACES_IN[0], c @ _ACES_IN[1], c @ _ACES_IN[2]], -1)
    a = c * (c + np.float32(0.0245786)) - np.float32(0.000090537)
    b = c * (np.float32(0.983729) * c + np.float32(0.4329510)) + np.float32(0.238081)
    c = a / b
    c = np.stack([c @ _ACES_OUT[0], c @ _ACES_OUT[1], c @ _ACES_OUT[2]], -1)
    c = np.clip(c, 0.0, 1.0)
    if srgb:
        c = np.where(c <= 0.0031308, c * 12.92, 1.055 * np.power(np.maximum(c, 1e-12), 1.0 / 2.4) - 0.055)
    return c.astype(np.float32)


def write_png(path, rgb01):
    """rgb01: (h, w, 3|4) in [0, 1], row 0 = bottom."""
    img = np.asarray(rgb01)
    img = (np.clip(img, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)[::-1]  # PNG rows run top to bottom
    h, w, ch = img.shape
    assert ch in (3, 4)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 6, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_pfm(path, rgb):
    """rgb: (h, w, 3) float32, row 0 = bottom -- which is PFM's own row order."""
    img = np.ascontiguousarray(np.asarray(rgb, np.float32)[..., :3])
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img.astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4")
    return data.reshape(h, w, 3).astype(np.float32)


def write_exr(path, img):
    """Uncompressed scan-line OpenEXR, 32-bit float channels.  img: (h, w, 3|4), row 0 = bottom."""
    a = np.asarray(img, np.float32)[::-1]  # EXR scan lines run top to bottom (increasing y)
    h, w, ch = a.shape
    names = ["B", "G", "R"] if ch == 3 else ["A", "B", "G", "R"]  # channels are stored alphabetically
    index = {"R": 0, "G": 1, "B": 2, "A": 3}

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0") +
              attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) +
              attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    magic = struct.pack("<ii", 20000630, 2)
    line_bytes = len(names) * w * 4
    table_pos = len(magic) + len(header)
    first = table_pos + 8 * h
    offsets = b"".join(struct.pack("<Q", first + y * (8 + line_bytes)) for y in range(h))
    with open(path, "wb") as f:
        f.write(magic + header + offsets)
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes))
            for n in names:
                f.write(np.ascontiguousarray(a[y, :, index[n]]).astype("<f4").tobytes())


def read_exr(path):
    """Reader for the files write_exr produces (uncompressed, float, scan-line)."""
    blob = open(path, "rb").read()
    assert struct.unpack("<i", blob[:4])[0] == 20000630
    pos = 8
    attrs = {}
    while blob[pos] != 0:
        end = blob.index(b"\0", pos)
        name = blob[pos:end].decode()
        pos = end + 1
        end = blob.index(b"\0", pos)
        typ = blob[pos:end].decode()
        pos = end + 1
        size = struct.unpack("<i", blob[pos:pos + 4])[0]
        pos += 4
        attrs[name] = (typ, blob[pos:pos + size])
        pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names = []
    c = attrs["channels"][1]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p)
        names.append(c[p:e].decode())
        p = e + 1 + 16
    assert attrs["compression"][1] == b"\0"
    offs = struct.unpack("<%dQ" % h, blob[pos:pos + 8 * h])
    out = np.zeros((h, w, len(names)), np.float32)
    index = {"R": 0, "G": 1, "B": 2, "A": 3}
    for y in range(h):
        q = offs[y]
        yy, size = struct.unpack("<ii", blob[q:q + 8])
        q += 8
        for n in names:
            out[yy, :, index[n]] = np.frombuffer(blob, "<f4", w, q)
            q += 4 * w
    return out[::-1]
