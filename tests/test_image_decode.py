"""The loader's own PNG and baseline-JPEG decoders (host/ImageDecode.cpp) against independently produced files.

PNG files are written here chunk by chunk with Python's zlib (stored, fixed-Huffman and dynamic-Huffman DEFLATE
streams; every scan-line filter type; every colour type the decoder accepts) and by Pillow; decoded texels must equal
the source exactly.  JPEG files are written by Pillow (libjpeg-turbo) at several qualities, chroma sub-samplings, sizes
that are not multiples of the MCU and with restart markers; the decoder follows the IJG release-6b arithmetic that
libjpeg-turbo keeps, so its output must equal Pillow's decode bit for bit.  Neither pins the reference's FreeImage
(not in the image; its bundled IJG 9a up-samples chroma differently): geometry/material import stays parity-unpinned.
"""
import ctypes as C
import io
import struct
import zlib

import numpy as np
import pytest

Image = pytest.importorskip("PIL.Image")


def _load(pkg, path):
    lib = pkg.load_library()
    w, h, a = C.c_uint32(), C.c_uint32(), C.c_uint32()
    msg = C.create_string_buffer(256)
    rc = lib.bdpt_image_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(a), None, 0, msg, 256)
    if rc != 0:
        return None, msg.value.decode()
    buf = np.zeros((h.value, w.value, 4), np.uint8)
    assert lib.bdpt_image_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(a), buf.ctypes.data, buf.size, msg, 256) == 0
    return buf, bool(a.value)


def _png(width, height, color_type, depth, rows, level=9, strategy=zlib.Z_DEFAULT_STRATEGY, filt=None, extra=b""):
    """rows: list of raw scan lines (bytes, unfiltered).  filt: filter type per row (None = 0)."""
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    samples = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = max(1, samples * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, line in enumerate(rows):
        ft = 0 if filt is None else filt[y % len(filt)]
        out = bytearray(len(line))
        for i, v in enumerate(line):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a
            elif ft == 2:
                p = b
            elif ft == 3:
                p = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (v - p) & 255
        raw.append(ft)
        raw += out
        prev = line
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    data = co.compress(bytes(raw)) + co.flush()
    half = len(data) // 2  # two IDAT chunks: the stream may be split anywhere
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color_type, 0, 0, 0)) + extra +
            chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (1, zlib.Z_HUFFMAN_ONLY)])
def test_png_colour_types_filters_and_deflate_block_kinds(pkg, tmp_path, level, strategy):
    rng = np.random.default_rng(7 + level)
    W, H = 37, 29
    smooth = (np.add.outer(np.arange(H), np.arange(W)) * 3 % 256).astype(np.uint8)  # compressible: long matches
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rgba[..., 0] = smooth
    rgba[5:20, 3:30, 1] = 17
    cases = {
        "rgba": (6, 8, [rgba[y].tobytes() for y in range(H)], rgba, True),
        "rgb": (2, 8, [rgba[y, :, :3].tobytes() for y in range(H)], np.dstack([rgba[..., :3], np.full((H, W), 255, np.uint8)]), False),
        "grey": (0, 8, [smooth[y].tobytes() for y in range(H)], np.dstack([smooth] * 3 + [np.full((H, W), 255, np.uint8)]), False),
        "grey_alpha": (4, 8, [np.stack([smooth[y], rgba[y, :, 3]], 1).tobytes() for y in range(H)],
                       np.dstack([smooth] * 3 + [rgba[..., 3]]), True),
    }
    for name, (ct, depth, rows, want, alpha) in cases.items():
        f = tmp_path / f"{name}.png"
        f.write_bytes(_png(W, H, ct, depth, rows, level, strategy, filt=[0, 1, 2, 3, 4]))
        got, a = _load(pkg, f)
        assert got is not None, a
        assert a == alpha and np.array_equal(got, want), name
        # Pillow reads the same file the same way (the writer above is not the only witness)
        ref = np.asarray(Image.open(f).convert("RGBA"))
        assert np.array_equal(got, ref), name


def test_png_palette_low_bit_depths_and_transparent_colour(pkg, tmp_path):
    rng = np.random.default_rng(3)
    W, H = 21, 9
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    plte = struct.pack(">I", 48) + b"PLTE" + pal.tobytes() + struct.pack(">I", zlib.crc32(b"PLTE" + pal.tobytes()) & 0xFFFFFFFF)
    for depth in (1, 2, 4, 8):
        idx = rng.integers(0, min(16, 1 << depth), (H, W), dtype=np.uint8)
        rows = []
        for y in range(H):
            bits = "".join(format(int(v), f"0{depth}b") for v in idx[y])
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        f = tmp_path / f"pal{depth}.png"
        f.write_bytes(_png(W, H, 3, depth, rows, extra=plte))
        got, a = _load(pkg, f)
        assert got is not None and not a
        assert np.array_equal(got[..., :3], pal[idx]) and (got[..., 3] == 255).all()
        g = tmp_path / f"grey{depth}.png"
        g.write_bytes(_png(W, H, 0, depth, rows))
        got, a = _load(pkg, g)
        assert not a and np.array_equal(got[..., 0], (idx.astype(np.int32) * 255 // ((1 << depth) - 1)).astype(np.uint8))
    # RGB with a tRNS colour key: FreeImage converts such files to 32 bits -> the material gets an alpha mask
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rgb[2, 3] = rgb[4, 5] = (9, 8, 7)
    body = struct.pack(">HHH", 9, 8, 7)
    trns = struct.pack(">I", 6) + b"tRNS" + body + struct.pack(">I", zlib.crc32(b"tRNS" + body) & 0xFFFFFFFF)
    f = tmp_path / "key.png"
    f.write_bytes(_png(W, H, 2, 8, [rgb[y].tobytes() for y in range(H)], extra=trns))
    got, a = _load(pkg, f)
    assert a and got[2, 3, 3] == 0 and got[4, 5, 3] == 0 and (got[..., 3] == 0).sum() == 2 and np.array_equal(got[..., :3], rgb)


def test_png_written_by_pillow_and_refusals(pkg, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 48, 4), dtype=np.uint8)
    img[10:40, 5:30] = (200, 100, 50, 128)
    for mode, arr in (("RGBA", img), ("RGB", img[..., :3]), ("L", img[..., 0]), ("LA", img[..., [0, 3]])):
        f = tmp_path / f"pil_{mode}.png"
        Image.fromarray(arr, mode).save(f, optimize=True)
        got, a = _load(pkg, f)
        assert got is not None and a == (mode in ("RGBA", "LA"))
        assert np.array_equal(got, np.asarray(Image.open(f).convert("RGBA"))), mode
    f = tmp_path / "interlaced.png"
    data = bytearray(_png(4, 4, 2, 8, [bytes(12)] * 4))
    ihdr = bytearray(data[8:8 + 25])
    ihdr[8 + 12] = 1  # interlace method
    ihdr[21:25] = struct.pack(">I", zlib.crc32(bytes(ihdr[4:21])) & 0xFFFFFFFF)
    f.write_bytes(bytes(data[:8]) + bytes(ihdr) + bytes(data[33:]))
    got, why = _load(pkg, f)
    assert got is None and "interlaced" in why
    f = tmp_path / "sixteen.png"
    Image.fromarray((rng.integers(0, 65536, (8, 8))).astype(np.uint16)).save(f)
    got, why = _load(pkg, f)
    assert got is None and "16-bit" in why
    f = tmp_path / "corrupt.png"
    good = bytearray((tmp_path / "pil_RGB.png").read_bytes())
    good[len(good) // 2] ^= 0x55
    f.write_bytes(bytes(good))
    got, why = _load(pkg, f)
    assert got is None


@pytest.mark.parametrize("size", [(64, 48), (37, 29), (17, 1), (1, 23), (200, 131)])
@pytest.mark.parametrize("subsampling,quality", [(0, 90), (1, 75), (2, 50), (2, 95)])
def test_jpeg_baseline_equals_libjpeg_turbo(pkg, tmp_path, size, subsampling, quality):
    rng = np.random.default_rng(size[0] * 131 + size[1] + subsampling)
    W, H = size
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 5 + yy * 3) % 256, (xx * 2 + 40 * np.sin(yy / 5.0) + 128) % 256, (yy * 7) % 256], -1).astype(np.uint8)
    img[H // 3: H // 2 + 1, W // 4: W // 2 + 1] = rng.integers(0, 256, 3, dtype=np.uint8)
    img = np.clip(img.astype(np.int32) + rng.integers(-20, 21, img.shape), 0, 255).astype(np.uint8)
    f = tmp_path / "a.jpg"
    Image.fromarray(img, "RGB").save(f, quality=quality, subsampling=subsampling, optimize=(quality == 95))
    got, a = _load(pkg, f)
    assert got is not None, a
    ref = np.asarray(Image.open(f).convert("RGB"))
    assert not a and got.shape == (H, W, 4) and (got[..., 3] == 255).all()
    assert np.array_equal(got[..., :3], ref), f"{(got[..., :3] != ref).any(-1).sum()} of {W * H} pixels differ, max {np.abs(got[..., :3].astype(int) - ref).max()}"


def test_jpeg_greyscale_restart_markers_and_refusals(pkg, tmp_path):
    rng = np.random.default_rng(11)
    g = (rng.integers(0, 256, (45, 70))).astype(np.uint8)
    f = tmp_path / "g.jpg"
    Image.fromarray(g, "L").save(f, quality=80)
    got, a = _load(pkg, f)
    ref = np.asarray(Image.open(f).convert("L"))
    assert not a and np.array_equal(got[..., 0], ref) and np.array_equal(got[..., 1], ref) and np.array_equal(got[..., 2], ref)
    rgb = rng.integers(0, 256, (50, 90, 3), dtype=np.uint8)
    f = tmp_path / "r.jpg"
    try:
        Image.fromarray(rgb, "RGB").save(f, quality=85, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    assert b"\xff\xdd" in f.read_bytes()
    got, a = _load(pkg, f)
    assert got is not None, a
    assert np.array_equal(got[..., :3], np.asarray(Image.open(f).convert("RGB")))
    f = tmp_path / "p.jpg"
    Image.fromarray(rgb, "RGB").save(f, quality=85, progressive=True)
    got, why = _load(pkg, f)
    assert got is None and "progressive" in why


def test_obj_material_with_png_and_jpeg_textures(pkg, tmp_path):
    """The loader end to end: an OBJ whose base-colour map is a PNG with an alpha channel becomes an alpha-masked
    material (Material.cpp:120-126), one with a JPEG map stays opaque; texels arrive as decoded."""
    rng = np.random.default_rng(2)
    rgba = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "leaf.png")
    Image.fromarray(rgba[..., :3].copy(), "RGB").save(tmp_path / "wall.jpg", quality=90)
    (tmp_path / "m.mtl").write_text("newmtl leaf\nKd 1 1 1\nmap_Kd leaf.png\nnewmtl wall\nKd 1 1 1\nmap_Kd wall.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\n"
                                    "usemtl leaf\nf 1/1 2/2 3/3\nusemtl wall\nf 2/2 4/4 3/3\n")
    scene = pkg.Scene.load(tmp_path / "m.obj")
    d = scene.desc
    assert d.numTriangles == 2 and d.numTextures == 2
    mats = {int(d.triMaterial[t]) for t in range(2)}
    assert len(mats) == 2
    alpha_modes = sorted((d.materials[m].flags >> 17) & 3 for m in mats)
    assert alpha_modes == [0, 1]  # opaque (JPEG) and mask (RGBA PNG)
    for i in range(2):
        t = d.textures[i]
        px = np.ctypeslib.as_array(C.cast(t.rgba8, C.POINTER(C.c_uint8)), (t.height, t.width, 4))
        if (px[..., 3] != 255).any():
            assert np.array_equal(px, rgba)
        else:
            assert np.array_equal(px[..., :3], np.asarray(Image.open(tmp_path / "wall.jpg").convert("RGB")))
    scene.close()


def _textured_obj(tmp_path):
    """A wall with a JPEG base-colour map behind a double-sided panel whose PNG base-colour map carries alpha holes."""
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:64, 0:64]
    wall = np.stack([(xx * 4) % 256, (yy * 4) % 256, ((xx + yy) * 2) % 256], -1).astype(np.uint8)
    Image.fromarray(wall, "RGB").save(tmp_path / "wall.jpg", quality=92, subsampling=2)
    panel = np.zeros((32, 32, 4), np.uint8)
    panel[..., :3] = (210, 160, 60)
    panel[..., 3] = np.where(((xx[:32, :32] % 8) >= 3) & ((yy[:32, :32] % 8) >= 3), 0, 255)
    Image.fromarray(panel, "RGBA").save(tmp_path / "panel.png")
    (tmp_path / "t.mtl").write_text("newmtl wall\nKd 1 1 1\nKs 0.04 0.04 0.04\nNs 0.5\nmap_Kd wall.jpg\n"
                                    "newmtl Panel.DoubleSided\nKd 1 1 1\nmap_Kd panel.png\nnewmtl floor\nKd 0.6 0.6 0.6\n")
    (tmp_path / "t.obj").write_text(
        "mtllib t.mtl\n"
        "v -2 0 -2\nv 2 0 -2\nv 2 3 -2\nv -2 3 -2\n"      # wall
        "v -1 0.2 -0.5\nv 1 0.2 -0.5\nv 1 2.2 -0.5\nv -1 2.2 -0.5\n"  # panel in front of it
        "v -2 0 2\nv 2 0 2\n"                                # floor corners
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 2 0\nvt 2 2\nvt 0 2\n"
        "vn 0 0 1\nvn 0 1 0\n"
        "usemtl wall\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"
        "usemtl Panel.DoubleSided\nf 5/1/1 6/5/1 7/6/1 8/7/1\n"
        "usemtl floor\nf 9/1/2 10/2/2 2/3/2 1/4/2\n")
    import json
    (tmp_path / "t.fscene").write_text(json.dumps({
        "version": 2, "models": [{"file": "t.obj", "name": "t", "instances": [{"name": "i", "translation": [0, 0, 0], "scaling": [1, 1, 1], "rotation": [0, 0, 0]}]}],
        "lights": [{"name": "l", "type": "point_light", "intensity": [12.0, 11.0, 10.0], "pos": [0.3, 2.6, 1.2], "direction": [0, -1, 0],
                    "opening_angle": 180.0, "penumbra_angle": 0.0}],
        "cameras": [{"name": "c", "pos": [0.3, 1.3, 3.6], "target": [0.0, 1.2, -1.0], "up": [0, 1, 0], "focal_length": 21.0}]}))
    return tmp_path / "t.fscene"


def test_oracle_renders_scene_with_png_and_jpeg_textures(pkg, ob, tmp_path):
    scene = pkg.Scene.load(_textured_obj(tmp_path))
    W, H = 48, 36
    orc = ob.OracleRender(pkg.abi, scene.desc, W, H)
    cam = scene.camera(W / H)
    gp = pkg.abi.GBufferParams()
    gp.frameCount, gp.lensRadius, gp.focalLen = 0xDEADBEEF, 0.0, 1.0
    gp.pixelJitter[0] = gp.pixelJitter[1] = 0.5
    orc.gbuffer(cam, gp)
    dif = orc.chan["materialDiffuse"].reshape(H, W, 4)
    hit = orc.chan["worldPosition"].reshape(H, W, 4)[..., 3] != 0
    assert hit.mean() > 0.3
    # through the panel's alpha holes the primary ray sees the JPEG-textured wall: many distinct diffuse colours
    assert len(np.unique(np.round(dif[hit][:, :3], 2), axis=0)) > 20
    orc.close()
    scene.close()


@pytest.mark.gpu
def test_png_jpeg_textured_scene_frame_matches_oracle(pkg, ob, tmp_path):
    """Loader -> decoded textures -> HIP path, against the oracle on the same descriptor: alpha-masked PNG panel
    (any-hit alpha test on primary, extension and shadow rays) over a JPEG-textured wall."""
    import torch
    scene = pkg.Scene.load(_textured_obj(tmp_path))
    for mat, depth in ((0, 4), (1, 6)):
        pipe = pkg.FramePipeline(scene, 96, 72, max_depth=depth, mat_index=mat)
        gp, p = pipe.render_frame()
        torch.cuda.synchronize()
        orc = ob.OracleRender(pkg.abi, scene.desc, pipe.W, pipe.H)
        orc.gbuffer(pipe.cam, gp)
        orc.bdpt(pipe.cam, p)
        orc.resolve()
        gpu, ref = pipe.output.cpu().numpy(), orc.image()
        g = pipe.channels["MaterialDiffuse"].float().cpu().numpy().reshape(-1, 4)
        assert np.array_equal(g.view(np.uint32), orc.chan["materialDiffuse"].view(np.uint32))
        assert np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)), f"{(gpu != ref).any(axis=-1).sum()} pixels differ"
        orc.close()
        pipe.close()
    scene.close()


def test_radiance_hdr_flat_scanline_with_a_dark_111_pixel(pkg, tmp_path):
    """A flat (uncompressed) scanline may hold a pixel whose mantissas are (1, 1, 1) — a legal dark grey and also
    Radiance's OLD-style repeat marker.  FreeImage's reader (the reference's) knows only new-style run-length scanlines
    and reads everything else as flat pixels, so the pixel decodes as a pixel: 1 * 2^(e - 136) per channel."""
    lib = pkg.load_library()
    h, w = 2, 9
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = 100
    rgbe[..., 3] = 130
    rgbe[0, 4] = (1, 1, 1, 121)   # the would-be marker, in the middle of a row
    rgbe[1, 0] = (1, 1, 1, 3)     # ... and as a row's first pixel
    path = tmp_path / "flat111.hdr"
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(rgbe.tobytes())
    ww, hh = C.c_uint32(), C.c_uint32()
    msg = C.create_string_buffer(256)
    out = np.zeros((h, w, 4), np.float32)
    assert lib.bdpt_image_load_hdr(str(path).encode(), C.byref(ww), C.byref(hh), out.ctypes.data, out.size, msg, 256) == 0, msg.value
    want = np.ones((h, w, 4), np.float32)
    want[..., :3] = (rgbe[..., :3].astype(np.float64) * np.ldexp(1.0, rgbe[..., 3].astype(np.int32) - 136)[..., None]).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert out[0, 4, 0] == np.float32(2.0 ** (121 - 136)) and out[1, 0, 2] == np.float32(2.0 ** (3 - 136))


def _write_hdr(path, img, rle, bottom_up=False, magic=b"#?RADIANCE"):
    """float image [h, w, 3] -> Radiance RGBE file (what the encoder of any HDR tool writes); returns the RGBE bytes."""
    h, w, _ = img.shape
    mx = img.max(axis=-1)
    e = np.zeros((h, w), np.int32)
    nz = mx > 1e-32
    e[nz] = np.floor(np.log2(mx[nz])).astype(np.int32) + 1
    scale = np.where(nz, 256.0 / np.exp2(e.astype(np.float64)), 0.0)
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(nz, e + 128, 0).astype(np.uint8)
    rows = rgbe[::-1] if bottom_up else rgbe
    with open(path, "wb") as f:
        f.write(magic + b"\n# written by the test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
        f.write(("%sY %d +X %d\n" % ("+" if bottom_up else "-", h, w)).encode())
        for r in rows:
            if not rle:
                f.write(r.tobytes())
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                ch = r[:, c]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and ch[x + run] == ch[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(ch[x])]))
                        x += run
                    else:
                        lit = 1
                        while x + lit < w and lit < 128 and not (x + lit + 2 < w and ch[x + lit] == ch[x + lit + 1] == ch[x + lit + 2]):
                            lit += 1
                        f.write(bytes([lit]) + ch[x:x + lit].tobytes())
                        x += lit
    return rgbe


@pytest.mark.parametrize("rle,bottom_up", [(False, False), (True, False), (True, True)])
def test_radiance_hdr_light_probe_decodes_to_the_rgbe_values(pkg, tmp_path, rle, bottom_up):
    """bdpt_image_load_hdr: what ResourceManager::updateEnvironmentMap gets for a .hdr light probe
    (SharedUtils/ResourceManager.cpp:96-110 -> FreeImage): flat and run-length scanlines, both row orders; float =
    mantissa * 2^(e - 136), alpha 1, row 0 = top."""
    lib = pkg.load_library()
    rng = np.random.default_rng(5)
    h, w = 12, 40
    img = (rng.random((h, w, 3)) ** 4 * 50.0).astype(np.float64)
    img[3:6, 5:30] = (7.5, 0.25, 1e-3)   # runs for the RLE path
    img[8, :] = 0.0                       # e = 0 -> exactly zero
    path = tmp_path / "probe.hdr"
    rgbe = _write_hdr(path, img, rle, bottom_up)
    ww, hh = C.c_uint32(), C.c_uint32()
    msg = C.create_string_buffer(256)
    assert lib.bdpt_image_load_hdr(str(path).encode(), C.byref(ww), C.byref(hh), None, 0, msg, 256) == 0, msg.value
    assert (ww.value, hh.value) == (w, h)
    out = np.zeros((h, w, 4), np.float32)
    assert lib.bdpt_image_load_hdr(str(path).encode(), C.byref(ww), C.byref(hh), out.ctypes.data, out.size, msg, 256) == 0
    want = np.ones((h, w, 4), np.float32)
    f = np.where(rgbe[..., 3] > 0, np.ldexp(1.0, rgbe[..., 3].astype(np.int32) - 136), 0.0)
    want[..., :3] = (rgbe[..., :3].astype(np.float64) * f[..., None]).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert np.abs(out[..., :3] - img).max() <= np.abs(img).max() / 128  # and those are the image, to RGBE precision
    # refusals: truncated data, a header that promises 60000 x 60000 texels, not an .hdr at all
    blob = open(path, "rb").read()
    for bad, why in ((blob[:len(blob) - 25], b"truncated"), (blob.replace(b"Y 12 +X 40", b"Y 60000 +X 60000"), b"limit"), (b"P6\n1 1\n255\nabc", b"not a Radiance")):
        open(tmp_path / "bad.hdr", "wb").write(bad)
        assert lib.bdpt_image_load_hdr(str(tmp_path / "bad.hdr").encode(), C.byref(ww), C.byref(hh), None, 0, msg, 256) != 0
        assert why in msg.value or (why == b"truncated" and b"corrupt .hdr run" in msg.value), (why, msg.value)


def test_decoders_refuse_oversized_headers_without_allocating(pkg, tmp_path):
    """A few bytes of header must not make the loader allocate gigabytes: PNG and JPEG dimensions above the
    loader's limit are refused before any plane is allocated."""
    import struct
    import zlib
    lib = pkg.load_library()
    w = h = C.c_uint32()
    msg = C.create_string_buffer(256)

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 30000, 30000, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b"")
    open(tmp_path / "big.png", "wb").write(png)
    assert lib.bdpt_image_load(str(tmp_path / "big.png").encode(), C.byref(w), C.byref(h), None, None, 0, msg, 256) != 0
    jpg = b"\xff\xd8" + b"\xff\xc0" + struct.pack(">HBHHB", 11, 8, 65535, 65535, 1) + b"\x01\x11\x00" + b"\xff\xd9"
    open(tmp_path / "big.jpg", "wb").write(jpg)
    assert lib.bdpt_image_load(str(tmp_path / "big.jpg").encode(), C.byref(w), C.byref(h), None, None, 0, msg, 256) != 0
    assert b"limit" in msg.value, msg.value


def test_reference_scene_texture_files_decode_to_their_pillow_digests(pkg):
    """The reference's OWN data for this path: the 28 texture files of its pink_room scene.  Each must decode through
    the loader (host/ImageDecode.cpp) to exactly the texels an independent decoder (Pillow) produces — committed as
    SHA-256 digests by tests/golden/make_pink_room_texture_digests.py — with the size and the 32-bit-image verdict
    (Utils/Bitmap.cpp:104-126: what makes a material's alpha mode Mask) the fixture records.  Needs the reference
    checkout (present where the CPU suite runs; the GPU box has neither the files nor this test's marker)."""
    import hashlib
    import json
    import os
    src = "/root/reference/src/CommonPasses/Data/pink_room/textures"
    if not os.path.isdir(src):
        pytest.skip("reference checkout not present")
    fixture = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pink_room_texture_digests.json")))
    lib = pkg.load_library()
    assert len(fixture) == 28
    for name, want in sorted(fixture.items()):
        path = os.path.join(src, name).encode()
        w, h, a = C.c_uint32(), C.c_uint32(), C.c_uint32()
        msg = C.create_string_buffer(256)
        assert lib.bdpt_image_load(path, C.byref(w), C.byref(h), C.byref(a), None, 0, msg, 256) == 0, (name, msg.value)
        assert (w.value, h.value, bool(a.value)) == (want["width"], want["height"], want["alpha"]), name
        px = np.zeros((h.value, w.value, 4), np.uint8)
        assert lib.bdpt_image_load(path, C.byref(w), C.byref(h), C.byref(a), px.ctypes.data, px.size, msg, 256) == 0
        texels = px if want["alpha"] else px[..., :3]
        assert hashlib.sha256(np.ascontiguousarray(texels).tobytes()).hexdigest() == want["sha256"], name
