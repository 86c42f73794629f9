"""Writes a small OBJ/MTL/PPM/TGA/.fscene scene set into a directory (test input for the scene loader).

Everything is generated from the seed below; nothing here comes from the reference tree.  The scene exercises
each import rule the loader restates: Kd/Ks/Ns/Ke/d, map_Kd (sRGB PPM), an alpha-carrying TGA (-> alpha mask),
map_bump (-> normal map slot, linear), map_Ks (dropped, reference bug), `.doublesided`, a model without
normals/texcoords/mtl (smooth normals, default material), polygons (fan), negative indices, two instances with
translation / yaw-pitch-roll rotation in degrees / non-uniform scaling, spot + directional lights, two cameras.
"""
import json
import os
import struct

import numpy as np


def write_ppm(path, img):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n# generated\n%d %d\n255\n" % (w, h))
        f.write(img.astype(np.uint8).tobytes())


def write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.astype(np.uint8).tobytes())


def write_tga(path, rgba, rle=False, top_down=False):
    """32-bit BGRA TGA, bottom-up by default (image row 0 is the TOP row of `rgba`)."""
    h, w, _ = rgba.shape
    rows = rgba if top_down else rgba[::-1]
    bgra = rows[..., [2, 1, 0, 3]].astype(np.uint8)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, 32, (0x20 if top_down else 0) | 8)
    with open(path, "wb") as f:
        f.write(hdr)
        if not rle:
            f.write(bgra.tobytes())
        else:
            flat = bgra.reshape(-1, 4)
            i = 0
            while i < len(flat):
                run = 1
                while i + run < len(flat) and run < 128 and (flat[i + run] == flat[i]).all():
                    run += 1
                if run > 1:
                    f.write(bytes([0x80 | (run - 1)]) + flat[i].tobytes())
                    i += run
                else:
                    lit = 1
                    while i + lit < len(flat) and lit < 128 and not (flat[i + lit] == flat[i + lit - 1]).all():
                        lit += 1
                    f.write(bytes([lit - 1]) + flat[i:i + lit].tobytes())
                    i += lit


def checker(n=32, a=(200, 190, 170), b=(60, 70, 90), cells=4):
    y, x = np.mgrid[0:n, 0:n]
    m = ((x * cells // n) + (y * cells // n)) % 2
    img = np.where(m[..., None] == 0, np.array(a), np.array(b))
    return img.astype(np.uint8)


def lattice_rgba(n=32):
    y, x = np.mgrid[0:n, 0:n]
    hole = ((x % 8) >= 3) & ((y % 8) >= 3)
    img = np.zeros((n, n, 4), np.uint8)
    img[..., 0] = 150
    img[..., 1] = 110
    img[..., 2] = 60
    img[..., 3] = np.where(hole, 0, 255)
    return img


def bumps_rgb(n=32, seed=7):
    rng = np.random.default_rng(seed)
    h = rng.random((n, n)).astype(np.float32)
    dx = np.roll(h, -1, 1) - np.roll(h, 1, 1)
    dy = np.roll(h, -1, 0) - np.roll(h, 1, 0)
    nrm = np.stack([-dx, -dy, np.full_like(h, 1.5)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return np.clip(nrm * 127.5 + 127.5, 0, 255).astype(np.uint8)


ROOM_MTL = """# materials for room.obj
newmtl floor
Kd 0.8 0.8 0.8
Ks 0.04 0.04 0.04
Ns 0.6
map_Kd floor.ppm
map_Ks spec_is_ignored.ppm

newmtl red
Kd 0.63 0.065 0.05
Ks 0 0 0

newmtl green
Kd 0.14 0.45 0.091
Ks 0.0 0.5 0.0
Ns 0.3

newmtl white
Kd 0.725 0.71 0.68
map_bump -bm 1.0 bumps.ppm

newmtl lamp
Kd 0 0 0
Ke 4.0 3.5 3.0

newmtl Fence.DoubleSided
Kd 1 1 1
d 0.75
map_Kd lattice.tga

newmtl black
Kd 0 0 0
Ks 0 0 0
"""


def room_obj():
    """A 4 x 3 x 4 room, open towards +z, with an emissive panel and an alpha-masked fence."""
    v, vt, vn, out = [], [], [], ["mtllib room.mtl"]

    def quad(mtl, p, n, uv=((0, 0), (1, 0), (1, 1), (0, 1))):
        base_v, base_t, base_n = len(v), len(vt), len(vn)
        v.extend(p)
        vt.extend(uv)
        vn.append(n)
        out.append(f"usemtl {mtl}")
        out.append("f " + " ".join(f"{base_v + i + 1}/{base_t + i + 1}/{base_n + 1}" for i in range(4)))

    quad("floor", [(-2, 0, 2), (2, 0, 2), (2, 0, -2), (-2, 0, -2)], (0, 1, 0), ((0, 0), (3, 0), (3, 3), (0, 3)))
    quad("white", [(-2, 3, -2), (2, 3, -2), (2, 3, 2), (-2, 3, 2)], (0, -1, 0))
    quad("white", [(-2, 0, -2), (2, 0, -2), (2, 3, -2), (-2, 3, -2)], (0, 0, 1), ((0, 0), (2, 0), (2, 1.5), (0, 1.5)))
    quad("red", [(-2, 0, 2), (-2, 0, -2), (-2, 3, -2), (-2, 3, 2)], (1, 0, 0))
    quad("green", [(2, 0, -2), (2, 0, 2), (2, 3, 2), (2, 3, -2)], (-1, 0, 0))
    quad("lamp", [(-0.5, 2.98, -0.5), (0.5, 2.98, -0.5), (0.5, 2.98, 0.5), (-0.5, 2.98, 0.5)], (0, -1, 0))
    quad("Fence.DoubleSided", [(-1.5, 0, 0.6), (0.2, 0, 0.9), (0.2, 1.4, 0.9), (-1.5, 1.4, 0.6)], (-0.17, 0, 0.98),
         ((0, 0), (2, 0), (2, 1), (0, 1)))
    quad("black", [(0.9, 0.001, 0.2), (1.7, 0.001, 0.2), (1.7, 0.001, -0.6), (0.9, 0.001, -0.6)], (0, 1, 0))
    quad("missing_material", [(0.9, 0.6, -1.9), (1.7, 0.6, -1.9), (1.7, 1.4, -1.9), (0.9, 1.4, -1.9)], (0, 0, 1))
    lines = [f"v {a:g} {b:g} {c:g}" for a, b, c in v] + [f"vt {a:g} {b:g}" for a, b in vt] + \
            [f"vn {a:g} {b:g} {c:g}" for a, b, c in vn]
    return "\n".join(["# generated room"] + lines[:0] + [out[0]] + lines + out[1:]) + "\n"


def pillar_obj(sides=5):
    """A prism without normals, texcoords or materials; n-gon caps (fan triangulation) and negative indices."""
    lines = ["# generated pillar", "o pillar"]
    for y in (0.0, 1.0):
        for i in range(sides):
            a = 2 * np.pi * i / sides
            lines.append(f"v {0.25 * np.cos(a):.6f} {y:g} {0.25 * np.sin(a):.6f}")
    for i in range(sides):
        j = (i + 1) % sides
        lines.append(f"f {i + 1} {i + 1 + sides} {j + 1 + sides} {j + 1}")
    lines.append("f " + " ".join(str(-(sides - i)) for i in reversed(range(sides))))  # top cap through negative indices
    lines.append("f " + " ".join(str(i + 1) for i in range(sides)))                   # bottom cap
    return "\n".join(lines) + "\n"


FSCENE = {
    "version": 2,
    "active_camera": "Second",
    "models": [
        {"file": "room.obj", "name": "room", "material": {"shading_model": "spec_gloss"},
         "instances": [{"name": "room0", "translation": [0, 0, 0], "scaling": [1, 1, 1], "rotation": [0, 0, 0]}]},
        {"file": "pillar.obj", "name": "pillar",
         "instances": [
             {"name": "p0", "translation": [-1.2, 0.0, -1.0], "scaling": [1.0, 2.0, 1.0], "rotation": [0, 0, 0]},
             {"name": "p1", "translation": [1.1, 0.3, -0.9], "scaling": [1.5, 1.2, 0.8], "rotation": [30.0, 20.0, 10.0]}]},
    ],
    "lights": [
        {"name": "spot", "type": "point_light", "intensity": [9.0, 8.5, 8.0], "pos": [0.0, 2.6, 0.6],
         "direction": [0.0, -2.0, -0.5], "opening_angle": 60.0, "penumbra_angle": 5.0},
        {"name": "sun", "type": "dir_light", "intensity": [0.6, 0.6, 0.5], "direction": [0.3, -2.0, -1.0]},
    ],
    "cameras": [
        {"name": "First", "pos": [0, 1, 8], "target": [0, 1, 0], "up": [0, 1, 0], "focal_length": 35.0},
        {"name": "Second", "pos": [0.2, 1.4, 5.5], "target": [0.0, 1.2, 0.0], "up": [0, 1, 0], "focal_length": 24.0,
         "aspect_ratio": 1.5},
    ],
}


def write_scene_set(directory):
    """Returns the path of the .fscene."""
    d = str(directory)
    os.makedirs(d, exist_ok=True)
    write_ppm(os.path.join(d, "floor.ppm"), checker())
    write_ppm(os.path.join(d, "bumps.ppm"), bumps_rgb())
    write_ppm(os.path.join(d, "spec_is_ignored.ppm"), checker(8))
    write_tga(os.path.join(d, "lattice.tga"), lattice_rgba(), rle=True)
    with open(os.path.join(d, "room.mtl"), "w") as f:
        f.write(ROOM_MTL)
    with open(os.path.join(d, "room.obj"), "w") as f:
        f.write(room_obj())
    with open(os.path.join(d, "pillar.obj"), "w") as f:
        f.write(pillar_obj())
    path = os.path.join(d, "courtyard.fscene")
    with open(path, "w") as f:
        json.dump(FSCENE, f, indent=2)
    return path
