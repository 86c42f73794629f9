#!/usr/bin/env python3
"""Writes tests/golden/pink_room_texture_digests.json: for every texture file the reference ships for its one scene
(/root/reference/src/CommonPasses/Data/pink_room/textures/*: 26 PNG + 2 JPEG; the WoodFloor base-colour pair is among
the blobs missing from the checkout), its size, whether it carries alpha, and the SHA-256 of its texels as decoded by
Pillow (libpng / libjpeg-turbo) — an independent decoder.  The loader's own PNG / JPEG code (host/ImageDecode.cpp) must
reproduce those texels bit for bit (tests/test_image_decode.py).  Data in, digests out: no reference source is read."""
import glob
import hashlib
import json
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/src/CommonPasses/Data/pink_room/textures"
out = {}
for f in sorted(glob.glob(os.path.join(SRC, "*"))):
    im = Image.open(f)
    has_alpha = im.mode in ("RGBA", "LA") or "transparency" in im.info
    px = np.asarray(im.convert("RGBA"))
    texels = px if has_alpha else px[..., :3]
    out[os.path.basename(f)] = {"width": im.size[0], "height": im.size[1], "format": im.format, "alpha": bool(has_alpha),
                                "sha256": hashlib.sha256(np.ascontiguousarray(texels).tobytes()).hexdigest()}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pink_room_texture_digests.json"), "w"), indent=1, sort_keys=True)
print(len(out), "files")
