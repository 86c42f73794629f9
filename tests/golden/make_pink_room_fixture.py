"""Extracts the light and camera VALUES of the one scene file the reference ships
(CommonPasses/Data/pink_room/pink_room.fscene) into tests/golden/pink_room_values.json.

Run in the build container (the reference tree is not on the GPU box):  python tests/golden/make_pink_room_fixture.py
Only numbers and names are kept (lights, cameras, active camera); the model entry is dropped because the
geometry blob it names is absent from the reference tree.
"""
import json
import os

SRC = "/root/reference/src/CommonPasses/Data/pink_room/pink_room.fscene"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pink_room_values.json")

if __name__ == "__main__":
    with open(SRC) as f:
        scene = json.load(f)
    out = {"source": "CommonPasses/Data/pink_room/pink_room.fscene", "version": scene.get("version"),
           "active_camera": scene.get("active_camera"), "lights": scene.get("lights", []), "cameras": scene.get("cameras", [])}
    with open(DST, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", DST, len(out["lights"]), "lights", len(out["cameras"]), "cameras")
