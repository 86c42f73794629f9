"""ctypes view of include/bdpt.h and include/bdpt_scene.h.

Plumbing only: struct layouts, prototypes and the loader for the in-tree
``libbdpt_amd.so``.  There is no CPU fallback — if the library is missing or was
built without its HIP kernels, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbdpt_amd.so")

BDPT_MAX_DEPTH = 16
BDPT_MAX_LIGHTS = 16
PARAM_COUNTERS = 1
PARAM_DEFER_RESOLVE = 2
PARAM_NO_NEE = 4
PARAM_NO_SPLAT = 8
PARAM_NO_CONNECT = 16
PARAM_SPECULAR_FROM_LOBE = 32
PARAM_MIS_POWER = 64
PARAM_MIS_LINEAR = 128
PARAM_DEFER_TAIL = 256
PARAM_KEEP_COUNTERS = 512
PARAM_ENV_ON_MISS = 1024
PARAM_EMISSIVE_HITS = 2048
PREPARE_PRIMARY = 1
PREPARE_BMFR = 2


class Material(C.Structure):
    _fields_ = [("baseColor", C.c_float * 4), ("specular", C.c_float * 4), ("emissive", C.c_float * 3),
                ("alphaThreshold", C.c_float), ("IoR", C.c_float), ("flags", C.c_uint32),
                ("texBaseColor", C.c_int16), ("texSpecular", C.c_int16), ("texEmissive", C.c_int16),
                ("texNormal", C.c_int16)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32),
                ("srgb", C.c_uint32), ("reserved", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("posW", C.c_float * 3), ("type", C.c_uint32), ("dirW", C.c_float * 3),
                ("openingAngle", C.c_float), ("intensity", C.c_float * 3), ("cosOpeningAngle", C.c_float),
                ("penumbraAngle", C.c_float), ("reserved", C.c_float * 3)]


LIGHT_POINT, LIGHT_DIRECTIONAL = 0, 1
BMFR_PREPROCESS, BMFR_REGRESSION, BMFR_POSTPROCESS, BMFR_KEEP_LD_FEATURES, BMFR_FULL_FRAME = 1, 2, 4, 8, 16


class BmfrParams(C.Structure):
    _fields_ = [("frameNumber", C.c_uint32), ("flags", C.c_uint32), ("prevViewProj", C.c_float * 16)]


class SceneDesc(C.Structure):
    _fields_ = [("numVertices", C.c_uint32), ("numTriangles", C.c_uint32), ("numMaterials", C.c_uint32),
                ("numTextures", C.c_uint32), ("numLights", C.c_uint32), ("reserved", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("bitangents", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_uint32)), ("triMaterial", C.POINTER(C.c_uint32)),
                ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)),
                ("lights", C.POINTER(Light))]


class Camera(C.Structure):
    _fields_ = [("posW", C.c_float * 3), ("cameraU", C.c_float * 3), ("cameraV", C.c_float * 3),
                ("cameraW", C.c_float * 3)]


class Params(C.Structure):
    _fields_ = [("minT", C.c_float), ("frameCount", C.c_uint32), ("matIndex", C.c_uint32),
                ("refractiveIndex", C.c_float), ("maxDepth", C.c_uint32), ("emitMult", C.c_float),
                ("clampUpper", C.c_float), ("pixelJitter", C.c_float * 2), ("flags", C.c_uint32)]


class GBufferParams(C.Structure):
    _fields_ = [("pixelJitter", C.c_float * 2), ("lensRadius", C.c_float), ("focalLen", C.c_float),
                ("frameCount", C.c_uint32), ("useThinLens", C.c_uint32), ("envWidth", C.c_uint32),
                ("envHeight", C.c_uint32), ("envMap", C.c_void_p), ("envColor", C.c_float * 4)]


class GBuffer(C.Structure):
    _fields_ = [("worldPosition", C.c_void_p), ("worldNormal", C.c_void_p), ("materialDiffuse", C.c_void_p),
                ("materialSpecRough", C.c_void_p), ("materialExtraParams", C.c_void_p), ("emissive", C.c_void_p)]


class Tile(C.Structure):
    _fields_ = [("y0", C.c_uint32), ("y1", C.c_uint32)]


class Stripes(C.Structure):
    _fields_ = [("stripeRows", C.c_uint32), ("numOwners", C.c_uint32), ("owner", C.c_uint32)]


class TileInfo(C.Structure):
    _fields_ = [("numRows", C.c_uint32), ("numPixels", C.c_uint32), ("chunkRows", C.c_uint32), ("numRowRanges", C.c_uint32),
                ("splatU64", C.c_uint64), ("chunkU64", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "raysPrimary", "raysEyeExtend", "raysLightExtend", "raysNee", "raysSplat", "raysConnect",
        "nodeVisitsClosest", "triTestsClosest", "nodeVisitsShadow", "triTestsShadow", "pixelsValid",
        "splatsLanded", "raysConnectLazy", "alphaTestsClosest", "alphaTestsShadow", "hintedNee", "hintedSplat")]

    def total_rays(self):
        return (self.raysPrimary + self.raysEyeExtend + self.raysLightExtend + self.raysNee + self.raysSplat +
                self.raysConnect)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Environment(C.Structure):
    _fields_ = [("envMap", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("color", C.c_float * 4)]


class BvhInfo(C.Structure):
    _fields_ = [("numNodes", C.c_uint32), ("numTriangles", C.c_uint32), ("maxDepth", C.c_uint32),
                ("nodeBytes", C.c_uint32), ("triBytes", C.c_uint32), ("sahCost", C.c_float), ("maxStack", C.c_uint32),
                ("reserved", C.c_uint32), ("numReferences", C.c_uint32), ("numDropped", C.c_uint32),
                ("numAlphaMode", C.c_uint32), ("numAlwaysPass", C.c_uint32)]


# name -> (restype, argtypes); every symbol include/*.h declares
PROTOTYPES = {
    "bdpt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bdpt_destroy": (None, [C.c_void_p]),
    "bdpt_last_error": (C.c_char_p, [C.c_void_p]),
    "bdpt_set_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "bdpt_get_bvh_info": (C.c_int, [C.c_void_p, C.POINTER(BvhInfo)]),
    "bdpt_set_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "bdpt_set_environment": (C.c_int, [C.c_void_p, C.POINTER(Environment)]),
    "bdpt_bvh_build_check": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(BvhInfo), C.c_char_p, C.c_uint32]),
    "bdpt_host_bvh_create": (C.c_void_p, [C.POINTER(SceneDesc), C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(BvhInfo)]),
    "bdpt_host_bvh_destroy": (None, [C.c_void_p]),
    "bdpt_host_bvh_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "bdpt_camera_look_at": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]),
    "bdpt_msaa_jitter": (None, [C.c_uint32, C.POINTER(C.c_float)]),
    "bdpt_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, Tile, C.c_uint32]),
    "bdpt_gbuffer_execute": (C.c_int, [C.c_void_p, C.POINTER(GBufferParams), C.POINTER(GBuffer), C.c_void_p]),
    "bdpt_execute": (C.c_int, [C.c_void_p, C.POINTER(Params), C.POINTER(GBuffer), C.c_void_p, C.c_void_p]),
    "bdpt_execute_tail": (C.c_int, [C.c_void_p, C.POINTER(Params), C.POINTER(GBuffer), C.c_void_p, C.c_void_p]),
    "bdpt_prepare": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bdpt_resize_stripes": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, Stripes, C.c_uint32]),
    "bdpt_stripe_rows": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "bdpt_get_tile_info": (C.c_int, [C.c_void_p, C.POINTER(TileInfo)]),
    "bdpt_tile_row_ranges": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]),
    "bdpt_resolve_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bdpt_accumulate_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "bdpt_test_tree_builder": (C.c_int, [C.c_int]),
    "bdpt_bvh_recs_hash": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_uint64), C.POINTER(BvhInfo)]),
    "bdpt_bvh_build_hash": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_uint64), C.POINTER(BvhInfo)]),
    "bdpt_splat_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "bdpt_set_splat_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bdpt_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bdpt_bmfr_execute": (C.c_int, [C.c_void_p, C.POINTER(BmfrParams), C.POINTER(GBuffer), C.c_void_p, C.c_void_p]),
    "bdpt_bmfr_reset": (C.c_int, [C.c_void_p]),
    "bdpt_bmfr_history_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bdpt_bmfr_save_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bdpt_bmfr_load_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bdpt_tile_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bdpt_tile_unpack": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bdpt_camera_view_proj": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "bdpt_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.c_void_p]),
    "bdpt_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "bdpt_get_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    "bdpt_enable_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "bdpt_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bdpt_test_rng": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bdpt_test_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "bdpt_test_trace_shadow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bdpt_test_bsdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "bdpt_scene_create_cornell": (C.c_void_p, []),
    "bdpt_scene_create_atrium": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "bdpt_scene_create_atrium_uneven": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "bdpt_scene_create_courtyard": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_float]),
    "bdpt_scene_create_soup": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_float]),
    "bdpt_scene_load": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_uint32]),
    "bdpt_scene_load_threads": (C.c_int, [C.c_int]),
    "bdpt_scene_destroy": (None, [C.c_void_p]),
    "bdpt_image_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64,
                                  C.c_char_p, C.c_uint32]),
    "bdpt_image_load_hdr": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint32]),
    "bdpt_scene_get_desc": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "bdpt_scene_get_camera": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(Camera)]),
}

_lib = None


def load_library(path=None):
    """Load libbdpt_amd.so and bind every prototype.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "There is no CPU fallback for the render pass.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
