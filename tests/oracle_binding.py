"""ctypes binding of oracle/liboracle.so — the CPU checker.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")
ORACLE_BRUTE_FORCE = 1
ORACLE_CONNECT_ALL_VISIBLE = 2


class OracleFrame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("y0", C.c_uint32), ("y1", C.c_uint32),
                ("worldPosition", C.c_void_p), ("worldNormal", C.c_void_p), ("materialDiffuse", C.c_void_p),
                ("materialSpecRough", C.c_void_p), ("materialExtra", C.c_void_p), ("emissive", C.c_void_p),
                ("out", C.c_void_p), ("splat", C.c_void_p)]


_lib = None


def load_oracle(abi):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_PATH):
        raise RuntimeError(f"{ORACLE_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(ORACLE_PATH)
    lib.oracle_scene_create.restype = C.c_void_p
    lib.oracle_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    lib.oracle_scene_destroy.argtypes = [C.c_void_p]
    lib.oracle_set_environment.argtypes = [C.c_void_p, C.POINTER(abi.Environment)]
    lib.oracle_gbuffer.restype = C.c_int
    lib.oracle_gbuffer.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.GBufferParams), C.c_void_p,
                                   C.POINTER(OracleFrame), C.c_uint32, C.c_int]
    lib.oracle_bdpt.restype = C.c_int
    lib.oracle_bdpt.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Params), C.POINTER(OracleFrame),
                                C.c_uint32, C.c_int, C.POINTER(abi.Counters)]
    lib.oracle_resolve.restype = C.c_int
    lib.oracle_resolve.argtypes = [C.POINTER(OracleFrame)]
    lib.oracle_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.oracle_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.oracle_bsdf.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.oracle_sincos2pi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.oracle_half_round.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.oracle_bmfr_create.restype = C.c_void_p
    lib.oracle_bmfr_create.argtypes = [C.c_uint32, C.c_uint32]
    lib.oracle_bmfr_destroy.argtypes = [C.c_void_p]
    lib.oracle_bmfr_reset.argtypes = [C.c_void_p]
    lib.oracle_bmfr_execute.restype = C.c_int
    lib.oracle_bmfr_execute.argtypes = [C.c_void_p, C.POINTER(abi.BmfrParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleRender:
    """Full-frame host buffers + the three oracle passes, mirroring the product's frame loop."""

    def __init__(self, abi, scene_desc, width, height, y0=0, y1=None):
        self.abi = abi
        self.lib = load_oracle(abi)
        self.scene = self.lib.oracle_scene_create(C.byref(scene_desc))
        assert self.scene, "oracle_scene_create failed"
        self.W, self.H = width, height
        self.y0, self.y1 = y0, height if y1 is None else y1
        n = width * height
        self.chan = {k: np.zeros((n, 4), np.float32) for k in
                     ("worldPosition", "worldNormal", "materialDiffuse", "materialSpecRough", "materialExtra",
                      "emissive", "out")}
        self.splat = np.zeros((n, 4), np.uint64)
        self.frame = OracleFrame(width, height, self.y0, self.y1, _p(self.chan["worldPosition"]),
                                 _p(self.chan["worldNormal"]), _p(self.chan["materialDiffuse"]),
                                 _p(self.chan["materialSpecRough"]), _p(self.chan["materialExtra"]),
                                 _p(self.chan["emissive"]), _p(self.chan["out"]), _p(self.splat))

    def close(self):
        if self.scene:
            self.lib.oracle_scene_destroy(self.scene)
            self.scene = None

    def set_environment(self, env_map=None, color=(0.0, 0.0, 0.0, 0.0)):
        """env_map: float32 [h, w, 4] host array (lat-long) or None for the constant colour."""
        e = self.abi.Environment()
        if env_map is not None:
            self._env_keep = np.ascontiguousarray(env_map, np.float32)
            e.envMap = self._env_keep.ctypes.data
            e.height, e.width = self._env_keep.shape[0], self._env_keep.shape[1]
        for i in range(4):
            e.color[i] = float(color[i])
        self.lib.oracle_set_environment(self.scene, C.byref(e))

    def gbuffer(self, cam, gparams, env=None, flags=0, threads=8):
        rc = self.lib.oracle_gbuffer(self.scene, C.byref(cam), C.byref(gparams), _p(env) if env is not None else None,
                                     C.byref(self.frame), flags, threads)
        assert rc == 0

    def bdpt(self, cam, params, flags=0, threads=8, zero_splat=True):
        if zero_splat:
            self.splat[:] = 0
        cnt = self.abi.Counters()
        rc = self.lib.oracle_bdpt(self.scene, C.byref(cam), C.byref(params), C.byref(self.frame), flags, threads,
                                  C.byref(cnt))
        assert rc == 0, rc
        return cnt

    def resolve(self):
        assert self.lib.oracle_resolve(C.byref(self.frame)) == 0

    def image(self):
        return self.chan["out"].reshape(self.H, self.W, 4)


class OracleBmfr:
    """The denoise pass of oracle/bmfr_oracle.cpp with its own history, mirroring bdpt_bmfr_execute."""

    def __init__(self, abi, width, height):
        self.lib = load_oracle(abi)
        self.h = self.lib.oracle_bmfr_create(width, height)
        self.W, self.H = width, height

    def reset(self):
        self.lib.oracle_bmfr_reset(self.h)

    def execute(self, params, cur_pos, cur_norm, albedo, noisy):
        """All arrays float32 [H*W, 4], C-contiguous; `noisy` is updated in place."""
        for a in (cur_pos, cur_norm, albedo, noisy):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == self.W * self.H * 4
        rc = self.lib.oracle_bmfr_execute(self.h, C.byref(params), _p(cur_pos), _p(cur_norm), _p(albedo), _p(noisy))
        assert rc == 0
        return noisy

    def close(self):
        if self.h:
            self.lib.oracle_bmfr_destroy(self.h)
            self.h = None
