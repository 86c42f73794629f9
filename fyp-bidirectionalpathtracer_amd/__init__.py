"""MI355X-native bidirectional path-tracing render pass — Python plumbing over the C ABI.

The product is ``csrc/libbdpt_amd.so`` (HIP kernels for gfx950 behind ``include/bdpt.h``) and
the C++ host mirror under ``host/``.  This module only provides what tests and ``bench.py``
need: ctypes bindings, torch tensors as device memory for the ResourceManager channels, HIP
streams, and the per-frame calling sequence of the reference pipeline
(SharedUtils/RenderingPipeline.cpp:666-682: G-buffer pass -> BDPT pass -> accumulation pass).
There is no CPU fallback anywhere in this package.
"""
import ctypes as C

from . import abi, tiling
from .abi import (Camera, Counters, GBuffer, GBufferParams, Params, SceneDesc, Stripes, Tile, TileInfo, load_library)

__all__ = ["abi", "tiling", "Scene", "Context", "FramePipeline", "load_library", "source_hash"]

# Channel names of the reference's ResourceManager (BDPTPass.cpp:27-29, LightProbeGBufferPass.cpp:46-51)
GBUFFER_CHANNELS = ("WorldPosition", "WorldNormal", "MaterialDiffuse", "MaterialSpecRough", "MaterialExtraParams",
                    "Emissive")
OUTPUT_CHANNEL = "PipelineOutput"  # ResourceManager::kOutputChannel, SharedUtils/ResourceManager.cpp:22


class BdptError(RuntimeError):
    pass


def source_hash():
    """SHA-256 over the sources libbdpt_amd.so is built from — exactly the Makefile's prerequisites: csrc/*.{hip,hpp,h,cpp},
    the host files linked into it (host/Scene.cpp, Atrium.cpp, SceneLoader.cpp, ImageDecode.cpp: the scene generators and
    the loader shape the measured workload), host/Scene.h, include/*.h and the Makefile.  Ties hardware-counter files
    (profiles/<round>/roofline_pmc.json) to the build they were measured on."""
    import glob
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.hpp")) +
                   glob.glob(os.path.join(here, "csrc", "*.h")) + glob.glob(os.path.join(here, "csrc", "*.cpp")) +
                   [os.path.join(here, "csrc", "Makefile")] +
                   [os.path.join(here, "host", f) for f in ("Scene.cpp", "Atrium.cpp", "SceneLoader.cpp", "ImageDecode.cpp")] +
                   [os.path.join(here, "host", "Scene.h")] + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


class Scene:
    """Host scene container (include/bdpt_scene.h)."""

    def __init__(self, handle):
        self._lib = load_library()
        if not handle:
            raise BdptError("scene creation failed")
        self._h = C.c_void_p(handle)
        self.desc = SceneDesc()
        if self._lib.bdpt_scene_get_desc(self._h, C.byref(self.desc)) != 0:
            raise BdptError("bdpt_scene_get_desc failed")

    @classmethod
    def cornell(cls):
        return cls(load_library().bdpt_scene_create_cornell())

    @classmethod
    def atrium(cls, seed=1, target_triangles=262144):
        return cls(load_library().bdpt_scene_create_atrium(seed, target_triangles))

    @classmethod
    def atrium_uneven(cls, seed=1, target_triangles=262144):
        """The atrium with heavy-tailed triangle areas (two-triangle walls and floors beside finely tessellated ornaments)."""
        return cls(load_library().bdpt_scene_create_atrium_uneven(seed, target_triangles))

    @classmethod
    def courtyard(cls, seed=1, target_triangles=262144, foliage_fraction=0.5):
        """Atrium + alpha-masked foliage (San Miguel stand-in, BASELINE config 5)."""
        return cls(load_library().bdpt_scene_create_courtyard(seed, target_triangles, foliage_fraction))

    @classmethod
    def soup(cls, seed, num_triangles, max_edge=0.25):
        return cls(load_library().bdpt_scene_create_soup(seed, num_triangles, max_edge))

    @classmethod
    def load(cls, path, threads=None):
        """`.fscene` / `.obj` ingestion (SharedUtils/SceneLoaderWrapper.cpp:56-103).  `threads`: host threads of the model
        loader for this call (None: the default; the scene does not depend on it)."""
        msg = C.create_string_buffer(512)
        lib = load_library()
        before = lib.bdpt_scene_load_threads(int(threads)) if threads is not None else None
        try:
            h = lib.bdpt_scene_load(str(path).encode(), msg, 512)
        finally:
            if before is not None:
                lib.bdpt_scene_load_threads(before)
        if not h:
            raise BdptError("bdpt_scene_load: " + msg.value.decode(errors="replace"))
        return cls(h)

    def camera(self, aspect):
        cam = Camera()
        if self._lib.bdpt_scene_get_camera(self._h, float(aspect), C.byref(cam)) != 0:
            raise BdptError("bdpt_scene_get_camera failed")
        return cam

    def close(self):
        if self._h:
            self._lib.bdpt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_view_proj(pos, target, up, focal_length_mm=21.0, frame_height_mm=24.0, aspect=1.7777, near=0.1, far=1000.0):
    """Falcor's jitter-free viewProjMat, row-major (Camera.cpp:60-105) -> 16 floats for BmfrParams.prevViewProj."""
    out = (C.c_float * 16)()
    v3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    rc = load_library().bdpt_camera_view_proj(v3(pos), v3(target), v3(up), focal_length_mm, frame_height_mm, aspect,
                                              near, far, out)
    if rc != 0:
        raise BdptError("bdpt_camera_view_proj failed")
    return list(out)


def msaa_jitter(counter_before_increment):
    j = (C.c_float * 2)()
    load_library().bdpt_msaa_jitter(counter_before_increment & 0xFFFFFFFF, j)
    return float(j[0]), float(j[1])


class Context:
    """One bdpt_ctx = one GPU (one ctx per GPU, not thread-safe)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.bdpt_create(int(device), C.byref(h))
        if rc != 0:
            raise BdptError(f"bdpt_create(device={device}) failed with {rc}: no usable HIP device — "
                            "the render pass has no CPU fallback")
        self._h = h
        self.device = device

    def _check(self, rc, what):
        if rc < 0:
            raise BdptError(f"{what} failed ({rc}): {self._lib.bdpt_last_error(self._h).decode()}")
        return rc

    def set_scene(self, desc):
        self._check(self._lib.bdpt_set_scene(self._h, C.byref(desc)), "bdpt_set_scene")

    def bvh_info(self):
        info = abi.BvhInfo()
        self._check(self._lib.bdpt_get_bvh_info(self._h, C.byref(info)), "bdpt_get_bvh_info")
        return info

    def set_camera(self, cam):
        self._check(self._lib.bdpt_set_camera(self._h, C.byref(cam)), "bdpt_set_camera")

    def set_environment(self, env_map_ptr=None, width=0, height=0, color=(0.0, 0.0, 0.0, 0.0)):
        """What BDPT_PARAM_ENV_ON_MISS looks up: a device RGBA32F lat-long map, or a constant colour."""
        e = abi.Environment()
        e.envMap = env_map_ptr
        e.width, e.height = int(width), int(height)
        for i in range(4):
            e.color[i] = float(color[i])
        self._check(self._lib.bdpt_set_environment(self._h, C.byref(e)), "bdpt_set_environment")

    def resize(self, width, height, y0, y1, max_depth):
        self._check(self._lib.bdpt_resize(self._h, width, height, Tile(y0, y1), max_depth), "bdpt_resize")

    def resize_stripes(self, width, height, stripe_rows, num_owners, owner, max_depth):
        self._check(self._lib.bdpt_resize_stripes(self._h, width, height, Stripes(stripe_rows, num_owners, owner), max_depth),
                    "bdpt_resize_stripes")

    def tile_info(self):
        info = TileInfo()
        self._check(self._lib.bdpt_get_tile_info(self._h, C.byref(info)), "bdpt_get_tile_info")
        return info

    def row_ranges(self):
        n = self.tile_info().numRowRanges
        buf = (C.c_uint32 * (2 * max(n, 1)))()
        got = self._check(self._lib.bdpt_tile_row_ranges(self._h, buf, n), "bdpt_tile_row_ranges")
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(got)]

    def resolve_tile(self, tile_splat_ptr, out_ptr, stream=None):
        self._check(self._lib.bdpt_resolve_tile(self._h, tile_splat_ptr, out_ptr, stream), "bdpt_resolve_tile")

    def accumulate_tile(self, last_ptr, cur_ptr, accum_count, max_count, stream=None):
        self._check(self._lib.bdpt_accumulate_tile(self._h, last_ptr, cur_ptr, accum_count, max_count, stream),
                    "bdpt_accumulate_tile")

    def gbuffer_execute(self, gparams, gbuffer, stream=None):
        self._check(self._lib.bdpt_gbuffer_execute(self._h, C.byref(gparams), C.byref(gbuffer), stream),
                    "bdpt_gbuffer_execute")

    def execute(self, params, gbuffer, out_ptr, stream=None):
        self._check(self._lib.bdpt_execute(self._h, C.byref(params), C.byref(gbuffer), out_ptr, stream), "bdpt_execute")

    def execute_tail(self, params, gbuffer, out_ptr, stream=None):
        self._check(self._lib.bdpt_execute_tail(self._h, C.byref(params), C.byref(gbuffer), out_ptr, stream),
                    "bdpt_execute_tail")

    def prepare(self, what):
        self._check(self._lib.bdpt_prepare(self._h, int(what)), "bdpt_prepare")

    def splat_buffer(self):
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self._lib.bdpt_splat_buffer(self._h, C.byref(p), C.byref(n)), "bdpt_splat_buffer")
        return p.value, n.value

    def set_splat_buffer(self, ptr, num_u64):
        self._check(self._lib.bdpt_set_splat_buffer(self._h, ptr, num_u64), "bdpt_set_splat_buffer")

    def resolve(self, splat_ptr, splat_row0, out_ptr, stream=None):
        self._check(self._lib.bdpt_resolve(self._h, splat_ptr, splat_row0, out_ptr, stream), "bdpt_resolve")

    def accumulate(self, last_ptr, cur_ptr, accum_count, max_count, num_texels, stream=None):
        self._check(self._lib.bdpt_accumulate(self._h, last_ptr, cur_ptr, accum_count, max_count, num_texels, stream),
                    "bdpt_accumulate")

    def bmfr_execute(self, params, gbuffer, noisy_ptr, stream=None):
        self._check(self._lib.bdpt_bmfr_execute(self._h, C.byref(params), C.byref(gbuffer), noisy_ptr, stream),
                    "bdpt_bmfr_execute")

    def bmfr_reset(self):
        self._check(self._lib.bdpt_bmfr_reset(self._h), "bdpt_bmfr_reset")

    def counters(self):
        c = Counters()
        self._check(self._lib.bdpt_get_counters(self._h, C.byref(c)), "bdpt_get_counters")
        return c

    def enable_stage_timing(self, on=True):
        self._check(self._lib.bdpt_enable_stage_timing(self._h, 1 if on else 0), "bdpt_enable_stage_timing")

    def stage_times(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = self._check(self._lib.bdpt_get_stage_times(self._h, names, ms, 64), "bdpt_get_stage_times")
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def sync(self, stream=None):
        self._check(self._lib.bdpt_sync(self._h, stream), "bdpt_sync")

    # test hooks -------------------------------------------------------------------------------
    def test_rng(self, val0, val1, draws):
        import numpy as np
        val0 = np.ascontiguousarray(val0, np.uint32)
        val1 = np.ascontiguousarray(val1, np.uint32)
        n = val0.size
        st = np.zeros((n, draws), np.uint32)
        fl = np.zeros((n, draws), np.float32)
        self._check(self._lib.bdpt_test_rng(self._h, val0.ctypes.data, val1.ctypes.data, n, draws, st.ctypes.data,
                                            fl.ctypes.data), "bdpt_test_rng")
        return st, fl

    def test_trace(self, rays, mode):
        import numpy as np
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = rays.shape[0]
        prim = np.zeros(n, np.int32)
        tuv = np.zeros((n, 3), np.float32)
        self._check(self._lib.bdpt_test_trace(self._h, rays.ctypes.data, n, mode, prim.ctypes.data, tuv.ctypes.data),
                    "bdpt_test_trace")
        return prim, tuv

    def test_trace_shadow(self, rays):
        """The persistent any-hit kernel over host rays (n x 8 float32): visibility bytes and the deepest stack reached."""
        import numpy as np
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[0]
        vis = np.zeros(n, np.uint8)
        deep = C.c_uint32(0)
        self._check(self._lib.bdpt_test_trace_shadow(self._h, rays.ctypes.data, n, vis.ctypes.data, C.byref(deep)),
                    "bdpt_test_trace_shadow")
        return vis, deep.value

    def test_bsdf(self, recs, mat_index):
        import numpy as np
        recs = np.ascontiguousarray(recs, np.float32).reshape(-1, 20)
        out = np.zeros((recs.shape[0], 16), np.float32)
        self._check(self._lib.bdpt_test_bsdf(self._h, recs.ctypes.data, recs.shape[0], mat_index, out.ctypes.data),
                    "bdpt_test_bsdf")
        return out

    def close(self):
        if self._h:
            self._lib.bdpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FramePipeline:
    """The reference's per-frame sequence for this path on one GPU / one tile.

    Pass 0 LightProbeGBufferPass (frame counter starts 0xdeadbeef, CommonPasses/LightProbeGBufferPass.h:79),
    pass 1 BDPTPass (0x1337, BDPTPass.h:44), pass 2 SimpleAccumulationPass (cap 100 by default,
    CommonPasses/SimpleAccumulationPass.h:70).  Channels live in torch tensors on ``cuda:<device>``.
    """

    def __init__(self, scene, width, height, max_depth=3, mat_index=0, device=0, tile=None, clamp_upper=0.9, min_t=1e-4,
                 accum_limit=100, flags=0, stripes=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise BdptError("no GPU visible to torch: the render pass cannot run (no CPU fallback)")
        self.W, self.H = int(width), int(height)
        self.y0, self.y1 = (0, self.H) if tile is None else (int(tile[0]), int(tile[1]))
        self.max_depth, self.mat_index = int(max_depth), int(mat_index)
        self.clamp_upper, self.min_t, self.flags = float(clamp_upper), float(min_t), int(flags)
        self.accum_limit = int(accum_limit)
        self.dev = torch.device("cuda", device)
        import time
        t0 = time.time()
        self.ctx = Context(device)
        t1 = time.time()
        # The frame's buffers first, the scene second: building the acceleration structure of a large scene takes and
        # frees some 15 GB of device scratch, and the driver clears VRAM that has been used before when it hands it out
        # again (~35 GB/s) — a 94 GB bdpt_resize issued right after such a bdpt_set_scene was measured 1-1.7 s slower
        # (0.01-0.03 s on untouched memory).
        # tile = (y0, y1): a contiguous band; stripes = (stripe_rows, num_owners, owner): interleaved stripes
        self.stripes = stripes
        if stripes is None:
            self.ctx.resize(self.W, self.H, self.y0, self.y1, self.max_depth)
        else:
            self.ctx.resize_stripes(self.W, self.H, int(stripes[0]), int(stripes[1]), int(stripes[2]), self.max_depth)
        self.rows = self.ctx.row_ranges()
        torch.cuda.synchronize(self.dev)
        t2 = time.time()
        self.scene = scene
        self.ctx.set_scene(scene.desc)
        self.cam = scene.camera(self.W / self.H)
        self.ctx.set_camera(self.cam)
        t3 = time.time()
        with torch.cuda.device(self.dev):
            self.channels = {"WorldPosition": torch.zeros(self.H, self.W, 4, dtype=torch.float32, device=self.dev)}
            for name in GBUFFER_CHANNELS[1:]:
                self.channels[name] = torch.zeros(self.H, self.W, 4, dtype=torch.float16, device=self.dev)
            self.channels[OUTPUT_CHANNEL] = torch.zeros(self.H, self.W, 4, dtype=torch.float32, device=self.dev)
            self.last_frame = torch.zeros(self.H, self.W, 4, dtype=torch.float32, device=self.dev)
        torch.cuda.synchronize(self.dev)
        # where the set-up time went: the scene (acceleration structure + uploads) and the frame's buffers are separate things
        self.setup_times = {"context_s": t1 - t0, "resize_s": t2 - t1, "set_scene_s": t3 - t2, "channels_s": time.time() - t3}
        self.gb = GBuffer(*[self.channels[n].data_ptr() for n in GBUFFER_CHANNELS])
        self.gbuffer_frame = 0xdeadbeef
        self.bdpt_frame = 0x1337
        self.accum_count = 0
        self.env_color = (0.5, 0.5, 0.8, 1.0)  # SharedUtils/ResourceManager.cpp:77-87 default environment
        self.use_jitter = True
        self.last_params = None

    @property
    def output(self):
        return self.channels[OUTPUT_CHANNEL]

    def _stream_ptr(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def gbuffer_params(self):
        gp = GBufferParams()
        if self.use_jitter:
            gp.pixelJitter[0], gp.pixelJitter[1] = msaa_jitter(self.gbuffer_frame)
        else:
            gp.pixelJitter[0], gp.pixelJitter[1] = 0.5, 0.5
        gp.frameCount = self.gbuffer_frame & 0xFFFFFFFF
        gp.useThinLens = 0
        gp.focalLen = 1.0
        gp.lensRadius = 1.0 / 64.0  # mFocalLength / (2 mFStop), LightProbeGBufferPass.cpp:117
        gp.envMap = None
        gp.envWidth = gp.envHeight = 128
        for i in range(4):
            gp.envColor[i] = self.env_color[i]
        return gp

    def bdpt_params(self, extra_flags=0):
        p = Params()
        p.minT = self.min_t
        p.frameCount = self.bdpt_frame & 0xFFFFFFFF
        p.matIndex = self.mat_index
        p.refractiveIndex = 1.0
        p.maxDepth = self.max_depth
        p.emitMult = 1.0
        p.clampUpper = self.clamp_upper
        p.pixelJitter[0], p.pixelJitter[1] = msaa_jitter(self.bdpt_frame)
        p.flags = self.flags | extra_flags
        return p

    def render_frame(self, accumulate=False, extra_flags=0, gbuffer=True):
        """One pipeline frame.  Returns the bdpt_params used (for the oracle to mirror)."""
        st = self._stream_ptr()
        gp = self.gbuffer_params()
        if gbuffer:
            self.ctx.gbuffer_execute(gp, self.gb, st)
        p = self.bdpt_params(extra_flags)
        self.ctx.execute(p, self.gb, C.c_void_p(self.output.data_ptr()), st)
        self.gbuffer_frame += 1
        self.bdpt_frame += 1
        if accumulate:
            n = self.accum_count if self.accum_count < self.accum_limit else self.accum_limit
            if self.accum_count < self.accum_limit:
                self.accum_count += 1
            if self.stripes is None:
                self.ctx.accumulate(C.c_void_p(self.last_frame.data_ptr()), C.c_void_p(self.output.data_ptr()), n,
                                    self.accum_limit, self.W * self.H, st)
            else:
                self.ctx.accumulate_tile(C.c_void_p(self.last_frame.data_ptr()), C.c_void_p(self.output.data_ptr()), n,
                                         self.accum_limit, st)
        self.last_params = (gp, p)
        return gp, p

    def close(self):
        self.ctx.close()
