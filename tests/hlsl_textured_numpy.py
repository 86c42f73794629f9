"""A second, independent reading — float64 numpy, written from the HLSL / Slang sources, not from the oracle — of the
part of the path the untextured Cornell cross-check (hlsl_integrator_numpy.py) never touches: vertex-attribute fetch,
material decode, texture sampling, the any-hit alpha test and the primary hit's normal map.

  getVertexAttributes               Falcor ShadingUtils/Raytracing.slang:60-106 (texC, normalW, bitangentW by barycentrics,
                                    both normalised; posW = origin + t * direction)
  sampleTexture                     Falcor ShadingUtils/Shading.slang:88-94 (unused -> 0, const -> factor, texture -> SampleLevel 0)
  simplePrepareShadingData          BDPT/Data/BDPTUtils.hlsli:2-52 (MetalRough / SpecGloss decode, roughness clamp, emissive,
                                    NO normal map on secondary hits, double-sided flip)
  _prepareShadingData + applyNormalMap   Shading.slang:135-157, 189-259 (the G-buffer's primary hit: RGB / RG normal maps)
  alphaTestFails / PrimaryAnyHit    BDPTUtils.hlsli:115-127, CP lightProbeGBuffer.rt.hlsl:77-90 (IgnoreHit when base alpha < threshold)
  GBufferRayGen / PrimaryClosestHit / PrimaryMiss   lightProbeGBuffer.rt.hlsl:63-159

What the shaders leave to the driver / sampler hardware is defined as DESIGN.md section 2 defines it: hits need
tmin < t < tmax, the closest ACCEPTED hit wins (ties to the lowest primitive), textures are mip 0, bilinear, wrap,
sRGB-decoded per texel before filtering (alpha is never sRGB), front face = det > 0 in Moeller-Trumbore.
"""
import numpy as np

import hlsl_integrator_numpy as hi

norm = hi.norm


def srgb_to_linear(c):
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


class TexturedScene(hi.Scene):
    def __init__(self, desc, tracer=None):
        super().__init__(desc, tracer)
        nv, nt = desc.numVertices, desc.numTriangles
        idx = np.ctypeslib.as_array(desc.indices, shape=(nt * 3,)).reshape(-1, 3)
        if desc.texcoords:
            uv = np.ctypeslib.as_array(desc.texcoords, shape=(nv * 3,)).reshape(-1, 3)[:, :2].astype(np.float64)
        else:
            uv = np.zeros((nv, 2))
        self.uv = (uv[idx[:, 0]], uv[idx[:, 1]], uv[idx[:, 2]])
        self.has_bitangents = bool(desc.bitangents)
        if self.has_bitangents:
            bt = np.ctypeslib.as_array(desc.bitangents, shape=(nv * 3,)).reshape(-1, 3).astype(np.float64)
            self.bt = (bt[idx[:, 0]], bt[idx[:, 1]], bt[idx[:, 2]])
        self.textures = []
        for i in range(desc.numTextures):
            t = desc.textures[i]
            px = np.ctypeslib.as_array(t.rgba8, shape=(t.height * t.width * 4,)).reshape(t.height, t.width, 4).astype(np.float64) / 255.0
            if t.srgb:
                px = np.concatenate([srgb_to_linear(px[..., :3]), px[..., 3:]], axis=-1)
            self.textures.append(px)
        flags = np.array([m.flags for m in self.mats], np.int64)
        self.tri_alpha_mode = ((flags >> 17) & 3)[self.mat_id]      # != 0: the any-hit shader runs
        self.tri_double_sided = (((flags >> 19) & 1)[self.mat_id]).astype(bool)

    # ---- textures
    def sample_bilinear(self, tex_id, uv):
        px = self.textures[tex_id]
        h, w = px.shape[0], px.shape[1]
        x, y = uv[0] * w - 0.5, uv[1] * h - 0.5
        x0, y0 = int(np.floor(x)), int(np.floor(y))
        fx, fy = x - x0, y - y0
        ix0, iy0 = x0 % w, y0 % h          # wrap addressing (Python's % is already non-negative)
        ix1, iy1 = (ix0 + 1) % w, (iy0 + 1) % h
        top = px[iy0, ix0] * (1.0 - fx) + px[iy0, ix1] * fx
        bot = px[iy1, ix0] * (1.0 - fx) + px[iy1, ix1] * fx
        return top * (1.0 - fy) + bot * fy

    def sample_texture(self, tex_id, uv, factor, mode):
        if mode == 0:      # ChannelTypeUnused
            return np.zeros(4)
        if mode == 1 or tex_id < 0:  # ChannelTypeConst (a texture channel without a texture falls back to the factor)
            return np.asarray(factor, np.float64)
        return self.sample_bilinear(tex_id, uv)

    def tex_coord(self, prim, u, v):
        return self.uv[0][prim] * (1.0 - u - v) + self.uv[1][prim] * u + self.uv[2][prim] * v

    def alpha_test_fails(self, prim, u, v):
        m = self.mats[self.mat_id[prim]]
        base = self.sample_texture(m.texBaseColor, self.tex_coord(prim, u, v), list(m.baseColor), (m.flags >> 3) & 7)
        return base[3] < m.alphaThreshold

    # ---- rays: closest ACCEPTED hit; the any-hit shader ignores candidates that fail the alpha test
    def intersect(self, o, d, tmin, tmax, cull_back=False):
        e1, e2 = self.v1 - self.v0, self.v2 - self.v0
        p = np.cross(d, e2)
        det = np.einsum("ij,ij->i", e1, p)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - self.v0
            u = np.einsum("ij,ij->i", tv, p) * inv
            q = np.cross(tv, e1)
            v = np.einsum("ij,j->i", q, d) * inv
            t = np.einsum("ij,ij->i", e2, q) * inv
        facing = ((det > 0) | (self.tri_double_sided & (det != 0))) if cull_back else (det != 0)
        ok = facing & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
        cand = np.nonzero(ok)[0]
        if cand.size == 0:
            return None
        for k in cand[np.argsort(t[cand], kind="stable")]:   # stable: ties stay in primitive order
            if self.tri_alpha_mode[k] != 0 and self.alpha_test_fails(int(k), float(u[k]), float(v[k])):
                continue
            return int(k), float(t[k]), float(u[k]), float(v[k])
        return None

    # ---- simplePrepareShadingData (secondary hits) / _prepareShadingData (primary hit, with the normal map)
    def shade(self, prim, u, v, o, d, t, view_pos, primary):
        m = self.mats[self.mat_id[prim]]
        flags = m.flags
        model, dif_t, spec_t, emis_t = flags & 7, (flags >> 3) & 7, (flags >> 6) & 7, (flags >> 9) & 7
        nmap_t, dbl = (flags >> 12) & 3, (flags >> 19) & 1
        uv = self.tex_coord(prim, u, v)
        base = self.sample_texture(m.texBaseColor, uv, list(m.baseColor), dif_t)
        spec = self.sample_texture(m.texSpecular, uv, list(m.specular), spec_t)
        pos = o + d * t
        b0 = 1.0 - u - v
        n = norm(norm(self.n0[prim] * b0 + self.n1[prim] * u + self.n2[prim] * v))  # getVertexAttributes, then sd.N
        vv = norm(view_pos - pos)
        if model == 0:   # ShadingModelMetalRough: R occlusion, G roughness, B metalness
            dif = base[:3] * (1.0 - spec[2])
            sp = 0.04 * (1.0 - spec[2]) + base[:3] * spec[2]
            lr = spec[1]
        else:            # ShadingModelSpecGloss
            dif, sp, lr = base[:3], spec[:3], 1.0 - spec[3]
        lr = max(0.08, lr)
        emissive = self.sample_texture(m.texEmissive, uv, [m.emissive[0], m.emissive[1], m.emissive[2], 1.0], emis_t)[:3]
        if primary and nmap_t != 0 and m.texNormal >= 0 and self.has_bitangents:
            bw = norm(self.bt[0][prim] * b0 + self.bt[1][prim] * u + self.bt[2][prim] * v)
            B = norm(bw - n * float(np.dot(bw, n)))
            T = norm(np.cross(B, n))
            mp = self.sample_bilinear(m.texNormal, uv)
            if nmap_t == 1:   # NormalMapRGB
                mn = norm(mp[:3] * 2.0 - 1.0)
            else:             # NormalMapRG: z from the RAW rg (Shading.slang:110-119)
                z = min(max(mp[0] * mp[0] + mp[1] * mp[1], 0.0), 1.0)
                mn = norm(np.array([mp[0] * 2.0 - 1.0, mp[1] * 2.0 - 1.0, np.sqrt(1.0 - z)]))
            n = T * mn[0] + B * mn[1] + n * mn[2]
        if float(np.dot(n, vv)) <= 0 and dbl:
            n = -n
        return dict(pos=pos, N=n, V=vv, dif=dif, spec=sp, linear_roughness=lr, roughness=lr * lr, emissive=emissive,
                    opacity=m.baseColor[3], ior=m.IoR, prim=prim)

    def shading(self, prim, u, v, o, d, t):
        """What hlsl_integrator_numpy.Renderer.shoot calls for a closest hit of a walk: V points at the ray origin."""
        s = self.shade(prim, u, v, o, d, t, o, False)
        return s["pos"], s["N"], s["V"], s["dif"], s["spec"], s["roughness"]


def gbuffer_pixel(scene, cam, gp, width, height, x, y, env_color):
    """GBufferRayGen + PrimaryAnyHit + PrimaryClosestHit / PrimaryMiss for textured, alpha-masked scenes (pinhole or thin
    lens, constant environment) -> dict of the channels written."""
    cam_pos = np.array(list(cam.posW), np.float64)
    U, V, W = (np.array(list(getattr(cam, k)), np.float64) for k in ("cameraU", "cameraV", "cameraW"))
    pc = np.array([(x + gp.pixelJitter[0]) / width, (y + gp.pixelJitter[1]) / height])
    ndc = np.array([2.0, -2.0]) * pc + np.array([-1.0, 1.0])
    ray_dir = (ndc[0] * U + ndc[1] * V + W) / np.linalg.norm(W)
    focal = cam_pos + gp.focalLen * ray_dir
    seed = hi.hm_init_rand(x + y * width, int(gp.frameCount))
    seed, r0 = hi.hm.next_rand(seed)
    seed, r1 = hi.hm.next_rand(seed)
    a, rad = 2.0 * hi.M_PI * r0, gp.lensRadius * r1
    origin = cam_pos + np.cos(a) * rad * norm(U) + np.sin(a) * rad * norm(V)
    o = origin if gp.useThinLens else cam_pos
    d = norm(focal - origin) if gp.useThinLens else norm(ray_dir)
    hit = scene.intersect(o, d, 0.0, 1e38, cull_back=True)
    if hit is None:
        return dict(hit=False, dif=np.array([env_color[0], env_color[1], env_color[2], 1.0]))
    prim, t, u, v = hit
    s = scene.shade(prim, u, v, o, d, t, cam_pos, True)   # prepareShadingData(vsOut, gMaterial, gCamera.posW, 0)
    s.update(hit=True, dist=float(np.linalg.norm(s["pos"] - cam_pos)))
    return s
