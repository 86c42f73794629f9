"""A second, independent reading of the reference's BDPT ray-generation shader in float64 numpy, used only to
cross-check the oracle (tests/test_oracle_cross_check.py).  Written from the HLSL, not from the oracle; one
pixel at a time, brute-force ray/triangle tests, constant-colour materials only (the Cornell box).

  SimpleDiffuseGIRayGen             BDPT/Data/BDPTMain.rt.hlsl:42-234
  shootRay / RayMiss / RayClosestHit / handleIndirectRayHit      BDPT/Data/globalIlluminationRay.hlsli
  shadowRayVisibility               BDPT/Data/standardShadowRay.hlsli:13-29
  PathVertex / RayPayload / initPayload / updateRayData          BDPT/Data/RayPathData.hlsli
  simplePrepareShadingData          BDPT/Data/BDPTUtils.hlsli:2-52   (constant channels, MetalRough / SpecGloss)
  getLaunchIndexFromDirection, sampleLight, evalGWithoutV, getUnweightedContribution   BDPTUtils.hlsli:129-224
  evalDirect / ggxDirect / lambertianDirect / getLightData / sampleUnitSphere          MaterialUtils.hlsli:5-185, 288-311
  evalPointLight / evalDirectionalLight                          Falcor ShadingUtils/Lights.slang:62-101

What the shader leaves to the driver is defined as the oracle defines it: a hit needs
tmin < t < tmax, the closest hit wins with ties to the lowest primitive; cross-pixel splats are returned as a
list instead of racing on gOutput, and out-of-frame splat targets are dropped.
"""
import math

import numpy as np

import hlsl_reference_math as hm

M_PI = hm.M_PI
M_1_PI = hm.M_1_PI


def norm(v):
    return v / np.linalg.norm(v)


class Scene:
    """Flat triangle list + constant materials + lights, from the ctypes bdpt_scene_desc."""

    def __init__(self, desc, tracer=None):
        # tracer(o, d, tmin, tmax) -> (prim, t, u, v) or None: lets the caller supply the ray/triangle arithmetic
        # (the part the shaders leave to the DXR driver); None = the float64 Moeller-Trumbore below
        self.tracer = tracer
        nv, nt = desc.numVertices, desc.numTriangles
        pos = np.ctypeslib.as_array(desc.positions, shape=(nv * 3,)).reshape(-1, 3).astype(np.float64)
        nrm = np.ctypeslib.as_array(desc.normals, shape=(nv * 3,)).reshape(-1, 3).astype(np.float64)
        idx = np.ctypeslib.as_array(desc.indices, shape=(nt * 3,)).reshape(-1, 3)
        self.v0, self.v1, self.v2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
        self.n0, self.n1, self.n2 = nrm[idx[:, 0]], nrm[idx[:, 1]], nrm[idx[:, 2]]
        self.mat_id = np.ctypeslib.as_array(desc.triMaterial, shape=(nt,)).copy()
        self.mats = [desc.materials[i] for i in range(desc.numMaterials)]
        self.lights = [desc.lights[i] for i in range(desc.numLights)]

    def intersect(self, o, d, tmin, tmax):
        """Moeller-Trumbore over all triangles; -> (prim, t, u, v) or None."""
        if self.tracer is not None:
            return self.tracer(o, d, tmin, tmax)
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
        ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
        if not ok.any():
            return None
        cand = np.nonzero(ok)[0]
        best = cand[np.argmin(t[cand])]  # argmin returns the first (lowest primitive) among equal t
        return int(best), float(t[best]), float(u[best]), float(v[best])

    def shading(self, prim, u, v, o, d, t):
        """simplePrepareShadingData for constant channels -> (posW, N, V, dif, spec, roughness)."""
        m = self.mats[self.mat_id[prim]]
        flags = m.flags
        model, dif_t, spec_t, dbl = flags & 7, (flags >> 3) & 7, (flags >> 6) & 7, (flags >> 19) & 1
        base = np.array(list(m.baseColor), np.float64) if dif_t == 1 else np.zeros(4)
        spec = np.array(list(m.specular), np.float64) if spec_t == 1 else np.zeros(4)
        assert dif_t != 2 and spec_t != 2, "textured materials are outside this cross-check"
        pos = o + d * t
        n = norm(self.n0[prim] * (1.0 - u - v) + self.n1[prim] * u + self.n2[prim] * v)
        vv = norm(o - pos)
        if model == 0:  # ShadingModelMetalRough
            dif = base[:3] * (1.0 - spec[2])
            sp = 0.04 + (base[:3] - 0.04) * spec[2]
            lr = spec[1]
        else:
            dif, sp, lr = base[:3], spec[:3], 1.0 - spec[3]
        lr = max(0.08, lr)
        if float(np.dot(n, vv)) <= 0 and dbl:
            n = -n
        return pos, n, vv, dif, sp, lr * lr


class Vertex:
    def __init__(self, color=None, pos=None, n=None, v=None, dif=None, spec=None, rough=0.0, is_spec=False, pdf=0.0):
        z = np.zeros(3)
        self.color = z if color is None else color
        self.pos = z if pos is None else pos
        self.N = z if n is None else n
        self.V = z if v is None else v
        self.dif = z if dif is None else dif
        self.spec = z if spec is None else spec
        self.rough, self.is_spec, self.pdf = rough, is_spec, pdf


class Renderer:
    def __init__(self, scene, cam, params, width, height, specular_from_lobe=False, mis=None):
        self.mis = mis  # None: the uniform 1/k the shader applies; "power" / "linear": getWeightPower / getWeightLinear
        self.s, self.W, self.H = scene, width, height
        self.cam_pos = np.array(list(cam.posW), np.float64)
        self.U, self.Vv, self.Wv = (np.array(list(getattr(cam, k)), np.float64) for k in ("cameraU", "cameraV", "cameraW"))
        self.p = params
        self.mat = int(params.matIndex)
        self.D = int(params.maxDepth)
        self.from_lobe = specular_from_lobe
        self.nl = len(scene.lights)

    # ---- rays
    def visible(self, o, d, tmin, tmax):
        if not (np.isfinite(o).all() and np.isfinite(d).all() and tmax > tmin):
            return True  # no triangle can satisfy tmin < t < tmax: the miss shader runs
        return self.s.intersect(o, d, tmin, tmax) is None

    def shoot(self, pl):
        """shootRay + closest-hit / miss shader on the payload dict."""
        hit = None
        if np.isfinite(pl["o"]).all() and np.isfinite(pl["d"]).all():
            hit = self.s.intersect(pl["o"], pl["d"], float(self.p.minT), 1.0e38)
        if hit is None:
            pl["color"] = np.zeros(3)
            pl["terminated"] = True
            return
        prim, t, u, v = hit
        pos, n, vv, dif, spec, rough = self.s.shading(prim, u, v, pl["o"], pl["d"], t)
        w, l, pdf, lobe, _ = hm.sample_brdf(self.mat, pl["seed"], n, vv, dif, spec, rough)  # seed passed BY VALUE
        pl["color"] = pl["color"] * w
        pl.update(o=pos, d=l, pos=pos, N=n, V=vv, dif=dif, spec=spec, rough=rough, pdf=pdf,
                  is_spec=(lobe and self.from_lobe) if self.mat == 0 else False)

    @staticmethod
    def payload(o, d, color, seed):
        z = np.zeros(3)
        return dict(o=o, d=d, seed=seed, color=color, pos=o, N=z, V=z, dif=z, spec=z, rough=0.0, is_spec=False, pdf=0.0,
                    terminated=False)

    @staticmethod
    def vertex_of(pl):
        return Vertex(pl["color"], pl["pos"], pl["N"], pl["V"], pl["dif"], pl["spec"], pl["rough"], pl["is_spec"], pl["pdf"])

    # ---- lights
    def light_data(self, index, hit):
        lt = self.s.lights[index]
        lpos, ldir, inten = (np.array(list(getattr(lt, k)), np.float64) for k in ("posW", "dirW", "intensity"))
        if lt.type == 1:  # directional
            L = -norm(ldir)
            dist = np.linalg.norm(hit - lpos)
            ls_pos = hit - ldir * dist
            diffuse = inten
        else:
            L = lpos - hit
            d2 = float(np.dot(L, L))
            L = norm(L) if d2 > 1e-5 else np.zeros(3)
            falloff = 1.0 / (0.01 * 0.01 + d2)
            cos_t = -float(np.dot(L, ldir))
            if cos_t < lt.cosOpeningAngle:
                falloff = 0.0
            elif lt.penumbraAngle > 0:
                delta = lt.openingAngle - math.acos(cos_t)
                falloff *= hm.saturate((delta - lt.penumbraAngle) / lt.penumbraAngle)
            ls_pos, diffuse = lpos, inten * falloff
        with np.errstate(invalid="ignore", divide="ignore"):
            to_light = L / np.linalg.norm(L)
        return to_light, diffuse, float(np.linalg.norm(ls_pos - hit))

    def eval_direct(self, seed, v):
        seed, r = hm.next_rand(seed)
        index = min(int(r * self.nl), self.nl - 1)
        L, inten, dist = self.light_data(index, v.pos)
        with np.errstate(invalid="ignore"):
            ndotl = hm.saturate(float(np.dot(v.N, L)))
        vis = self.visible(v.pos, L, float(self.p.minT), dist)
        shadow = float(self.nl) if vis else 0.0
        if self.mat == 1:
            return seed, shadow * ndotl * inten * v.dif / M_PI
        with np.errstate(invalid="ignore", divide="ignore"):
            h = (v.V + L) / np.linalg.norm(v.V + L)
            ndoth, ldoth = hm.saturate(float(np.dot(v.N, h))), hm.saturate(float(np.dot(L, h)))
            ndotv = hm.saturate(float(np.dot(v.N, v.V)))
            d = hm.ggx_d(ndoth, v.rough)
            g = hm.ggx_g(ndotl, ndotv, v.rough)
            f = hm.schlick(v.spec, ldoth)
            term = d * g * f / np.float64(4 * ndotv)
            return seed, shadow * inten * (term + ndotl * v.dif / M_PI)

    def sample_light(self, seed):
        seed, r = hm.next_rand(seed)
        index = min(int(r * self.nl), self.nl - 1)
        lt = self.s.lights[index]
        origin = np.array(list(lt.posW), np.float64)
        inten = np.array(list(lt.intensity), np.float64)
        if lt.type == 1:
            d = np.array(list(lt.dirW), np.float64)
        else:
            d = np.array([2.0, 2.0, 2.0])
            while np.linalg.norm(d) > 1.0:
                seed, a = hm.next_rand(seed)
                seed, b = hm.next_rand(seed)
                seed, c = hm.next_rand(seed)
                d = np.array([a * 2.0 - 1.0, b * 2.0 - 1.0, c * 2.0 - 1.0])
        seed, d = hm.cos_hemisphere(seed, d)
        return seed, origin, d, inten

    # ---- helpers of the contribution loops
    def clamp_vec(self, c):
        # HLSL clamp = min(max(x, lo), hi) with the "other operand wins over NaN" rule: clamp(NaN) = lo
        return np.fmin(np.fmax(c, 0.0), float(self.p.clampUpper))

    def pixel_of(self, d):
        d1 = np.dot(d, self.U) / np.dot(self.U, self.U)
        d2 = np.dot(d, self.Vv) / np.dot(self.Vv, self.Vv)
        d3 = np.dot(d, self.Wv) / np.dot(self.Wv, self.Wv)
        cx = (d1 / d3) * 0.5 + 0.5
        cy = (-d2 / d3) * 0.5 + 0.5
        fx, fy = cx * self.W - self.p.pixelJitter[0], cy * self.H - self.p.pixelJitter[1]
        # HLSL round() = round half to even; uint() of a negative value is out of range -> dropped here
        ix, iy = np.rint(fx), np.rint(fy)
        if not (np.isfinite(ix) and np.isfinite(iy)) or ix < 0 or iy < 0 or ix >= self.W or iy >= self.H:
            return None
        return int(ix), int(iy)

    def g_without_v(self, a, b):
        with np.errstate(invalid="ignore", divide="ignore"):
            vec = b.pos - a.pos
            inv = np.float64(1.0) / np.linalg.norm(vec)
            d = vec * inv
            return abs(float(np.dot(a.N, d))) * abs(float(np.dot(b.N, d))) * inv * inv

    def weight(self, cp, lp, ci, li):
        """getWeightPower / getWeightLinear (BDPTUtils.hlsli:226-278); `i == cameraIndex, j == lightIndex` is a comma
        expression, i.e. `j == lightIndex` — the same condition, since i + j is fixed."""
        total = ci + li
        total_pdf, current = 0.0, 1.0
        with np.errstate(all="ignore"):
            for i in range(total + 1):
                j = total - i
                pe = np.float64(cp[0].pdf)
                for x in range(1, i + 1):
                    pe = pe * (cp[x].pdf * self.g_without_v(cp[x - 1], cp[x]))
                pl = np.float64(lp[0].pdf)
                for x in range(1, j + 1):
                    pl = pl * (lp[x].pdf * self.g_without_v(lp[x - 1], lp[x]))
                term = pe * pe * pl * pl if self.mis == "power" else pe * pl
                total_pdf = total_pdf + term
                if j == li:
                    current = term
            return current / np.float64(total_pdf)

    def weighted(self, c, k, cp, lp, ci, li):
        with np.errstate(all="ignore"):
            return c / k if self.mis is None else c * self.weight(cp, lp, ci, li)

    def contribution(self, cp, lp, ci, li, g):
        if ci == 0 or li == 0:
            return np.zeros(3)
        ce, le = cp[ci], lp[li]
        a_e, a_l = cp[ci - 1].color, lp[ci - 1].color  # sic: the light throughput is indexed with cameraIndex
        with np.errstate(invalid="ignore", divide="ignore"):
            cd = (ce.pos - le.pos) / np.linalg.norm(ce.pos - le.pos)
            wo = (lp[li - 1].pos - le.pos) / np.linalg.norm(lp[li - 1].pos - le.pos)
            fs_l = hm.eval_brdf(self.mat, cd, wo, le.N, le.dif, le.spec, le.rough, le.is_spec)
            if (fs_l == 0).all():
                return fs_l
            wo = (cp[ci - 1].pos - ce.pos) / np.linalg.norm(cp[ci - 1].pos - ce.pos)
            fs_e = hm.eval_brdf(self.mat, -cd, wo, ce.N, ce.dif, ce.spec, ce.rough, ce.is_spec)
            if (fs_e == 0).all():
                return fs_e
            return a_l * (fs_l * g * fs_e) * a_e

    # ---- the ray-generation shader for one pixel
    def pixel(self, x, y, world_pos, world_norm, dif4, spec4, emissive4, nee=True, splat=True, connect=True,
              connect_all_visible=False):
        """-> (rgba written to this pixel, [(target x, target y, rgb)] splats); inputs are the G-buffer texels."""
        D, minT = self.D, float(self.p.minT)
        if world_pos[3] == 0.0:
            return np.array([dif4[0], dif4[1], dif4[2], 1.0]), []
        out = np.zeros(4)  # getClearedTexture: the channel starts at 0 (BDPTPass.cpp:73)
        rough = spec4[3] * spec4[3]
        wp, wn = world_pos[:3].astype(np.float64), world_norm[:3].astype(np.float64)
        dif, spec = dif4[:3].astype(np.float64), spec4[:3].astype(np.float64)
        V = norm(self.cam_pos - wp)
        seed = hm_init_rand(x + y * self.W, int(self.p.frameCount))
        cam = [Vertex() for _ in range(9)]
        lig = [Vertex() for _ in range(9)]
        cam[0].pos, cam[0].N, cam[0].color, cam[0].pdf = self.cam_pos, norm(self.Wv), np.ones(3), 1.0
        w, out_dir, pdf, lobe, _ = hm.sample_brdf(self.mat, seed, wn, V, dif, spec, rough)
        cam[1] = Vertex(w, wp, wn, V, dif, spec, rough, (lobe and self.from_lobe) if self.mat == 0 else False, pdf)
        pl = self.payload(wp, out_dir, w, seed)
        depth = 1
        while depth < D and not pl["terminated"]:
            self.shoot(pl)
            cam[depth + 1] = self.vertex_of(pl)
            depth += 1
        seed = pl["seed"]
        seed, l_origin, l_dir, l_int = self.sample_light(seed)
        take = [True] * 9
        lig[0].pos, lig[0].color, lig[0].pdf = l_origin, l_int, 1.0 / self.nl
        lp = self.payload(l_origin, l_dir, l_int, seed)
        depth = 0
        while depth < D and not lp["terminated"]:
            self.shoot(lp)
            lig[depth + 1] = self.vertex_of(lp)
            take[depth + 1] = not lp["terminated"]
            depth += 1
        seed = lp["seed"]
        if (emissive4[:3] > 0.0).any():
            out = out + emissive4.astype(np.float64)
        for i in range(D if nee else 0):
            seed, direct = self.eval_direct(seed, cam[i + 1])
            c = self.clamp_vec(self.weighted(cam[i].color * direct, i + 2, cam, lig, i + 1, 0))
            out = out + np.append(np.zeros(3) if np.isnan(c).any() else c, 1.0)
        splats = []
        cam_n = norm(self.Wv)
        i = 0
        while splat and i < D and take[i + 1]:
            last_pos, last_n = lig[i + 1].pos, lig[i + 1].N
            to_cam = norm(self.cam_pos - last_pos)
            dist = float(np.linalg.norm(self.cam_pos - last_pos))
            if float(np.dot(cam_n, to_cam)) < 0 and self.visible(last_pos, to_cam, minT, dist):
                target = self.pixel_of(to_cam)
                t1 = hm.saturate(abs(float(np.dot(to_cam, cam_n))))
                t2 = hm.saturate(abs(float(np.dot(to_cam, last_n))))
                inv = 1.0 / dist
                g = t1 * t2 * inv * inv
                v = lig[i + 1]
                f = hm.eval_brdf(self.mat, v.V, norm(self.cam_pos - v.pos), v.N, v.dif, v.spec, v.rough, v.is_spec)
                c = self.clamp_vec(self.weighted((lig[i].color * f) * g, i + 2, cam, lig, 0, i + 1))
                if target is not None:
                    splats.append((target[0], target[1], np.zeros(3) if np.isnan(c).any() else c))
            i += 1
        for total in range(2, (D + 1) if connect else 2):
            for cl in range(1, D):
                ll = total - cl
                if ll < 0 or ll > 8:
                    continue  # uint underflow / out-of-range index in the shader: skipped
                g = self.g_without_v(cam[cl], lig[ll])
                a, b = cam[cl].pos, lig[ll].pos
                with np.errstate(invalid="ignore", divide="ignore"):
                    length = float(np.linalg.norm(b - a))
                    d = (b - a) / np.float64(length)
                if connect_all_visible or self.visible(a, d, minT, length):
                    c = self.clamp_vec(self.weighted(self.contribution(cam, lig, cl, ll, g), total, cam, lig, cl, ll))
                    out = np.fmin(np.fmax(out + np.append(np.zeros(3) if np.isnan(c).any() else c, 1.0), 0.0), 1.0)  # saturate
        return out, splats


def hm_init_rand(val0, val1, backoff=16):  # BDPTUtils.hlsli:91-103
    v0, v1, s0 = val0 & 0xFFFFFFFF, val1 & 0xFFFFFFFF, 0
    for _ in range(backoff):
        s0 = (s0 + 0x9e3779b9) & 0xFFFFFFFF
        v0 = (v0 + ((((v1 << 4) + 0xa341316c) & 0xFFFFFFFF) ^ ((v1 + s0) & 0xFFFFFFFF) ^ (((v1 >> 5) + 0xc8013ea4) & 0xFFFFFFFF))) & 0xFFFFFFFF
        v1 = (v1 + ((((v0 << 4) + 0xad90777d) & 0xFFFFFFFF) ^ ((v0 + s0) & 0xFFFFFFFF) ^ (((v0 >> 5) + 0x7e95761e) & 0xFFFFFFFF))) & 0xFFFFFFFF
    return v0


def gbuffer_pixel(scene, cam, gp, width, height, x, y, env_color):
    """GBufferRayGen + PrimaryClosestHit / PrimaryMiss (CommonPasses/Data/CommonPasses/lightProbeGBuffer.rt.hlsl:63-159)
    for constant materials and a constant environment -> dict of the channels written (None for a miss).

    Back faces are culled (RAY_FLAG_CULL_BACK_FACING_TRIANGLES); which winding is "front" is the driver's business,
    the build defines it as det > 0 in Moeller-Trumbore, and double-sided materials are exempt (Falcor marks their
    instances TRIANGLE_CULL_DISABLE)."""
    cam_pos = np.array(list(cam.posW), np.float64)
    U, V, W = (np.array(list(getattr(cam, k)), np.float64) for k in ("cameraU", "cameraV", "cameraW"))
    pc = np.array([(x + gp.pixelJitter[0]) / width, (y + gp.pixelJitter[1]) / height])
    ndc = np.array([2.0, -2.0]) * pc + np.array([-1.0, 1.0])
    ray_dir = ndc[0] * U + ndc[1] * V + W
    ray_dir = ray_dir / np.linalg.norm(W)
    focal = cam_pos + gp.focalLen * ray_dir
    seed = hm_init_rand(x + y * width, int(gp.frameCount))
    seed, r0 = hm.next_rand(seed)
    seed, r1 = hm.next_rand(seed)
    a, rad = 2.0 * M_PI * r0, gp.lensRadius * r1
    uv = np.array([math.cos(a) * rad, math.sin(a) * rad])
    origin = cam_pos + uv[0] * norm(U) + uv[1] * norm(V)
    o = origin if gp.useThinLens else cam_pos
    d = norm(focal - origin) if gp.useThinLens else norm(ray_dir)
    # closest front-facing hit, brute force
    e1, e2 = scene.v1 - scene.v0, scene.v2 - scene.v0
    p = np.cross(d, e2)
    det = np.einsum("ij,ij->i", e1, p)
    dbl = np.array([(scene.mats[m].flags >> 19) & 1 for m in scene.mat_id], bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = o - scene.v0
        u = np.einsum("ij,ij->i", tv, p) * inv
        q = np.cross(tv, e1)
        v = np.einsum("ij,j->i", q, d) * inv
        t = np.einsum("ij,ij->i", e2, q) * inv
    ok = ((det > 0) | (dbl & (det != 0))) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0.0) & (t < 1e38)
    if not ok.any():
        return dict(hit=False, dif=np.array([env_color[0], env_color[1], env_color[2], 1.0]))
    cand = np.nonzero(ok)[0]
    prim = int(cand[np.argmin(t[cand])])
    pos, n, vv, dif, spec, rough = scene.shading(prim, float(u[prim]), float(v[prim]), o, d, float(t[prim]))
    # the G-buffer's V is toward gCamera.posW (prepareShadingData(vsOut, gMaterial, gCamera.posW, 0)), also for thin-lens origins
    m = scene.mats[scene.mat_id[prim]]
    n2 = norm(scene.n0[prim] * (1.0 - u[prim] - v[prim]) + scene.n1[prim] * u[prim] + scene.n2[prim] * v[prim])
    v_cam = norm(cam_pos - pos)
    if float(np.dot(n2, v_cam)) <= 0 and ((m.flags >> 19) & 1):
        n2 = -n2
    return dict(hit=True, pos=pos, N=n2, dist=float(np.linalg.norm(pos - cam_pos)), dif=np.append(dif, m.baseColor[3]),
                spec=np.append(spec, math.sqrt(rough)), ior=m.IoR, prim=prim)
