"""The oracle against a second, independent reading of the shader code (tests/hlsl_reference_math.py: float64,
true cos/sin/pow, written from the HLSL files).  The reference pins nothing for these functions, so this is the
strongest check available that the oracle's BSDF arithmetic says what BRDFUtils.hlsli / MaterialUtils.hlsli say —
and it bounds the error of the oracle's fixed polynomial transcendental forms."""
import os

import numpy as np

import hlsl_reference_math as hm

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz"))


def _rel(a, b, floor):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert (np.isfinite(a) == np.isfinite(b)).all()
    m = np.isfinite(a)
    return float(np.max(np.abs(a[m] - b[m]) / (floor + np.abs(b[m])))) if m.any() else 0.0


def test_bsdf_records_match_float64_reading_of_the_hlsl(pkg, ob):
    rin = GOLD["bsdf_in"]
    seeds = rin.view(np.uint32)[:, 17]
    for mat in (0, 1, 2):
        out = np.zeros((rin.shape[0], 16), np.float32)
        ob.load_oracle(pkg.abi).oracle_bsdf(rin.ctypes.data, rin.shape[0], mat, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), GOLD[f"bsdf_out_{mat}"].view(np.uint32))  # the live oracle == the fixture
        worst = dict(w=0.0, L=0.0, pdf=0.0, f=0.0)
        checked = 0
        for i in range(rin.shape[0]):
            r = rin[i, :17].astype(np.float64)  # [17] holds the seed bits
            n, v, l, dif, spec, rough = r[0:3], r[3:6], r[6:9], r[9:12], r[12:15], r[15]
            w, L, pdf, lobe, margin = hm.sample_brdf(mat & 1, int(seeds[i]), n, v, dif, spec, rough)
            if margin < 1e-6:
                continue  # the lobe choice sits on a rounding boundary
            f = hm.eval_brdf(mat & 1, v, l, n, dif, spec, rough, rin[i, 16] != 0)
            o = out[i].astype(np.float64)
            worst["w"] = max(worst["w"], _rel(o[0:3], w, 1e-4))
            worst["L"] = max(worst["L"], float(np.max(np.abs(o[3:6] - L))))
            worst["pdf"] = max(worst["pdf"], _rel(o[6], pdf, 1e-4))
            worst["f"] = max(worst["f"], _rel(o[8:11], f, 1e-4))
            # isSpecular out-parameter: never written by sampleGGXBRDF -> defined false (mat 0), "lobe picked" with the flag (mat 2)
            assert o[7] == (1.0 if (mat == 2 and lobe) else 0.0)
            checked += 1
        assert checked > 0.99 * rin.shape[0]
        assert worst["w"] < 5e-4 and worst["pdf"] < 5e-4 and worst["f"] < 5e-4 and worst["L"] < 1e-4, (mat, worst)


def test_deterministic_sincos_against_libm(ob, pkg):
    """det_sincos2pi (the stand-in for HLSL sin/cos of 2*pi*u) against numpy's double-precision functions."""
    u = GOLD["sincos_u"].astype(np.float64)
    assert np.max(np.abs(GOLD["sincos_s"] - np.sin(2 * np.pi * u))) < 2e-6
    assert np.max(np.abs(GOLD["sincos_c"] - np.cos(2 * np.pi * u))) < 2e-6


def _frame_params(pkg, depth, mat, flags, min_t):
    A = pkg.abi
    gp = A.GBufferParams()
    gp.pixelJitter[0], gp.pixelJitter[1] = pkg.msaa_jitter(0xdeadbeef)
    gp.frameCount, gp.focalLen, gp.lensRadius = 0xdeadbeef, 1.0, 1.0 / 64.0
    gp.envWidth = gp.envHeight = 128
    for i, c in enumerate((0.5, 0.5, 0.8, 1.0)):
        gp.envColor[i] = c
    p = A.Params()
    p.minT, p.frameCount, p.matIndex, p.maxDepth = min_t, 0x1337, mat, depth
    p.refractiveIndex, p.emitMult, p.clampUpper, p.flags = 1.0, 1.0, 0.9, flags
    p.pixelJitter[0], p.pixelJitter[1] = pkg.msaa_jitter(0x1337)
    return gp, p


import pytest


@pytest.mark.parametrize("size,depth,mat,mis", [(16, 3, 1, None), (16, 3, 0, None), (12, 5, 1, None), (10, 4, 0, None),
                                                  (12, 3, 1, "power"), (12, 4, 0, "linear")])
def test_integrator_matches_float64_reading_of_the_raygen_shader(pkg, ob, size, depth, mat, mis):
    """The whole ray-generation shader (BDPTMain.rt.hlsl:42-234: both walks, NEE, light-tracing splats, vertex
    connections with all its indexing quirks) re-read independently in float64 (tests/hlsl_integrator_numpy.py, own
    brute-force intersection) against the oracle, stage by stage on a small Cornell frame.

    Two things make a cross-precision comparison possible at all: gMinT = 0.05 instead of 1e-4 (in the 555-unit box a
    hit point is only known to ~1e-4, so with the default every grazing ray's self-intersection is a coin flip), and
    the ORACLE_CONNECT_ALL_VISIBLE hook (the reference ends connection rays exactly ON the far surface, so that
    surface occludes them or not depending on the last bit).  With those, every pixel, splat target and splat count
    agrees.  The last two cases switch the uniform 1/k for getWeightPower / getWeightLinear (BDPTUtils.hlsli:226-278),
    which the reference defines but never calls.  (Power at depth 3 only: its pE^2 * pL^2 products of per-vertex
    pdf * G factors (G ~ 1e-6 in the 555-unit box) leave the fp32 range from depth 4 on, so there the fp32 weights of
    oracle and HIP path — identical to each other — are not comparable with a float64 evaluation.)"""
    import hlsl_integrator_numpy as hi
    A = pkg.abi
    mis_flag = {None: 0, "power": A.PARAM_MIS_POWER, "linear": A.PARAM_MIS_LINEAR}[mis]
    scene = pkg.Scene.cornell()
    cam = scene.camera(1.0)
    stages = (("nee", A.PARAM_NO_SPLAT | A.PARAM_NO_CONNECT, dict(splat=False, connect=False), 0),
              ("splat", A.PARAM_NO_NEE | A.PARAM_NO_CONNECT, dict(nee=False, connect=False), 0),
              ("connect", A.PARAM_NO_NEE | A.PARAM_NO_SPLAT, dict(nee=False, splat=False, connect_all_visible=True),
               ob.ORACLE_CONNECT_ALL_VISIBLE))
    for name, flags, kw, oflags in stages:
        gp, p = _frame_params(pkg, depth, mat, flags | mis_flag, 0.05)
        orc = ob.OracleRender(A, scene.desc, size, size)
        orc.gbuffer(cam, gp)
        orc.bdpt(cam, p, flags=oflags)
        own = orc.image().astype(np.float64)           # own-pixel terms (before the splat fold-in)
        splat = orc.splat.astype(np.float64)
        splat[:, :3] /= 2.0 ** 32
        R = hi.Renderer(hi.Scene(scene.desc), cam, p, size, size, mis=mis)
        img = np.zeros((size, size, 4))
        spl = np.zeros((size * size, 4))
        for y in range(size):
            for x in range(size):
                i = y * size + x
                with np.errstate(all="ignore"):
                    o, ss = R.pixel(x, y, orc.chan["worldPosition"][i], orc.chan["worldNormal"][i], orc.chan["materialDiffuse"][i],
                                    orc.chan["materialSpecRough"][i], orc.chan["emissive"][i], **kw)
                img[y, x] = o
                for tx, ty, c in ss:
                    spl[ty * size + tx, :3] += c
                    spl[ty * size + tx, 3] += 1
        assert np.abs(img - own).max() < 2e-5, (name, np.abs(img - own).max())
        assert np.array_equal(spl[:, 3], splat[:, 3]), name     # same splats land on the same pixels
        assert np.abs(spl[:, :3] - splat[:, :3]).max() < 2e-5, name
        if name == "splat":
            assert splat[:, 3].sum() > size * size / 2
        orc.close()
    scene.close()


@pytest.mark.parametrize("thin_lens", [0, 1])
def test_gbuffer_matches_float64_reading_of_the_primary_shaders(pkg, ob, thin_lens):
    """lightProbeGBuffer.rt.hlsl:63-159 (pinhole and thin-lens ray generation, closest-hit channel writes, miss colour)
    re-read in float64 against the oracle's G-buffer on the Cornell box; half-precision channels to half precision."""
    import hlsl_integrator_numpy as hi
    size = 24
    scene = pkg.Scene.cornell()
    cam = scene.camera(1.0)
    gp, _ = _frame_params(pkg, 3, 0, 0, 1e-4)
    gp.useThinLens, gp.lensRadius, gp.focalLen = thin_lens, 4.0, 700.0
    orc = ob.OracleRender(pkg.abi, scene.desc, size, size)
    orc.gbuffer(cam, gp)
    sc = hi.Scene(scene.desc)
    hits = 0
    for y in range(size):
        for x in range(size):
            i = y * size + x
            with np.errstate(all="ignore"):
                g = hi.gbuffer_pixel(sc, cam, gp, size, size, x, y, (0.5, 0.5, 0.8))
            wp, wn = orc.chan["worldPosition"][i], orc.chan["worldNormal"][i]
            if not g["hit"]:
                assert wp[3] == 0.0 and np.allclose(orc.chan["materialDiffuse"][i], g["dif"], atol=1e-3)
                continue
            hits += 1
            assert wp[3] == 1.0 and np.abs(wp[:3] - g["pos"]).max() < 2e-3, (x, y, wp, g["pos"])
            assert np.abs(wn[:3] - g["N"]).max() < 2e-3 and abs(wn[3] - g["dist"]) < 0.6           # fp16: 0.5 at ~800
            assert np.abs(orc.chan["materialDiffuse"][i] - g["dif"]).max() < 1e-3
            assert np.abs(orc.chan["materialSpecRough"][i] - g["spec"]).max() < 1e-3
            assert abs(orc.chan["materialExtra"][i][0] - g["ior"]) < 1e-3
    assert hits > 0.9 * size * size
    orc.close()
    scene.close()


def _relit_desc(pkg, scene):
    """The scene's description with three lights of the three kinds Falcor's evalLight distinguishes: a spot light with
    a penumbra (Lights.slang:75-102), a plain point light and a directional light."""
    import ctypes as C
    import math
    A = pkg.abi
    d = A.SceneDesc()
    C.memmove(C.byref(d), C.byref(scene.desc), C.sizeof(A.SceneDesc))
    lights = (A.Light * 3)()

    def fill(l, typ, pos, direction, intensity, opening=math.pi, penumbra=0.0):
        l.type = typ
        n = math.sqrt(sum(x * x for x in direction))
        for k in range(3):
            l.posW[k], l.dirW[k], l.intensity[k] = pos[k], direction[k] / n, intensity[k]
        l.openingAngle, l.cosOpeningAngle, l.penumbraAngle = opening, math.cos(opening), penumbra

    fill(lights[0], A.LIGHT_POINT, (-3.0, 6.5, 0.4), (0.35, -1.0, -0.1), (60.0, 52.0, 40.0), opening=0.75, penumbra=0.3)
    fill(lights[1], A.LIGHT_POINT, (6.0, 3.0, -0.8), (0.0, -1.0, 0.0), (14.0, 16.0, 22.0))
    fill(lights[2], A.LIGHT_DIRECTIONAL, (0.0, 30.0, 0.0), (0.3, -1.0, 0.25), (0.9, 0.85, 0.7))
    d.lights, d.numLights = lights, 3
    return d, lights


def test_textured_alpha_masked_spot_lit_frame_matches_float64_reading(pkg, ob):
    """The rows the Cornell cross-check cannot reach: vertex
    fetch + material decode + sRGB / bilinear / wrap texture sampling (MetalRough and SpecGloss materials), the any-hit
    alpha test on primary, extension and shadow rays, the primary hit's normal map, and the spot-light / penumbra branch
    of evalPointLight — re-read in float64 from the HLSL / Slang (tests/hlsl_textured_numpy.py) and compared with the
    oracle on a small frame of the foliage courtyard: every G-buffer channel, then the NEE image of a depth-3 pass.

    Pixels are allowed to disagree only where a float32 and a float64 ray fall on different sides of a discontinuity
    (triangle and leaf-silhouette edges, shadow boundaries, the light pick): a small share, asserted below."""
    import hlsl_integrator_numpy as hi
    import hlsl_textured_numpy as ht
    A = pkg.abi
    W, H, depth = 44, 28, 3
    scene = pkg.Scene.courtyard(5, 7000, 0.35)
    desc, keep = _relit_desc(pkg, scene)
    cam = scene.camera(W / H)
    gp, p = _frame_params(pkg, depth, 0, A.PARAM_NO_SPLAT | A.PARAM_NO_CONNECT, 2e-3)
    orc = ob.OracleRender(A, desc, W, H)
    orc.gbuffer(cam, gp)
    orc.bdpt(cam, p)
    nee = orc.image().astype(np.float64)
    sc = ht.TexturedScene(desc)
    assert len(sc.textures) >= 7 and (sc.tri_alpha_mode != 0).sum() > 1000
    models = {int(m.flags) & 7 for m in sc.mats}
    assert models == {0, 2}, models                       # both shading models occur
    assert any(((int(m.flags) >> 12) & 3) != 0 for m in sc.mats)  # and a normal-mapped material
    R = hi.Renderer(sc, cam, p, W, H)
    f16 = lambda a: np.asarray(a, np.float64).astype(np.float16).astype(np.float64)  # the channel's storage format
    rel = lambda a, b: np.max(np.abs(np.asarray(a, np.float64) - b) / (1e-3 + np.abs(b)))

    def half_err(stored, exact):
        """A half-precision channel against the float64 value it should hold: the error beyond half a unit in the last
        place of the half format (what rounding alone may cost), relative to the vector's largest component — the
        float32 and float64 values differ by ~1e-6 of that and may round to neighbouring halves."""
        stored, exact = np.asarray(stored, np.float64), np.asarray(exact, np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(exact), 2.0 ** -14))) - 10)
        return float(np.max(np.maximum(0.0, np.abs(stored - exact) - 0.5 * ulp)) / max(1e-3, float(np.max(np.abs(exact)))))
    n_hit = n_gb_ok = n_pix = n_nee_ok = n_alpha_pixels = 0
    worst_gb = worst_nee = 0.0
    for y in range(H):
        for x in range(W):
            i = y * W + x
            n_pix += 1
            with np.errstate(all="ignore"):
                g = ht.gbuffer_pixel(sc, cam, gp, W, H, x, y, (0.5, 0.5, 0.8))
            wp = orc.chan["worldPosition"][i]
            if not g["hit"] or wp[3] == 0.0:
                same = (not g["hit"]) and wp[3] == 0.0 and np.allclose(orc.chan["materialDiffuse"][i], f16(g["dif"]), atol=1e-3)
                n_gb_ok += bool(same)
                n_nee_ok += bool(same)  # the pass copies the environment colour through (BDPTMain.rt.hlsl:62-66)
                continue
            n_hit += 1
            n_alpha_pixels += int(sc.tri_alpha_mode[g["prim"]] != 0)
            errs = [rel(wp[:3], g["pos"]),
                    half_err(orc.chan["worldNormal"][i][:3], g["N"]), half_err(orc.chan["worldNormal"][i][3:], [g["dist"]]),
                    half_err(orc.chan["materialDiffuse"][i], np.append(g["dif"], g["opacity"])),
                    half_err(orc.chan["materialSpecRough"][i], np.append(g["spec"], g["linear_roughness"])),
                    half_err(orc.chan["materialExtra"][i][:1], [g["ior"]]),
                    half_err(orc.chan["emissive"][i][:3], g["emissive"])]
            gb_ok = max(errs) < 1e-4
            n_gb_ok += gb_ok
            if gb_ok:
                worst_gb = max(worst_gb, max(errs))
                # the BDPT pass of the float64 reading starts from the ORACLE's G-buffer texels, as the shader would
                with np.errstate(all="ignore"):
                    o, _ = R.pixel(x, y, wp, orc.chan["worldNormal"][i], orc.chan["materialDiffuse"][i], orc.chan["materialSpecRough"][i],
                                   orc.chan["emissive"][i], splat=False, connect=False)
                e = float(np.max(np.abs(o[:3] - nee[y, x, :3]) / (1e-3 + np.abs(nee[y, x, :3]))))
                ok = e < 5e-4 and o[3] == nee[y, x, 3]
                n_nee_ok += ok
                if ok:
                    worst_nee = max(worst_nee, e)
    assert n_hit > 0.6 * n_pix and n_alpha_pixels > 10, (n_hit, n_alpha_pixels)
    assert n_gb_ok >= 0.985 * n_pix, (n_gb_ok, n_pix)
    assert n_nee_ok >= 0.97 * n_pix, (n_nee_ok, n_pix)
    assert worst_gb < 1e-4 and worst_nee < 5e-4
    lit = nee[..., :3].sum(axis=-1) > 1e-4
    assert lit.mean() > 0.2  # the comparison is not about black pixels
    print("textured cross-check: %d px, %d hits (%d on alpha-masked triangles), G-buffer agrees on %d, NEE image on %d; worst %.2e / %.2e" % (
        n_pix, n_hit, n_alpha_pixels, n_gb_ok, n_nee_ok, worst_gb, worst_nee))
    orc.close()
    scene.close()


def test_textured_light_walk_and_connections_match_float64_reading(pkg, ob):
    """The same second reading (tests/hlsl_textured_numpy.py) for the stages whose vertices come from the LIGHT walk —
    light-tracing splats and vertex connections — on the textured, alpha-masked, spot-lit courtyard: the light
    sub-path's hits go through vertex fetch, material decode and texture sampling too, and its rays through the
    any-hit alpha test.  Connections with ORACLE_CONNECT_ALL_VISIBLE (their rays end exactly on the far surface)."""
    import hlsl_integrator_numpy as hi
    import hlsl_textured_numpy as ht
    A = pkg.abi
    W, H, depth = 26, 16, 3
    scene = pkg.Scene.courtyard(5, 7000, 0.35)
    desc, keep = _relit_desc(pkg, scene)
    cam = scene.camera(W / H)
    sc = ht.TexturedScene(desc)
    stages = (("splat", A.PARAM_NO_NEE | A.PARAM_NO_CONNECT, dict(nee=False, connect=False), 0),
              ("connect", A.PARAM_NO_NEE | A.PARAM_NO_SPLAT, dict(nee=False, splat=False, connect_all_visible=True), ob.ORACLE_CONNECT_ALL_VISIBLE))
    for name, flags, kw, oflags in stages:
        gp, p = _frame_params(pkg, depth, 0, flags, 2e-3)
        orc = ob.OracleRender(A, desc, W, H)
        orc.gbuffer(cam, gp)
        orc.bdpt(cam, p, flags=oflags)
        own = orc.image().astype(np.float64)
        splat = orc.splat.astype(np.float64)
        splat[:, :3] /= 2.0 ** 32
        R = hi.Renderer(sc, cam, p, W, H)
        img = np.zeros((H, W, 4))
        spl = np.zeros((W * H, 4))
        for y in range(H):
            for x in range(W):
                i = y * W + x
                with np.errstate(all="ignore"):
                    o, ss = R.pixel(x, y, orc.chan["worldPosition"][i], orc.chan["worldNormal"][i], orc.chan["materialDiffuse"][i],
                                    orc.chan["materialSpecRough"][i], orc.chan["emissive"][i], **kw)
                img[y, x] = o
                for tx, ty, c in ss:
                    spl[ty * W + tx, :3] += c
                    spl[ty * W + tx, 3] += 1
        n = W * H
        if name == "splat":
            # a light path is per PIXEL (its seed), its splat lands anywhere: compare per source pixel via the totals and
            # per target pixel where the same number of splats landed
            same_count = spl[:, 3] == splat[:, 3]
            assert same_count.mean() > 0.97, same_count.mean()
            err = np.abs(spl[same_count, :3] - splat[same_count, :3]) / (1e-3 + np.abs(splat[same_count, :3]))
            assert (err.max(axis=1) < 1e-3).mean() > 0.99 and splat[:, 3].sum() > n / 4
            assert abs(spl[:, 3].sum() - splat[:, 3].sum()) <= 0.03 * splat[:, 3].sum()
        else:
            err = np.abs(img[..., :3] - own[..., :3]).max(axis=-1) / (1e-3 + np.abs(own[..., :3]).max(axis=-1))
            ok = (err < 1e-3) & (img[..., 3] == own[..., 3])
            assert ok.mean() > 0.97, ok.mean()
            assert (own[..., :3].sum(axis=-1) > 1e-5).mean() > 0.2
        orc.close()
    scene.close()
