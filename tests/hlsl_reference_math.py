"""A second, independent reading of the reference's BSDF shader code in float64 numpy with the true
cos/sin/pow — used only to cross-check the oracle's restatement (which is fp32 with fixed polynomial
transcendental forms).  Written from the HLSL, not from the oracle:

  nextRand                          BDPT/Data/BDPTUtils.hlsli:105-110
  probabilityToSampleDiffuse        BDPT/Data/MaterialUtils.hlsli:22-27
  getPerpendicularVector            MaterialUtils.hlsli:31-38
  getCosHemisphereSample            MaterialUtils.hlsli:41-54
  ggxNormalDistribution / ggxSchlickMaskingTerm / schlickFresnel / getGGXMicrofacet / ggxLighting
                                    BDPT/Data/BRDFUtils.hlsli:5-76
  evalGGXBRDF / sampleGGXBRDF       MaterialUtils.hlsli:185-252
  evalLambertianBRDF / sampleLambertianBRDF   MaterialUtils.hlsli:313-330
"""
import math

import numpy as np

M_PI = 3.14159265358979323846
M_1_PI = 0.318309886183790671538


def next_rand(s):
    s = (1664525 * s + 1013904223) & 0xFFFFFFFF
    return s, float(s & 0x00FFFFFF) / float(0x01000000)


def luminance(c):  # Falcor Data/HostDeviceSharedCode.h / Helpers.slang: Rec.709 weights
    return 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2]


def saturate(x):
    if x != x:  # HLSL saturate(NaN) = 0
        return 0.0
    return min(max(x, 0.0), 1.0)


def prob_diffuse(dif, spec):
    ld = max(0.01, luminance(dif))
    ls = max(0.01, luminance(spec))
    return ld / (ld + ls)


def perpendicular(u):
    a = np.abs(u)
    xm = 1 if ((a[0] - a[1]) < 0 and (a[0] - a[2]) < 0) else 0
    ym = (1 ^ xm) if (a[1] - a[2]) < 0 else 0
    zm = 1 ^ (xm | ym)
    return np.cross(u, np.array([xm, ym, zm], np.float64))


def cos_hemisphere(seed, n):
    seed, r0 = next_rand(seed)
    seed, r1 = next_rand(seed)
    b = perpendicular(n)
    t = np.cross(b, n)
    r = math.sqrt(r0)
    phi = 2.0 * 3.14159265 * r1
    return seed, t * (r * math.cos(phi)) + b * (r * math.sin(phi)) + n * math.sqrt(max(0.0, 1.0 - r0))


def ggx_d(ndoth, rough):
    a2 = rough * rough
    d = (ndoth * a2 - ndoth) * ndoth + 1
    return a2 / max(0.001, d * d * M_PI)


def ggx_g(ndotl, ndotv, rough):
    k = np.float64(rough * rough / 2)
    with np.errstate(divide="ignore", invalid="ignore"):  # 0/0 -> NaN as on the GPU, not an exception
        return (np.float64(ndotv) / (ndotv * (1 - k) + k)) * (np.float64(ndotl) / (ndotl * (1 - k) + k))


def schlick(f0, u):
    return f0 + (1.0 - f0) * (1.0 - u) ** 5.0


def ggx_microfacet(seed, rough, n):
    seed, r0 = next_rand(seed)
    seed, r1 = next_rand(seed)
    b = perpendicular(n)
    t = np.cross(b, n)
    a2 = rough * rough
    cos_t = math.sqrt(max(0.0, (1.0 - r0) / ((a2 - 1.0) * r0 + 1)))
    sin_t = math.sqrt(max(0.0, 1.0 - cos_t * cos_t))
    phi = r1 * M_PI * 2.0
    return seed, t * (sin_t * math.cos(phi)) + b * (sin_t * math.sin(phi)) + n * cos_t


def ggx_lighting(h, l, n, ndotl, ndotv, rough, spec):
    ndoth = saturate(float(np.dot(n, h)))
    ldoth = saturate(float(np.dot(l, h)))
    d = ggx_d(ndoth, rough)
    g = ggx_g(ndotl, ndotv, rough)
    f = schlick(spec, ldoth)
    with np.errstate(divide="ignore", invalid="ignore"):
        prob = np.float64(d * ndoth) / np.float64(4 * ldoth)
        val = d * g * f / np.float64(4 * ndotl * ndotv)
    return val, float(prob)


def sample_brdf(mat_index, seed, n, v, dif, spec, rough):
    """-> (weight3, L3, pdf, lobe_was_specular, margin) ; margin = |rand - probDiffuse| of the lobe choice."""
    if mat_index == 1:
        seed, l = cos_hemisphere(seed, n)
        return dif.copy(), l, saturate(float(np.dot(n, l))) * M_1_PI, False, 1.0
    pd = prob_diffuse(dif, spec)
    seed, r = next_rand(seed)
    choose_diffuse = r < pd
    ndotv = saturate(float(np.dot(n, v)))
    if choose_diffuse:
        seed, l = cos_hemisphere(seed, n)
        if float(np.dot(n, l)) <= 0.0:
            return np.zeros(3), l, 0.0, False, abs(r - pd)
        ndotl = saturate(float(np.dot(n, l)))
        return dif / pd, l, (ndotl * M_1_PI) * pd, False, abs(r - pd)
    seed, h = ggx_microfacet(seed, rough, n)
    l = 2.0 * float(np.dot(v, h)) * h - v
    l = l / np.linalg.norm(l)
    if float(np.dot(n, l)) <= 0.0:
        return np.zeros(3), l, 0.0, True, abs(r - pd)
    ndotl = saturate(float(np.dot(n, l)))
    term, prob = ggx_lighting(h, l, n, ndotl, ndotv, rough, spec)
    pdf = prob * (1.0 - pd)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = ndotl * term / np.float64(prob * (1.0 - pd))
    return w, l, pdf, True, abs(r - pd)


def eval_brdf(mat_index, v, l, n, dif, spec, rough, is_specular):
    if mat_index == 1:
        return dif.copy()
    if not is_specular:
        if float(np.dot(n, l)) <= 0.0:
            return np.zeros(3)
        return dif * M_1_PI
    h = (l + v) / np.linalg.norm(l + v)
    if float(np.dot(n, l)) <= 0.0:
        return np.zeros(3)
    ndotl = saturate(float(np.dot(n, l)))
    ndotv = saturate(float(np.dot(n, v)))
    val, _ = ggx_lighting(h, l, n, ndotl, ndotv, rough, spec)
    return val
