#!/usr/bin/env python3
"""bench.py — Mrays/s of the BDPT render pass on MI355X (BASELINE.json metric).

One "step" = one pipeline frame of the hot path (G-buffer pass + BDPT pass + accumulation) at
1 spp over the whole 1920x1080 image, depth 8, on the Sponza stand-in scene (seeded procedural
atrium, 262,144 triangles, textured GGX — the real Sponza asset is in neither the reference
tree nor this image).  With N GPUs the image is dealt to the ranks in interleaved
stripes of rows (scene replicated); each rank renders its stripes, the fixed-point splat buffers
are summed with one RCCL reduce-scatter per frame, and each rank resolves + accumulates its rows.
value = rays actually traced by all ranks / max-over-ranks wall time of the K timed steps.

The timed loop never synchronises with the host: frames are handed to `frames_in_flight` contexts
round-robin (three by default, also at N = 1: the tail of one frame's persistent launches overlaps
the next frame's walks), the running mean is applied in frame order through an event chain, and the
ray tallies add up on the device.  Stage times, visit counts and the roofline block come from
untimed frames run ALONE before the timed region (their HIP events see one frame's kernels only).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--scene atrium|cornell] ...
`python bench.py --gpus N` with N > 1 starts its own ranks (one process per GPU, a torch.distributed.run child started
before anything in this process touches a GPU); launched under torch.distributed.run it is one of the ranks.
The per-rank frame loop is the package's (fyp-bidirectionalpathtracer_amd/tiling.py TileRenderer).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAY_KEYS = ("raysEyeExtend", "raysLightExtend", "raysNee", "raysSplat", "raysConnect")
# any-hit queries answered "occluded" by their occluder hint (one triangle of the scene tried before the traversal:
# DESIGN.md section 4): queries the integrator issued and the pass answered, counted in `value` like every other any-hit
# query and reported on their own as well (config.rays_hinted_per_frame, config.value_traversals_only)
HINT_KEYS = ("hintedNee", "hintedSplat")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; its measured float4 copy rate is 6290 GB/s)
# Rates the traversal kernels are priced against (MI355X_MICROARCH.md "Indexed rows: gather", chip-wide; the low ends):
L2_GATHER_GBS = 16800.0         # rows served by the XCDs' L2s
MALL_GATHER_GBS = 8600.0        # uniformly random rows served by the Infinity Cache
# Measured here (tools/microbench, profiles/r2/microbench/): a divergent wave costs the vector-memory address unit one
# clock per lane and 16-byte load — 1.03 lane-loads per clock per CU, flat from 8 to 32 waves per CU; VALU issue on one
# SIMD: v_fma / v_mul / v_add / v_sub_f32, v_add_u32, v_and, v_mov 2.2-2.7 clk per wave-instruction, everything else
# the kernels use (v_min / v_max / v_max3 / v_cvt_f32_ubyte / v_cmp / v_cndmask / v_lshl_add / v_bfi) 4.2-4.4
LANE_LOADS_PER_CLK_PER_CU = 1.03
VALU_CLK_FAST, VALU_CLK_SLOW = 2.4, 4.3
SHADER_CLK_HZ = 2.4e9
NUM_CUS, NUM_SIMDS = 256, 1024
PROFILE_ROUND = "r5"   # profiles/<round>/roofline_pmc.json holds the PMC counts the roofline block quotes


SURVEY_NODE_BYTES, SURVEY_TRI_BYTES = 64, 48  # algorithmic bytes per interior-node / triangle visit


def num_connect_pairs(D):
    """Connection pairs the reference defines for depth D (BDPTMain.rt.hlsl:212-216)."""
    return sum(min(t, D - 1) for t in range(2, D + 1))


def algorithmic_bytes(cnt, n_pairs_eval, n_pix_tile, node_b, tri_b, D):
    """SURVEY.md §8(d), per frame and per kernel: B_ray = 32 + nodeBytes*n_int + triBytes*n_tri + O (O = 4 shadow /
    20 closest); 388 B per closest-hit shade; 168 B per connection pair and 84 B per NEE/splat term for the generators.
    The gather side of every term (SURVEY: "+ 32 B pixel RMW" per term) is costed where it happens, in the per-pixel
    kernels, at what one term really needs there: 4 B slot id per slot, 1 B visibility per ray,
    12 B contribution per ray, ONE 32 B read-modify-write of the pixel (the sums run in registers), 4 B target index per
    splat slot and 32 B of atomics per landed splat; init_paths: 56 B G-buffer + 16 B output per pixel and, per valid
    pixel, two 96-B vertices, 24 B ray directions, 8 B seeds, 3 flag bytes, 4 B queue entry; clear + resolve: 32 B of
    splat accumulator written and read per pixel, 32 B pixel RMW per pixel a splat landed on (<= splats)."""
    closest = cnt["raysEyeExtend"] + cnt["raysLightExtend"]
    shadow = cnt["raysNee"] + cnt["raysSplat"] + cnt["raysConnect"]
    valid = cnt["pixelsValid"]
    pairs = num_connect_pairs(D)
    b = {}
    # walk_kernel: every extension ray of both walks and the hit/miss shader that follows it
    b["walk_kernel"] = 52 * closest + node_b * cnt["nodeVisitsClosest"] + tri_b * cnt["triTestsClosest"] + 388 * closest
    # trace_shadow_kernel: main launch + lazy rounds
    b["trace_shadow_kernel"] = 36 * shadow + node_b * cnt["nodeVisitsShadow"] + tri_b * cnt["triTestsShadow"]
    # gen_nee + gen_splat + gen_connect (+ lazy_gen): the vertex reads of every term and pair, the ray they append (28 B + 12 B)
    # (a term its occluder hint answered read its vertex like any other, plus 4 B of hint word and the 48-B triangle record)
    hinted = cnt.get("hintedNee", 0) + cnt.get("hintedSplat", 0)
    b["gen_kernels"] = 84 * (cnt["raysNee"] + cnt["raysSplat"] + hinted) + 52 * hinted + 168 * n_pairs_eval + 40 * shadow
    # clear + init_paths + gather + lazy_check + resolve
    b["per_pixel_kernels"] = (n_pix_tile * (56 + 16 + 32 + 32) + valid * (2 * 96 + 24 + 8 + 3 + 4) +
                              valid * (4 * (2 * D + pairs) + 32 + 4 * D) + shadow * (1 + 12) + cnt["splatsLanded"] * (32 + 32))
    return b


def load_pmc(pkg, sub=""):
    """Per-kernel, per-frame hardware counts of the committed rocprofv3 --pmc passes of a workload (tools/prof_pmc.sh for
    the bench frame, tools/prof_config_pmc.sh for the configs[3] / configs[4] shapes -> tools/roofline_pmc.py):
    {kernel base name: {counter: value per frame}} from profiles/<round>/<sub>/roofline_pmc.json.  The file names the
    sources it was measured on (source_hash); counts of another build are not quoted (pmc_build_match false)."""
    pj = os.path.join(ROOT, "profiles", PROFILE_ROUND, sub, "roofline_pmc.json")
    if not os.path.exists(pj):
        return {}
    with open(pj) as f:
        raw = json.load(f)
    if raw.get("source_hash") != pkg.source_hash():
        return {"_mismatch": True}
    # counts per frame by kernel NAME: a frame runs one template variant of each kernel (the statistics frame the variant
    # with visit counters), so the variants of a name add up to whole frames (tools/roofline_pmc.py "by_name")
    out = dict(raw["by_name"])
    out["_meta"] = {k: v for k, v in raw.items() if k not in ("kernels", "by_name")}
    return out


KERNEL_MEMBERS = {"walk_kernel": ["walk_kernel"], "trace_shadow_kernel": ["trace_shadow_kernel"],
                  "gen_kernels": ["gen_nee_kernel", "gen_splat_kernel", "gen_connect_kernel", "lazy_gen_kernel"],
                  "per_pixel_kernels": ["init_paths_kernel", "gather_kernel", "lazy_check_kernel", "resolve_kernel"]}


def kernel_durations(ms):
    """Stage times (bdpt_get_stage_times: one stage = one kernel of the launch stream, a wait for the second stream, or a
    "side:" kernel of the second stream) -> summed KERNEL time per roofline block, the waits, and the critical path."""
    k = {"walk_kernel": ms.get("walk", 0.0),
         "trace_shadow_kernel": ms.get("trace_terms", 0.0) + ms.get("trace_pairs", 0.0) + ms.get("lazy_trace", 0.0),
         # gen_splat and gen_connect run on the second stream beside gen_nee / trace_terms: their own event pairs
         "gen_kernels": ms.get("gen_nee", 0.0) + ms.get("gen_splat", 0.0) + ms.get("gen_connect", 0.0) +
                        ms.get("side:gen_splat", 0.0) + ms.get("side:gen_connect", 0.0) + ms.get("lazy_gen", 0.0),
         "per_pixel_kernels": ms.get("clear", 0.0) + ms.get("init_paths", 0.0) + ms.get("mis_prefix", 0.0) + ms.get("gather", 0.0) +
                              ms.get("lazy_check", 0.0) + ms.get("resolve", 0.0)}
    waits = ms.get("splat_wait", 0.0) + ms.get("connect_wait", 0.0) + ms.get("tail_wait", 0.0)
    critical = sum(v for n, v in ms.items() if not n.startswith("side:"))
    return k, waits, critical


def roofline_kernels(kernel_ms, alg, fetched, pmc):
    """One roofline entry per kernel block: the contract's figure (algorithmic bytes / kernel time / HBM peak) and, when
    PMC counts of this build are committed, the HBM-side traffic and the fractions of the ceilings that can bind."""
    def roof(name):
        t_ms, by = kernel_ms[name], alg[name]
        sec = t_ms * 1e-3
        ach = by / sec / 1e9 if sec > 0 else 0.0
        r = {"ms_per_frame": round(t_ms, 3), "bytes_per_frame": int(by), "achieved": round(ach, 1),
             "frac": round(ach / HBM_PEAK_GBS, 4)}
        if r["frac"] > 1.0:
            r["frac_note"] = "above 1: cache-served (algorithmic bytes exceed what leaves the chip's caches; see hbm_side_frac and fractions)"
        have = [pmc[k] for k in KERNEL_MEMBERS[name] if k in pmc]
        if not have or sec <= 0:
            r["traffic"] = None
            return r

        def tot(c):
            return sum(p.get(c, 0.0) for p in have)

        # HBM / fabric side: 2 x FETCH_SIZE + WRITE_SIZE.  FETCH_SIZE tallies 64 B per 128-byte request on gfx950
        # (MI355X_MICROARCH.md, measured there on coalesced streams); tools/microbench/fetch_calib.hip shows the same
        # for 16-byte gathers (profiles/r3/fetch_calib.txt), so the factor applies to the traversal kernels too.
        fetch_factor = pmc["_meta"].get("fetch_factor", {}).get(name, 2.0)
        if not tot("FETCH_SIZE_bytes") and not tot("WRITE_SIZE_bytes"):
            r["traffic"] = None
        else:
            traffic = fetch_factor * tot("FETCH_SIZE_bytes") + tot("WRITE_SIZE_bytes")
            r["traffic"] = int(traffic)
            r["traffic_over_algorithmic"] = round(traffic / by, 3)
            # counter bytes / kernel time / 8 TB/s: what of the HBM roof this kernel really uses
            r["hbm_side_frac"] = round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4)
        fr = {"hbm": r.get("hbm_side_frac")}
        if tot("TCP_TOTAL_CACHE_ACCESSES_sum"):
            if name in fetched:
                fb = fetched[name]
                r["fetched"] = {"bytes_per_frame": int(fb), "rate_gbs": round(fb / sec / 1e9, 1),
                                "note": "bytes at the record sizes this build loads (48-B nodes and triangles) / time; most are served by L1 / L2"}
            req = tot("TCP_TCC_READ_REQ_sum")
            l1 = tot("TCP_TOTAL_CACHE_ACCESSES_sum")
            miss = tot("TCC_MISS_sum")
            r["l2_side"] = {"l1_to_l2_requests_per_frame": int(req), "request_bytes": 128,
                            "rate_gbs": round(req * 128 / sec / 1e9, 1), "l2_gather_peak_gbs": L2_GATHER_GBS,
                            "l2_misses_per_frame": int(miss), "fabric_rate_gbs": round(miss * 128 / sec / 1e9, 1),
                            "infinity_cache_gather_peak_gbs": MALL_GATHER_GBS,
                            "l1_hit": round(1.0 - req / l1, 3) if l1 else None,
                            "l2_hit": round(1.0 - miss / max(1.0, tot("TCC_HIT_sum") + miss), 3)}
            gathers = name in fetched  # the gather ceilings apply to the traversal kernels; the others stream
            if gathers:
                fr["l2_gather"] = round(req * 128 / sec / 1e9 / L2_GATHER_GBS, 3)
                fr["infinity_cache_gather"] = round(miss * 128 / sec / 1e9 / MALL_GATHER_GBS, 3)
            clks = sec * SHADER_CLK_HZ
            lane_loads = l1 / (NUM_CUS * clks) if clks else 0.0
            fast = tot("SQ_INSTS_VALU_FAST")
            valu = tot("SQ_INSTS_VALU")
            slow = max(0.0, valu - fast)
            r["issue"] = {"lane_loads_per_clk_per_cu": round(lane_loads, 3), "lane_load_ceiling": LANE_LOADS_PER_CLK_PER_CU,
                          "valu_wave_instructions_per_frame": int(valu), "valu_fast_share": round(fast / valu, 3) if valu else None,
                          "valu_clk_fast_slow": [VALU_CLK_FAST, VALU_CLK_SLOW],
                          "lane_utilisation": round(tot("SQ_THREAD_CYCLES_VALU") / (64.0 * tot("SQ_ACTIVE_INST_VALU")), 3)
                          if tot("SQ_ACTIVE_INST_VALU") else None,
                          "wait_any": round(tot("SQ_WAIT_ANY") / tot("SQ_WAVE_CYCLES"), 3) if tot("SQ_WAVE_CYCLES") else None,
                          "shader_clock_hz_assumed": SHADER_CLK_HZ}
            if gathers:
                fr["vmem_address_unit"] = round(lane_loads / LANE_LOADS_PER_CLK_PER_CU, 3)
            if valu and tot("SQ_INSTS_VALU_FAST"):
                fr["valu_issue"] = round((fast * VALU_CLK_FAST + slow * VALU_CLK_SLOW) / (NUM_SIMDS * clks), 3) if clks else None
        r["fractions"] = fr
        r["bound"] = max(((v, k) for k, v in fr.items() if v is not None), default=(0, "hbm"))[1]
        top = max((v for v in fr.values() if v is not None), default=0.0)
        if top < 0.5 and "issue" in r:
            r["bound_note"] = ("no ceiling above half: short kernels of dependent loads (vertex record -> hint / material -> queue cursor), "
                               "waves waiting %.2f of their cycles — latency, not a rate, sets their time; they run beside the traversal "
                               "kernels and take little of what those are short of" % (r["issue"]["wait_any"] or 0.0))
        return r

    return {k: roof(k) for k in kernel_ms}


def fetched_bytes(stat, info):
    """bytes the build really loads per unit: 48-byte records for nodes and triangles (bvh.h), a 28-byte ray + 1 visibility
    byte per any-hit ray, origin + direction + 16-byte hit record per closest-hit ray, SURVEY's 388 B per shade"""
    closest_rays = stat["raysEyeExtend"] + stat["raysLightExtend"]
    n_shadow = stat["raysNee"] + stat["raysSplat"] + stat["raysConnect"]
    return {"walk_kernel": 44 * closest_rays + info.nodeBytes * stat["nodeVisitsClosest"] + info.triBytes * stat["triTestsClosest"] + 388 * closest_rays,
            "trace_shadow_kernel": 29 * n_shadow + info.nodeBytes * stat["nodeVisitsShadow"] + info.triBytes * stat["triTestsShadow"]}


def solo_frames(R, pkg, torch, n, flags=0):
    """n frames run alone (host-synchronised after each) with stage timing on: summed stage times, ray tallies per
    stage, and the last frame's counters."""
    stage_ms, rays, last = {}, {}, None
    for pp in R.pipes:
        pp.ctx.enable_stage_timing(True)
    for _ in range(n):
        cx = R.step(flags)
        torch.cuda.synchronize(R.dev)
        for name, ms in cx.stage_times():
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
        last = cx.counters().as_dict()
        for k in RAY_KEYS + HINT_KEYS:
            rays[k] = rays.get(k, 0) + last[k]
    for pp in R.pipes:
        pp.ctx.enable_stage_timing(False)
    return stage_ms, rays, last


def timed_frames(R, pkg, K):
    """K frames, no host synchronisation inside; returns (seconds, ray queries of this rank, of which answered by a hint)."""
    R.barrier()
    first = R.state["frame"]
    t0 = time.perf_counter()
    for k in range(K):
        R.step(pkg.abi.PARAM_KEEP_COUNTERS if k >= R.inflight else 0)  # first use of each context zeroes its tallies
    R.barrier()
    dt = time.perf_counter() - t0
    rays = hinted = 0
    for i, pp in enumerate(R.pipes):
        used = sum(1 for k in range(K) if (first + k) % R.inflight == i)  # frames this context rendered
        if used:
            c = pp.ctx.counters().as_dict()
            h = sum(c[k] for k in HINT_KEYS)
            rays += sum(c[k] for k in RAY_KEYS) + h + used * R.num_pixels  # + one primary ray per pixel
            hinted += h
    return dt, rays, hinted


_CLOSE_LAST = []  # renderers of the side measurements: nothing is freed on the device before the last shape has been set up


def other_config(pkg, torch, name, scene, W, H, D, mat, keep, frames=3, pmc_sub=None):
    """One more BASELINE shape on this GPU, one frame in flight: ms per frame, Mrays/s, visits per ray, stage times.
    The pipeline is left open (appended to `keep`): the driver wipes freed VRAM before it hands it out again, and the
    next shape's set-up would be charged the wait for this one's tens of GB (measured: 0.9-1.2 s per shape)."""
    # (unless the device is short of memory: path state grows with pixels x depth^2, the build takes scratch on top)
    need = W * H * (580 * D + 12 * D * D) + scene.desc.numTriangles * 2000 + (2 << 30)  # measured: 8.5 / 62 / 93 GiB of path state at depth 8 / 12 / 16, 18 GiB of scene + build scratch at 10 M triangles
    free_before = torch.cuda.mem_get_info()[0]
    if free_before < need:
        for q in keep + _CLOSE_LAST:
            q.close()
        del keep[:], _CLOSE_LAST[:]
        torch.cuda.synchronize()
        # ... and then the wipe of what was just freed (~30 GB/s, tools/vram_wipe_probe.py) is the bench's housekeeping, not
        # this shape's set-up: wait it out before the clock starts
        time.sleep(1.0 + max(0, torch.cuda.mem_get_info()[0] - free_before) / 25e9)
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    pipe = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat, accum_limit=1 << 30)
    keep.append(pipe)
    try:
        torch.cuda.synchronize()
        setup = time.time() - t0
        pipe.render_frame(accumulate=True)
        torch.cuda.synchronize()
        pipe.ctx.enable_stage_timing(True)
        agg = {}
        pipe.render_frame(accumulate=True)
        for k, v in pipe.ctx.stage_times():
            agg[k] = agg.get(k, 0.0) + v  # (a stage name occurs once per launch: the lazy rounds' three times)
        pipe.ctx.enable_stage_timing(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            pipe.render_frame(accumulate=True, extra_flags=pkg.abi.PARAM_KEEP_COUNTERS if k else 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        c = pipe.ctx.counters().as_dict()
        hinted = sum(c[k] for k in HINT_KEYS) / frames
        rays = (sum(c[k] for k in RAY_KEYS) + frames * W * H) / frames + hinted
        pipe.render_frame(extra_flags=pkg.abi.PARAM_COUNTERS)
        torch.cuda.synchronize()
        s = pipe.ctx.counters().as_dict()
        shadow = max(1, s["raysNee"] + s["raysSplat"] + s["raysConnect"])
        closest = max(1, s["raysEyeExtend"] + s["raysLightExtend"])
        info = pipe.ctx.bvh_info()
        free, total = torch.cuda.mem_get_info()
        out = {"workload": name, "resolution": [W, H], "max_depth": D, "mat_index": mat, "frames_timed": frames, "frames_in_flight": 1,
               "ms_per_frame": round(dt * 1e3, 3), "value": round(rays / dt / 1e6, 1), "unit": "Mrays/s", "rays_per_frame": int(rays),
               "rays_hinted_per_frame": int(hinted),
               "visits_per_ray": {"closest_nodes": round(s["nodeVisitsClosest"] / closest, 2), "closest_tris": round(s["triTestsClosest"] / closest, 2),
                                  "shadow_nodes": round(s["nodeVisitsShadow"] / shadow, 2), "shadow_tris": round(s["triTestsShadow"] / shadow, 2),
                                  # any-hit shader invocations (alpha tests) per ray: the "any-hit rate" of BASELINE.md section 3 row 5
                                  "closest_alpha_tests": round(s["alphaTestsClosest"] / closest, 3), "shadow_alpha_tests": round(s["alphaTestsShadow"] / shadow, 3)},
               "stage_ms": {k: round(v, 2) for k, v in agg.items() if v >= 0.05},
               "bvh": {"nodes": info.numNodes, "references": info.numReferences, "triangles": info.numTriangles,
                       "alpha_mode_triangles": info.numAlphaMode, "always_pass": info.numAlwaysPass, "dropped": info.numDropped},
               "setup_s": round(setup, 2),
               # scene_setup_s = context + bdpt_set_scene (the acceleration structure, built on the device, + uploads); the rest of
               # setup_s is bdpt_resize: hipMalloc of the frame's path state, which the driver clears at ~35 GB/s whenever the
               # VRAM it hands out has been used before — by this process or an earlier one (0.03 s for 94 GB on a fresh box)
               "scene_setup_s": round(pipe.setup_times["context_s"] + pipe.setup_times["set_scene_s"], 2),
               "setup_breakdown_s": {k: round(v, 2) for k, v in pipe.setup_times.items()}, "device_memory_gb": round((free0 - free) / 2 ** 30, 1)}
        # the same roofline block as the headline's: algorithmic bytes from this shape's own tallies, kernel times from its
        # stage events (one frame run alone), HBM-side bytes from the committed PMC passes of tools/big_configs.py when
        # they were measured on this build (profiles/<round>/<pmc_sub>/roofline_pmc.json)
        alg = algorithmic_bytes(s, s["pixelsValid"] * num_connect_pairs(D), W * H, SURVEY_NODE_BYTES, SURVEY_TRI_BYTES, D)
        kernel_ms, wait_ms, critical_ms = kernel_durations(agg)
        pmc = load_pmc(pkg, pmc_sub) if pmc_sub else {}
        pmc_match = None if not pmc else not pmc.get("_mismatch", False)
        if pmc_match is False:
            pmc = {}
        kernels = roofline_kernels(kernel_ms, alg, fetched_bytes(s, info), pmc)
        dominant = max(kernel_ms.items(), key=lambda kv: kv[1])[0]
        dom = kernels[dominant]
        out["roofline"] = {"kernel": dominant, "bound": dom.get("bound", "hbm"), "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": dom["frac"], "traffic": dom.get("traffic"), "hbm_side_frac": dom.get("hbm_side_frac"),
                           "kernels": kernels, "waits_ms_per_frame": round(wait_ms, 3),
                           "frame": {"ms_per_frame": round(critical_ms, 3), "bytes_per_frame": int(sum(alg.values())),
                                     "frac": round(sum(alg.values()) / (critical_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if critical_ms > 0 else 0.0,
                                     "traffic": sum(k["traffic"] for k in kernels.values() if k.get("traffic")) or None},
                           "pmc_build_match": pmc_match}
    except Exception:
        keep.remove(pipe)
        pipe.close()
        raise
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="atrium", choices=["atrium", "atrium_uneven", "cornell"],
                    help="built-in scene: the Sponza stand-in (default), the same hall with heavy-tailed triangle areas, the Cornell box")
    ap.add_argument("--scene-file", default=None, help="a supplied asset instead: .fscene or .obj, read through bdpt_scene_load "
                    "(the reference's import rules); config.workload names the file and data says \"file\"")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--mat", type=int, default=None, help="0 GGX (default for atrium), 1 Lambertian (default for cornell)")
    ap.add_argument("--triangles", type=int, default=262144)
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight (contexts / streams per rank)")
    ap.add_argument("--no-single-pass", dest="single_pass", action="store_false",
                    help="skip the informational one-frame-in-flight pass at N=1")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip the other BASELINE shapes (configs 2, 4 and 5 on this one GPU) at N=1")
    ap.add_argument("--dump-frames", type=int, default=0, help="render this many frames from a fresh state, write the "
                    "accumulated image of all ranks to --dump-path (rank 0) and exit: lets runs with different N be compared")
    ap.add_argument("--dump-path", default="bench_image.npy")
    ap.add_argument("--plain-loop", action="store_true", help="with --dump-frames at N=1: the reference's plain sequence instead "
                    "(one whole-frame context, G-buffer pass -> BDPT pass -> accumulation pass, one frame at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if "RANK" in os.environ:  # said before torch is imported: a rank that is ended because a peer failed has still shown it ran
        print("bench.py: rank %s of %s started" % (os.environ["RANK"], os.environ.get("WORLD_SIZE", "?")), file=sys.stderr, flush=True)

    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launched under torch.distributed.run with another rank count?)"
                         % (args.gpus, world))
    dist = None
    # BDPT_BENCH_TILED_AT_1=1 under torch.distributed.run makes a single rank go through the RCCL exchange too, so the
    # N>1 code can be exercised on a one-GPU box; otherwise N=1 runs without any collective.
    use_dist = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and os.environ.get("BDPT_BENCH_TILED_AT_1", "0") == "1")
    backend = None
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # BDPT_BENCH_BACKEND=gloo + BDPT_BENCH_DEVICE=0 rehearse several ranks on a one-GPU box (the exchange then goes
        # through gloo instead of RCCL; everything else is the N>1 path)
        backend = os.environ.get("BDPT_BENCH_BACKEND", "nccl")
        shared_device = "BDPT_BENCH_DEVICE" in os.environ
        if shared_device:
            local_rank = int(os.environ["BDPT_BENCH_DEVICE"])
        elif world > 1 and torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks but %d visible GPUs: every rank needs a GPU of its own "
                             "(BDPT_BENCH_DEVICE=<ordinal> with BDPT_BENCH_BACKEND=gloo rehearses on a shared one)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render pass has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if use_dist and world > 1 and "BDPT_BENCH_DEVICE" not in os.environ:
        # no two ranks on one device: gather (hostname-free) PCI bus ids
        mine = torch.cuda.get_device_properties(dev)
        ident = "%s/%s" % (getattr(mine, "pci_bus_id", local_rank), getattr(mine, "uuid", local_rank))
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if len(set(idents)) != world:
            raise SystemExit("bench.py: two ranks share a device (%s)" % idents)

    W, H, D = args.width, args.height, args.depth
    scene_load_s = None
    if args.scene_file:  # a supplied asset (e.g. the real Sponza OBJ, were it dropped on the box)
        t0 = time.time()
        scene = pkg.Scene.load(args.scene_file)
        scene_load_s = time.time() - t0
        scene_name = "file"
        workload = "%s (%d triangles, %d materials, %d textures, %d lights; read by bdpt_scene_load in %.2f s)" % (
            os.path.basename(args.scene_file), scene.desc.numTriangles, scene.desc.numMaterials, scene.desc.numTextures, scene.desc.numLights, scene_load_s)
        metric_scene = "scene file %s" % os.path.basename(args.scene_file)
    elif args.scene == "cornell":
        scene, scene_name = pkg.Scene.cornell(), "cornell"
        workload, metric_scene = "Cornell box, 32 triangles, Lambertian", "Cornell box"
    elif args.scene == "atrium_uneven":
        scene, scene_name = pkg.Scene.atrium_uneven(1, args.triangles), "atrium_uneven"
        workload = "procedural atrium with heavy-tailed triangle areas (two-triangle walls and floors), %d triangles, textured GGX" % scene.desc.numTriangles
        metric_scene = "Sponza-class scene"
    else:
        scene, scene_name = pkg.Scene.atrium(1, args.triangles), "atrium"
        workload = "procedural atrium (Sponza stand-in), %d triangles, textured GGX" % scene.desc.numTriangles
        metric_scene = "Sponza-class scene"
    mat = args.mat if args.mat is not None else (1 if scene_name == "cornell" else 0)

    if args.dump_frames > 0 and args.plain_loop:
        if world != 1 or dist is not None:
            raise SystemExit("--plain-loop is the single-GPU reference sequence")
        pp = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat, device=local_rank, accum_limit=1 << 30)
        for _ in range(args.dump_frames):
            pp.render_frame(accumulate=True)
        torch.cuda.synchronize(dev)
        import numpy as np
        np.save(args.dump_path, pp.last_frame.cpu().numpy())
        print(json.dumps({"dumped": args.dump_path, "frames": args.dump_frames, "n_gpus": 1, "frames_in_flight": 1, "tiled": False,
                          "backend": None}), flush=True)
        pp.close()
        return

    R = pkg.tiling.TileRenderer(scene, W, H, D, mat, local_rank, world, rank, dist, args.inflight)
    pipe, ctx, inflight = R.pipe, R.ctx, R.inflight
    n_pix_tile = R.num_pixels
    info = ctx.bvh_info()

    if args.dump_frames > 0:
        for _ in range(args.dump_frames):
            R.step()
        full = R.gather()
        if rank == 0:
            import numpy as np
            np.save(args.dump_path, full.cpu().numpy())
            print(json.dumps({"dumped": args.dump_path, "frames": args.dump_frames, "n_gpus": world, "frames_in_flight": inflight,
                              "tiled": True, "backend": dist.get_backend() if dist is not None else None}), flush=True)
        R.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- untimed: warm-up, then frames run alone: node / triangle visit statistics (deterministic per frame), stage
    # times from HIP events on the launch stream, exchange timing; the sequence is rewound afterwards
    for _ in range(args.warmup):
        R.step()
    R.barrier()
    mark = (R.state["frame"], R.state["accum"])
    _, _, stat = solo_frames(R, pkg, torch, 1, pkg.abi.PARAM_COUNTERS)
    R.rewind(*mark)
    stage_frames = 3
    R.exchange_events = []
    stage_ms, per_stage_rays, _ = solo_frames(R, pkg, torch, stage_frames)
    ex_ev, R.exchange_events = R.exchange_events, None
    tail_ms = sum(e[0].elapsed_time(e[1]) for e in ex_ev) / max(1, len(ex_ev))
    ex_plus_tail_ms = sum(e[2].elapsed_time(e[3]) for e in ex_ev) / max(1, len(ex_ev))
    R.rewind(*mark)

    # ---- timed region: exactly K steps, no host synchronisation inside
    elapsed, rays_total, hinted_total = timed_frames(R, pkg, args.steps)

    t = torch.tensor([elapsed, float(rays_total), float(hinted_total)], dtype=torch.float64, device=dev)
    per_rank = None
    if dist is not None:
        gdev = dev if backend == "nccl" else torch.device("cpu")
        tg = t.to(gdev)
        allv = [torch.zeros_like(tg) for _ in range(world)]
        dist.all_gather(allv, tg)
        elapsed_max = max(float(v[0]) for v in allv)
        rays_all = sum(float(v[1]) for v in allv)
        hinted_all = sum(float(v[2]) for v in allv)
        rows_all = [None] * world
        dist.all_gather_object(rows_all, n_pix_tile // W)
        per_rank = {"ms_per_step": [round(float(v[0]) / args.steps * 1e3, 3) for v in allv],
                    "ms_per_step_min": round(min(float(v[0]) for v in allv) / args.steps * 1e3, 3),
                    "ms_per_step_max": round(elapsed_max / args.steps * 1e3, 3),
                    "mrays": [round(float(v[1]) / float(v[0]) / 1e6, 1) for v in allv], "rows": rows_all}
    else:
        elapsed_max, rays_all, hinted_all = elapsed, float(rays_total), float(hinted_total)

    out = None
    if rank == 0:
        K = args.steps
        mrays = rays_all / elapsed_max / 1e6
        n_shadow = max(1, stat["raysNee"] + stat["raysSplat"] + stat["raysConnect"])
        n_int_s = stat["nodeVisitsShadow"] / n_shadow
        n_tri_s = stat["triTestsShadow"] / n_shadow
        n_closest = max(1, stat["raysEyeExtend"] + stat["raysLightExtend"])
        n_int_c = stat["nodeVisitsClosest"] / n_closest
        n_tri_c = stat["triTestsClosest"] / n_closest
        # ---- roofline per kernel and for the whole frame.  Algorithmic bytes from the device tallies
        # of one frame of the same sequence; durations = HIP events on the launch stream around each kernel's launches
        # (bdpt_get_stage_times), summed over the frame, averaged over the frames named in stage_timing.
        alg = algorithmic_bytes(stat, stat["pixelsValid"] * num_connect_pairs(D), n_pix_tile, SURVEY_NODE_BYTES, SURVEY_TRI_BYTES, D)
        ms = {k: v / stage_frames for k, v in stage_ms.items()}
        # one stage = one kernel (or a wait for the second stream); gen_splat / gen_connect run on the second stream and
        # carry event pairs of their own ("side:" stages): a block's time is the sum of its KERNELS' times
        kernel_ms, wait_ms, critical_ms = kernel_durations(ms)
        pmc = load_pmc(pkg) if (scene_name == "atrium" and (W, H, D, world) == (1920, 1080, 8, 1)) else {}
        pmc_match = None if not pmc else not pmc.get("_mismatch", False)
        if pmc_match is False:
            pmc = {}
        kernels = roofline_kernels(kernel_ms, alg, fetched_bytes(stat, info), pmc)
        dominant = max(kernel_ms.items(), key=lambda kv: kv[1])[0]
        frame_ms = critical_ms  # the launch stream's stages, waits included: what one frame run alone takes
        frame_bytes = sum(alg.values())
        dom = kernels[dominant]
        closest_per_frame = (per_stage_rays.get("raysEyeExtend", 0) + per_stage_rays.get("raysLightExtend", 0)) / stage_frames
        shadow_per_frame = (per_stage_rays.get("raysNee", 0) + per_stage_rays.get("raysSplat", 0) +
                            per_stage_rays.get("raysConnect", 0)) / stage_frames
        out = {
            "metric": "Mrays/s + RMSE vs the scalar oracle, BDPT pass, %s %dx%d depth %d" % (metric_scene, W, H, D),
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / K * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # value counts every ray query of the integrator the pass answered; the ones an occluder hint answered by one
            # triangle test (no traversal) are shown apart, and so is the rate without them
            "value_traversals_only": round((rays_all - hinted_all) / elapsed_max / 1e6, 2),
            "rays_hinted_per_frame": int(hinted_all / K),
            "dtype": "f32",
            "data": "file" if args.scene_file else "synthetic",
            "config": {
                "workload": workload,
                "resolution": [W, H], "max_depth": D, "spp_per_step": 1, "mat_index": mat,
                "parallelism": "tile%d" % world,
                "tiling": {"stripe_rows": pkg.tiling.stripe_rows(H, world), "rows_this_rank": n_pix_tile // W,
                           "exchange": "reduce_scatter_tensor(int64 SUM) of the owner-major splat buffer" if dist is not None else None,
                           "exchange_bytes_per_rank_per_frame": R.exchange_bytes if dist is not None else 0,
                           "backend": backend,
                           # one frame run alone on rank 0: the zero-valued connection rounds that run beside the exchange,
                           # exchange + those rounds (enqueue of the collective to its completion on the render stream),
                           # and what of the exchange the rounds did not cover
                           "tail_ms": round(tail_ms, 3), "exchange_plus_tail_ms": round(ex_plus_tail_ms, 3),
                           "exchange_exposed_ms": round(max(0.0, ex_plus_tail_ms - tail_ms), 3),
                           "per_rank": per_rank},
                "rays_per_frame": int(rays_all / K),
                # of which answered by an occluder hint (a triangle test instead of a traversal), and the rate without them
                "rays_hinted_per_frame": int(hinted_all / K),
                "value_traversals_only": round((rays_all - hinted_all) / elapsed_max / 1e6, 2),
                "rays_reference_equivalent_per_frame": int(stat["pixelsValid"] * ((D + 1) ** 2 - 1) + n_pix_tile)
                if world == 1 else None,
                "bvh": {"nodes": info.numNodes, "node_bytes": info.nodeBytes, "tri_bytes": info.triBytes,
                        "references": info.numReferences, "max_depth": info.maxDepth, "sah_cost": round(info.sahCost, 2)},
                "visits_per_ray": {"shadow_nodes": round(n_int_s, 2), "shadow_tris": round(n_tri_s, 2),
                                   "closest_nodes": round(n_int_c, 2), "closest_tris": round(n_tri_c, 2),
                                   "closest_alpha_tests": round(stat["alphaTestsClosest"] / n_closest, 3),
                                   "shadow_alpha_tests": round(stat["alphaTestsShadow"] / n_shadow, 3)},
                "frames_in_flight": inflight,
                "stage_timing": "HIP events over %d untimed frames run alone on rank 0's tile before the timed region "
                                "(timed frames overlap each other); one stage = one kernel of the launch stream, a wait for the "
                                "second stream (*_wait) or a kernel of the second stream (side:*, beside the stages before it)" % stage_frames,
                "stage_ms_per_step": {k: round(v, 3) for k, v in ms.items()},
                "stage_mrays": {"walk_kernel": round(closest_per_frame / (kernel_ms["walk_kernel"] * 1e-3) / 1e6, 1)
                                if kernel_ms["walk_kernel"] > 0 else None,
                                "trace_shadow_kernel": round(shadow_per_frame / (kernel_ms["trace_shadow_kernel"] * 1e-3) / 1e6, 1)
                                if kernel_ms["trace_shadow_kernel"] > 0 else None},
                "dominant_kernel": dominant,
            },
            # the kernel with the largest summed duration in the frame; every figure below can be recomputed from
            # profiles/<round>/ (kernel_stats.csv for the durations, roofline_pmc.json <- the pmc_*.csv passes for the counts)
            "roofline": {
                "kernel": dominant, "bound": dom.get("bound", "hbm"), "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["frac"], "traffic": dom.get("traffic"),
                # HBM-side counter bytes (2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s: beside the contract's `frac`,
                # which prices cache-served bytes against HBM
                "hbm_side_frac": dom.get("hbm_side_frac"),
                "frac_note": dom.get("frac_note", "algorithmic bytes over HBM peak; see `fractions` for the ceilings that can bind"),
                "fractions": dom.get("fractions"),
                "peak_achievable_copy": 6290.0,  # float4 copy rate MI355X_MICROARCH.md reports (79 % of spec)
                "bytes_per_frame": dom["bytes_per_frame"], "ms_per_frame": dom["ms_per_frame"],
                "kernels": kernels,
                "waits_ms_per_frame": round(wait_ms, 3),
                "frame": {"ms_per_frame": round(frame_ms, 3), "bytes_per_frame": int(frame_bytes),
                          "achieved": round(frame_bytes / (frame_ms * 1e-3) / 1e9, 1) if frame_ms > 0 else 0.0,
                          "frac": round(frame_bytes / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if frame_ms > 0 else 0.0,
                          "traffic": sum(k["traffic"] for k in kernels.values() if k.get("traffic")) or None},
                # True: the PMC counts behind traffic / fractions were collected on exactly these sources; False: the
                # committed counts belong to another build and are not quoted; None: no counts for this workload
                "pmc_build_match": pmc_match,
                "node_bytes_costed": SURVEY_NODE_BYTES, "node_bytes_fetched": info.nodeBytes,
                "note": "achieved / frac = ALGORITHMIC bytes (SURVEY.md §8d: no cache credit, an interior-node visit costed at 64 B) / "
                        "measured time against HBM peak: the contract's figure; it exceeds the HBM-side traffic (PMC) because the BVH "
                        "is served from L1 / L2 / Infinity Cache, so it can pass 1 and does not say what binds.  `fractions` does: each "
                        "is a measured rate over the ceiling that applies to it — hbm: 2 x FETCH_SIZE + WRITE_SIZE over 8 TB/s; l2_gather: "
                        "L1-to-L2 read requests x 128 B (the request size, profiles/r3/fetch_calib.txt) over the guide's L2 gather rate; "
                        "infinity_cache_gather: L2 misses x 128 B over its random-row Infinity-Cache rate; vmem_address_unit: L1 accesses (lane-loads) per clock per CU over this repo's "
                        "microbenchmarked ceiling; valu_issue: VALU wave-instructions x measured issue clocks (2.4 for the fma/mul/add "
                        "class, 4.3 for the rest) over SIMD clocks.  `bound` names the largest; none of them can exceed 1 by "
                        "construction except through the assumed 2.4 GHz clock.  Counts come from the committed PMC passes of this "
                        "command (profiles/%s/), durations from this run." % PROFILE_ROUND,
            },
        }

    # ---- informational passes at N = 1 (rank 0 only; the headline above is already final: whatever goes wrong below is
    # recorded in the line instead of losing it)
    def guarded(key, fn):
        try:
            fn()
        except Exception as e:  # noqa: BLE001 — the driver needs the JSON line whatever an informational leg does
            out["config"].setdefault("informational_errors", {})[key] = "%s: %s" % (type(e).__name__, str(e)[:300])

    if rank == 0 and world == 1 and dist is None:
        if args.single_pass:
            def single():
                R1 = pkg.tiling.TileRenderer(scene, W, H, D, mat, local_rank, 1, 0, None, 1)
                try:
                    for _ in range(2):
                        R1.step()
                    R1.barrier()
                    R1.rewind(*mark)
                    dt1, rays1, _ = timed_frames(R1, pkg, args.steps)
                    out["config"]["single_frame_in_flight"] = {
                        "frames_in_flight": 1, "value": round(rays1 / dt1 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt1 / args.steps * 1e3, 3),
                        "note": "same frames, one context, no host synchronisation either; informational (the kernels' solo durations of the roofline block add up to this loop)"}
                finally:
                    _CLOSE_LAST.append(R1)  # (closed at the end: see other_config)
            guarded("single_frame_in_flight", single)
        if not args.no_cpu_baseline:  # rank 0 at N=1 only
            def cpu():
                base, parity = cpu_baseline(pkg, scene, pipe, W, H, D, mat, args.cpu_seconds, mark[0])
                out["cpu_baseline"] = base
                out.update(parity)
            guarded("cpu_baseline", cpu)
    keep = []  # the other shapes' pipelines, closed together with the bench's own at the end (see other_config)
    if rank == 0 and world == 1 and dist is None and args.other_configs and scene_name == "atrium" and (W, H, D) == (1920, 1080, 8):
        others = []

        def other(name, make, w, h, d, m, pmc_sub=None):
            def run():
                sc = make()
                try:
                    others.append(other_config(pkg, torch, name, sc, w, h, d, m, keep, pmc_sub=pmc_sub))
                finally:
                    sc.close()
            guarded(name.split(":")[0], run)

        other("BASELINE configs[1]: Cornell box 1920x1080 depth 8, Lambertian", pkg.Scene.cornell, 1920, 1080, 8, 1)
        # what even tessellation hides: the same hall and triangle count with areas spread over six decades,
        # as built by default and with opaque pre-splitting switched on (the builder's BDPT_SPLIT_BUDGET environment knob)
        other("stress: atrium with heavy-tailed triangle areas (two-triangle walls and floors), 262144 triangles, 1920x1080 depth 8",
              lambda: pkg.Scene.atrium_uneven(1, 262144), 1920, 1080, 8, 0)

        def split_on():
            os.environ["BDPT_SPLIT_BUDGET"] = "1"
            return pkg.Scene.atrium_uneven(1, 262144)
        try:
            other("stress+split: the same scene built with BDPT_SPLIT_BUDGET=1 (one extra reference per opaque triangle on average)",
                  split_on, 1920, 1080, 8, 0)
        finally:
            os.environ.pop("BDPT_SPLIT_BUDGET", None)
        other("BASELINE configs[3] shape on ONE GPU: 2.8 M triangles (atrium generator, Bistro stand-in) 3840x2160 depth 12",
              lambda: pkg.Scene.atrium(1, 2800000), 3840, 2160, 12, 0, pmc_sub="config4")
        other("BASELINE configs[4] shape on ONE GPU: 10 M triangles, half of them alpha-masked leaf cards "
              "(courtyard generator, San Miguel stand-in) 3840x2160 depth 16", lambda: pkg.Scene.courtyard(2, 10000000, 0.5), 3840, 2160, 16, 0,
              pmc_sub="config5")
        out["config"]["other_configs"] = others
    for p in keep + _CLOSE_LAST:
        p.close()
    R.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (never an exec: a
    process that has initialised the GPU must not be replaced, and nothing here has touched one yet — torch is not even
    imported), stream its output through and return its exit code.  Rank 0 prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port on the loop-back interface (the hostname may not resolve)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def effective_cpus():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, scene, pipe, W, H, D, mat, budget_s, frame):
    """The oracle (scalar C++ restatement, kind "port") on this box's host cores, on a band of rows of the same frame
    in the middle of the image, sized from a short probe to take about budget_s.  The rows it rendered are then
    rendered by the HIP path as a tile with the same frame counters and jitter, and compared: `rmse` (linear fp32
    RGB, the metric's second half) and the fraction of bit-identical pixels."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = effective_cpus()
    pipe.gbuffer_frame, pipe.bdpt_frame = 0xdeadbeef + frame, 0x1337 + frame
    gp = pipe.gbuffer_params()
    p = pipe.bdpt_params()

    def run(rows, keep=False):
        ya = max(0, H // 2 - rows // 2)
        yb = min(H, ya + rows)
        orc = ob.OracleRender(pkg.abi, scene.desc, W, H, ya, yb)
        t0 = time.perf_counter()
        orc.gbuffer(pipe.cam, gp, threads=cores)
        cnt = orc.bdpt(pipe.cam, p, threads=cores)
        dt = time.perf_counter() - t0
        rays = cnt.total_rays() + (yb - ya) * W
        img = None
        if keep:
            orc.resolve()
            img = orc.image()[ya:yb].copy()
        orc.close()
        return rays, dt, yb - ya, (ya, yb), img

    rays, dt, r, band, img = run(max(2, min(H, cores // 4 + 2)), keep=True)
    for _ in range(2):  # the short probe overestimates the rate (thread start-up, sky rows): size twice
        rate = rays / max(dt, 1e-6)
        rows = int(max(r, min(H, budget_s * rate / max(1.0, rays / r))))
        if rows <= r or dt >= 0.6 * budget_s:
            break
        rays, dt, r, band, img = run(rows, keep=True)
    base = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d middle rows of the same %dx%d depth-%d frame (%d rays, %.1f s); the oracle traces every ray the "
                      "reference issues, incl. zero-contribution ones the GPU path skips" % (r, W, H, D, rays, dt)}
    # ---- parity leg: the same rows, same counters, through the HIP path
    ya, yb = band
    tp = pkg.FramePipeline(scene, W, H, max_depth=D, mat_index=mat, device=pipe.dev.index or 0, tile=(ya, yb))
    tp.gbuffer_frame, tp.bdpt_frame = pipe.gbuffer_frame, pipe.bdpt_frame
    tp.render_frame()
    torch.cuda.synchronize(tp.dev)
    gpu = tp.output[ya:yb].cpu().numpy()
    _CLOSE_LAST.append(tp)  # (closed at the end: see other_config)
    d = gpu[..., :3].astype(np.float64) - img[..., :3].astype(np.float64)
    parity = {"rmse": float(np.sqrt(np.mean(d * d))),
              "bit_exact_frac": float((gpu.view(np.uint32) == img.view(np.uint32)).all(axis=-1).mean()),
              "rmse_sample": "rows [%d, %d) of the benchmark frame (1 spp, frame counter 0x%x), HIP tile vs the oracle over the same "
                             "rows; the oracle is parity-unpinned against the DXR reference (DESIGN.md section 2)" % (ya, yb, p.frameCount)}
    return base, parity


if __name__ == "__main__":
    main()
