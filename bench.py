#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + secondary) and frame ms on scene/bunny.json @1920x1080.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run, one rank per GPU (RCCL).  A *step* is one pass of
  the hot path over one batch: N frames of the workload, every frame row-tiled over the N ranks
  (interleaved 8-row blocks), finished by ONE gather of the packed buffers to rank 0, which
  re-interleaves them into the final row-major frames.  Per-GPU work is one frame-equivalent at
  every N  ->  "scaling": "weak".  (--scaling strong keeps one frame per step.)

  value   = rays of the reference algorithm for all frames of the K timed steps / wall seconds
            (max over ranks), in Mrays/s; inputs (the uploaded scene) are resident in HBM.
  rays    = ray_cast invocations the REFERENCE algorithm performs (incl. its duplicated primary
            cast), the fixed numerator SURVEY.md §8(d) defines; counted by the kernel itself and
            equal to the oracle's count (64 278 888 for the default workload; tested).
  roofline= the bound that binds this kernel is VALU ISSUE (the scene, <1 MB, lives in scalar cache / L2;
            HBM carries 0.007 of its peak):  achieved = wave-level VALU instructions per launch (PMC
            SQ_INSTS_VALU of the same command, profiles/r04/counters_<workload>.json) x the mean issue cost of
            the instructions the kernel EXECUTES (profiles/r04/valu_mix_dynamic_<workload>.json: every
            straight-line segment of the source counted at run time by a -DCTR_PROFILE build, its instructions
            classified in the ISA of a -DCTR_MARKS build, scripts/dynamic_mix.py; cross-checked against
            SQ_INSTS_VALU / SALU / SMEM of the shipped build) priced with the per-kind costs MEASURED on the box
            by scripts/valu_issue.hip — 2.2 cycles VGPR-only, 4.1 with an SGPR operand / compare / packed /
            min3, 8.1 transcendental — / the kernel's mean duration from HIP events in this run;  peak = 1024
            SIMDs x the effective shader clock (cycles per launch of the PMC pass / this run's kernel time).
            frac = achieved / peak <= 1; it is withheld (stale) when the kernel sources are not the ones the
            profiles were taken on.  The same fraction for the 64 000-triangle mesh and the C4 grid:
            config.dense_64k_roofline / config.c4_roofline.  The HBM view (counter traffic / time / 8 TB/s) is
            kept as roofline.hbm_frac; SURVEY §8(d)'s "bytes the reference's flat traversal streams" is a
            property of the workload, reported in config, not a rate.
  config.c4_strong = BASELINE config 5 measured in the same run at whatever N the driver chose: the 4x4 bunny grid
            @4096x4096, one frame per step row-tiled over the N ranks, gathered to rank 0 (frame ms, Mrays/s).
  cpu_baseline = the CPU checker (oracle/_ref = the reference's own headers built for the host
            when present, else the plain-C port) on a bounded row sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scene", default="scene/bunny.json")
    p.add_argument("--workload", choices=["bunny", "c4"], default="bunny",
                   help="bunny: --scene (default scene/bunny.json@1920x1080, BASELINE config C2, the metric's workload); "
                        "c4: BASELINE config 5, the 4x4 bunny grid @4096x4096 (16 meshes), generated on the fly — meant for "
                        "--scaling strong --roots rank0 (one frame per step, row-tiled over the ranks, gathered to rank 0)")
    p.add_argument("--width", type=int, default=0)
    p.add_argument("--height", type=int, default=0)
    p.add_argument("--bounces", type=int, default=5)
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    p.add_argument("--roots", choices=["rotate", "rank0"], default="rotate",
                   help="which rank a frame is gathered to when a step has several frames: frame f -> rank f mod N "
                        "(balanced xGMI links and re-interleave work), or always rank 0")
    p.add_argument("--variant", type=int, default=0)
    p.add_argument("--in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                   help="2: consecutive steps alternate between two scene handles on two streams, so the first waves of step k+1 "
                        "fill the slots the tail of step k leaves empty (throughput of a frame sequence; every step still "
                        "completes inside the timed region).  1 (default, the headline): one launch after the other")
    p.add_argument("--skip-probe", action="store_true",
                   help="skip the untimed image-order launches after the timed region (profiling runs)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--check", action="store_true",
                   help="rank 0 compares every gathered frame bitwise with a single-process render")
    p.add_argument("--cpu-sample-div", type=int, default=0, help="CPU baseline renders 1/div of the row blocks (0=auto)")
    p.add_argument("--of", type=int, default=0, help="diagnostic: time the tiled batch render of one rank of N (no gather)")
    p.add_argument("--as-rank", type=int, default=0)
    p.add_argument("--counters-json", default="", help="PMC-derived per-launch counters (default profiles/r02/counters.json)")
    p.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (first launch, host-buffer "
                   "call, lane statistics, dense mesh): profiling runs")
    return p.parse_args()


def cpu_baseline(ca, host_scene, bounces, div):
    """Bounded CPU sample: every `div`-th 8-row block of the same frame, all host threads."""
    import oracle  # the checker libraries: this leg is the only place bench.py touches them
    w, h = host_scene.size
    # all hardware threads the process may use ("1 thread and all hardware threads"; the box's 64-core EPYC shows
    # 128 with SMT — round 3 capped this at 64), but no more than the checker's own limit of 256 workers
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 256))
    use_ref = oracle.ref_lib() is not None
    fn = oracle.ref_render if use_ref else oracle.oracle_render
    if div <= 0:
        # aim at ~10-30 s: ~0.14 Mrays/s per thread measured for this workload
        est_full = 64.3e6 * (w * h / 2073600.0) / (0.14e6 * threads)
        div = max(1, int(round(est_full / 15.0)))
    rows = (0, h, 8, 0, div)
    t0 = time.perf_counter()
    r = fn(host_scene, bounces=bounces, rows=rows, threads=threads, hit_ids=False)
    dt = time.perf_counter() - t0
    # one thread on a 1/threads-th of that sample (1 thread and all hardware threads)
    rows1 = (0, h, 8, 0, div * threads)
    t1 = time.perf_counter()
    r1 = fn(host_scene, bounces=bounces, rows=rows1, threads=1, hit_ids=False)
    dt1 = time.perf_counter() - t1
    # ... and, where the host shows more than 64 hardware threads (2 x 64 cores x SMT on the box: 256), the figure at 64 threads as
    # well, on a quarter of the sample: on a host shared with seven other jobs more threads than cores did not mean more rays
    at64 = None
    if threads > 64:
        t2 = time.perf_counter()
        r2 = fn(host_scene, bounces=bounces, rows=(0, h, 8, 0, 4 * div), threads=64, hit_ids=False)
        dt2 = time.perf_counter() - t2
        at64 = {"value": r2["ray_count"] / dt2 / 1e6, "unit": "Mrays/s", "cores": 64,
                "sample": f"every {4 * div}th 8-row block ({r2['depth'].shape[0]} rows, {r2['ray_count']} rays) in {dt2:.2f} s"}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": r["ray_count"] / dt / 1e6, "unit": "Mrays/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "sample": f"every {div}th 8-row block of the {w}x{h} frame ({r['depth'].shape[0]} rows, "
                  f"{r['ray_count']} rays) in {dt:.2f} s",
        "seconds": dt,
        "cpu_model": model, "hardware_threads_visible": os.cpu_count(), "at_64_threads": at64,
        "one_thread": {"value": r1["ray_count"] / dt1 / 1e6 if dt1 > 0 else 0.0, "unit": "Mrays/s",
                       "sample": f"every {div * threads}th 8-row block ({r1['depth'].shape[0]} rows, "
                                 f"{r1['ray_count']} rays) in {dt1:.2f} s"},
    }


def campath(ca, hs, device):
    import ctypes as C
    import math
    import torch
    from cutrace_amd import _lib
    w, h = hs.size
    cam0 = hs.desc.contents.cam
    N = 90
    cams = []
    for k in range(N):
        c = ca.Camera()
        C.memmove(C.byref(c), C.byref(cam0), C.sizeof(ca.Camera))
        a = math.radians(-22.5 + 0.35 * k)
        eye = _lib.Vec3(1.0 - 0.015 * k * 0.5, 0.1 * math.sin(k / 15.0), 2.0)
        ang = math.radians(67.5) - a + math.radians(-22.5)
        look = _lib.Vec3(-math.sin(ang), 0.0, -math.cos(ang))
        _lib.host_lib().ctr_camera_look_at(C.byref(c), eye, _lib.Vec3(0, 1, 0), look)
        cams.append(c)
    dev = torch.device("cuda", device)
    depth = torch.zeros(h * w, dtype=torch.float32, device=dev)
    color = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    normal = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    counters = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    res = {}
    for var, key in ((ca.VAR_NO_REORDER, "campath_ms_image_order"), (0, "campath_ms")):
        x = ca.DeviceScene(hs, device=device)
        x.set_cameras(cams)
        x.set_variant(var)
        ms = []
        for rep_ in range(2):
            evs = []
            for k in range(N):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                x.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=1, frame_stride_px=h * w,
                                      first_frame=k, d_counters=counters.data_ptr(), stream=stream.cuda_stream)
                e1.record(stream)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in evs]
        res[key] = sum(ms) / N
        x.close()
    res["campath"] = "90-frame camera path through the bunny room, one launch per frame, kernel ms per frame (mean)"
    return res


def _live_summary(st):
    """ctr_debug_lane_stats, condensed for the bench line: the live fraction over all trips (= lane_usefulness.cast), the
    trips by number of live lanes, and per kind of cast the mean number of live lanes in the trips that have any."""
    return {"cast": st["live_fraction"], "wave_trips": st["wave_trips"],
            "trips_by_live_lanes_1_8_to_57_64": st["trips_by_live_lanes_1_8_to_57_64"],
            "mean_live_lanes_by_kind": {k: round(v["lanes_per_trip"], 1) for k, v in st["by_kind"].items()},
            "trips_a_perfect_repacking_would_save": 1.0 - (st["live_fraction"] or 0.0)}


def extras(ca, hs, args, ds):
    """Untimed side measurements reported in `config` (never part of `value`)."""
    import statistics
    import tempfile
    from cutrace_amd import scenes
    w, h = hs.size
    out = {}
    # (1) what the drop-in `cutrace <scene.json>` gets (main.cu:30: one frame per process): the FIRST launch of the
    #     shape on a fresh scene handle, before any costs are known — the kernel alone (device buffers), and the whole host-buffer
    #     call into page-locked memory, which the kernel delivers itself while it renders (= the reference's
    #     total_ms, kernel.hpp:88,126); then the same call in steady state, with the frame leaving by one DMA after
    #     the kernel instead (CTR_VAR_NO_DIRECT), and through pageable memory
    fresh = ca.DeviceScene(hs, device=ds.device)
    fresh.render(bounces=args.bounces, rows=(0, 8), pinned=True)   # another shape: code object + clocks warm
    fresh.set_variant(ca.VAR_NO_DIRECT)
    out["first_launch_kernel_ms"] = fresh.render(bounces=args.bounces, pinned=True)["kernel_ms"]
    fresh.close()
    fresh = ca.DeviceScene(hs, device=ds.device)
    fresh.render(bounces=args.bounces, rows=(0, 8), pinned=True)
    out["first_launch_total_ms"] = fresh.render(bounces=args.bounces, pinned=True)["total_ms"]
    for _ in range(3):
        fresh.render(bounces=args.bounces, pinned=True)
    rr = [fresh.render(bounces=args.bounces, pinned=True) for _ in range(5)]
    out["host_call_total_ms_pinned"] = statistics.median(x["total_ms"] for x in rr)
    fresh.set_variant(ca.VAR_NO_DIRECT)
    rr = [fresh.render(bounces=args.bounces, pinned=True) for _ in range(7)][2:]
    out["host_call_total_ms_pinned_dma"] = statistics.median(x["total_ms"] for x in rr)
    # ordinary (pageable) buffers: three hipMemcpy after the kernel — into buffers reused from call to call, and into
    # freshly allocated ones (numpy.empty per call: every destination page is touched for the first time by the copy)
    fresh.set_variant(0)
    bufs = fresh.render(bounces=args.bounces)
    for _ in range(3):
        fresh.render(bounces=args.bounces, into=bufs)
    out["host_call_total_ms_pageable"] = statistics.median(fresh.render(bounces=args.bounces, into=bufs)["total_ms"] for _ in range(5))
    out["host_call_total_ms_pageable_fresh_buffers"] = statistics.median(fresh.render(bounces=args.bounces)["total_ms"] for _ in range(3))
    # (2) how many of a wave's 64 lanes the wave-level work serves (CTR_VAR_STATS build of the same kernel)
    fresh.set_variant(ca.VAR_STATS)
    fresh.render(bounces=args.bounces)
    c = [int(x) for x in fresh.last_counters()]
    if c[5] and c[6] and c[7]:
        out["work_per_launch"] = {"wave_casts": c[4], "bvh_nodes": c[5], "tri_prefilters": c[6], "tri_exact": c[7],
                                  "mesh_entries": c[8]}
        out["lane_usefulness"] = {"cast": c[9] / (64.0 * c[4]), "bvh_node": c[10] / (64.0 * c[5]),
                                  "tri_prefilter": c[11] / (64.0 * c[6]), "tri_exact": c[12] / (64.0 * c[7])}
    fresh.close()
    # (3) BASELINE.json words C2 as "~70k tris"; scene/bunny.stl holds 1000.  The same mesh subdivided to 64 000
    #     triangles (same surface, same 64 278 888 rays): steady-state and first-launch kernel ms
    if os.path.basename(args.scene) == "bunny.json":
        d = tempfile.mkdtemp()
        dense = ca.HostScene.load(scenes.make_dense_bunny(d, 3, width=w, height=h))
        dd = ca.DeviceScene(dense, device=ds.device)
        dd.render(bounces=args.bounces, rows=(0, 8))
        first = dd.render(bounces=args.bounces)
        for _ in range(3):
            dd.render(bounces=args.bounces)
        out["dense_64k_ms"] = statistics.median(dd.render(bounces=args.bounces)["kernel_ms"] for _ in range(7))
        out["dense_64k_first_launch_ms"] = first["kernel_ms"]
        out["dense_64k_rays"] = first["ray_count"]
        dd.close()
    # (3b) the first launch of a shape (image / centre-out order: no costs known yet) against the steady state for the
    #      other single-GPU configs too — what a one-frame-per-process `cutrace <scene.json>` pays
    if os.path.basename(args.scene) == "bunny.json":
        d = tempfile.mkdtemp()
        fl, live = {}, {}
        for name, path, b in (("C1 sphere_plane.json", os.path.join(ROOT, "scene", "sphere_plane.json"), 5),
                              ("C3 mirror.json b8", os.path.join(ROOT, "scene", "mirror.json"), 8),
                              ("C3-deep (walls reflect 0.5) b8", scenes.make_mirror_deep(d), 8)):
            sc = ca.HostScene.load(path)
            x = ca.DeviceScene(sc, device=ds.device)
            x.render(bounces=b, rows=(0, 8))
            first = x.render(bounces=b)["kernel_ms"]
            for _ in range(3):
                x.render(bounces=b)
            steady = statistics.median(x.render(bounces=b)["kernel_ms"] for _ in range(7))
            fl[name] = {"first_launch_kernel_ms": first, "steady_kernel_ms": steady}
            # how many of a wave's 64 lanes are alive per trip on this config: one lane = one pixel for
            # the pixel's whole life, so lanes whose recursion ends early idle (shading.hpp:126-150)
            x.set_variant(ca.VAR_STATS)
            ca.DeviceScene.lane_stats(reset=True)
            x.render(bounces=b)
            live[name] = _live_summary(ca.DeviceScene.lane_stats(reset=True))
            x.close()
        out["first_launch_by_config"] = fl
        out["live_lanes_by_config"] = live
        # (3c) a MOVING camera (90 frames, eye ~1.5 cm and view ~0.35 deg per frame): every frame is ordered by the costs
        #      of the previous, different frame — against image order (scripts/gpu_campath.py's loop)
        out.update(campath(ca, hs, ds.device))
    # (3d) what a one-shot `cutrace <scene.json>` pays BEFORE its render (main.cu:21-30: load, cpu_to_gpu, then render):
    #      ctr_scene_create = host BVH build + flat records + device allocation + upload, for C2, C2-dense and C4; and the
    #      drop-in CLI's whole wall time for scene/bunny.json (process start, HIP initialisation, load, create, render, three JPGs)
    if os.path.basename(args.scene) == "bunny.json":
        import subprocess
        d = tempfile.mkdtemp()
        sc = {}
        for name, path in (("C2 bunny.json (1000 triangles)", os.path.join(ROOT, "scene", "bunny.json")),
                           ("C2-dense (64000 triangles)", scenes.make_dense_bunny(d, 3, width=w, height=h)),
                           ("C4 4x4 grid (16 x 1000 triangles)", scenes.make_bunny_grid(tempfile.mkdtemp()))):
            t0 = time.perf_counter()
            hsx = ca.HostScene.load(path)
            t1 = time.perf_counter()
            ms = []
            for _ in range(3):
                t2 = time.perf_counter()
                x = ca.DeviceScene(hsx, device=ds.device)
                ms.append((time.perf_counter() - t2) * 1e3)
                x.close()
            sc[name] = {"scene_create_ms": min(ms), "scene_create_ms_first": ms[0], "json_and_stl_load_ms": (t1 - t0) * 1e3}
        out["scene_create_ms"] = sc
        cli = os.path.join(ROOT, "cutrace_amd", "cutrace")
        if os.path.exists(cli):
            cwd = tempfile.mkdtemp()
            os.symlink(os.path.join(ROOT, "scene"), os.path.join(cwd, "scene"))
            walls = []
            line = ""
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([cli, "scene/bunny.json"], cwd=cwd, capture_output=True, text=True, timeout=300)
                walls.append((time.perf_counter() - t0) * 1e3)
                line = next((l for l in r.stdout.splitlines() if "ms" in l and "ender" in l), line)
            out["cli_bunny_json"] = {"wall_ms": min(walls), "wall_ms_first": walls[0], "its_own_timing_line": line.strip(),
                                     "what": "`cutrace scene/bunny.json` as a process: start, HIP initialisation, JSON + STL load, ctr_scene_create, "
                                             "ctr_render into page-locked grids, three JPG files"}
    # (4) SURVEY §8(d): bytes the REFERENCE's flat traversal streams for this frame (56 B x objects per ray_cast +
    #     48 B x triangles of every mesh whose AABB the ray hits + 28 B per pixel) — a workload property
    alg_bytes, alg_rays = ds.algorithmic_bytes(bounces=args.bounces)
    out["reference_equivalent_bytes_per_frame"] = alg_bytes
    out["reference_rays_per_frame"] = alg_rays
    return out


PROFILE_WORKLOADS = {"bunny": "bunny.json@1920x1080b5", "dense64k": "bunny_dense3.json@1920x1080b5",
                     "c4": "bunny_grid4x4.json@4096x4096b5"}
PROFILE_ROUND = "r04"
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)


def kernel_source_hash():
    """What the render kernel's ISA is made from: the PMC counters and the instruction mix under profiles/ describe ONE
    build; a roofline fraction computed from them for another build would be a stale constant."""
    import hashlib
    from cutrace_amd import build
    hh = hashlib.sha256()
    for f in ("cutrace_amd/csrc/render_kernel.hip", "cutrace_amd/csrc/scene_device.h", "cutrace_amd/csrc/bvh.h",
              "include/cutrace_amd.h"):
        hh.update(open(os.path.join(ROOT, f), "rb").read())
    hh.update(" ".join(x for x in build.HIP_FLAGS if not x.startswith("-I")).encode())
    return hh.hexdigest()[:16]


def roofline_from_profiles(workload, kern_avg_ms, counters_json):
    """VALU-issue roofline of the render kernel for `workload` from the committed PMC passes (counters_<tag>.json) and the
    execution-weighted instruction mix (valu_mix_dynamic_<tag>.json, scripts/dynamic_mix.py).  kern_avg_ms > 0: this
    run's own mean kernel time (HIP events) turns the cycle counts into rates; 0: only the fraction, which needs none."""
    tag = next((t for t, wl in PROFILE_WORKLOADS.items() if wl == workload), None)
    roof = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "G SIMD-cycles/s", "frac": None,
            "traffic": None, "kernel": "render_kernel", "kernel_ms_avg": kern_avg_ms or None,
            "kernel_variant_bits": "1 PREFILTER | 2 ANYHIT | 8 BVH | 32 FASTPOW | 64 built for 6 waves per SIMD"}
    cj = counters_json or (os.path.join(PROFILE_DIR, f"counters_{tag}.json") if tag else "")
    mj = os.path.join(PROFILE_DIR, f"valu_mix_dynamic_{tag}.json") if tag else ""
    if not (cj and os.path.exists(cj) and mj and os.path.exists(mj)):
        return roof
    cnt, mix = json.load(open(cj)), json.load(open(mj))
    roof["kernel"] = cnt.get("kernel", "render_kernel")
    if cnt.get("workload") != workload:
        return roof
    have, want = cnt.get("kernel_source_sha256"), kernel_source_hash()
    if have != want:
        roof["stale"] = f"profiles/{PROFILE_ROUND} describes kernel source {have}, this tree is {want}: no fraction is claimed"
        return roof
    cost = mix["mean_issue_cycles_per_valu"]
    cycles = cnt["cycles_per_launch"]
    traffic = cnt.get("hbm_bytes_per_launch")
    # The kernel's CYCLE count per launch is what the counters pin (same code, same work); the clock the chip holds
    # differs between a profiled and an un-profiled run (MI355X_MICROARCH.md, DVFS).  So
    #   frac = VALU issue cycles the launch needs / SIMD cycles it had = valu_insts x mean issue cost / (1024 x cycles_per_launch)
    # is recomputable from profiles/<round> alone; this run's kernel time only scales achieved and peak by the same clock.
    frac = cnt["valu_insts_per_launch"] * cost / (1024.0 * cycles)
    roof.update({
        "frac": frac, "traffic": traffic, "cycles_per_launch": cycles, "kernel_ms_in_pmc_pass": cnt["kernel_ns_in_pmc_pass"] * 1e-6,
        "clock_ghz_in_pmc_pass": cnt["effective_clock_ghz"], "valu_insts_per_launch": cnt["valu_insts_per_launch"],
        "salu_insts_per_launch": cnt.get("salu_insts_per_launch"), "smem_insts_per_launch": cnt.get("smem_insts_per_launch"),
        "mean_issue_cycles_per_valu": cost, "valu_mix_dynamic": mix.get("class_shares"),
        "valu_mix_cross_check": mix.get("pmc_check"),
        "issue_cost_cycles": {"F_vgpr_only": 2.2, "H_sgpr_operand_cmp_packed_min3": 4.1, "Q_transcendental": 8.1},
        "simds": 1024, "hbm_bytes_per_launch": traffic, "hbm_peak_gbs": HBM_PEAK_GBS,
        "wait_any_share_of_wave_cycles": (cnt["wait_any_quadcycles"] / cnt["wave_quadcycles_per_launch"]) if cnt.get("wave_quadcycles_per_launch") else None,
        "source": f"profiles/{PROFILE_ROUND}/counters_{tag}.json (rocprofv3 --pmc passes of bench.py on this workload), "
                  f"profiles/{PROFILE_ROUND}/valu_mix_dynamic_{tag}.json (execution-weighted mix, scripts/dynamic_mix.py), profiles/r02/valu_issue.txt"})
    ij = os.path.join(PROFILE_DIR, "issue_mix_measured.json")
    if os.path.exists(ij):
        # the additive pricing (sum of per-kind costs) against a MEASURED stream of the kernel's mix: mixing plain and
        # half-rate instructions costs a little less than the sum (3.12 vs 3.25 cycles per VALU at 6 waves per SIMD), and
        # the scalar instructions between them are not free (3.49 with the kernel's 0.65 SALU per VALU)
        im = json.load(open(ij))
        scale = cost / im["additive_model_cycles_per_valu"]  # the kernel's mix differs slightly from the stream's
        spent = 1024.0 * cycles / cnt["valu_insts_per_launch"]
        roof["simd_cycles_spent_per_valu"] = spent
        roof["frac_valu_issue_measured_stream"] = im["measured_valu_only"] * scale / spent
        roof["frac_issue_incl_scalar_measured_stream"] = im["measured_with_salu"] * scale / spent
        roof["issue_mix_measured"] = {k: im[k] for k in ("waves_per_simd", "measured_valu_only", "measured_with_salu", "additive_model_cycles_per_valu")}
    if kern_avg_ms and kern_avg_ms > 0:
        clock = cycles / (kern_avg_ms * 1e-3) / 1e9
        roof.update({"achieved": cnt["valu_insts_per_launch"] * cost / (kern_avg_ms * 1e-3) / 1e9, "peak": 1024.0 * clock,
                     "clock_ghz_this_run": clock,
                     "hbm_frac": (traffic / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None})
    return roof


def self_launch(args):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves, as
    a CHILD process — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same
    arguments>` — wait for it, and exit with its code.  This runs before `import torch` and before anything touches a
    device (never replace a process that has initialised the GPU); the child inherits stdout / stderr, so rank 0's ONE
    JSON line is this command's line."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this image
    env["CUTRACE_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist
    import cutrace_amd as ca
    from cutrace_amd.tiling import FrameTiler

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:   # (RANK / WORLD_SIZE=1 set by hand: self_launch() above covers the plain command)
            raise SystemExit("bench.py --gpus %d: WORLD_SIZE=1 in the environment; unset RANK / WORLD_SIZE or use "
                             "torch.distributed.run" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # Rehearsal on a one-GPU box (never the measured configuration): CUTRACE_BENCH_SHARE_GPU=1 puts every
    # rank on device 0 and CUTRACE_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one GPU.
    if os.environ.get("CUTRACE_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("CUTRACE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # what the process group itself says it is (the record shows RCCL saw N ranks on N distinct devices)
    dist_info = {"world_size": 1, "backend": None, "launched_by": "single process", "devices": [local_rank]}
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        every = torch.zeros(world, 5, dtype=torch.int64, device=dev)   # (a SUM of one-hot rows: gloo has no GPU all_gather)
        every[rank] = torch.tensor([rank, local_rank, torch.cuda.current_device(), int(getattr(props, "pci_bus_id", -1)),
                                    int(getattr(props, "pci_domain_id", -1))], dtype=torch.int64, device=dev)
        dist.all_reduce(every)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "launched_by": "bench.py itself (child torch.distributed.run)" if os.environ.get("CUTRACE_BENCH_SELF_LAUNCHED") == "1"
                                    else "external launcher",
                     "devices": [int(t[2]) for t in every],
                     "ranks": [{"rank": int(t[0]), "local_rank": int(t[1]), "cuda_device": int(t[2]), "pci_bus_id": int(t[3]),
                                "pci_domain_id": int(t[4])} for t in every],
                     "shared_gpu_rehearsal": os.environ.get("CUTRACE_BENCH_SHARE_GPU") == "1"}

    if args.workload == "c4":
        import tempfile
        from cutrace_amd import scenes
        args.scene = scenes.make_bunny_grid(tempfile.mkdtemp(prefix=f"c4_r{rank}_"))
    hs = ca.HostScene.load(args.scene)
    assert hs.ok, "scene failed to load"
    if args.width and args.height:
        hs.set_size(args.width, args.height)
    w, h = hs.size
    sim = args.of if (world == 1 and args.of > 1) else 0

    def measure(hs, scaling, roots, steps, warmup, probe, in_flight=1):
        """One workload through the tiler: -> (ds, tiler, frames, dt_max, rays_step, kern_avg, kern_io)."""
        w, h = hs.size
        # in_flight == 2: two scene handles (each keeps its own counters, tile costs and dispatch order: launches on ONE handle
        # must not overlap) on two streams, used alternately
        handles = [ca.DeviceScene(hs, device=local_rank) for _ in range(in_flight)]  # raises when the HIP library is missing
        ds = handles[0]
        if args.variant:
            for x in handles:
                x.set_variant(args.variant)
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(in_flight - 1)]
        frames = world if scaling == "weak" else 1
        n_slots = 2 if in_flight == 1 else 2 * in_flight
        if sim:  # one process plays rank `as_rank` of `of` ranks: same launches, no collective
            frames = sim if scaling == "weak" else 1
            tiler = FrameTiler(w, h, frames, args.as_rank, sim, dev, roots=roots, slots=n_slots)
            tiler.gather = lambda slot: None
            tiler.begin = lambda slot: None
            tiler.finish = lambda: torch.cuda.synchronize(dev)
        else:
            tiler = FrameTiler(w, h, frames, rank, world, dev, roots=roots, slots=n_slots)
        if frames > 1:
            import ctypes
            cam0 = hs.desc.contents.cam
            cams = []
            for _ in range(frames):  # synthetic camera path: the scene camera repeated
                c = ca.Camera()
                ctypes.memmove(ctypes.byref(c), ctypes.byref(cam0), ctypes.sizeof(ca.Camera))
                cams.append(c)
            for x in handles:
                x.set_cameras(cams)
        counters_all = [torch.zeros(16, dtype=torch.int64, device=dev) for _ in range(in_flight)]
        counters = counters_all[0]
        step_no = [0]

        def render_step(events=None):
            k = step_no[0]
            slot = k % tiler.slots
            step_no[0] += 1
            hnd, stream = handles[k % in_flight], streams[k % in_flight]
            with torch.cuda.stream(stream):
                tiler.begin(slot)  # the gather that last used this part of the buffer ring must be done
                buf = tiler.local[slot]
                d0, c0, n0, _ = tiler.sec
                esz = buf.element_size()
                if events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                # ONE launch renders this rank's rows of all `frames` frames of the step (camera path batch)
                hnd.render_device_batch(buf.data_ptr() + d0 * esz, buf.data_ptr() + c0 * esz, buf.data_ptr() + n0 * esz,
                                        n_frames=frames, frame_stride_px=tiler.cap * w, d_counters=counters_all[k % in_flight].data_ptr(),
                                        stream=stream.cuda_stream, fudge=1e-3, bounces=args.bounces, rows=tiler.rows,
                                        part_stride=tiler.part_stride)
                if events is not None:
                    e1.record(stream)
                    events.append((e0, e1))
                tiler.gather(slot)  # asynchronous: overlaps with the next step's rendering

        def barrier():
            tiler.finish()      # every outstanding gather / re-interleave completes INSIDE the timed region
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # rays per step for this rank (counting launch, untimed)
        counters.zero_()
        render_step()
        barrier()
        rays_rank_step = int(counters[0].item())
        for _ in range(max(0, warmup - 1)):
            render_step()
        barrier()
        events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            render_step(events)
        barrier()
        dt = time.perf_counter() - t0

        kern_ms = [a.elapsed_time(b) for a, b in events]
        # for the record (untimed): the same launch with tiles dispatched in image order, i.e. what the
        # first launch of a shape costs before the scheduling feedback exists
        kern_io = None
        if probe:
            for x in handles:
                x.set_variant(args.variant | ca.VAR_NO_REORDER)
            ev_io = []
            for _ in range(3):
                render_step(ev_io)
                barrier()
            kern_io = min(a.elapsed_time(b) for a, b in ev_io)
            for x in handles:
                x.set_variant(args.variant)
        t = torch.tensor([dt, float(rays_rank_step), sum(kern_ms) / max(1, len(kern_ms))], dtype=torch.float64, device=dev)
        if world > 1:
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dt_max, rays_step, kern_avg = float(tmax[0]), float(tsum[1]), float(tsum[2]) / world
        else:
            dt_max, rays_step, kern_avg = dt, float(rays_rank_step), float(t[2])
        return ds, tiler, frames, dt_max, rays_step, kern_avg, kern_io

    ds, tiler, frames, dt_max, rays_step, kern_avg, kern_io = measure(hs, args.scaling, args.roots, args.steps, args.warmup,
                                                                       not args.skip_probe, args.in_flight)

    if args.check and not sim:
        import numpy as np
        n_ok = torch.zeros(1, dtype=torch.int64, device=dev)
        if tiler.final_frames:  # every root rank checks the frames it assembled
            ref = ca.DeviceScene(hs, device=local_rank)
            ref.set_variant(ca.VAR_NO_REORDER)
            want = ref.render(bounces=args.bounces)
            for i, f in enumerate(tiler.final_frames):
                for k in ("depth", "color", "normal"):
                    got = tiler.final[k][i].cpu().numpy()
                    assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), f"--check: rank {rank} frame {f} {k} differs"
                n_ok += 1
        if world > 1:
            dist.all_reduce(n_ok)
        assert int(n_ok.item()) == frames, f"--check: {int(n_ok.item())} of {frames} frames were assembled"
        if rank == 0:
            print(f"check: {frames} gathered frame(s) bitwise equal to the single-process render", file=sys.stderr, flush=True)

    # ---- the line's core: everything the timed region produced ----
    out = None
    if rank == 0:
        total_rays = rays_step * args.steps
        value = total_rays / dt_max / 1e6
        workload = f"{os.path.basename(args.scene)}@{w}x{h}b{args.bounces}"
        # roofline of the dominant (only) kernel, per launch: VALU issue (the committed counters describe ONE launch of ONE whole
        # frame: no fraction is claimed for a step of several frames or a part of a frame)
        roof = roofline_from_profiles(workload if (world == 1 and frames == 1 and not sim) else "", kern_avg, args.counters_json)
        config = {"workload": f"{os.path.basename(args.scene)}@{w}x{h} bounces={args.bounces} fudge=1e-3, "
                              f"{frames} frame(s)/step row-tiled over {world} GPU(s), "
                              + ("frame f gathered to rank f mod N" if tiler.rotate_roots else "gather to rank 0"),
                  "tile_order": "expensive tiles first, costs recorded by the previous launch of the same "
                                "shape (first launch of a shape: blocks of tiles from the image centre outwards)",
                  "kernel_ms_image_order": kern_io,
                  "dist": dist_info,
                  "frames_per_step": frames, "rays_per_step": rays_step,
                  "frame_ms": dt_max / args.steps * 1e3 / frames,
                  "unique_mrays_per_s": (rays_step - frames * w * h) * args.steps / dt_max / 1e6,
                  "steps_in_flight": args.in_flight}
        for tag, key in (("dense64k", "dense_64k_roofline"), ("c4", "c4_roofline")):
            r_ = roofline_from_profiles(PROFILE_WORKLOADS[tag], 0.0, "")
            if r_.get("frac") is not None:  # (constants of the committed PMC passes: that scene is not timed live here)
                config[key] = {k: r_[k] for k in ("frac", "valu_insts_per_launch", "mean_issue_cycles_per_valu", "cycles_per_launch",
                                                  "kernel_ms_in_pmc_pass", "hbm_bytes_per_launch", "source")}
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config,
            "roofline": roof,
        }

    # ---- untimed side measurements.  They involve collectives that have never run on more than one GPU: should one of them
    #      hang, a watchdog prints the line as it stands (config.extras says so) and ends every rank, so that the timed result —
    #      what the run was for — is never lost to an extra ----
    import threading
    printed = threading.Lock()

    def emit_and_leave(reason):
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            out["config"]["extras"] = reason
            print(json.dumps(out), flush=True)
        else:
            time.sleep(3.0)   # (rank 0 prints first)
        os._exit(0)

    deadline = float(os.environ.get("CUTRACE_BENCH_EXTRAS_DEADLINE_S", "900" if world == 1 else "420"))
    watchdog = threading.Timer(deadline, emit_and_leave, args=(f"abandoned after {deadline:.0f} s (a side measurement did not return)",))
    watchdog.daemon = True
    if not args.no_extras:
        watchdog.start()

    try:
        # BASELINE config 5 next to the metric's workload, at every N the driver runs: the 4x4 bunny grid @4096x4096, ONE frame
        # per step row-tiled over all ranks and gathered to rank 0 (strong scaling), untimed for `value`
        c4_leg = None
        if not args.no_extras and args.workload != "c4" and not sim and os.path.basename(args.scene) == "bunny.json" and not (args.width or args.height):
            import tempfile
            from cutrace_amd import scenes
            c4_hs = ca.HostScene.load(scenes.make_bunny_grid(tempfile.mkdtemp(prefix=f"c4_r{rank}_")))
            # (12 steps after 4 of warm-up: the dispatch order is rebuilt from measured costs over the first launches of a shape, and
            #  a part of a frame — 5 waves per slot at N = 8 — feels an order that has not settled: 6 steps after 2 read 6 % slower)
            # (a gloo rehearsal moves every 470 MB frame through host memory: three steps there)
            c4_steps, c4_warm = (12, 4) if (world == 1 or backend == "nccl") else (3, 1)
            _, c4_tiler, _, c4_dt, c4_rays, c4_kern, _ = measure(c4_hs, "strong", "rank0", c4_steps, c4_warm, False)
            c4_leg = {"workload": "4x4 bunny grid (16 meshes x 1000 triangles) @4096x4096 bounces=%d, one frame per step row-tiled "
                                  "over %d GPU(s), gathered to rank 0" % (args.bounces, world),
                      "n_gpus": world, "steps": c4_steps, "frame_ms": c4_dt / c4_steps * 1e3, "mrays_per_s": c4_rays * c4_steps / c4_dt / 1e6,
                      "rays_per_frame": c4_rays, "kernel_ms_avg_over_ranks": c4_kern}
            if rank == 0:
                out["config"]["c4_strong"] = c4_leg
            # the same with two frames in flight (--in-flight 2): a rank's part of ONE frame is 5 waves per slot at N = 8 and ends in a
            # tail of its dearest tiles (one 8x8 tile = one wave, up to 5x the mean); the next frame's first waves fill that tail
            del c4_tiler
            _, c4_tiler, _, c4_dt2, c4_rays2, _, _ = measure(c4_hs, "strong", "rank0", c4_steps, c4_warm, False, 2)
            c4_leg["frame_ms_two_in_flight"] = c4_dt2 / c4_steps * 1e3
            c4_leg["mrays_per_s_two_in_flight"] = c4_rays2 * c4_steps / c4_dt2 / 1e6
            if world == 1:
                c4_ds = ca.DeviceScene(c4_hs, device=local_rank)
                c4_ds.set_variant(ca.VAR_STATS)
                ca.DeviceScene.lane_stats(reset=True)
                c4_ds.render(bounces=args.bounces)
                c4_leg["live_lanes"] = _live_summary(ca.DeviceScene.lane_stats(reset=True))
                c4_ds.close()
            del c4_tiler
        if not args.no_extras and args.in_flight == 1 and not sim:
            # the timed workload once more with two steps in flight (--in-flight 2), untimed for `value`
            _, t2_tiler, _, t2_dt, t2_rays, _, _ = measure(hs, args.scaling, args.roots, args.steps, args.warmup, False, 2)
            if rank == 0:
                out["config"]["two_steps_in_flight"] = {
                    "ms_per_step": t2_dt / args.steps * 1e3, "mrays_per_s": t2_rays * args.steps / t2_dt / 1e6,
                    "what": "consecutive steps alternate between two scene handles on two streams: the first waves of step k+1 fill the "
                            "slots the tail of step k leaves empty; `value` above is measured with one launch after the other"}
            del t2_tiler
        if rank == 0:
            if world == 1 and not args.no_extras:
                out["config"].update(extras(ca, hs, args, ds))
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(ca, hs, args.bounces, args.cpu_sample_div)
    except Exception as e:  # noqa: BLE001 — whatever goes wrong in a side measurement must not cost the timed result
        import traceback
        traceback.print_exc()
        emit_and_leave(f"a side measurement failed on rank {rank}: {type(e).__name__}: {e}")
    watchdog.cancel()
    if printed.acquire(blocking=False):
        if rank == 0:
            print(json.dumps(out), flush=True)
    else:
        time.sleep(10.0)   # (the watchdog is printing and will end the process)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
