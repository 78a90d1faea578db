#!/usr/bin/env python3
"""Benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5]

A step is one frame of the configuration, from the launch of its kernels to the frame's arrival in (pinned) host
memory: trace + resolve kernels, with several GPUs the gather of the tile buffers and the assembly on the first one,
and the copy of the frame to the host.  The scene, camera and buffers are resident before the timed region.

  c2 (default)  the Game1 scene (Game1.cs:96-144), 1920x1080, 16 spp, depth 8, traced as the reference does
                (GetRayColor on the BVHNode world, Game1.cs:293): the configuration the metric is quoted on
  c3            teapot.dae subdivided to 101,120 triangles, 1920x1080, 64 spp, depth 4
  c4            983,040-triangle mesh + 10,000 random spheres, 3840x2160, 256 spp, depth 8
  c5            the 983,040-triangle mesh, 7680x4320, 1024 spp, depth 8
  (--spp / --size shrink c3..c5 for quick runs; the workload string then says so.)

Several GPUs (--scaling weak|strong; default: weak for c1..c3, strong for c4 and c5, whose configurations name one
fixed frame over 2 / 4 / 8 GPUs).  weak: the configuration's spp per GPU, N times that per frame.  strong: the
configuration's own frame, its tiles shared out over the N GPUs.
  * started as a plain command (`python bench.py --gpus N`), one process drives all N GPUs through one libsrt context:
    32x32 screen tiles interleaved over the devices, tile buffers gathered on the first device by grouped
    ncclSend / ncclRecv, assembled there;
  * started by torch.distributed.run with N ranks, every rank owns one GPU and one context, and the tile buffers are
    gathered on rank 0 with torch.distributed (backend nccl = RCCL), then assembled by srt_assemble_tiles.
Either way the line reports, per step: gather_ms and assemble_ms (HIP events), the device time of every GPU's kernels,
and the tail (slowest minus median GPU).

Metric: Mrays/s, rays = GetRayColor calls with depth >= 1 (primary + bounce rays, one closest-hit query each), counted
on the device.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import simple_raytracer_amd as srt  # noqa: E402
from simple_raytracer_amd import game1, scenes  # noqa: E402

TILE = 32
PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 (= f32 MFMA) peak; it counts a packed fused multiply-add, 4 flops per lane and cycle
PEAK_FP32_UNFUSED_TFLOPS = PEAK_FP32_TFLOPS / 4.0   # one unfused, unpacked operation per lane and cycle: what this arithmetic may use
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E peak
PEAK_L2_GBS = 34500.0      # aggregate L2 bandwidth (8 XCDs)

# name -> (scene factory(width, height), world kind, width, height, spp per GPU, depth, description)
CONFIGS = {
    "c1": (lambda w, h: game1.default_scene(w, h), srt.WORLD_BVHNODE, 1280, 720, 500, 4,
           "default Game1 scene at the reference's own settings (Game1.cs:41-43, 78-79), bvhnode world"),
    "c2": (lambda w, h: game1.default_scene(w, h), srt.WORLD_BVHNODE, 1920, 1080, 16, 8,
           "default Game1 scene (5 spheres, 2 point lights), bvhnode world"),
    "c3": (lambda w, h: scenes.teapot_scene(w, h, 2), srt.WORLD_SURFACELIST, 1920, 1080, 64, 4,
           "teapot.dae subdivided to 101,120 triangles, 2 point lights, surfacelist world (library tree)"),
    "c4": (lambda w, h: scenes.sphere_mesh_scene(w, h, 5, 10000), srt.WORLD_SURFACELIST, 3840, 2160, 256, 8,
           "983,040-triangle mesh + 10,000 random spheres, 2 point lights, surfacelist world (library tree)"),
    "c5": (lambda w, h: scenes.sphere_mesh_scene(w, h, 5), srt.WORLD_SURFACELIST, 7680, 4320, 1024, 8,
           "983,040-triangle mesh, 2 point lights, surfacelist world (library tree)"),
}

# Memory-side numbers of the dominant kernel on the N=1 workloads come from counter passes of this same command
# (tools/pmc_bench.sh: separate `rocprofv3 --pmc` runs for FETCH_SIZE, WRITE_SIZE and the L2 request / hit / miss counters;
# FETCH_SIZE and WRITE_SIZE are in KiB, and FETCH_SIZE is doubled, the correction for gfx950).  A normal run has no
# profiler attached, so it reads the digest of those passes (tools/pmc_bench_digest.py writes it): per configuration the
# bytes per frame and the L2 request rate, with the profile they come from.
TRAFFIC_DIGEST = os.path.join(ROOT, "bench_traffic.json")   # at the top of the tree: profiles/ does not travel to the GPU box

# Algorithmic work per event: bytes fetched / FP32 operations (add, sub, mul, div, sqrt, min, max, compare = 1 each)
# of the reference algorithm, excluding work the GPU legitimately skips.
EVENT_COST = {                 # (bytes, flops)
    "box_tests": (32, 24),     # AABB.hit, RaytracingClasses.cs:773-789
    "sphere_tests": (16, 29),  # Sphere.hit up to the root test, RaytracingClasses.cs:461-478
    "tri_tests": (36, 62),     # Tri.hit, RaytracingClasses.cs:599-626
    "rect_tests": (20, 10),    # RectAxis.hit, RaytracingClasses.cs:518-528
    "hit_nodes": (72, 145),    # hit record (normals 36 + material 36) + 30 normal/point + 25 blend + 90 shading of GetRayColor
    "rays_shadow": (28, 140),  # PointLight.calculateLight, RaytracingClasses.cs:96-188, + occluder point (30)
    "pixels": (4, 6),          # gamma + pack, Game1.cs:301-306
}


def algorithmic_work(stats, pixels, scale=1.0):
    ev = {k: stats[k] * scale for k in EVENT_COST if k != "pixels"}
    ev["pixels"] = pixels
    b = sum(ev[k] * c[0] for k, c in EVENT_COST.items())
    f = sum(ev[k] * c[1] for k, c in EVENT_COST.items())
    return b, f


def cpu_baseline(name, world, lights, cam, width, height, spp, depth):
    """The CPU oracle (a C++ restatement of the reference's algorithm, incl. its own BVH builder, threaded by row bands
    like Game1.cs:376-380) timed on this host.  c2: one whole frame (about 20 s on 16 cores).  Mesh configurations:
    a few columns x `threads` rows at the centre of the frame at 1 spp -- the reference's tree costs 10^4..10^5 box tests per ray."""
    from oracle import orc   # checker / baseline only
    threads = min(16, os.cpu_count() or 1)
    if name in ("c1", "c2"):
        sc, ocam = orc.default_scene(), orc.default_camera(width, height)
        cpu_spp = min(spp, 64)   # a whole frame at up to 64 spp bounds the sample (the rate does not depend on the sample count)
        _, st = sc.render(ocam, width, height, samples=cpu_spp, max_depth=depth, seed=1, window=(0, 0, width, height), threads=threads)
        sample = "one whole %dx%d %d-spp depth-%d frame (all rows, row bands over %d threads), %d rays in %.2f s" % (
            width, height, cpu_spp, depth, threads, st["rays_path"], st["seconds"])
    else:
        flat = world.flatten()
        mats = [orc.material(m.albedo, m.metalness, m.smoothness, m.emission) for m in flat["materials"]]
        tris = orc.tris_from_arrays(flat["tri_pos"], flat["tri_nrm"], 0)
        tris["material"] = flat["tri_mat"]
        sph = [orc.Sphere(orc.f3(s[:3]), float(s[3]), int(m)) for s, m in zip(flat["spheres"], flat["sphere_mat"])]
        t = time.time()
        sc = orc.Scene(mats, spheres=sph, tris=tris, lights=[orc.point_light(l.position, l.brightness, None) for l in lights.lights])
        build_s = time.time() - t
        ocam = orc.Camera(orc.f3(cam.position), orc.f3(cam.lowerLeftCorner), orc.f3(cam.horizontal), orc.f3(cam.vertical))
        # one row per thread; the width bounds the sample to some tens of seconds (1 to 1,000 rays per second and thread)
        cols = {"c3": 32, "c4": 2, "c5": 4}[name]
        x0, y0 = ((width // 2) // 32) * 32, ((height // 2) // 32) * 32
        rows = threads
        _, st = sc.render(ocam, width, height, samples=1, max_depth=depth, seed=1, window=(x0, y0, x0 + cols, y0 + rows), threads=threads)
        if st["seconds"] < 5.0:
            # the probe was quick: a second window sized for about 20 s at the probe's rate (wider first, then taller)
            grow = min(512.0, 20.0 / max(st["seconds"], 1e-3))
            wider = int(min(grow, max(1, (width - x0) // cols), 16))
            taller = int(min(max(1.0, grow / wider), max(1, (height - y0) // rows), 32))
            cols, rows = cols * wider, rows * taller
            _, st = sc.render(ocam, width, height, samples=1, max_depth=depth, seed=1, window=(x0, y0, x0 + cols, y0 + rows), threads=threads)
        sample = "%dx%d window at the centre of the %dx%d frame, 1 spp, depth %d, the reference's own tree (built in %.1f s): %d rays in %.1f s" % (
            cols, rows, width, height, depth, build_s, st["rays_path"], st["seconds"])
    out = {"value": round(st["rays_path"] / st["seconds"] / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port", "sample": sample,
           "rays_in_sample": int(st["rays_path"])}
    if st["rays_path"] < 1000:
        # the reference's tree costs 10^4..10^5 box tests per ray on these scenes: a bounded sample holds few rays, and a
        # rate from so few says how slow that tree is, nothing about the GPU kernels
        out["unit"] = "Mrays/s (from only %d rays: an order of magnitude, not a measurement)" % st["rays_path"]
    return out


def pinned_frame(pixels):
    """Page-locked host memory for the frame, as a numpy view (the C# host pins its Color[] the same way)."""
    t = torch.empty(pixels, dtype=torch.int32).pin_memory()
    return t, t.numpy().view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel and GPU (default: the configuration's)")
    ap.add_argument("--size", default="", help="WxH (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", choices=["rccl", "peer"], default="rccl", help="in-library multi-GPU: how tiles reach the first device")
    ap.add_argument("--virtual-devices", action="store_true", help="rehearsal: all --gpus entries name device 0 (one context, several replicas)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="", help="several GPUs: weak = the configuration's spp per GPU, "
                    "strong = the configuration's own frame over all GPUs (default: strong for c4 and c5, weak otherwise)")
    args = ap.parse_args()
    scaling = args.scaling or ("strong" if args.config in ("c4", "c5") else "weak")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # world == 1 and --gpus N > 1: this process drives all N GPUs through one context (the C# host's situation)
    in_library = world == 1 and args.gpus > 1
    n_gpus = args.gpus

    make, kind, width, height, spp_per_gpu, depth, what = CONFIGS[args.config]
    reduced = []
    if args.spp:
        spp_per_gpu = args.spp
        reduced.append("spp")
    if args.size:
        width, height = (int(v) for v in args.size.lower().split("x"))
        reduced.append("size")
    spp = spp_per_gpu * n_gpus if scaling == "weak" else spp_per_gpu

    # SRT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (the ranks then share devices and
    # the tile buffers are gathered through host memory); the measured configuration is nccl = RCCL, one GPU per rank.
    backend = os.environ.get("SRT_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    if in_library:
        devices = [0] * n_gpus if args.virtual_devices else list(range(n_gpus))
        r = srt.Renderer(devices=devices, gather=srt.GATHER_RCCL if args.gather == "rccl" else srt.GATHER_PEER_COPY)
    else:
        r = srt.Renderer(device=device_index)
    info = r.device_info()
    if in_library and not args.virtual_devices and args.gather == "rccl" and info["rccl_ranks"] != n_gpus:
        raise SystemExit("the RCCL communicator has %d ranks, expected %d" % (info["rccl_ranks"], n_gpus))
    scene_world, lights, cam = make(width, height)
    t0 = time.perf_counter()
    r.set_scene(scene_world, lights, cam, world_kind=kind)
    commit_s = time.perf_counter() - t0

    stream = torch.cuda.current_stream().cuda_stream
    tf = None
    if world == 1:
        p = r.params(width, height, samples=spp, max_depth=depth, seed=1)
        local = width * height
        host_t, host = pinned_frame(local)
    else:
        from simple_raytracer_amd.distributed import TiledFrame
        tf = TiledFrame(r, width, height, rank=rank, world_size=world, tile=TILE, device="cuda", samples=spp, max_depth=depth, seed=1)
        p, local = tf.params, tf.local_pixels
        host_t = torch.empty(width * height, dtype=torch.int32).pin_memory() if rank == 0 else None

    import ctypes as C
    device_ms = []     # per step: device time of the frame's trace + resolve kernels (HIP events on the launch stream)
    counts = [0.0, 0.0]  # path rays, shadow rays of the timed steps
    gather_ms, assemble_ms, per_device = [], [], []   # several GPUs, per step (in-library) or from one extra step (one process per GPU)

    timed_stats = {}      # the library's statistics of the last timed step

    def step(record=False):
        if tf is None:
            # blocking: kernels (+ gather + assemble) + copy of the frame into pinned host memory
            r._check(r._lib.srt_render(r._ctx, C.byref(p), host.ctypes.data_as(C.c_void_p)))
            if record:
                st = r.stats()   # collected by the blocking render; no extra wait
                timed_stats.update(st)
                device_ms.append(st["kernel_ms"])
                counts[0] += st["rays_path"]
                counts[1] += st["rays_shadow"]
                if in_library:
                    gather_ms.append(st["gather_ms"])
                    assemble_ms.append(st["assemble_ms"])
                    per_device.append([r.device_stats(i)["kernel_ms"] for i in range(len(info["devices"]))])
        else:
            tf.render_local()
            tf.gather()      # RCCL gather of the tile buffers to rank 0
            frame = tf.assemble()    # rank 0: tiles -> frame
            if frame is not None:
                host_t.copy_(frame, non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    r.stats()          # drop the warmup's counters
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    rays, shadow = counts
    if tf is not None:
        st = r.stats()     # counters accumulated over the timed launches; kernel_ms is the last launch's
        timed_stats.update(st)
        rays, shadow = float(st["rays_path"]), float(st["rays_shadow"])
        device_ms.append(st["kernel_ms"])
        # the parts of a step, from one more (untimed) step with HIP events around them on this rank's stream
        fence()
        r.stats()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf.render_local()
        k_ms = r.stats()["kernel_ms"]            # waits for this rank's kernels
        if world > 1:
            dist.barrier()                       # every rank has its tiles: what follows is the exchange alone
        ev[0].record()
        tf.gather()
        ev[1].record()
        frame = tf.assemble()
        ev[2].record()
        torch.cuda.synchronize()
        parts = torch.tensor([k_ms, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])], dtype=torch.float64, device=reduce_device)
        every = [torch.zeros_like(parts) for _ in range(world)]
        dist.all_gather(every, parts)
        per_device.append([float(e[0].item()) for e in every])
        gather_ms.append(float(every[0][1].item()))      # rank 0 receives
        assemble_ms.append(float(every[0][2].item()))

    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    rr = torch.tensor([rays, shadow], dtype=torch.float64, device=reduce_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays, total_shadow = float(rr[0].item()), float(rr[1].item())

    if rank == 0:
        # Algorithmic work of ONE frame on this rank's devices, from the counting build of the same kernels.  The
        # big configurations count a 16-spp render of the same frame and scale by the sample ratio.
        count_spp = spp if args.config == "c2" else min(spp, 16)   # the counting render of the other configurations is bounded; counts scale with spp
        pc = r.params(width, height, samples=count_spp, max_depth=depth, seed=1)
        if tf is not None:
            pc = r.params(width, height, samples=count_spp, max_depth=depth, seed=1, layout=srt.LAYOUT_TILES, tile_w=TILE, tile_h=TILE,
                          rank=rank, world_size=world)
        scratch = np.zeros(r.local_pixels(pc), dtype=np.uint32)
        r._check(r._lib.srt_render_counted(r._ctx, C.byref(pc), scratch.ctypes.data_as(C.c_void_p)))
        cst = r.stats()
        px = width * height if tf is None else local
        abytes, aflops = algorithmic_work(cst, px, spp / count_spp)
        avg_ms = float(np.mean(device_ms))
        tflops = aflops / (avg_ms * 1e-3) / 1e12
        gbs = abytes / (avg_ms * 1e-3) / 1e9
        if in_library:
            tflops, gbs = tflops / n_gpus, gbs / n_gpus     # per GPU: the frame's work is spread over n_gpus devices
        # Which roof applies is decided by where the scene lives and by what the counters show.  The Game1 scene is 5
        # nodes + 5 spheres: its "algorithmic bytes" are scalar-cache hits and the kernel is bound by vector FP32
        # issue.  The mesh scenes (17 MB of nodes + 47 MB of triangles) live in L2 + Infinity Cache: their SQ counters
        # show VALU issue ~ 96 % busy and 0.3 TB/s from beyond L2, so vector FP32 issue is their roof as well, and
        # the node / triangle bytes are reported against the L2 bandwidth as a second figure.
        roofline = {"bound": "valu_fp32", "achieved": round(tflops, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / PEAK_FP32_TFLOPS, 4),
                    # the peak counts packed fused multiply-adds (4 flops per lane and cycle); this arithmetic may not contract
                    # a * b + c and mostly cannot pack, so one operation per lane and cycle is the most it could ever issue
                    "peak_unfused": PEAK_FP32_UNFUSED_TFLOPS, "frac_unfused": round(tflops / PEAK_FP32_UNFUSED_TFLOPS, 4)}
        # memory side, from the counter passes of this command (a constant: a normal run has no profiler attached)
        traffic, traffic_source, memory = None, None, {}
        if world == 1 and not in_library and not reduced and os.path.exists(TRAFFIC_DIGEST):
            d = json.load(open(TRAFFIC_DIGEST)).get(args.config)
            if d:
                traffic = int((2.0 * d["fetch_kib"] + d["write_kib"]) * 1024)
                traffic_source = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, 2*FETCH+WRITE summed over the %d kernel launches of a frame (%s); "
                                  "constant, not re-measured in this run" % (d["launches_per_frame"], d["source"]))
                memory = {"hbm_side_gbs": round(traffic / (avg_ms * 1e-3) / 1e9, 1),
                          "hbm_side_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                          "l2_request_gbs": round(d["tcc_req"] * 128.0 / (avg_ms * 1e-3) / 1e9, 1),
                          "l2_frac": round(d["tcc_req"] * 128.0 / (avg_ms * 1e-3) / 1e9 / PEAK_L2_GBS, 4),
                          "l2_hit_rate": round(d["tcc_hit"] / max(1.0, d["tcc_hit"] + d["tcc_miss"]), 3)}
        # scenes with deep ray trees switch to the level-by-level tracer after their first frame (the warm-up): the timed
        # steps then consist of its kernels, and the frame's device time is the sum over them
        level_by_level = bool(timed_stats.get("wavefront_items", 0))
        kernel = "trace_kernel" if args.config in ("c1", "c2") else ("wf_walk + wf_hit + wf_shade + wf_combine (level-by-level tracer)" if level_by_level else "mesh_trace_kernel")
        roofline.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": kernel + " (+ resolve_kernel) of one frame",
                         "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_flops_per_launch": aflops, "algorithmic_bytes_per_launch": abytes,
                         "algorithmic_tflops": round(tflops, 3), "counted_spp": count_spp})
        roofline.update(memory)
        # the algorithmic bytes (nodes, primitives, hit records) are served by the scalar cache, L1 and L2, not by HBM: above the
        # HBM peak the figure can only be a cache-level rate, and it is named so
        roofline["algorithmic_cache_level_gbs" if gbs > PEAK_HBM_GBS or args.config in ("c1", "c2") else "algorithmic_gbs"] = round(gbs, 2)
        multi = None
        if n_gpus > 1 and per_device:
            pd = np.array(per_device, dtype=np.float64)          # steps x devices
            mean_pd = pd.mean(axis=0)
            multi = {"scaling": scaling, "rccl_ranks": info["rccl_ranks"] if in_library else (world if world > 1 and backend == "nccl" else 0),
                     "gather_ms": round(float(np.mean(gather_ms)), 4), "assemble_ms": round(float(np.mean(assemble_ms)), 4),
                     "per_device_kernel_ms": [round(float(v), 4) for v in mean_pd],
                     "tail_ms": round(float(np.mean(pd.max(axis=1) - np.median(pd, axis=1))), 4),
                     "measured_over": "the timed steps" if in_library else "one more step after the timed ones (a barrier before the exchange)",
                     "gather_ms_is": ("from the end of the first device's own tiles to the arrival of all tiles there: transfers + the wait for the slowest device"
                                      if in_library else ("the %s gather to rank 0 after a barrier: transfers alone" % ("RCCL" if backend == "nccl" else "gloo (through host memory)")))}
        how = ("single frame buffer" if n_gpus == 1 else
               ("one process, one context over %d devices %s: 32x32 tiles interleaved, %s, assembled on device %d" % (
                   n_gpus, info["devices"], "grouped ncclSend/ncclRecv (communicator of %d ranks)" % info["rccl_ranks"] if info["gather"] == srt.GATHER_RCCL
                   else "peer copies", info["devices"][0]) if in_library else
                ("one process per GPU: 32x32 tiles interleaved over ranks + RCCL gather to rank 0" if backend == "nccl" else
                 "one process per GPU: 32x32 tiles interleaved over ranks + gloo gather through host memory (REHEARSAL, ranks may share GPUs)")))
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": "Mrays/sec (primary+bounce)", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_step_device_only": round(avg_ms, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, %dx%d, %s, depth %d; %s; the step ends with the frame in pinned host memory%s" % (
                           args.config, what, width, height,
                           ("%d spp per GPU (%d total)" % (spp_per_gpu, spp)) if scaling == "weak" else ("%d spp, one frame shared out over the GPUs" % spp), depth, how,
                           "; REDUCED %s against the named configuration" % "+".join(reduced) if reduced else ""),
                       "name": args.config, "width": width, "height": height, "spp": spp, "max_depth": depth, "seed": 1,
                       "devices": info["devices"] if in_library else [device_index], "rccl_ranks": info["rccl_ranks"] if in_library else (world if world > 1 and backend == "nccl" else 0),
                       "path_rays_per_step": total_rays / args.steps, "shadow_rays_per_step": total_shadow / args.steps,
                       "scene_commit_s": round(commit_s, 3), "bvh_nodes": cst["bvh_nodes"], "surfaces": cst["n_prims"],
                       "walk_tune": cst["walk_tune"], "tracer": "level-by-level" if level_by_level else "one kernel",
                       "node_records_per_ray": round(timed_stats.get("wavefront_nodes", 0) / max(1, timed_stats.get("rays_path", 1)), 3) if level_by_level else None,
                       "chunks_given_up": int(timed_stats.get("wavefront_given_up", 0)),
                       # path rays (counted in path_rays_per_step like every GetRayColor call) whose closest-hit query was the query of
                       # the call before them -- the metalness and the reflectivity pass follow one ray -- and was not walked twice
                       "path_rays_sharing_a_query": int(timed_stats.get("wavefront_twins", 0))},
            "rays_per_step": total_rays / args.steps, "shadow_rays_per_step": total_shadow / args.steps,
            "traversals_per_sec_M": round((total_rays + total_shadow) / elapsed / 1e6, 2),
            "roofline": roofline,
        }
        if multi is not None:
            line["multi_gpu"] = multi
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, scene_world, lights, cam, width, height, spp, depth)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
