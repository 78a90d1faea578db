#!/usr/bin/env python3
"""Benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one frame of the headline workload -- the default Game1 scene (Game1.cs:96-144)
at 1920x1080, depth 8, traced as the reference does (GetRayColor on the BVHNode world, Game1.cs:293) -- with 16
samples per pixel PER GPU (weak scaling: N GPUs render 16*N spp; the frame is split into 32x32 screen tiles
interleaved over the ranks, each rank traces its tiles, the tile buffers are gathered on rank 0 over RCCL and
scattered into the frame by a kernel).  The scene, camera and framebuffers are resident in HBM before the timed
region; the frame stays on the device (the PCIe copy of a 1080p frame is 8.3 MB).

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
from simple_raytracer_amd import game1  # noqa: E402

WIDTH, HEIGHT, SPP_PER_GPU, MAX_DEPTH, TILE = 1920, 1080, 16, 8, 32
PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 (= f32 MFMA) peak
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E peak

# Memory-side bytes per launch of the trace kernel on the N=1 workload, from separate `rocprofv3 --pmc FETCH_SIZE` and
# `--pmc WRITE_SIZE` passes over this command (tools/pmc_traffic.sh; the counters are in KiB, and FETCH_SIZE is
# doubled, the correction for gfx950).  They are not re-measured by a normal run: a counter pass needs the profiler.
# Almost all of it is private-memory traffic of the ray-tree frames below the levels kept in LDS, plus 0.4 GB of samples.
MEASURED_TRAFFIC_KIB = {"bvhnode": (3106033.2, 12299515.6)}   # (FETCH_SIZE, WRITE_SIZE) of trace_kernel<true,false,true>

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


def algorithmic_work(stats, pixels):
    ev = dict(stats)
    ev["pixels"] = pixels
    b = sum(ev[k] * c[0] for k, c in EVENT_COST.items())
    f = sum(ev[k] * c[1] for k, c in EVENT_COST.items())
    return b, f


def cpu_baseline(width, height, spp, depth):
    """The CPU oracle (a C++ restatement of the reference's algorithm, threaded by row bands like Game1.cs:376-380)
    timed on this host on one frame of the same workload (about 20 s on 16 cores)."""
    from oracle import orc   # checker / baseline only
    threads = min(16, os.cpu_count() or 1)
    win = (0, 0, width, height)
    sc, cam = orc.default_scene(), orc.default_camera(width, height)
    _, st = sc.render(cam, width, height, samples=spp, max_depth=depth, seed=1, window=win, threads=threads)
    return {"value": round(st["rays_path"] / st["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "one whole %dx%d %d-spp depth-%d frame (all rows, row bands over %d threads), %d rays in %.2f s" % (
                width, height, spp, depth, threads, st["rays_path"], st["seconds"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--world-kind", choices=["bvhnode", "surfacelist"], default="bvhnode")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
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

    spp = SPP_PER_GPU * world
    r = srt.Renderer(device=device_index)
    scene_world, lights, cam = game1.default_scene(WIDTH, HEIGHT)
    kind = srt.WORLD_BVHNODE if args.world_kind == "bvhnode" else srt.WORLD_SURFACELIST
    r.set_scene(scene_world, lights, cam, world_kind=kind)

    stream = torch.cuda.current_stream().cuda_stream
    tf = None
    if world == 1:
        p = r.params(WIDTH, HEIGHT, samples=spp, max_depth=MAX_DEPTH, seed=1)
        local = r.local_pixels(p)
        out = torch.zeros(local, dtype=torch.int32, device="cuda")
    else:
        from simple_raytracer_amd.distributed import TiledFrame
        tf = TiledFrame(r, WIDTH, HEIGHT, rank=rank, world_size=world, tile=TILE, device="cuda", samples=spp, max_depth=MAX_DEPTH, seed=1)
        p, local = tf.params, tf.local_pixels

    def step(events=None):
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if tf is None:
            r.render_device(p, out.data_ptr(), out.numel(), stream)
        else:
            tf.render_local()
        if events is not None:
            e1.record()
            events.append((e0, e1))
        if tf is not None:
            tf.gather()      # RCCL gather of the tile buffers to rank 0
            tf.assemble()    # rank 0: tiles -> frame

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    r.stats()          # drop the warmup's counters
    events = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()     # counters accumulated over the timed launches
    launch_ms = [a.elapsed_time(b) for a, b in events]

    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    rays = torch.tensor([float(st["rays_path"]), float(st["rays_shadow"])], dtype=torch.float64, device=reduce_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays, total_shadow = float(rays[0].item()), float(rays[1].item())

    if rank == 0:
        # algorithmic work of ONE launch of this rank's trace kernel, from the counting build of the same kernel
        host = np.zeros(local, dtype=np.uint32)
        import ctypes as C
        r._check(r._lib.srt_render_counted(r._ctx, C.byref(p), host.ctypes.data_as(C.c_void_p)))
        cst = r.stats()
        px = WIDTH * HEIGHT if world == 1 else local
        abytes, aflops = algorithmic_work(cst, px)
        avg_ms = float(np.mean(launch_ms))
        tflops = aflops / (avg_ms * 1e-3) / 1e12
        gbs = abytes / (avg_ms * 1e-3) / 1e9
        # Which roof applies is decided by where the scene lives: geometry that fits the scalar / L1 caches (the Game1
        # scene is 5 nodes + 5 spheres) is never re-fetched from HBM, so its "algorithmic bytes" are cache hits and
        # the kernel is bound by vector FP32 issue; a BVH that exceeds the 32 MiB of L2 is bound by memory.
        scene_bytes = cst["bvh_nodes"] * 32 + cst["n_prims"] * 48
        valu_bound = scene_bytes < (1 << 20) or tflops / PEAK_FP32_TFLOPS >= gbs / PEAK_HBM_GBS
        roofline = ({"bound": "valu_fp32", "achieved": round(tflops, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tflops / PEAK_FP32_TFLOPS, 4)} if valu_bound else
                    {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
        traffic = None
        if world == 1 and args.world_kind in MEASURED_TRAFFIC_KIB:
            fetch_kib, write_kib = MEASURED_TRAFFIC_KIB[args.world_kind]
            traffic = int((2.0 * fetch_kib + write_kib) * 1024)
        roofline.update({"traffic": traffic, "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, 2*FETCH+WRITE "
                         "(profiles/r1_v6_pmc_traffic.txt); constant, not re-measured in this run", "kernel": "trace_kernel", "avg_launch_ms": round(avg_ms, 4),
                         "algorithmic_flops_per_launch": aflops, "algorithmic_bytes_per_launch": abytes,
                         "algorithmic_gbs": round(gbs, 2), "algorithmic_tflops": round(tflops, 3)})
        line = {
            "metric": "Mrays/sec (primary+bounce)", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "default Game1 scene (5 spheres, 2 point lights), %dx%d, %d spp per GPU (%d total), depth %d, %s world, "
                                   "%s" % (WIDTH, HEIGHT, SPP_PER_GPU, spp, MAX_DEPTH, args.world_kind,
                                           "single frame buffer" if world == 1 else
                                           ("32x32 tiles interleaved over ranks + RCCL gather to rank 0" if backend == "nccl" else
                                            "32x32 tiles interleaved over ranks + gloo gather through host memory (REHEARSAL, ranks may share GPUs)")),
                       "width": WIDTH, "height": HEIGHT, "spp": spp, "max_depth": MAX_DEPTH, "seed": 1},
            "rays_per_step": total_rays / args.steps, "shadow_rays_per_step": total_shadow / args.steps,
            "traversals_per_sec_M": round((total_rays + total_shadow) / elapsed / 1e6, 2),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(WIDTH, HEIGHT, spp, MAX_DEPTH)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
