#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing hot path.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one Pt_Trace call (rendering/path_tracer.c:2557-2585): one sample per pixel of the whole
image through the wavefront integrator (raygen -> {traverse, shade, traverse, shade} x bounces ->
resolve), scene and BVH resident in HBM before the timed region starts.

Workloads (BASELINE.json configs):
  cornell_1080p   (default) config #2: cornell_box, 1920x1080, 8-bounce clamp, pt_test camera
  cornell_360p    config #1 shape (640x360, 4 bounces) -- the CPU-runnable case
  synth1m_4k      config #3: synthetic 1M-triangle scene, 3840x2160, 8 bounces
  synth1m_1080p   the same scene at 1920x1080

Multi-GPU (N > 1, one process per GPU under torch.distributed.run, RCCL): the scene is replicated,
samples are sharded by index (rank r renders samples r, r+N, ...: per-GPU work is fixed, "weak"), and
the ONE exchange step of the path -- the sum of the per-GPU mean images onto rank 0 -- is an RCCL
reduce inside the timed region.  value = rays traced by all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pim_amd import api, scenes  # noqa: E402
from pim_amd import dist as pdist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    "cornell_1080p": dict(scene="cornell", width=1920, height=1080, bounces=8),
    "cornell_360p": dict(scene="cornell", width=640, height=360, bounces=4),
    "synth1m_4k": dict(scene="synth1m", width=3840, height=2160, bounces=8),
    "synth1m_1080p": dict(scene="synth1m", width=1920, height=1080, bounces=8),
}


def build_scene(kind):
    if kind == "cornell":
        return scenes.cornell_box(), scenes.pt_test_camera()
    if kind == "synth1m":
        sc = scenes.synthetic_scene(target_tris=1_000_000, seed=1234, extent=100.0, emissive_attempts=64)
        return sc, scenes.synthetic_camera(100.0)
    raise ValueError(kind)


def cpu_baseline(sc, cam, wl, threads):
    """TEST-INFRASTRUCTURE leg: the CPU oracle (oracle/liborc_p.so, a C restatement of the reference's
    path; Embree itself is not in the reference snapshot) timed on the host cores on a bounded sample:
    whole frames of the same workload until ~10 s have passed (at least one)."""
    from tests.util import OracleScene
    o = OracleScene(sc.desc, accel=1, threads=threads)
    W, H = wl["width"], wl["height"]
    if sc.tri_count > 200000:          # keep the sample bounded on the big scene: a 640x360 crop of the same camera
        W, H = 640, 360
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dof = scenes.default_dof()
    rays = 0
    frames = 0
    t0 = time.perf_counter()
    while True:
        st = o.trace_image(img, W, H, cam, dof, scenes.trace_params(frames, seed=1, max_bounces=wl["bounces"]))
        rays += st.extensionRays + st.shadowRays + st.otherRays
        frames += 1
        if time.perf_counter() - t0 > 10.0 or frames >= 64:
            break
    dt = time.perf_counter() - t0
    o.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} frame(s) of {W}x{H}, {wl['bounces']} bounces, CPU oracle with binned-SAH BVH2 "
                      f"(Embree binary absent from the reference snapshot), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="cornell_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    api.init(local_rank if world > 1 else 0)
    sc, cam = build_scene(wl["scene"])
    scene = api.Scene(sc.desc)
    info = scene.info()
    W, H = wl["width"], wl["height"]
    trace = api.Trace(W, H)
    dof = scenes.default_dof()
    reduce_buf = None
    if world > 1:
        reduce_buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")

    def step(i):
        # rank r renders samples r, r + world, ...; sampleWeight makes each rank's image its own running mean
        p = scenes.trace_params(i, seed=1, max_bounces=wl["bounces"])
        p.sampleIndex = pdist.sample_indices(rank, world, i + 1)[i]
        trace.trace(scene, cam, dof, p)
        s = trace.stats()
        return s.extensionRays + s.shadowRays + s.otherRays, s.traceMs, s.totalMs, s.tailRays, 2 * s.bounces

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    trace.clear()
    sync()
    t0 = time.perf_counter()
    rays = 0
    tail_rays = 0        # traced inside the tail kernel (deep bounces), not by the k_trace launches that trace_ms covers
    launches = 0
    trace_ms = 0.0
    total_ms = 0.0
    for i in range(args.steps):
        r, tm, tot, tr, nl = step(i)
        rays += r; trace_ms += tm; total_ms += tot; tail_rays += tr; launches += nl
    if world > 1:
        # the path's one exchange step: sum of the per-GPU mean images on rank 0 (RCCL reduce over xGMI)
        trace.copy_dev(reduce_buf.data_ptr())
        pdist.reduce_mean_image(reduce_buf, world, dist, dst=0)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, float(rays)], dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); rays_all = float(tsum[1])
    else:
        rays_all = float(rays)

    if rank == 0:
        mean, stddev = trace.stddev()
        # roofline of the dominant kernel (the closest-hit traversal, 2 launches per bounce): algorithmic bytes
        # per ray = 32 (ray) + 16 (hit) + 64 per BVH4 node fetched (the 64-byte compressed node) + 48 per
        # triangle tested, with the per-ray node / triangle counts MEASURED by the counting build of the kernel
        # on one more frame of the same workload (outside the timed region)
        api.work_counters(True)
        step(args.steps)
        nodes, tris, crays, worst = api.work_counters(False)
        crays = max(crays, 1)
        b_ray = 48.0 + 64.0 * nodes / crays + 48.0 * tris / crays
        batched = rays - tail_rays          # rays of the k_trace launches
        launches = max(launches, 1)
        achieved = batched * b_ray / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        # HBM-side bytes per traversal launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
        # rocprofv3 runs of this same command: scripts/profile_bench.sh -> profiles/traffic.json); counters cannot
        # be read inside this process, so the figure is the committed measurement of this workload, or null
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = (json.load(open(tpath)).get(args.workload) or {}).get("bytes_per_launch")
        out = {
            "metric": "Mrays/sec (incoherent, per GPU + node) and spp/sec at 1080p; % HBM BW peak",
            "value": rays_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene_tris": info.triCount, "width": W, "height": H,
                       "max_bounces": wl["bounces"], "spp_per_step": 1, "sharding": "sample index" if world > 1 else "none"},
            "spp_per_s": args.steps * world / dt,
            "paths_per_s": args.steps * world * W * H / dt,
            "rays_per_path": rays / (args.steps * W * H),
            "image_mean_luminance": mean, "pt_stddev": stddev,
            "bvh_build_ms": info.bvhBuildMs, "scene_setup_ms": info.sceneSetupMs,
            "device_ms_per_step": total_ms / args.steps, "traversal_ms_per_step": trace_ms / args.steps,
            "roofline": {"bound": "hbm", "kernel": "k_trace<closest>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": batched / launches * b_ray,
                         "bytes_per_ray": b_ray, "nodes_per_ray": nodes / crays, "tris_per_ray": tris / crays, "max_nodes_one_ray": worst,
                         "rays_per_launch": batched / launches, "launches_per_step": launches / args.steps,
                         "avg_launch_ms": trace_ms / launches, "tail_kernel_rays_per_step": tail_rays / args.steps},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(sc, cam, wl, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    trace.close(); scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    api.shutdown()


if __name__ == "__main__":
    main()
