#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing hot path.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--total-spp S]

A "step" is one Pt_Trace call (rendering/path_tracer.c:2557-2585): one sample per pixel of the whole
image through the wavefront integrator (raygen -> {traverse, shade, traverse, shade} x bounces -> tail ->
resolve), scene and BVH resident in HBM before the timed region starts.

Workloads (BASELINE.json configs):
  synth1m_4k      (default) config #3 on one GPU: synthetic 1M-triangle scene, 3840x2160, 8 bounces
  synth1m_1080p   the same scene at 1920x1080
  cornell_1080p   config #2: cornell_box, 1920x1080, 8-bounce clamp, pt_test camera
  cornell_360p    config #1 shape (640x360, 4 bounces) -- the CPU-runnable case

At N = 1 the line also carries, as sub-records ("other_workloads"), cornell_1080p, synth1m_1080p and the north
star's incoherent-ray microbenchmark (2^26 shuffled surface rays into the 1M-triangle tree, one
closest-hit launch); --no-sub skips them.

Multi-GPU (N > 1, one process per GPU under torch.distributed.run, RCCL): the scene is replicated, samples are
sharded by index, and the ONE exchange step of the path -- the sum of the per-GPU mean images onto rank 0 -- is an
RCCL reduce inside the timed region.  Default: rank r renders samples r, r+N, ... for K steps each (per-GPU work fixed,
"weak").  --total-spp S (config 3's "64 spp sharded"): the S samples are split over the ranks, rank r renders
[r*S/N, (r+1)*S/N) ("strong"; --steps is then S/N per rank).  value = rays traced by all ranks / max-over-ranks time.

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
    "synth1m_4k": dict(scene="synth1m", width=3840, height=2160, bounces=8),
    "synth1m_1080p": dict(scene="synth1m", width=1920, height=1080, bounces=8),
    "cornell_1080p": dict(scene="cornell", width=1920, height=1080, bounces=8),
    "cornell_360p": dict(scene="cornell", width=640, height=360, bounces=4),
}
DEFAULT_WORKLOAD = "synth1m_4k"

# algorithmic bytes per unit of the shade stages (path state read + written, NEE ray / hit, queue entry)
B_SHADE_A = 260.0
B_SHADE_B = 340.0

_SCENES = {}


def build_scene(kind):
    """host arrays + GPU scene, built once per process"""
    if kind not in _SCENES:
        if kind == "cornell":
            sc, cam = scenes.cornell_box(), scenes.pt_test_camera()
        elif kind == "synth1m":
            sc = scenes.synthetic_scene(target_tris=1_000_000, seed=1234, extent=100.0, emissive_attempts=64)
            cam = scenes.synthetic_camera(100.0)
        else:
            raise ValueError(kind)
        _SCENES[kind] = (sc, cam, api.Scene(sc.desc))
    return _SCENES[kind]


def load_traffic(workload):
    """HBM-side bytes per k_trace launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 runs
    of this same command: scripts/profile_bench.sh -> profiles/traffic.json).  Counters cannot be read inside this
    process: the figure is the COMMITTED measurement of this workload (tagged as such), or None."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get(workload)


def cpu_baseline(sc, cam, wl, threads):
    """TEST-INFRASTRUCTURE leg: the CPU oracle (oracle/liborc_p.so, a C restatement of the reference's
    path; Embree itself is not in the reference snapshot) timed on the host cores on a bounded sample:
    whole frames of the same workload until ~10 s have passed (at least one)."""
    from tests.util import OracleScene
    o = OracleScene(sc.desc, accel=1, threads=threads)
    W, H = wl["width"], wl["height"]
    if sc.tri_count > 200000:          # keep the sample bounded on the big scene: a 640x360 image from the same camera
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
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"{frames} frame(s) of {W}x{H}, {wl['bounces']} bounces, same scene and camera, CPU oracle (-O2, portable libm, "
                      f"binned-SAH BVH2; Embree binary absent from the reference snapshot) on {threads} threads, {dt:.1f} s"}


class Run:
    """one workload on this rank: warm-up, K timed steps, statistics"""

    def __init__(self, name, rank=0, world=1):
        self.name, self.wl = name, WORKLOADS[name]
        self.rank, self.world = rank, world
        self.sc, self.cam, self.scene = build_scene(self.wl["scene"])
        self.W, self.H = self.wl["width"], self.wl["height"]
        self.trace = api.Trace(self.W, self.H)
        self.dof = scenes.default_dof()
        self.acc = {}

    def step(self, i, sample_index=None):
        p = scenes.trace_params(i, seed=1, max_bounces=self.wl["bounces"])
        p.sampleIndex = sample_index if sample_index is not None else pdist.sample_indices(self.rank, self.world, i + 1)[i]
        self.trace.trace(self.scene, self.cam, self.dof, p)
        return self.trace.stats()

    def add(self, s):
        a = self.acc
        for k in ("extensionRays", "shadowRays", "otherRays", "tailRays", "traceMs", "totalMs", "shadeAMs", "shadeBMs", "tailMs",
                  "traceLaunches", "wavefrontPathBounces", "tailPaths", "tailPathBounces"):
            a[k] = a.get(k, 0) + getattr(s, k)

    def work_per_ray(self, i):
        """per-ray node / triangle counts, MEASURED by the counting build of the traversal kernel on two more frames of the same
        workload (outside the timed region): one under the library's own tail policy -- what the batched launches of the timed
        steps traced -- and one with every bounce as a wavefront, i.e. ALL rays of a sample; the difference is what the tail
        kernel traced (the same rays, whoever walks them)"""
        api.work_counters(True)
        self.step(i)
        nb, tb, rb, worst = api.work_counters(False)
        pinned = os.environ.get("PIMGPU_TAIL_BOUNCE")
        os.environ["PIMGPU_TAIL_BOUNCE"] = "-1"
        try:
            api.work_counters(True)
            self.step(i)
            na, ta, ra, worst_all = api.work_counters(False)
        finally:
            if pinned is None: os.environ.pop("PIMGPU_TAIL_BOUNCE", None)
            else: os.environ["PIMGPU_TAIL_BOUNCE"] = pinned
        batched = (nb / max(rb, 1), tb / max(rb, 1), worst)
        rt = max(ra - rb, 1)
        tail = (max(na - nb, 0) / rt, max(ta - tb, 0) / rt, worst_all)
        return batched, tail

    def rooflines(self, steps, dt_ms_per_step):
        """one entry per kernel class with its share of the step.  `achieved` = ALGORITHMIC bytes / device time of the kernel's
        launches (events attached to the dispatches, on the library's stream)."""
        a = self.acc
        (nodes, tris, worst), (tnodes, ttris, tworst) = self.work_per_ray(steps)
        b_ray = 48.0 + 64.0 * nodes + 48.0 * tris
        b_ray_tail = 48.0 + 64.0 * tnodes + 48.0 * ttris
        rays = a["extensionRays"] + a["shadowRays"] + a["otherRays"]
        batched = rays - a["tailRays"]
        launches = max(a["traceLaunches"], 1)
        device_ms = a["totalMs"] / steps
        out = []

        def entry(kernel, ms_total, bytes_total, units, unit_name, bytes_per_unit, **extra):
            ach = bytes_total / (ms_total * 1e-3) / 1e9 if ms_total > 0 else 0.0
            e = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                 "ms_per_step": ms_total / steps, "share_of_device_step": (ms_total / steps) / device_ms if device_ms > 0 else 0.0,
                 "units_per_step": units / steps, "unit_name": unit_name, "algorithmic_bytes_per_unit": bytes_per_unit}
            e.update(extra)
            return e

        trw = load_traffic(self.name) or {}
        per_kernel = trw.get("kernels") or ({trw["kernel"]: trw} if trw.get("kernel") else {})

        def attach_traffic(e, label, launch_ms):
            """HBM-side bytes per launch of this kernel class from the committed PMC passes (not this run), and the DRAM-side rate they give"""
            tr = per_kernel.get(label)
            e["traffic"] = tr.get("bytes_per_launch") if tr else None
            if tr and launch_ms > 0:
                e["hbm_measured"] = {"bytes_per_launch": tr["bytes_per_launch"], "GBps": tr["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9,
                                     "frac_of_peak": tr["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "l2_hit_rate": tr.get("l2_hit_rate"),
                                     "source": f"profiles/traffic.json [{tr.get('tag')}]: rocprofv3 --pmc passes of `{tr.get('command')}`, NOT measured in this run"}
                e["traffic_source"] = "profiles/traffic.json (committed PMC measurement of this workload, not this run)"

        avg_launch_ms = a["traceMs"] / launches
        t = entry("k_trace<closest>", a["traceMs"], batched * b_ray, batched, "ray", b_ray,
                  nodes_per_ray=nodes, tris_per_ray=tris, max_nodes_one_ray=worst, rays_per_launch=batched / launches,
                  launches_per_step=launches / steps, avg_launch_ms=avg_launch_ms, bytes_per_launch=batched / launches * b_ray)
        # the honest ceiling: the algorithmic bytes are mostly cache hits (the tree lives in L2 / the Infinity Cache), so the
        # HBM-side rate is the measured one, and what binds the kernel is the vector L1's lookup rate and VALU issue
        t["algorithmic_note"] = ("achieved/frac count 64 B per node visit and 48 B per triangle test as if each came from HBM (SURVEY 8d's "
                                 "contract); most are L2 / Infinity Cache hits -- see hbm_measured for the DRAM-side rate")
        attach_traffic(t, "k_trace<closest>", avg_launch_ms if a["traceMs"] > 0 and batched > launches else 0.0)
        t["binding_resource"] = "vector-L1 (TCP) lookups and VALU issue at partial lane utilisation, not HBM bytes (profiles/*_pmc_summary.txt)"
        out.append(t)
        wpb = a["wavefrontPathBounces"]
        out.append(entry("k_shade_a<FAST+SLOW>", a["shadeAMs"], wpb * B_SHADE_A, wpb, "path-bounce", B_SHADE_A))
        out.append(entry("k_shade_b<FAST+SLOW>", a["shadeBMs"], wpb * B_SHADE_B, wpb, "path-bounce", B_SHADE_B))
        tail_bytes = a["tailRays"] * b_ray_tail + a["tailPathBounces"] * (B_SHADE_A + B_SHADE_B)
        out.append(entry("k_tail", a["tailMs"], tail_bytes, a["tailRays"], "ray (+ its share of the bounce's shading)",
                         tail_bytes / max(a["tailRays"], 1), paths_per_step=a["tailPaths"] / steps,
                         nodes_per_ray=tnodes, tris_per_ray=ttris,
                         binding_resource="VALU issue (about 80 % busy) at about 29 of 64 lanes per instruction (profiles/*_pmc_summary.txt)",
                         note="one lane per path, the two rays of a bounce walked by the whole wave (DESIGN.md section 4); nodes / triangles per ray: "
                              "the counting build's figures for all rays of a sample minus those of the batched launches"))
        attach_traffic(out[-1], "k_tail", a["tailMs"] / steps)
        return out

    def close(self):
        self.trace.close()


def run_single(name, steps, warmup):
    """a sub-record: the workload on one GPU, no collective"""
    r = Run(name)
    for i in range(warmup):
        r.step(i)
    r.trace.clear()
    t0 = time.perf_counter()
    for i in range(steps):
        r.add(r.step(i))
    dt = time.perf_counter() - t0
    a = r.acc
    rays = a["extensionRays"] + a["shadowRays"] + a["otherRays"]
    mean, stddev = r.trace.stddev()
    rec = {"workload": name, "scene_tris": r.sc.tri_count, "width": r.W, "height": r.H, "max_bounces": r.wl["bounces"], "steps": steps,
           "warmup": warmup, "value": rays / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt * 1e3 / steps, "spp_per_s": steps / dt,
           "rays_per_path": rays / (steps * r.W * r.H), "device_ms_per_step": a["totalMs"] / steps,
           "image_mean_luminance": mean, "pt_stddev": stddev, "rooflines": r.rooflines(steps, dt * 1e3 / steps)}
    r.close()
    return rec


def incoherent_microbench(log2_rays=26, reps=4):
    """incoherent closest-hit rays -- origins uniformly on random triangles of the 1M-triangle scene (+1e-4 N),
    directions uniform in the hemisphere of the geometric normal, in random order; 2^26 rays, ONE launch of the traversal
    kernel through the C-ABI (pimgpu_intersect_packed_dev), rays and hits resident in HBM.  The rays are generated on the
    device (torch is used for device memory and random numbers only)."""
    import torch
    sc, cam, scene = build_scene("synth1m")
    lib = api.load_library()
    dev = torch.device("cuda", 0)
    n = 1 << log2_rays
    g = torch.Generator(device=dev); g.manual_seed(1234)
    pos = torch.from_numpy(np.ascontiguousarray(sc.positions[:, :3]).reshape(-1, 3, 3)).to(dev)
    rayO = torch.empty((n, 4), dtype=torch.float32, device=dev)
    rayD = torch.empty((n, 4), dtype=torch.float32, device=dev)
    chunk = 1 << 23
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        tri = torch.randint(0, pos.shape[0], (m,), device=dev, generator=g)
        P = pos[tri]
        w = -torch.log(torch.rand((m, 3), device=dev, generator=g).clamp_min(1e-12))
        w = w / w.sum(1, keepdim=True)
        p = (P * w[:, :, None]).sum(1)
        nrm = torch.linalg.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
        nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-20)
        d = torch.randn((m, 3), device=dev, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        d = torch.where((d * nrm).sum(1, keepdim=True) < 0, -d, d)
        rayO[b:b + m, :3] = p + nrm * 1e-4; rayO[b:b + m, 3] = 0.0
        rayD[b:b + m, :3] = d; rayD[b:b + m, 3] = api.RCP_EPSILON
        del tri, P, w, p, nrm, d
    hits = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def launch():
        api.check(lib.pimgpu_intersect_packed_dev(scene.handle, ctypes.c_void_p(rayO.data_ptr()), ctypes.c_void_p(rayD.data_ptr()),
                                                  ctypes.c_int32(n), ctypes.c_void_p(hits.data_ptr())), "pimgpu_intersect_packed_dev")
    launch()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()                       # synchronous: returns when the hits are in HBM
    dt = (time.perf_counter() - t0) / reps
    api.work_counters(True); launch(); nodes, tris, rays, worst = api.work_counters(False)
    rays = max(rays, 1)
    b_ray = 48.0 + 64.0 * nodes / rays + 48.0 * tris / rays
    ach = n * b_ray / dt / 1e9
    rec = {"workload": "incoherent_closest_hit_1m", "scene_tris": sc.tri_count, "rays_per_launch": n, "launches_timed": reps,
           "ms_per_launch": dt * 1e3, "value": n / dt / 1e6, "unit": "Mrays/s", "north_star_target_Mrays_per_s": 2000.0,
           "hit_fraction": float((hits[:, 0] >= 0).float().mean()),
           "roofline": {"kernel": "k_trace<closest>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_ray": b_ray, "nodes_per_ray": nodes / rays, "tris_per_ray": tris / rays, "max_nodes_one_ray": worst,
                        "traffic": None,
                        "algorithmic_note": "algorithmic bytes (64 B per node visit, 48 B per triangle test); the 67 MB tree is served from L2 / the "
                                            "Infinity Cache, so a fraction near or above 1 is a cache-bandwidth statement, not DRAM traffic"}}
    del rayO, rayD, hits, pos
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--total-spp", type=int, default=0, help="strong scaling: this many samples in total, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub-records (other workloads, microbenchmark)")
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

    if world == 1 and not args.no_sub:
        # the microbenchmark leg keeps its rays in torch tensors; torch's HIP runtime has to come up BEFORE libpimgpu's
        # (the other order leaves torch without a device in this process)
        import torch
        torch.cuda.init()
        torch.zeros(1, device="cuda")
    api.init(local_rank if world > 1 else 0)
    run = Run(args.workload, rank, world)
    info = run.scene.info()
    W, H = run.W, run.H
    strong = args.total_spp > 0
    if strong:
        lo, hi = pdist.sample_range(rank, world, args.total_spp)
        my_samples = list(range(lo, hi))
        steps = len(my_samples)
    else:
        steps = args.steps
        my_samples = [pdist.sample_indices(rank, world, i + 1)[i] for i in range(steps)]
    reduce_buf = None
    if world > 1:
        reduce_buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        run.step(i)
    run.trace.clear()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        run.add(run.step(i, my_samples[i]))         # sampleWeight 1/(i+1): this rank's image is the running mean of ITS samples
    if world > 1:
        # the path's one exchange step: sum of the per-GPU mean images on rank 0 (RCCL reduce over xGMI), weighted by the
        # number of samples each rank rendered
        run.trace.copy_dev(reduce_buf.data_ptr())
        pdist.reduce_mean_image(reduce_buf, world, dist, dst=0, weight=steps, total=args.total_spp if strong else steps * world)
    sync()
    dt = time.perf_counter() - t0
    a = run.acc
    rays = a["extensionRays"] + a["shadowRays"] + a["otherRays"] if a else 0
    total_steps = steps
    if world > 1:
        t = torch.tensor([dt, float(rays), float(steps)], dtype=torch.float64, device="cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); rays_all = float(tsum[1]); total_steps = int(tsum[2])
    else:
        rays_all = float(rays)

    if rank == 0:
        mean, stddev = run.trace.stddev()
        steps_r0 = max(steps, 1)
        roofs = run.rooflines(steps_r0, dt * 1e3 / steps_r0)
        dominant = max(roofs, key=lambda e: e["ms_per_step"])
        # the contract's `roofline` object is the dominant kernel's entry; `rooflines` has every kernel class of the step
        out = {
            "metric": "Mrays/sec (incoherent, per GPU + node) and spp/sec at 1080p; % HBM BW peak",
            "value": rays_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / steps_r0, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene_tris": info.triCount, "width": W, "height": H,
                       "max_bounces": wl["bounces"], "spp_per_step": 1,
                       "sharding": ("sample ranges of --total-spp %d" % args.total_spp if strong else "sample index, interleaved") if world > 1 else "none"},
            "spp_per_s": total_steps / dt,
            "paths_per_s": total_steps * W * H / dt,
            "rays_per_path": rays / (steps_r0 * W * H),
            "image_mean_luminance": mean, "pt_stddev": stddev,
            "bvh_build_ms": info.bvhBuildMs, "scene_setup_ms": info.sceneSetupMs,
            "device_ms_per_step": a["totalMs"] / steps_r0, "traversal_ms_per_step": a["traceMs"] / steps_r0,
            "roofline": dominant, "rooflines": roofs,
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)        # the box gives a one-GPU job 16 cores, whatever cpu_count says
            out["cpu_baseline"] = cpu_baseline(run.sc, run.cam, wl, threads)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_sub:
            subs = []
            for name in ("cornell_1080p", "synth1m_1080p"):
                if name != args.workload:
                    subs.append(run_single(name, 16, 3))
            try:
                subs.append(incoherent_microbench())
            except Exception as e:                      # a sub-record must not lose the headline line
                subs.append({"workload": "incoherent_closest_hit_1m", "error": repr(e)})
            out["other_workloads"] = subs
        print(json.dumps(out), flush=True)

    run.close()
    for sc_, cam_, scene_ in _SCENES.values():
        scene_.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    api.shutdown()


if __name__ == "__main__":
    main()
