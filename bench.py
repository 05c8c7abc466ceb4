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
  synth20m_1080p  config #5 on one GPU: 20M-triangle diffuse scene, the BVH rebuilt inside every timed step

At N = 1 the line also carries, as sub-records ("other_workloads"), cornell_1080p, synth1m_1080p, synth20m_1080p and the
north star's incoherent-ray microbenchmark (shuffled surface rays in one closest-hit launch) on the 1M-triangle
tree (2^26 rays) and on the 20M-triangle tree (2^24 rays), and the default workload with two whole frames in flight on the one
GPU (pimgpu_trace_image_samples over the device list [0, 0], run in a child process); --no-sub skips them, PIM_BENCH_SKIP_20M=1
skips the 20M pair.
--plain: no counting-build frame and no sub-records (what the profiler scripts run); --microbench-only: the two
microbenchmarks alone.

Every kernel entry of "rooflines" has three ceilings (valu issue / HBM-side bytes / algorithmic bytes against L2) and names
the binding one in "bound"; the instruction and byte counts per unit come from the committed PMC runs (profiles/traffic.json,
scripts/profile_bench.sh) and are scaled by this run's units and device time -- DESIGN.md section 5.

Multi-GPU (N > 1, one process per GPU under torch.distributed.run, RCCL): the scene is replicated, samples are
sharded by index, and the exchange steps of the path -- the sum of the per-GPU mean images onto rank 0 and, after every
step, the SUM all-reduce of the ranks' light-histogram increments (so that every rank samples lights as one process that
had traced all samples would) -- are RCCL collectives inside the timed region.  Default: rank r renders samples r, r+N, ... for K steps each (per-GPU work fixed,
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

# Peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # HBM3E, spec
L2_PEAK_GBS = 34500.0             # aggregate L2 bandwidth (8 XCDs x 4 MiB), "L2 (per XCD)"
IC_GATHER_PEAK_GBS = 8600.0       # rows gathered at random from a 38 MB table that lives in the Infinity Cache, "Indexed rows"
# VALU issue: the guide's fp32 vector peak (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz) is one wave64
# instruction per SIMD every 2 cycles: 1228.8 G wave-instructions/s.  What this pool's chips sustain with every SIMD full of
# independent instructions and no memory traffic (scripts/micro/valu_peak.hip -> profiles/r03_valu_issue_peak.txt; the counter
# SQ_INSTS_VALU counts exactly one per wave64 instruction, profiles/r03_valu_issue_peak_pmc_calibration.txt): 797 G/s on
# v_fma_f32 alone, 490-520 G/s on v_max3_f32 / v_cvt_f32_ubyte0 -- reported beside the spec ceiling as `attainable`.
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
VALU_ATTAINABLE_GINST = 796.6
# ... and the opcodes fall into three issue classes there (scripts/micro/valu_classes.hip -> profiles/r03_valu_instruction_classes.txt):
# ~3 cycles per wave64 instruction per SIMD (mov, and/or/xor, right shifts, add/sub/mul/fma f32, add/sub u32), ~4.7 (min/max, compares,
# selects, conversions, left shifts, 3-operand integer ops, packed fp32, fp64) and ~8.9 (rcp, sqrt, ...).  scripts/valu_mix.py prices
# each kernel's STATIC mix of loop instructions with those costs -> profiles/valu_mix.json: the rate a kernel made of these opcodes
# can reach when its SIMDs issue VALU instructions back to back, reported as `attainable_for_mix`.


def load_valu_mix():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    except (OSError, ValueError):
        return {}


def add_mix(ceiling, kernel_class):
    """the static-mix estimate of scripts/valu_mix.py beside a kernel's measured VALU issue rate"""
    m = load_valu_mix().get(kernel_class)
    if m:
        ceiling["attainable_for_mix"] = m["attainable_Ginst_per_s"]
        ceiling["frac_of_mix_attainable"] = ceiling["achieved"] / m["attainable_Ginst_per_s"]
        ceiling["mix"] = {"fast": m["fast_fraction"], "normal": m["normal_fraction"], "transcendental": m["transcendental_fraction"],
                          "source": "profiles/valu_mix.json (static mix of the loop instructions; an estimate: a run whose dynamic mix is lighter can exceed it)"}
    return ceiling

WORKLOADS = {
    "synth1m_4k": dict(scene="synth1m", width=3840, height=2160, bounces=8),
    "synth1m_1080p": dict(scene="synth1m", width=1920, height=1080, bounces=8),
    "cornell_1080p": dict(scene="cornell", width=1920, height=1080, bounces=8),
    "cornell_360p": dict(scene="cornell", width=640, height=360, bounces=4),
    # config #5 on one GPU: the 20M-triangle diffuse scene, FULL BVH rebuild + 1 spp per step (the HBM-bound case: 2.9 GB of
    # vertices + 1.7 GB of tree do not fit the 256 MiB Infinity Cache)
    "synth20m_1080p": dict(scene="synth20m", width=1920, height=1080, bounces=8, rebuild=True),
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
        elif kind == "synth20m":
            sc = scenes.synthetic_scene(target_tris=20_000_000, seed=1234, extent=100.0, emissive_attempts=4, diffuse_only=True)
            cam = scenes.synthetic_camera(100.0)
        else:
            raise ValueError(kind)
        _SCENES[kind] = (sc, cam, api.Scene(sc.desc))
    return _SCENES[kind]


def load_profile(workload):
    """per-kernel-class figures of this workload from the committed rocprofv3 --pmc passes (scripts/profile_bench.sh ->
    scripts/pmc_per_unit.py -> profiles/traffic.json): VALU wave-instructions, HBM-side bytes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE), vector-L1 accesses per unit of work, lanes per VALU instruction, L2 hit rate.  Counters cannot be read inside
    this process: these are the COMMITTED measurements of the same command (tagged with their source), or nothing."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}
    return (json.load(open(tpath)).get(workload) or {}).get("kernels") or {}


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
        self.build_ms = 0.0

    def step(self, i, sample_index=None, update_dists=1):
        p = scenes.trace_params(i, seed=1, max_bounces=self.wl["bounces"], update_dists=update_dists)
        p.sampleIndex = sample_index if sample_index is not None else pdist.sample_indices(self.rank, self.world, i + 1)[i]
        if self.wl.get("rebuild"):
            self.scene.rebuild_bvh()                      # config 5: rtcCommitScene every frame (path_tracer.c:687)
            self.build_ms = self.build_ms + self.scene.info().bvhBuildMs
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

    def rooflines(self, steps, dt_ms_per_step, plain=False):
        """One entry per kernel class of the step.  Every entry carries THREE ceilings and names the one that binds:
          valu  wave-instructions per second against the VALU issue peak (instructions per unit from the committed PMC passes of
                this workload x the units of THIS run / the kernel's device time in THIS run, from events attached to the dispatches);
          hbm   HBM-side bytes per second (FETCH_SIZE x2 + WRITE_SIZE per unit, same source, same scaling) against 8 TB/s;
          cache the ALGORITHMIC bytes (64 B per node visit, 48 B per triangle test, the path state a stage streams)
                against the L2's 34.5 TB/s -- most of them are cache hits, so against HBM they can exceed 1 and say nothing.
        `bound` / `achieved` / `peak` / `unit` / `frac` are those of the ceiling with the largest fraction."""
        a = self.acc
        if plain:
            (nodes, tris, worst), (tnodes, ttris, tworst) = (0.0, 0.0, 0), (0.0, 0.0, 0)
        else:
            (nodes, tris, worst), (tnodes, ttris, tworst) = self.work_per_ray(steps)
        b_ray = 48.0 + 64.0 * nodes + 48.0 * tris
        b_ray_tail = 48.0 + 64.0 * tnodes + 48.0 * ttris
        rays = a["extensionRays"] + a["shadowRays"] + a["otherRays"]
        batched = rays - a["tailRays"]
        launches = max(a["traceLaunches"], 1)
        device_ms = a["totalMs"] / steps
        prof = load_profile(self.name)
        out = []

        def entry(kernel, ms_total, bytes_total, units, unit_name, bytes_per_unit, **extra):
            sec = ms_total * 1e-3
            alg = bytes_total / sec / 1e9 if sec > 0 else 0.0
            e = {"kernel": kernel, "ms_per_step": ms_total / steps, "share_of_device_step": (ms_total / steps) / device_ms if device_ms > 0 else 0.0,
                 "units_per_step": units / steps, "unit_name": unit_name,
                 "algorithmic_bytes_per_unit": bytes_per_unit if not plain else None,
                 "algorithmic_GBps": alg if not plain else None,
                 "algorithmic_frac_of_hbm_peak": alg / HBM_PEAK_GBS if not plain else None}
            ceilings = {}
            if not plain and alg > 0:
                ceilings["cache"] = {"achieved": alg, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": alg / L2_PEAK_GBS,
                                     "what": "algorithmic bytes against the L2's aggregate bandwidth (the data is served from L1 / L2 / the Infinity Cache)"}
            pk = prof.get(kernel.split("<")[0])
            if pk and sec > 0 and units > 0:
                src = f"profiles/traffic.json [{pk.get('tag')}] = rocprofv3 --pmc passes of `{pk.get('command')}` ({pk.get('summary_file')}); per-unit figures of THAT run, scaled by the units and device time of THIS run"
                if pk.get("valu_insts_per_unit"):
                    g = units * pk["valu_insts_per_unit"] / sec / 1e9
                    ceilings["valu"] = {"achieved": g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": g / VALU_PEAK_GINST,
                                        "wave_insts_per_unit": pk["valu_insts_per_unit"], "lanes_per_valu_inst": pk.get("lanes_per_valu_inst"),
                                        "attainable_peak": VALU_ATTAINABLE_GINST, "frac_of_attainable": g / VALU_ATTAINABLE_GINST,
                                        "what": "VALU issue: wave64 instructions per second against 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles (the fp32 vector peak); attainable = the v_fma_f32 rate measured on this pool", "source": src}
                    add_mix(ceilings["valu"], kernel.split("<")[0])
                if pk.get("hbm_bytes_per_unit") is not None:
                    g = units * pk["hbm_bytes_per_unit"] / sec / 1e9
                    ceilings["hbm"] = {"achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                       "bytes_per_unit": pk["hbm_bytes_per_unit"], "l2_hit_rate": pk.get("l2_hit_rate"),
                                       "what": "HBM-side bytes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; counted between L2 and the fabric, so Infinity-Cache hits are included)", "source": src}
                    e["traffic"] = units / steps * pk["hbm_bytes_per_unit"]          # HBM-side bytes per step of this kernel class
                    if not plain and bytes_per_unit > 0:
                        e["traffic_over_algorithmic"] = pk["hbm_bytes_per_unit"] / bytes_per_unit
                if pk.get("l1_accesses_per_unit"):
                    e["vector_l1_accesses_per_unit"] = pk["l1_accesses_per_unit"]
            e.setdefault("traffic", None)
            e["ceilings"] = ceilings
            if ceilings:
                name = max(ceilings, key=lambda k: ceilings[k]["frac"])
                # the cache ceiling is a statement about algorithmic bytes; when a measured ceiling (valu / hbm) exists it names the bound
                measured = {k: v for k, v in ceilings.items() if k != "cache"}
                if measured:
                    name = max(measured, key=lambda k: measured[k]["frac"])
                c = ceilings[name]
                e.update({"bound": name, "achieved": c["achieved"], "peak": c["peak"], "unit": c["unit"], "frac": c["frac"]})
            else:
                e.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None})
            e.update(extra)
            return e

        avg_launch_ms = a["traceMs"] / launches
        out.append(entry("k_trace<closest>", a["traceMs"], batched * b_ray, batched, "ray", b_ray,
                         nodes_per_ray=nodes, tris_per_ray=tris, max_nodes_one_ray=worst, rays_per_launch=batched / launches,
                         launches_per_step=launches / steps, avg_launch_ms=avg_launch_ms, bytes_per_launch=batched / launches * b_ray))
        wpb = a["wavefrontPathBounces"]
        out.append(entry("k_shade_a<FAST+SLOW>", a["shadeAMs"], wpb * B_SHADE_A, wpb, "path-bounce", B_SHADE_A))
        out.append(entry("k_shade_b<FAST+SLOW>", a["shadeBMs"], wpb * B_SHADE_B, wpb, "path-bounce", B_SHADE_B))
        tail_bytes = a["tailRays"] * b_ray_tail + a["tailPathBounces"] * (B_SHADE_A + B_SHADE_B)
        out.append(entry("k_tail", a["tailMs"], tail_bytes, a["tailRays"], "ray (+ its share of the bounce's shading)",
                         tail_bytes / max(a["tailRays"], 1), paths_per_step=a["tailPaths"] / steps,
                         nodes_per_ray=tnodes, tris_per_ray=ttris,
                         note="one lane per path, the two rays of a bounce walked by the whole wave (DESIGN.md section 4); nodes / triangles per ray: "
                              "the counting build's figures for all rays of a sample minus those of the batched launches"))
        return out

    def close(self):
        self.trace.close()


def run_single(name, steps, warmup, release_scene=False):
    """a sub-record: the workload on one GPU, no collective"""
    r = Run(name)
    for i in range(warmup):
        r.step(i)
    r.trace.clear()
    r.build_ms = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        r.add(r.step(i))
    dt = time.perf_counter() - t0
    r.build_ms_timed = r.build_ms
    a = r.acc
    rays = a["extensionRays"] + a["shadowRays"] + a["otherRays"]
    mean, stddev = r.trace.stddev()
    rec = {"workload": name, "scene_tris": r.sc.tri_count, "width": r.W, "height": r.H, "max_bounces": r.wl["bounces"], "steps": steps,
           "warmup": warmup, "value": rays / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt * 1e3 / steps, "spp_per_s": steps / dt,
           "rays_per_path": rays / (steps * r.W * r.H), "device_ms_per_step": a["totalMs"] / steps, "traversal_ms_per_step": a["traceMs"] / steps,
           "image_mean_luminance": mean, "pt_stddev": stddev}
    if r.wl.get("rebuild"):
        # config 5: a step is a FULL BVH rebuild plus one sample; the two parts are reported separately as BASELINE asks
        info = r.scene.info()
        rec["bvh_rebuild_ms_per_step"] = r.build_ms_timed / steps
        rec["trace_only_Mrays_per_s"] = rays / (a["totalMs"] * 1e-3) / 1e6 if a["totalMs"] > 0 else None
        rec["bvh_nodes"] = info.bvhNodeCount
        rec["bvh_build_algorithmic_bytes_per_tri"] = 600.0
        rec["bvh_build_GBps_algorithmic"] = r.sc.tri_count * 600.0 / (rec["bvh_rebuild_ms_per_step"] * 1e-3) / 1e9 if rec["bvh_rebuild_ms_per_step"] > 0 else None
    rec["rooflines"] = r.rooflines(steps, dt * 1e3 / steps)
    r.close()
    if release_scene:
        _SCENES.pop(r.wl["scene"])[2].close()
    return rec


def microbench_roofline(name, n, dt, b_ray, alg, nodes, tris, worst):
    """the one launch of the microbenchmark against the same three ceilings as the step's kernels (Run.rooflines)"""
    pk = load_profile(name).get("k_trace") or {}
    e = {"kernel": "k_trace<closest>", "algorithmic_bytes_per_ray": b_ray, "algorithmic_GBps": alg, "algorithmic_frac_of_hbm_peak": alg / HBM_PEAK_GBS,
         "nodes_per_ray": nodes, "tris_per_ray": tris, "max_nodes_one_ray": worst, "traffic": None,
         "ceilings": {"cache": {"achieved": alg, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": alg / L2_PEAK_GBS,
                                "what": "algorithmic bytes (64 B per node visit, 48 B per triangle test) against the L2's aggregate bandwidth (a 1 M-triangle "
                                        "tree, 67 MB, lives in L2 and the Infinity Cache; the 20 M-triangle one, 1.3 GB, does not: its binding ceiling is hbm)"}}}
    src = f"profiles/traffic.json [{pk.get('tag')}] = rocprofv3 --pmc passes of `{pk.get('command')}` ({pk.get('summary_file')})"
    if pk.get("valu_insts_per_unit"):
        g = n * pk["valu_insts_per_unit"] / dt / 1e9
        e["ceilings"]["valu"] = {"achieved": g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": g / VALU_PEAK_GINST,
                                 "wave_insts_per_unit": pk["valu_insts_per_unit"], "lanes_per_valu_inst": pk.get("lanes_per_valu_inst"),
                                 "attainable_peak": VALU_ATTAINABLE_GINST, "frac_of_attainable": g / VALU_ATTAINABLE_GINST, "source": src}
        add_mix(e["ceilings"]["valu"], "k_trace")
    if pk.get("hbm_bytes_per_unit") is not None:
        g = n * pk["hbm_bytes_per_unit"] / dt / 1e9
        e["ceilings"]["hbm"] = {"achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS, "bytes_per_unit": pk["hbm_bytes_per_unit"],
                                "l2_hit_rate": pk.get("l2_hit_rate"), "source": src,
                                # the fabric moves 128-byte lines, and serves random line requests at a measured 46-57 G/s (a coalesced
                                # stream: 45-47 G/s): profiles/r03_gather_calibration.txt, scripts/micro/gather.hip
                                "line_requests_per_s": g * 1e9 / 128.0, "random_line_request_ceiling_measured": [46e9, 57e9]}
        e["traffic"] = n * pk["hbm_bytes_per_unit"]
    measured = {k: v for k, v in e["ceilings"].items() if k != "cache"} or e["ceilings"]
    name = max(measured, key=lambda k: measured[k]["frac"])
    c = e["ceilings"][name]
    e.update({"bound": name, "achieved": c["achieved"], "peak": c["peak"], "unit": c["unit"], "frac": c["frac"]})
    return e


def incoherent_microbench(log2_rays=26, reps=4, kind="synth1m", name="incoherent_closest_hit_1m"):
    """incoherent closest-hit rays -- origins uniformly on random triangles of the 1M-triangle scene (+1e-4 N),
    directions uniform in the hemisphere of the geometric normal, in random order; 2^26 rays, ONE launch of the traversal
    kernel through the C-ABI (pimgpu_intersect_packed_dev), rays and hits resident in HBM.  The rays are generated on the
    device (torch is used for device memory and random numbers only)."""
    import torch
    sc, cam, scene = build_scene(kind)
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
    rec = {"workload": name, "scene_tris": sc.tri_count, "rays_per_launch": n, "launches_timed": reps,
           "ms_per_launch": dt * 1e3, "value": n / dt / 1e6, "unit": "Mrays/s", "north_star_target_Mrays_per_s": 2000.0,
           "hit_fraction": float((hits[:, 0] >= 0).float().mean()),
           "roofline": microbench_roofline(name, n, dt, b_ray, ach, nodes / rays, tris / rays, worst)}
    del rayO, rayD, hits, pos
    torch.cuda.empty_cache()
    return rec


def frames_in_flight(workload, contexts, samples):
    """BASELINE config 3 renders MANY samples of one image.  pimgpu_trace_image_samples shards them by sample index over the device
    list; with THIS GPU listed `contexts` times that many whole frames are in flight on it, each on its own stream with its own work
    buffers, and one frame's launches fill the machine while another's drain (every launch ends with its longest rays).  Same
    parity bar (tests/test_gpu_multi_device.py plays the schedule on the oracle, bit for bit); UpdateDists runs once per round of
    `contexts` samples instead of once per sample.  The device list is per process, so the measurement runs in a child
    (scripts/measure_concurrent_samples.py); a sub-record, NOT the headline: the headline step is one Pt_Trace call."""
    import subprocess
    rec = {"workload": "%s_%d_frames_in_flight" % (workload, contexts), "contexts_on_one_gpu": contexts, "samples_per_call": samples}
    try:
        run = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", "measure_concurrent_samples.py"), workload, str(contexts), str(samples)],
                             cwd=ROOT, capture_output=True, text=True, timeout=300)
        line = [l for l in run.stdout.splitlines() if l.startswith("{")]
        if run.returncode != 0 or not line:
            rec["error"] = (run.stderr or run.stdout)[-400:]
        else:
            d = json.loads(line[-1])
            rec.update({"value": d["Mrays_per_s"], "unit": "Mrays/s", "ms_per_step": d["ms_per_sample"], "ms_per_call": d["ms_per_call"]})
    except Exception as e:
        rec["error"] = repr(e)
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
    ap.add_argument("--plain", action="store_true", help="profiling target: warm-up + timed steps and nothing else (no counting frames, no CPU baseline, "
                    "no sub-records), so that every frame a profiler sees is one step of the workload")
    ap.add_argument("--microbench-only", action="store_true", help="profiling target: the incoherent-ray microbenchmark alone")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.plain:
        args.no_sub = args.no_cpu_baseline = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if os.environ.get("PIM_BENCH_REHEARSE_GLOO"):
            # rehearsal of the N-process control flow on a box with fewer GPUs than ranks: the ranks share the devices and the
            # collectives run over gloo through host copies (pim_amd/dist.py).  Timings of such a run mean nothing.
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if world == 1 and not args.no_sub:
        # the microbenchmark leg keeps its rays in torch tensors; torch's HIP runtime has to come up BEFORE libpimgpu's
        # (the other order leaves torch without a device in this process)
        import torch
        torch.cuda.init()
        torch.zeros(1, device="cuda")
    api.init(local_rank if world > 1 else 0)
    if args.microbench_only:
        if wl["scene"] == "synth20m":
            print(json.dumps(incoherent_microbench(24, 2, "synth20m", "incoherent_closest_hit_20m")), flush=True)
        else:
            print(json.dumps(incoherent_microbench(reps=2)), flush=True)
        for sc_, cam_, scene_ in _SCENES.values():
            scene_.close()
        api.shutdown()
        return
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
    live_t = base_t = None
    if world > 1:
        reduce_buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
        # the light histograms of the replicas are summed before every UpdateDists (pim_amd/dist.py): every rank then samples its
        # lights from the distributions ONE process tracing all the ranks' paths would hold
        nlive = max(1, run.scene.live_count())
        live_t = torch.zeros(nlive, dtype=torch.int32, device="cuda")
        base_t = torch.zeros(nlive, dtype=torch.int32, device="cuda")
        run.scene.live_get_dev(base_t.data_ptr())

    def exchange():
        """N > 1, before every step: all-reduce of the histogram increments, UpdateDists on the sum.  A collective: every rank
        calls it the same number of times (a rank whose sample range is one short still takes part in the last one)."""
        run.scene.live_get_dev(live_t.data_ptr())
        summed = pdist.allreduce_light_histograms(live_t, base_t, world, dist)
        torch.cuda.current_stream().synchronize()            # the library copies on its own stream
        run.scene.live_set_dev(summed.data_ptr())
        run.scene.update_dists()
        run.scene.live_get_dev(base_t.data_ptr())

    def step(i, sample):
        """one step of this rank"""
        if world == 1:
            return run.step(i, sample)
        exchange()
        return run.step(i, sample, update_dists=0)

    def sync():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, None)
    run.trace.clear()
    run.build_ms = 0.0
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        run.add(step(i, my_samples[i]))             # sampleWeight 1/(i+1): this rank's image is the running mean of ITS samples
    if world > 1 and strong:
        for _ in range(steps, -(-args.total_spp // world)):     # ranges differ by at most one sample: the short ranks' last exchange
            exchange()
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
        t = torch.tensor([dt, float(rays), float(steps)], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); rays_all = float(tsum[1]); total_steps = int(tsum[2])
    else:
        rays_all = float(rays)

    if rank == 0:
        mean, stddev = run.trace.stddev()
        steps_r0 = max(steps, 1)
        roofs = run.rooflines(steps_r0, dt * 1e3 / steps_r0, plain=args.plain)
        dominant = max(roofs, key=lambda e: e["ms_per_step"])
        # the contract's `roofline` object is the dominant kernel's entry; `rooflines` has every kernel class of the step
        out = {
            "metric": "Mrays/sec (incoherent, per GPU + node) and spp/sec at 1080p; % HBM BW peak",
            "value": rays_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / steps_r0, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene_tris": info.triCount, "width": W, "height": H,
                       "max_bounces": wl["bounces"], "spp_per_step": 1,
                       "sharding": ("sample ranges of --total-spp %d" % args.total_spp if strong else "sample index, interleaved") if world > 1 else "none",
                       "light_histograms": "SUM all-reduce of the ranks' increments before every UpdateDists" if world > 1 else "one replica"},
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
            if args.workload != "synth20m_1080p" and not os.environ.get("PIM_BENCH_SKIP_20M"):
                try:
                    subs.append(run_single("synth20m_1080p", 8, 2))                          # config 5 on one GPU: the HBM-bound case
                    subs.append(incoherent_microbench(24, 3, "synth20m", "incoherent_closest_hit_20m"))
                except Exception as e:
                    subs.append({"workload": "synth20m_1080p", "error": repr(e)})
                if "synth20m" in _SCENES:
                    _SCENES.pop("synth20m")[2].close()
            if args.workload == DEFAULT_WORKLOAD and not os.environ.get("PIM_BENCH_SKIP_FRAMES_IN_FLIGHT"):
                subs.append(frames_in_flight(args.workload, 2, 8))
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
