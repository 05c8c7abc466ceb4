"""measurements of BASELINE configs that bench.py's default line does not cover (one GPU; JSON lines on stdout):
  rays   <ntri> <log2 rays>   incoherent closest-hit microbenchmark (surface + 1e-4 N, random hemisphere
                              direction, shuffled order), rays resident in HBM
  bake   <pages> <spp>        config 4 shape: lightmap bake of the 1M-triangle scene at lm_density 16, 1024^2 pages
                              (bounded: the first <pages> pages of the per-triangle charts), host pages in / out
  big    <ntri>               config 5 shape: N-triangle diffuse scene, BVH build + one 1080p 8-bounce sample"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pim_amd import api, scenes

mode = sys.argv[1]
api.init(0)
lib = api.load_library()

def emit(d):
    print(json.dumps(d), flush=True)

if mode == "rays":
    ntri, lg = int(sys.argv[2]), int(sys.argv[3])
    nray = 1 << lg
    sc = scenes.synthetic_scene(target_tris=ntri, seed=1234, extent=100.0, emissive_attempts=16)
    g = api.Scene(sc.desc); info = g.info()
    chunk = 1 << 22
    tro = torch.empty((nray, 4), dtype=torch.float32, device="cuda"); trd = torch.empty((nray, 4), dtype=torch.float32, device="cuda")
    for b in range(0, nray, chunk):          # generated in chunks (host memory), uploaded once
        ro, rd = scenes.surface_rays(sc, min(chunk, nray - b), seed=1000 + b // chunk)
        ro[:, 3] = 0.0; rd[:, 3] = api.RCP_EPSILON
        tro[b:b + ro.shape[0]] = torch.from_numpy(ro).cuda(); trd[b:b + ro.shape[0]] = torch.from_numpy(rd).cuda()
    hits = torch.zeros((nray, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    def run():
        api.check(lib.pimgpu_intersect_packed_dev(g.handle, ctypes.c_void_p(tro.data_ptr()), ctypes.c_void_p(trd.data_ptr()), ctypes.c_int32(nray), ctypes.c_void_p(hits.data_ptr())))
    run()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): run()
    dt = (time.perf_counter() - t0) / reps
    api.work_counters(True); run(); nodes, tris, rays, worst = api.work_counters(False)
    b_ray = 48.0 + 64.0 * nodes / rays + 48.0 * tris / rays
    emit({"what": "incoherent closest-hit microbenchmark", "scene_tris": info.triCount, "rays_per_launch": nray, "ms_per_launch": dt * 1e3,
          "Mrays_per_s": nray / dt / 1e6, "nodes_per_ray": nodes / rays, "tris_per_ray": tris / rays, "max_nodes_one_ray": worst,
          "bytes_per_ray": b_ray, "algorithmic_GBps": nray * b_ray / dt / 1e9, "frac_of_8TBps": nray * b_ray / dt / 8e12,
          "hit_fraction": float((hits[:, 0] >= 0).float().mean()), "bvh_build_ms": info.bvhBuildMs})
elif mode == "bake":
    npages, spp = int(sys.argv[2]), int(sys.argv[3])
    sc = scenes.synthetic_scene(target_tris=1000000, seed=1234, extent=100.0, emissive_attempts=16)
    g = api.Scene(sc.desc)
    S = 1024
    pages = scenes.lightmap_pages_for(sc, S, 16.0, max_pages=npages)
    if "--pageable" not in sys.argv:                 # as Lightmap_New of the host layer allocates them: page-locked
        locked = [api.host_array(p.size) for p in pages]
        for d, p in zip(locked, pages): d[:] = p
        pages = locked
    texels = int(sum((p[S * S * 26:] > 0).sum() for p in pages))
    g.lightmap_bake(pages, S, 16.0, 1.0, 1, passIndex=0, seed=1, maxBounces=8)        # warm-up pass
    t0 = time.perf_counter()
    g.lightmap_bake(pages, S, 16.0, 1.0, spp, passIndex=1, seed=1, maxBounces=8)
    dt = time.perf_counter() - t0
    emit({"what": "lightmap bake (LmPack_Bake), 1M-triangle scene, lm_density 16, 1024^2 pages, host pages in/out (PCIe inclusive)",
          "pages": len(pages), "host_memory": "pageable" if "--pageable" in sys.argv else "page-locked", "mapped_texels": texels, "spp": spp, "seconds": dt, "texel_samples_per_s": texels * spp / dt})
elif mode == "big":
    ntri = int(sys.argv[2])
    t0 = time.perf_counter()
    sc = scenes.synthetic_scene(target_tris=ntri, seed=1234, extent=100.0, emissive_attempts=4, diffuse_only=True)
    tgen = time.perf_counter() - t0
    t0 = time.perf_counter(); g = api.Scene(sc.desc); tcreate = time.perf_counter() - t0
    info = g.info()
    builds = []
    for _ in range(3):
        g.rebuild_bvh(); builds.append(g.info().bvhBuildMs)
    W, H = 1920, 1080
    cam = scenes.synthetic_camera(100.0)
    tr = api.Trace(W, H); dof = scenes.default_dof()
    rays = 0; ms = 0.0; tms = 0.0
    for i in range(6):
        tr.trace(g, cam, dof, scenes.trace_params(i, seed=1, max_bounces=8))
        st = tr.stats()
        if i >= 2: rays += st.extensionRays + st.shadowRays + st.otherRays; ms += st.totalMs; tms += st.traceMs
    emit({"what": "config 5 shape: per-frame BVH rebuild + 1 spp 1080p 8 bounces", "scene_tris": info.triCount, "host_generate_s": tgen,
          "scene_create_s": tcreate, "bvh_build_ms": builds, "bvh_nodes": info.bvhNodeCount, "bvh_depth": info.bvhMaxDepth,
          "scene_setup_ms": info.sceneSetupMs, "ms_per_spp": ms / 4, "traversal_ms_per_spp": tms / 4, "Mrays_per_s": rays / (ms * 1e-3) / 1e6,
          "frame_ms_with_rebuild": ms / 4 + min(builds)})
