"""incoherent closest-hit microbenchmark: random rays resident in HBM through pimgpu_intersect_packed_dev"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pim_amd import api, scenes

ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nray = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
api.init(0)
lib = api.load_library()
sc = scenes.synthetic_scene(target_tris=ntri, seed=1234, extent=100.0, emissive_attempts=16)
g = api.Scene(sc.desc)
i = g.info()
print("tris", i.triCount, "depth", i.bvhMaxDepth, "build ms", round(i.bvhBuildMs, 2), flush=True)
for kind in ("uniform", "surface"):
    rng = np.random.default_rng(1)
    if kind == "uniform":
        ro, rd = scenes.random_rays(sc, nray, seed=1)
    else:
        tri = rng.integers(0, sc.tri_count, nray)
        P = sc.positions[:, :3].reshape(-1, 3, 3)[tri]
        w = rng.dirichlet([1, 1, 1], nray).astype(np.float32)
        p = (P * w[:, :, None]).sum(1)
        nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        d = rng.normal(size=(nray, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = np.where((d * nrm).sum(1, keepdims=True) < 0, -d, d)
        ro = np.zeros((nray, 4), np.float32); rd = np.zeros((nray, 4), np.float32)
        ro[:, :3] = p + nrm * 1e-4; rd[:, :3] = d
    ro[:, 3] = 0.0; rd[:, 3] = api.RCP_EPSILON
    tro = torch.from_numpy(ro).cuda(); trd = torch.from_numpy(rd).cuda()
    hits = torch.zeros((nray, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    def run(n):
        api.check(lib.pimgpu_intersect_packed_dev(g.handle, ctypes.c_void_p(tro.data_ptr()), ctypes.c_void_p(trd.data_ptr()), ctypes.c_int32(n), ctypes.c_void_p(hits.data_ptr())))
    run(nray)
    for n in (nray, nray // 8, nray // 64):
        t0 = time.perf_counter(); run(n); dt = time.perf_counter() - t0
        api.work_counters(True); run(n); nodes, tris, rays, worst = api.work_counters(False)
        print(f"{kind:8s} n={n:8d} {n / dt / 1e6:8.1f} Mrays/s  {dt * 1e3:7.3f} ms  nodes/ray {nodes / max(rays, 1):6.1f} tris/ray {tris / max(rays, 1):6.1f} worst ray {worst} nodes", flush=True)
    h = hits.cpu().numpy()
    # a checksum of the records: every build variant of the kernel must print the same line (the result does not depend on the tree walk)
    print("   hit fraction", (h[:, 0] >= 0).mean(), "checksum", hex(int(np.bitwise_xor.reduce(h.view(np.uint32).astype(np.uint64).sum(axis=1) & 0xffffffff))))
