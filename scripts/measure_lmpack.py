"""LmPack_Pack (rendering/lightmap.c:1047-1102) at scale: the objects of the synthetic scene (the room's 100 m walls left out: at
16 texels/m a wall is wider than an atlas and the reference's packer never returns on it) as one world-space mesh, through the
host layer's steps one by one, timed as whole calls (host arrays in and out).
usage: python scripts/measure_lmpack.py [tris] [texels_per_meter]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from pim_amd import api, scenes

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tpu = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
S = 1024
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
host.PtSys_Ok.restype = C.c_bool
host.PtSys_Init()
assert host.PtSys_Ok()
sc = scenes.synthetic_scene(target_tris=tris, seed=1234, extent=100.0, emissive_attempts=4, diffuse_only=True)
P = np.ascontiguousarray(sc.positions, np.float32).reshape(-1, 3, 4)
e1, e2 = P[:, 1, :3] - P[:, 0, :3], P[:, 2, :3] - P[:, 0, :3]
area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
keep = np.repeat(area < 4.0, 3)
P = np.ascontiguousarray(sc.positions[keep], np.float32); N = np.ascontiguousarray(sc.normals[keep], np.float32)
U = np.zeros((P.shape[0], 4), np.float32); T = np.full((P.shape[0], 4), -1, np.int32)
m = api.PimMaterial(); m.flags = 0
M = np.zeros((1, 4, 4), np.float32); M[0, 0, 0] = M[0, 1, 1] = M[0, 2, 2] = M[0, 3, 3] = 1.0
dr = api.Drawables([(P, N, U, T)], np.zeros(1, np.int32), M, [m])
out = {"what": "LmPack_Pack step by step", "triangles": int(P.shape[0] // 3), "texels_per_meter": tpu, "atlas": S}

def timed(name, fn):
    t0 = time.perf_counter(); r = fn(); out[name + "_s"] = round(time.perf_counter() - t0, 4); return r

api.lightmap_chartnodes(dr, tpu)
nodes = timed("chartnodes_gpu", lambda: api.lightmap_chartnodes(dr, tpu))
host.LmPack_GroupCharts.restype = C.c_int32
host.LmPack_GroupCharts.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_void_p]
first = np.zeros(nodes.shape[0] + 1, np.int32)
nc = timed("chart_group_host", lambda: host.LmPack_GroupCharts(nodes.ctypes.data, nodes.shape[0], 0.1, 15.0, S / 3.0, 1, first.ctypes.data))
first = first[:nc + 1].copy()
out["charts"] = int(nc); out["largest_chart_nodes"] = int(np.diff(first).max())
areas, sizes, offsets, masks = timed("chart_masks_gpu", lambda: api.lightmap_chart_masks(nodes, first))
out["mask_bytes"] = int(masks.shape[0])
order = np.zeros(nc, np.int32)
host.LmPack_SortCharts.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
timed("chart_sort_host", lambda: host.LmPack_SortCharts(areas.ctypes.data, nc, order.ctypes.data))
est = int(np.ceil(np.ceil(areas).astype(np.int64).sum() / (S * S))) or 1
out["atlas_estimate"] = est
tr, idx, used = timed("atlas_place_gpu", lambda: api.lightmap_atlas_place(masks, offsets, sizes, areas, order, S, est, est * 4 + 16))
import zlib
out["placement_crc32"] = zlib.crc32(np.ascontiguousarray(tr).tobytes() + np.ascontiguousarray(idx).tobytes())
out["atlases_used"] = int(used); out["charts_per_s_placed"] = round(nc / out["atlas_place_gpu_s"], 1)
print(json.dumps(out))
