"""BASELINE config 4 end to end on one GPU: the 1 M-triangle synthetic scene, lm_density 16, 1024^2 pages -- LmPack_Pack (host
layer: chart nodes, grouping, masks, atlas search, EmbedAttributes) on the scene's objects, then LmPack_Bake at <spp> samples
per texel with the pages resident in HBM (mode 2), then LmPack_Sync.  The room's 100 m walls are traced but not packed: at 16
texels/m a wall is wider than an atlas and the reference's packer never returns on it (scripts/measure_lmpack.py).
usage: python scripts/measure_config4.py [spp] [tris]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from pim_amd import api, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
S, tpu, seed, bounces = 1024, 16.0, 1, 8
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))


class Lightmap(C.Structure):
    _fields_ = [("probes", C.c_void_p * 5), ("position", C.c_void_p), ("normal", C.c_void_p), ("sampleCounts", C.c_void_p),
                ("size", C.c_int32), ("slot", C.c_uint32)]


class LmPack(C.Structure):
    _fields_ = [("axii", C.c_float * 20), ("lightmaps", C.POINTER(Lightmap)), ("lmCount", C.c_int32), ("lmSize", C.c_int32),
                ("texelsPerMeter", C.c_float)]


host.PtSys_Ok.restype = C.c_bool
host.PtSys_Configure((C.c_int32 * 5)(seed, bounces, 0, 1, -1)); host.PtSys_Init()
assert host.PtSys_Ok()
sc = scenes.synthetic_scene(target_tris=tris, seed=1234, extent=100.0, emissive_attempts=16)
P = np.ascontiguousarray(sc.positions, np.float32).reshape(-1, 3, 4)
area = 0.5 * np.linalg.norm(np.cross(P[:, 1, :3] - P[:, 0, :3], P[:, 2, :3] - P[:, 0, :3]), axis=1)
keep = np.repeat(area < 4.0, 3)
P = np.ascontiguousarray(sc.positions[keep], np.float32); N = np.ascontiguousarray(sc.normals[keep], np.float32)
U = np.zeros((P.shape[0], 4), np.float32); T = np.full((P.shape[0], 4), -1, np.int32)
m = api.PimMaterial(); m.flags = 0
M = np.zeros((1, 4, 4), np.float32); M[0, 0, 0] = M[0, 1, 1] = M[0, 2, 2] = M[0, 3, 3] = 1.0
dr = api.Drawables([(P, N, U, T)], np.zeros(1, np.int32), M, [m])
out = {"what": "config 4 end to end: LmPack_Pack + LmPack_Bake, 1024^2 pages at 16 texels/m, pages resident in HBM", "scene_triangles": sc.tri_count,
       "packed_triangles": int(P.shape[0] // 3), "spp": spp, "max_bounces": bounces}
host.LmPack_PackEx.restype = LmPack
host.LmPack_PackEx.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_uint32]
host.LmPack_LastError.restype = C.c_char_p
t0 = time.perf_counter()
pack = host.LmPack_PackEx(C.byref(dr.desc), S, tpu, 0.1, 15.0, seed)
out["pack_s"] = round(time.perf_counter() - t0, 3)
assert pack.lmCount >= 1, host.LmPack_LastError()
out["pages"] = pack.lmCount
texels = 0
for i in range(pack.lmCount):
    counts = np.ctypeslib.as_array(C.cast(pack.lightmaps[i].sampleCounts, C.POINTER(C.c_float)), shape=(S * S,))
    texels += int((counts > 0).sum())
out["mapped_texels"] = texels
host.LmPack_Get.restype = C.POINTER(LmPack)
C.memmove(host.LmPack_Get(), C.byref(pack), C.sizeof(LmPack))
host.PtScene_NewFromSnapshot.restype = C.c_void_p
host.PtScene_NewFromSnapshot.argtypes = [C.c_void_p]
host.LmPack_Bake.argtypes = [C.c_void_p, C.c_float, C.c_int32]
scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
host.LmPack_SetResident(C.c_int32(2))
t0 = time.perf_counter()
host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(1))            # first call: the pages go up (7 GB), one sample
out["first_call_with_upload_s"] = round(time.perf_counter() - t0, 3)
print(json.dumps(dict(out, progress="baking")), file=sys.stderr, flush=True)
done = 0
t0 = time.perf_counter()
while done < spp:                                                 # calls of 32 samples: a progress line per call
    k = min(32, spp - done)
    host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(k)); done += k
    print(f"  {done} / {spp} spp, {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
dt = time.perf_counter() - t0
assert host.PtSys_Ok(), host.LmPack_LastError()
out["bake_s"] = round(dt, 3); out["texel_samples_per_s"] = texels * spp / dt
t0 = time.perf_counter(); host.LmPack_Sync(); out["sync_to_host_s"] = round(time.perf_counter() - t0, 3)
counts = np.ctypeslib.as_array(C.cast(pack.lightmaps[0].sampleCounts, C.POINTER(C.c_float)), shape=(S * S,))
out["sample_count_of_a_mapped_texel"] = float(counts.max())
print(json.dumps(out), flush=True)
