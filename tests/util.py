"""Shared helpers for the tests: library loading and ctypes mirrors of include/pimgpu.h."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = {}


def load_oracle(portable=False):
    """TEST INFRASTRUCTURE ONLY: the CPU oracle.

    portable=False: oracle/liborc.so   (glibc libm; bit-exact to the reference's headers)
    portable=True:  oracle/liborc_p.so (portable libm of oracle/orc_libm.h; the variant the
                    HIP path is compared with bit for bit)"""
    name = "liborc_p.so" if portable else "liborc.so"
    if name not in _ORACLE:
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _ORACLE[name] = ctypes.CDLL(path)
    return _ORACLE[name]


# ---------------------------------------------------------------------------------------------
# ctypes face of oracle/pim_oracle.h (TEST INFRASTRUCTURE ONLY)

import numpy as np  # noqa: E402

from pim_amd import api  # noqa: E402

C = ctypes


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """orc_scene_create .. orc_scene_destroy; accel 0 = brute force, 1 = binned-SAH BVH2"""

    def __init__(self, desc, accel=1, threads=8, portable=True):
        self.lib = load_oracle(portable)
        self.lib.orc_scene_create.restype = C.c_void_p
        self.threads = threads
        self.handle = C.c_void_p(self.lib.orc_scene_create(C.byref(desc), C.c_int(accel), C.c_int(threads)))
        assert self.handle, "orc_scene_create failed"

    def close(self):
        if self.handle:
            self.lib.orc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = api.PimSceneInfo()
        self.lib.orc_scene_info(self.handle, C.byref(out))
        return out

    def download_lights(self):
        info = self.info()
        E, L, V = info.emissiveCount, info.gridLen, info.vertCount
        out = dict(vertToEmit=np.zeros(V, np.int32), emitToVert=np.zeros(E, np.int32), cellLength=np.zeros(L, np.int32),
                   pdf=np.zeros(L * E, np.float32), cdf=np.zeros(L * (E + 1), np.float32), live=np.zeros(L * E, np.uint32))
        self.lib.orc_scene_download_lights(self.handle, _p(out["vertToEmit"]), _p(out["emitToVert"]), _p(out["cellLength"]),
                                           _p(out["pdf"]), _p(out["cdf"]), _p(out["live"]))
        return out

    def update_dists(self):
        self.lib.orc_scene_update_dists(self.handle)

    def set_live(self, live):
        live = np.ascontiguousarray(live, np.uint32)
        self.lib.orc_scene_set_live(self.handle, _p(live))

    def intersect(self, ro, rd, tNear=None, tFar=None):
        ro, rd = api.f32(ro), api.f32(rd)
        n = ro.shape[0]
        tn = api.f32(tNear) if tNear is not None else np.zeros(n, np.float32)
        tf = api.f32(tFar) if tFar is not None else np.full(n, api.RCP_EPSILON, np.float32)
        out = np.zeros(n, api.RAYHIT_DTYPE)
        self.lib.orc_intersect(self.handle, _p(ro), _p(rd), _p(tn), _p(tf), C.c_int32(n), _p(out), C.c_int(self.threads))
        return out

    def intersect_raw(self, ro, rd, tNear=None, tFar=None, brute=False):
        ro, rd = api.f32(ro), api.f32(rd)
        n = ro.shape[0]
        tn = api.f32(tNear) if tNear is not None else np.zeros(n, np.float32)
        tf = api.f32(tFar) if tFar is not None else np.full(n, api.RCP_EPSILON, np.float32)
        prim = np.zeros(n, np.int32)
        tuv = np.zeros((n, 3), np.float32)
        self.lib.orc_intersect_raw(self.handle, _p(ro), _p(rd), _p(tn), _p(tf), C.c_int32(n), _p(prim), _p(tuv),
                                   C.c_int(1 if brute else 0), C.c_int(self.threads))
        return prim, tuv

    def intersect_f64(self, ro, rd, tNear=None, tFar=None):
        ro, rd = api.f32(ro), api.f32(rd)
        n = ro.shape[0]
        tn = api.f32(tNear) if tNear is not None else np.zeros(n, np.float32)
        tf = api.f32(tFar) if tFar is not None else np.full(n, api.RCP_EPSILON, np.float32)
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float64)
        self.lib.orc_intersect_f64(self.handle, _p(ro), _p(rd), _p(tn), _p(tf), C.c_int32(n), _p(prim), _p(t), C.c_int(self.threads))
        return prim, t

    def trace_rays(self, ro, rd, sampleIndex=0, seed=0, maxBounces=0):
        ro, rd = api.f32(ro), api.f32(rd)
        n = ro.shape[0]
        out = np.zeros(n, api.RESULT_DTYPE)
        self.lib.orc_trace_rays(self.handle, _p(ro), _p(rd), C.c_int32(n), C.c_uint32(sampleIndex), C.c_uint32(seed),
                                C.c_int32(maxBounces), _p(out), C.c_int(self.threads))
        return out

    def raygen(self, origin, count, sampleIndex=0, seed=0, maxBounces=0):
        origin = api.f32(origin)
        colors = np.zeros((count, 4), np.float32)
        directions = np.zeros((count, 4), np.float32)
        self.lib.orc_raygen(self.handle, _p(origin), C.c_int32(count), C.c_uint32(sampleIndex), C.c_uint32(seed),
                            C.c_int32(maxBounces), _p(colors), _p(directions), C.c_int(self.threads))
        return colors, directions

    def cubemap_bake(self, faces, size, origin, weight, sampleIndex=0, seed=0, maxBounces=0):
        arr = (C.c_void_p * 6)()
        for f in range(6):
            arr[f] = faces[f].ctypes.data
        origin = api.f32(origin)
        self.lib.orc_cubemap_bake(self.handle, arr, C.c_int32(size), _p(origin), C.c_float(weight), C.c_uint32(sampleIndex),
                                  C.c_uint32(seed), C.c_int32(maxBounces), C.c_int(self.threads))

    def ambcube_bake(self, cube, origin, samples, prevSampleCount=0, sampleIndex=0, seed=0, maxBounces=0):
        origin = api.f32(origin)
        self.lib.orc_ambcube_bake.restype = C.c_int32
        return self.lib.orc_ambcube_bake(self.handle, _p(cube), _p(origin), C.c_int32(samples), C.c_int32(prevSampleCount),
                                         C.c_uint32(sampleIndex), C.c_uint32(seed), C.c_int32(maxBounces), C.c_int(self.threads))

    def trace_image(self, images, width, height, camera, dof, params):
        """images = (color, albedo, normal) float32 (n,3) arrays updated in place (running mean); returns stats"""
        stats = api.PimTraceStats()
        self.lib.orc_trace_image(self.handle, _p(images[0]), _p(images[1]), _p(images[2]), C.c_int32(width), C.c_int32(height),
                                 C.byref(camera), C.byref(dof), C.byref(params), C.c_int(self.threads), C.byref(stats))
        return stats

    def stddev(self, color, width, height):
        m, s = C.c_float(), C.c_float()
        self.lib.orc_stddev(_p(api.f32(color)), C.c_int32(width), C.c_int32(height), C.byref(m), C.byref(s))
        return m.value, s.value

    def lightmap_bake(self, pages, lmSize, texelsPerMeter, timeSlice, spp, passIndex=0, seed=0, maxBounces=0,
                      texelBegin=0, texelEnd=-1):
        arr = (api.PimLightmapPage * len(pages))()
        for i, p in enumerate(pages):
            arr[i].block = p.ctypes.data
        self.lib.orc_lightmap_bake(self.handle, arr, C.c_int32(len(pages)), C.c_int32(lmSize), C.c_float(texelsPerMeter),
                                   C.c_float(timeSlice), C.c_int32(spp), C.c_uint32(passIndex), C.c_uint32(seed),
                                   C.c_int32(maxBounces), C.c_int64(texelBegin), C.c_int64(texelEnd), C.c_int(self.threads))


def copy_dof(d):
    out = api.PimDofInfo()
    C.memmove(C.byref(out), C.byref(d), C.sizeof(d))
    return out


def bits(a):
    """float32 array -> uint32 view for bit-exact comparisons"""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def brute_positions(positions, ro, rd, tNear=None, tFar=None, want_f64=False, threads=16, portable=True):
    """orc_brute_positions: the contract's fp32 closest hit over ALL triangles of a bare soup -> (prim, tuv);
    with want_f64 also the fp64 ground truth -> (prim, tuv, prim64, t64)"""
    lib = load_oracle(portable)
    positions = api.f32(positions); ro, rd = api.f32(ro), api.f32(rd)
    n = ro.shape[0]
    tn = api.f32(tNear) if tNear is not None else np.zeros(n, np.float32)
    tf = api.f32(tFar) if tFar is not None else np.full(n, api.RCP_EPSILON, np.float32)
    prim = np.zeros(n, np.int32); tuv = np.zeros((n, 3), np.float32)
    prim64 = np.zeros(n, np.int32) if want_f64 else None
    t64 = np.zeros(n, np.float64) if want_f64 else None
    lib.orc_brute_positions.restype = None
    lib.orc_brute_positions(_p(positions), C.c_int32(positions.shape[0]), _p(ro), _p(rd), _p(tn), _p(tf), C.c_int32(n),
                            _p(prim), _p(tuv), _p(prim64), _p(t64), C.c_int(threads))
    return (prim, tuv, prim64, t64) if want_f64 else (prim, tuv)


def oracle_cubemap_elem_count(size):
    lib = load_oracle(True)
    lib.orc_cubemap_elem_count.restype = C.c_int32
    return int(lib.orc_cubemap_elem_count(C.c_int32(size)))


def oracle_cubemap_convolve(faces, convolved, size, sample_count, weight, sampleIndex=0, seed=0, threads=8):
    """Cubemap_Convolve on the oracle (oracle/pim_oracle.c: orc_cubemap_convolve); convolved is blended into in place"""
    lib = load_oracle(True)
    fa, ca = (C.c_void_p * 6)(), (C.c_void_p * 6)()
    for f in range(6):
        fa[f] = faces[f].ctypes.data; ca[f] = convolved[f].ctypes.data
    lib.orc_cubemap_convolve.restype = None
    lib.orc_cubemap_convolve(fa, ca, C.c_int32(size), C.c_uint32(sample_count), C.c_float(weight), C.c_uint32(sampleIndex), C.c_uint32(seed), C.c_int(threads))
