"""ctypes mirror of include/pimgpu.h and a thin loader for pim_amd/libpimgpu.so.

This is the Python face of the C-ABI used by tests/, bench.py and __graft_entry__.py; a C host
would include include/pimgpu.h (or the Pim-shaped wrappers in pim_amd/host/pim_host.h) instead.
The library is hand-written HIP for gfx950; loading fails loudly when it is missing -- there is no
CPU fallback.
"""
import ctypes
import os

import numpy as np

C = ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIMGPU_LIB", os.path.join(HERE, "libpimgpu.so"))     # override: A/B experiments only

PIMGPU_OK = 0
MATFLAG_EMISSIVE = 1 << 0
MATFLAG_SKY = 1 << 1
MATFLAG_REFRACTIVE = 1 << 5
HIT_NOTHING, HIT_BACKFACE, HIT_TRIANGLE = 0, 1, 2
TEX_RGBA8, TEX_XY16 = 0, 1
RCP_EPSILON = 8388608.0


class PimMaterial(C.Structure):
    _fields_ = [("albedo", C.c_int32), ("rome", C.c_int32), ("normal", C.c_int32), ("flags", C.c_uint32),
                ("meanFreePath", C.c_float * 4), ("ior", C.c_float), ("bumpiness", C.c_float), ("_pad", C.c_float * 2)]


class PimTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("format", C.c_int32), ("_pad", C.c_int32),
                ("texels", C.c_void_p)]


class PimMediaDesc(C.Structure):
    _fields_ = [("constantColor", C.c_float * 4), ("noiseColor", C.c_float * 4), ("constantMfp", C.c_float),
                ("noiseMfp", C.c_float), ("absorption", C.c_float), ("noiseOctaves", C.c_int32),
                ("noiseGain", C.c_float), ("noiseLacunarity", C.c_float), ("noiseFreq", C.c_float),
                ("noiseScale", C.c_float), ("noiseHeight", C.c_float), ("phaseDirA", C.c_float),
                ("phaseDirB", C.c_float), ("phaseBlend", C.c_float)]


class PimSceneDesc(C.Structure):
    _fields_ = [("vertCount", C.c_int32), ("matCount", C.c_int32), ("texCount", C.c_int32), ("skySize", C.c_int32),
                ("positions", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p), ("matIds", C.c_void_p),
                ("materials", C.c_void_p), ("textures", C.c_void_p), ("skyFaces", C.c_void_p * 6),
                ("brdfLut", C.c_void_p), ("brdfLutSize", C.c_int32), ("emissiveAttempts", C.c_int32),
                ("metersPerCell", C.c_float), ("seed", C.c_uint32), ("media", PimMediaDesc)]


class PimSceneInfo(C.Structure):
    _fields_ = [("vertCount", C.c_int32), ("triCount", C.c_int32), ("matCount", C.c_int32), ("emissiveCount", C.c_int32),
                ("gridSize", C.c_int32 * 3), ("gridLen", C.c_int32), ("activeCells", C.c_int32),
                ("bvhNodeCount", C.c_int32), ("bvhMaxDepth", C.c_int32), ("boundsLo", C.c_float * 3),
                ("boundsHi", C.c_float * 3), ("bvhBuildMs", C.c_float), ("sceneSetupMs", C.c_float)]


class PimCamera(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("rotation", C.c_float * 4), ("zNear", C.c_float), ("zFar", C.c_float),
                ("fovy", C.c_float), ("_pad", C.c_float)]


class PimDofInfo(C.Structure):
    _fields_ = [("aperture", C.c_float), ("focalLength", C.c_float), ("bladeCount", C.c_int32), ("bladeRot", C.c_float),
                ("focalPlaneCurvature", C.c_float), ("autoFocusSpeed", C.c_float), ("autoFocus", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class PimTraceParams(C.Structure):
    _fields_ = [("sampleWeight", C.c_float), ("frameDt", C.c_float), ("sampleIndex", C.c_uint32), ("seed", C.c_uint32),
                ("maxBounces", C.c_int32), ("updateDists", C.c_int32)]


class PimTraceStats(C.Structure):
    _fields_ = [("extensionRays", C.c_uint64), ("shadowRays", C.c_uint64), ("otherRays", C.c_uint64),
                ("paths", C.c_uint64), ("bounces", C.c_int32), ("traceMs", C.c_float), ("totalMs", C.c_float),
                ("tailRays", C.c_uint64), ("shadeAMs", C.c_float), ("shadeBMs", C.c_float), ("tailMs", C.c_float),
                ("traceLaunches", C.c_int32), ("wavefrontPathBounces", C.c_uint64), ("tailPaths", C.c_uint64), ("tailPathBounces", C.c_uint64)]


class PimMesh(C.Structure):
    _fields_ = [("length", C.c_int32), ("_pad", C.c_int32), ("positions", C.c_void_p), ("normals", C.c_void_p),
                ("uvs", C.c_void_p), ("texIndices", C.c_void_p)]


class PimDrawables(C.Structure):
    _fields_ = [("count", C.c_int32), ("meshCount", C.c_int32), ("meshes", C.c_void_p), ("matrices", C.c_void_p),
                ("materials", C.c_void_p), ("meshTable", C.c_void_p)]


class PimChartNode(C.Structure):
    _fields_ = [("plane", C.c_float * 4), ("triCoord", C.c_float * 6), ("area", C.c_float), ("drawableIndex", C.c_int32),
                ("vertIndex", C.c_int32), ("_pad", C.c_int32 * 3)]


CHARTNODE_DTYPE = np.dtype([("plane", np.float32, 4), ("triCoord", np.float32, 6), ("area", np.float32),
                            ("drawableIndex", np.int32), ("vertIndex", np.int32), ("_pad", np.int32, 3)])


class PimLightmapPage(C.Structure):
    _fields_ = [("block", C.c_void_p)]


RAYHIT_DTYPE = np.dtype([("wuvt", np.float32, 4), ("normal", np.float32, 4), ("type", np.int32), ("iVert", np.int32),
                         ("flags", np.uint32), ("_pad", np.int32)])
RESULT_DTYPE = np.dtype([("color", np.float32, 3), ("albedo", np.float32, 3), ("normal", np.float32, 3)])
RAWHIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])
assert RAYHIT_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 36 and C.sizeof(PimMaterial) == 48

# every symbol include/pimgpu.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "pimgpu_init", "pimgpu_shutdown", "pimgpu_last_error", "pimgpu_host_alloc", "pimgpu_host_free",
    "pimgpu_scene_create", "pimgpu_scene_destroy", "pimgpu_scene_rebuild_bvh", "pimgpu_scene_update_dists",
    "pimgpu_scene_info", "pimgpu_scene_download_lights", "pimgpu_scene_live_get_dev", "pimgpu_scene_live_set_dev",
    "pimgpu_intersect", "pimgpu_intersect_dev", "pimgpu_intersect_packed_dev", "pimgpu_trace_rays", "pimgpu_raygen",
    "pimgpu_trace_create", "pimgpu_trace_destroy", "pimgpu_trace_clear", "pimgpu_trace_image", "pimgpu_trace_image_samples",
    "pimgpu_trace_download", "pimgpu_trace_image_host", "pimgpu_trace_buffers", "pimgpu_trace_stddev", "pimgpu_trace_stats",
    "pimgpu_cubemap_bake", "pimgpu_cubemap_convolve", "pimgpu_cubemap_elem_count", "pimgpu_ambcube_bake",
    "pimgpu_flatten_counts", "pimgpu_flatten", "pimgpu_brdf_lut_bake", "pimgpu_lightmap_embed", "pimgpu_lightmap_chartnodes", "pimgpu_lightmap_chart_layout", "pimgpu_lightmap_chart_masks",
    "pimgpu_lightmap_bake", "pimgpu_stream", "pimgpu_work_counters", "pimgpu_trace_copy_dev", "pimgpu_trace_stddev_noalloc", "pimgpu_init_devices", "pimgpu_device_count",
    "pimgpu_lightmap_set_create", "pimgpu_lightmap_set_destroy", "pimgpu_lightmap_set_upload", "pimgpu_lightmap_set_download", "pimgpu_lightmap_set_bake", "pimgpu_lightmap_atlas_place",
]

_LIB = None


def cubemap_elem_count(size):
    """float4 elements of one face's convolved mip chain (Cubemap_New, rendering/cubemap.c:103-114)"""
    lib = load_library()
    lib.pimgpu_cubemap_elem_count.restype = C.c_int32
    return int(lib.pimgpu_cubemap_elem_count(C.c_int32(size)))


def cubemap_convolve(faces, convolved, size, sample_count, weight, sampleIndex=0, seed=0):
    """Cubemap_Convolve (rendering/cubemap.c:264-303); faces: six float32 arrays of size*size*3 (read), convolved: six float32
    arrays of cubemap_elem_count(size)*4, blended into in place"""
    lib = load_library()
    elems = cubemap_elem_count(size)
    fa, ca = (C.c_void_p * 6)(), (C.c_void_p * 6)()
    for f in range(6):
        assert faces[f].dtype == np.float32 and faces[f].flags["C_CONTIGUOUS"] and faces[f].size == size * size * 3
        assert convolved[f].dtype == np.float32 and convolved[f].flags["C_CONTIGUOUS"] and convolved[f].size == elems * 4
        fa[f] = faces[f].ctypes.data; ca[f] = convolved[f].ctypes.data
    check(lib.pimgpu_cubemap_convolve(fa, ca, C.c_int32(size), C.c_uint32(sample_count), C.c_float(weight), C.c_uint32(sampleIndex), C.c_uint32(seed)),
          "pimgpu_cubemap_convolve")


class PimGpuError(RuntimeError):
    pass


def load_library():
    """dlopen pim_amd/libpimgpu.so (built by __graft_entry__.build() / pim_amd/csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PimGpuError(f"{LIB_PATH} is missing: build it with `make -C pim_amd/csrc` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        lib.pimgpu_last_error.restype = C.c_char_p
        lib.pimgpu_stream.restype = C.c_void_p
        _LIB = lib
    return _LIB


def check(rc, what=""):
    if rc != PIMGPU_OK:
        msg = load_library().pimgpu_last_error().decode(errors="replace")
        raise PimGpuError(f"{what} failed with status {rc}: {msg}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """PtScene_New .. PtScene_Del (rendering/path_tracer.h:70-74) on the GPU."""

    def __init__(self, desc):
        self.lib = load_library()
        self.handle = C.c_void_p()
        check(self.lib.pimgpu_scene_create(C.byref(desc), C.byref(self.handle)), "pimgpu_scene_create")
        self.vertCount = desc.vertCount

    def close(self):
        if self.handle:
            self.lib.pimgpu_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = PimSceneInfo()
        check(self.lib.pimgpu_scene_info(self.handle, C.byref(out)), "pimgpu_scene_info")
        return out

    def rebuild_bvh(self):
        check(self.lib.pimgpu_scene_rebuild_bvh(self.handle), "pimgpu_scene_rebuild_bvh")

    def update_dists(self):
        check(self.lib.pimgpu_scene_update_dists(self.handle), "pimgpu_scene_update_dists")

    def live_count(self):
        info = self.info()
        return info.gridLen * info.emissiveCount

    def live_get_dev(self, dst_ptr):
        """the light histograms (u32[gridLen * E]) into caller-owned HBM"""
        check(self.lib.pimgpu_scene_live_get_dev(self.handle, C.c_void_p(dst_ptr)), "pimgpu_scene_live_get_dev")

    def live_set_dev(self, src_ptr):
        check(self.lib.pimgpu_scene_live_set_dev(self.handle, C.c_void_p(src_ptr)), "pimgpu_scene_live_set_dev")

    def download_lights(self):
        info = self.info()
        E, L, V = info.emissiveCount, info.gridLen, info.vertCount
        out = dict(vertToEmit=np.zeros(V, np.int32), emitToVert=np.zeros(E, np.int32), cellLength=np.zeros(L, np.int32),
                   pdf=np.zeros(L * E, np.float32), cdf=np.zeros(L * (E + 1), np.float32), live=np.zeros(L * E, np.uint32))
        check(self.lib.pimgpu_scene_download_lights(self.handle, _ptr(out["vertToEmit"]), _ptr(out["emitToVert"]),
              _ptr(out["cellLength"]), _ptr(out["pdf"]), _ptr(out["cdf"]), _ptr(out["live"])), "pimgpu_scene_download_lights")
        return out

    def intersect(self, ro, rd, tNear=None, tFar=None):
        """batched Pt_Intersect: ro/rd (n,4) float32 -> structured array of PtRayHit"""
        ro, rd = f32(ro), f32(rd)
        n = ro.shape[0]
        tn = f32(tNear) if tNear is not None else np.zeros(n, np.float32)
        tf = f32(tFar) if tFar is not None else np.full(n, RCP_EPSILON, np.float32)
        out = np.zeros(n, RAYHIT_DTYPE)
        check(self.lib.pimgpu_intersect(self.handle, _ptr(ro), _ptr(rd), _ptr(tn), _ptr(tf), C.c_int32(n), _ptr(out)), "pimgpu_intersect")
        return out

    def trace_rays(self, ro, rd, sampleIndex=0, seed=0, maxBounces=0):
        ro, rd = f32(ro), f32(rd)
        n = ro.shape[0]
        out = np.zeros(n, RESULT_DTYPE)
        check(self.lib.pimgpu_trace_rays(self.handle, _ptr(ro), _ptr(rd), C.c_int32(n), C.c_uint32(sampleIndex), C.c_uint32(seed),
              C.c_int32(maxBounces), _ptr(out)), "pimgpu_trace_rays")
        return out

    def raygen(self, origin, count, sampleIndex=0, seed=0, maxBounces=0):
        origin = f32(origin)
        colors = np.zeros((count, 4), np.float32)
        directions = np.zeros((count, 4), np.float32)
        check(self.lib.pimgpu_raygen(self.handle, _ptr(origin), C.c_int32(count), C.c_uint32(sampleIndex), C.c_uint32(seed),
              C.c_int32(maxBounces), _ptr(colors), _ptr(directions)), "pimgpu_raygen")
        return colors, directions

    def cubemap_bake(self, faces, size, origin, weight, sampleIndex=0, seed=0, maxBounces=0):
        """Cubemap_Bake (rendering/cubemap.c:162-190); faces: six float32 arrays of size*size*3, updated in place"""
        arr = (C.c_void_p * 6)()
        for f in range(6):
            assert faces[f].dtype == np.float32 and faces[f].flags["C_CONTIGUOUS"] and faces[f].size == size * size * 3
            arr[f] = faces[f].ctypes.data
        origin = f32(origin)
        check(self.lib.pimgpu_cubemap_bake(self.handle, arr, C.c_int32(size), _ptr(origin), C.c_float(weight), C.c_uint32(sampleIndex),
              C.c_uint32(seed), C.c_int32(maxBounces)), "pimgpu_cubemap_bake")

    def ambcube_bake(self, cube, origin, samples, prevSampleCount=0, sampleIndex=0, seed=0, maxBounces=0):
        """AmbCube_Bake (math/ambcube.c:5-32); cube: float32[6,4] updated in place; returns the new sample count"""
        assert cube.dtype == np.float32 and cube.flags["C_CONTIGUOUS"] and cube.size == 24
        origin = f32(origin)
        out = C.c_int32(0)
        check(self.lib.pimgpu_ambcube_bake(self.handle, _ptr(cube), _ptr(origin), C.c_int32(samples), C.c_int32(prevSampleCount),
              C.c_uint32(sampleIndex), C.c_uint32(seed), C.c_int32(maxBounces), C.byref(out)), "pimgpu_ambcube_bake")
        return out.value

    def lightmap_bake(self, pages, lmSize, texelsPerMeter, timeSlice, spp, passIndex=0, seed=0, maxBounces=0,
                      texelBegin=0, texelEnd=-1):
        """pages: list of float32 arrays of lmSize*lmSize*27 floats each (the reference's page layout), updated in place"""
        arr = (PimLightmapPage * len(pages))()
        for i, p in enumerate(pages):
            assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"] and p.size == lmSize * lmSize * 27
            arr[i].block = p.ctypes.data
        check(self.lib.pimgpu_lightmap_bake(self.handle, arr, C.c_int32(len(pages)), C.c_int32(lmSize), C.c_float(texelsPerMeter),
              C.c_float(timeSlice), C.c_int32(spp), C.c_uint32(passIndex), C.c_uint32(seed), C.c_int32(maxBounces),
              C.c_int64(texelBegin), C.c_int64(texelEnd)), "pimgpu_lightmap_bake")


def _page_table(pages, lmSize):
    arr = (PimLightmapPage * len(pages))()
    for i, p in enumerate(pages):
        assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"] and p.size == lmSize * lmSize * 27
        arr[i].block = p.ctypes.data
    return arr


class LightmapSet:
    """lightmap pages that stay in HBM between bake calls (pimgpu_lightmap_set_*): upload once, bake many times, download
    probes | sampleCounts when the host wants them"""

    def __init__(self, lmCount, lmSize):
        self.lib = load_library()
        self.lmCount, self.lmSize = lmCount, lmSize
        self.handle = C.c_void_p()
        check(self.lib.pimgpu_lightmap_set_create(C.c_int32(lmCount), C.c_int32(lmSize), C.byref(self.handle)), "pimgpu_lightmap_set_create")

    def close(self):
        if self.handle:
            self.lib.pimgpu_lightmap_set_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, pages):
        assert len(pages) == self.lmCount
        check(self.lib.pimgpu_lightmap_set_upload(self.handle, _page_table(pages, self.lmSize)), "pimgpu_lightmap_set_upload")

    def download(self, pages):
        assert len(pages) == self.lmCount
        check(self.lib.pimgpu_lightmap_set_download(self.handle, _page_table(pages, self.lmSize)), "pimgpu_lightmap_set_download")

    def bake(self, scene, texelsPerMeter, timeSlice, spp, passIndex=0, seed=0, maxBounces=0, texelBegin=0, texelEnd=-1):
        check(self.lib.pimgpu_lightmap_set_bake(scene.handle, self.handle, C.c_float(texelsPerMeter), C.c_float(timeSlice), C.c_int32(spp),
              C.c_uint32(passIndex), C.c_uint32(seed), C.c_int32(maxBounces), C.c_int64(texelBegin), C.c_int64(texelEnd)), "pimgpu_lightmap_set_bake")


class Trace:
    """PtTrace_New .. PtTrace_Del + Pt_Trace (rendering/path_tracer.h:76-77,101)."""

    def __init__(self, width, height):
        self.lib = load_library()
        self.width, self.height = width, height
        self.handle = C.c_void_p()
        check(self.lib.pimgpu_trace_create(C.c_int32(width), C.c_int32(height), C.byref(self.handle)), "pimgpu_trace_create")

    def close(self):
        if self.handle:
            self.lib.pimgpu_trace_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(self.lib.pimgpu_trace_clear(self.handle), "pimgpu_trace_clear")

    def trace(self, scene, camera, dof, params):
        check(self.lib.pimgpu_trace_image(scene.handle, self.handle, C.byref(camera), C.byref(dof), C.byref(params)), "pimgpu_trace_image")

    def trace_samples(self, scene, camera, dof, params, count):
        """`count` samples (indices params.sampleIndex ...) sharded by sample index over the device list, blended in as one"""
        check(self.lib.pimgpu_trace_image_samples(scene.handle, self.handle, C.byref(camera), C.byref(dof), C.byref(params), C.c_int32(count)),
              "pimgpu_trace_image_samples")

    def trace_to_host(self, scene, camera, dof, params, color, albedo=None, normal=None):
        """pimgpu_trace_image_host: the sample and the copy of the images to the host arrays (float32 [W*H, 3]) in one call"""
        check(self.lib.pimgpu_trace_image_host(scene.handle, self.handle, C.byref(camera), C.byref(dof), C.byref(params),
                                               _ptr(color), _ptr(albedo) if albedo is not None else None, _ptr(normal) if normal is not None else None),
              "pimgpu_trace_image_host")

    def download(self):
        n = self.width * self.height
        color, albedo, normal = (np.zeros((n, 3), np.float32) for _ in range(3))
        check(self.lib.pimgpu_trace_download(self.handle, _ptr(color), _ptr(albedo), _ptr(normal)), "pimgpu_trace_download")
        return color, albedo, normal

    def buffers(self):
        c, a, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.lib.pimgpu_trace_buffers(self.handle, C.byref(c), C.byref(a), C.byref(n)), "pimgpu_trace_buffers")
        return c.value, a.value, n.value

    def copy_dev(self, color_ptr, albedo_ptr=None, normal_ptr=None):
        """device-to-device copy of the running-mean images into caller-owned HBM (e.g. torch tensors)"""
        check(self.lib.pimgpu_trace_copy_dev(self.handle, C.c_void_p(color_ptr), C.c_void_p(albedo_ptr), C.c_void_p(normal_ptr)), "pimgpu_trace_copy_dev")

    def stddev(self):
        m, s = C.c_float(), C.c_float()
        check(self.lib.pimgpu_trace_stddev(self.handle, C.byref(m), C.byref(s)), "pimgpu_trace_stddev")
        return m.value, s.value

    def stddev_noalloc(self):
        m, s = C.c_float(), C.c_float()
        check(self.lib.pimgpu_trace_stddev_noalloc(self.handle, C.byref(m), C.byref(s)), "pimgpu_trace_stddev_noalloc")
        return m.value, s.value

    def stats(self):
        out = PimTraceStats()
        check(self.lib.pimgpu_trace_stats(self.handle, C.byref(out)), "pimgpu_trace_stats")
        return out


class Drawables:
    """the Entities columns + mesh table FlattenDrawables reads (rendering/drawable.h:14-27, mesh.h:25-33), kept alive
    as numpy arrays behind a PimDrawables struct.  meshes: list of (positions, normals, uvs) float32 (len,4) arrays,
    optionally with a fourth int32 (len,4) array, texIndices (lightmap id in .w; only lightmap_embed reads it);
    mesh_ids: per entity index into meshes (-1 = no mesh); matrices: (count,4,4) float32 with matrices[i][c] = column c."""

    def __init__(self, meshes, mesh_ids, matrices, materials):
        self.keep = []
        self.mesh_arrays = []                     # per mesh: (positions, normals, uvs, texIndices or None) as uploaded
        self.table = (PimMesh * max(1, len(meshes)))()
        for i, mesh in enumerate(meshes):
            P, N, U = f32(mesh[0]), f32(mesh[1]), f32(mesh[2])
            assert P.shape == N.shape == U.shape and (P.ndim == 2 and P.shape[1] == 4 or P.size == 0)
            self.keep += [P, N, U]
            self.table[i].length = P.shape[0]
            self.table[i].positions, self.table[i].normals, self.table[i].uvs = P.ctypes.data, N.ctypes.data, U.ctypes.data
            T = None
            if len(mesh) > 3 and mesh[3] is not None:
                T = np.ascontiguousarray(mesh[3], np.int32)
                assert T.shape == P.shape
                self.keep.append(T)
                self.table[i].texIndices = T.ctypes.data
            self.mesh_arrays.append((P, N, U, T))
        self.mesh_ids = np.ascontiguousarray(mesh_ids, np.int32)
        self.matrices = f32(matrices).reshape(-1, 16)
        self.materials = (PimMaterial * max(1, len(materials)))(*materials)
        assert self.matrices.shape[0] == self.mesh_ids.shape[0] == len(materials)
        self.desc = PimDrawables()
        self.desc.count = self.mesh_ids.shape[0]
        self.desc.meshCount = len(meshes)
        self.desc.meshes = self.mesh_ids.ctypes.data
        self.desc.matrices = self.matrices.ctypes.data
        self.desc.materials = C.cast(self.materials, C.c_void_p)
        self.desc.meshTable = C.cast(self.table, C.c_void_p)


def flatten_counts(drawables):
    v, m = C.c_int32(), C.c_int32()
    check(load_library().pimgpu_flatten_counts(C.byref(drawables.desc), C.byref(v), C.byref(m)), "pimgpu_flatten_counts")
    return v.value, m.value


def flatten(drawables):
    """FlattenDrawables on the GPU -> positions (V,4), normals (V,4), uvs (V,2), matIds (V,), materials (ctypes array)"""
    V, M = flatten_counts(drawables)
    P, N = np.zeros((V, 4), np.float32), np.zeros((V, 4), np.float32)
    U, ids = np.zeros((V, 2), np.float32), np.zeros(V, np.int32)
    mats = (PimMaterial * max(1, M))()
    check(load_library().pimgpu_flatten(C.byref(drawables.desc), _ptr(P), _ptr(N), _ptr(U), _ptr(ids), mats), "pimgpu_flatten")
    return P, N, U, ids, mats


def host_array(count, dtype=np.float32):
    """a zeroed numpy array in page-locked host memory (pimgpu_host_alloc): host <-> device copies of it run at link rate.
    The memory is freed when the array is collected."""
    lib = load_library()
    lib.pimgpu_host_alloc.restype = C.c_void_p
    nbytes = int(count) * np.dtype(dtype).itemsize
    p = lib.pimgpu_host_alloc(C.c_size_t(nbytes))
    if not p:
        raise PimGpuError("pimgpu_host_alloc failed: " + lib.pimgpu_last_error().decode())
    buf = (C.c_char * nbytes).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(count))
    import weakref
    weakref.finalize(buf, lib.pimgpu_host_free, C.c_void_p(p))
    return arr


def _page_array(pages, lmSize):
    arr = (PimLightmapPage * max(1, len(pages)))()
    for i, p in enumerate(pages):
        assert p.dtype == np.float32 and p.flags["C_CONTIGUOUS"] and p.size == lmSize * lmSize * 27
        arr[i].block = p.ctypes.data
    return arr


def lightmap_embed(drawables, pages, lmSize, page_tex_ids=None):
    """EmbedAttributes (rendering/lightmap.c:941-1045) on the GPU: fills position | normal | sampleCounts of the pages
    (float32 arrays of lmSize*lmSize*27 floats each, the reference's page layout) from the meshes' uvs.zw / texIndices.w"""
    ids = None if page_tex_ids is None else np.ascontiguousarray(page_tex_ids, np.int32)
    assert ids is None or ids.shape == (len(pages),)
    check(load_library().pimgpu_lightmap_embed(C.byref(drawables.desc), None if ids is None else _ptr(ids),
          _page_array(pages, lmSize), C.c_int32(len(pages)), C.c_int32(lmSize)), "pimgpu_lightmap_embed")
    return pages


def lightmap_chartnodes(drawables, texelsPerUnit):
    """chartnodes_create (rendering/lightmap.c:723-767) -> structured array of CHARTNODE_DTYPE, one per mapped triangle"""
    lib = load_library()
    n = C.c_int32()
    check(lib.pimgpu_lightmap_chartnodes(C.byref(drawables.desc), C.c_float(texelsPerUnit), None, C.c_int32(0), C.byref(n)), "pimgpu_lightmap_chartnodes")
    nodes = np.zeros(n.value, CHARTNODE_DTYPE)
    check(lib.pimgpu_lightmap_chartnodes(C.byref(drawables.desc), C.c_float(texelsPerUnit), _ptr(nodes), C.c_int32(n.value), C.byref(n)), "pimgpu_lightmap_chartnodes")
    return nodes


def lightmap_chart_masks(nodes, chartFirst):
    """ChartMaskFn (rendering/lightmap.c:527-561): nodes (CHARTNODE_DTYPE, charts contiguous) are moved to their chart's
    origin in place; returns (areas, maskSizes (charts, 2), maskOffsets, masks uint8)"""
    lib = load_library()
    first = np.ascontiguousarray(chartFirst, np.int32)
    nc = first.shape[0] - 1
    areas, sizes = np.zeros(nc, np.float32), np.zeros((nc, 2), np.int32)
    check(lib.pimgpu_lightmap_chart_layout(_ptr(nodes), _ptr(first), C.c_int32(nc), _ptr(areas), _ptr(sizes)), "pimgpu_lightmap_chart_layout")
    lens = sizes[:, 0].astype(np.int64) * sizes[:, 1]
    offsets = np.zeros(nc, np.int64); offsets[1:] = np.cumsum(lens)[:-1]
    masks = np.full(int(lens.sum()), 7, np.uint8)
    check(lib.pimgpu_lightmap_chart_masks(_ptr(nodes), _ptr(first), C.c_int32(nc), _ptr(sizes), _ptr(offsets), _ptr(masks)), "pimgpu_lightmap_chart_masks")
    return areas, sizes, offsets, masks


def lightmap_atlas_place(masks, offsets, sizes, areas, order, atlasSize, atlasCount, atlasCap=0):
    """atlases_create (rendering/lightmap.c:845-874) as one worker runs it -> (translations (charts, 2), atlasIndex, usedAtlases)"""
    nc = int(np.asarray(order).shape[0])
    masks = np.ascontiguousarray(masks, np.uint8); offsets = np.ascontiguousarray(offsets, np.int64)
    sizes = np.ascontiguousarray(sizes, np.int32); areas = f32(areas); order = np.ascontiguousarray(order, np.int32)
    tr, idx, used = np.full((nc, 2), -7, np.int32), np.full(nc, -7, np.int32), C.c_int32(-1)
    check(load_library().pimgpu_lightmap_atlas_place(_ptr(masks), _ptr(offsets), _ptr(sizes), _ptr(areas), _ptr(order), C.c_int32(nc),
          C.c_int32(atlasSize), C.c_int32(atlasCount), C.c_int32(atlasCap), _ptr(tr), _ptr(idx), C.byref(used)), "pimgpu_lightmap_atlas_place")
    return tr, idx, used.value


def brdf_lut_bake(texels, prevSamples, numSamples, seed=0):
    """BakeBrdfFn (math/lighting.c:31-84) over a float32 (height, width, 2) table, in place"""
    assert texels.dtype == np.float32 and texels.flags["C_CONTIGUOUS"] and texels.ndim == 3 and texels.shape[2] == 2
    check(load_library().pimgpu_brdf_lut_bake(_ptr(texels), C.c_int32(texels.shape[1]), C.c_int32(texels.shape[0]),
          C.c_int32(prevSamples), C.c_int32(numSamples), C.c_uint32(seed)), "pimgpu_brdf_lut_bake")
    return texels


def work_counters(enable):
    """switch the counting build of the traversal kernel on/off; returns (nodes, triangles, rays, worst ray's nodes) since the last call"""
    out = (C.c_uint64 * 4)()
    check(load_library().pimgpu_work_counters(C.c_int(1 if enable else 0), out), "pimgpu_work_counters")
    return int(out[0]), int(out[1]), int(out[2]), int(out[3])


def init(device=0):
    check(load_library().pimgpu_init(C.c_int(device)), "pimgpu_init")


def init_devices(devices):
    """pimgpu_init_devices: one context per entry of the list (a device may repeat)"""
    arr = (C.c_int * len(devices))(*devices)
    check(load_library().pimgpu_init_devices(arr, C.c_int(len(devices))), "pimgpu_init_devices")


def shutdown():
    load_library().pimgpu_shutdown()
