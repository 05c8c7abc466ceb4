"""Pt_Trace through the plain-C host layer (libpimhost.so), host images included: ms per call at 1080p"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pim_amd import api, scenes
from tests.test_gpu_host_api import PtTrace, Int2, ROOT
host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
host.PtSys_Init.restype = C.c_bool; host.PtScene_New.restype = C.c_void_p
assert host.PtSys_Init()
sc = scenes.cornell_box(); scene = C.c_void_p(host.PtScene_New(C.byref(sc.desc)))
trace = PtTrace(); host.PtTrace_New(C.byref(trace), Int2(1920, 1080)); trace.seed, trace.maxBounces = 1, 8
keep = None
if "--pageable" in sys.argv:                       # the images in ordinary (pageable) host memory, as calloc would give
    import numpy as np
    keep = (trace.color, trace.albedo, trace.normal)
    arrs = [np.zeros(1920 * 1080 * 3, np.float32) for _ in range(3)]
    trace.color, trace.albedo, trace.normal = [a.ctypes.data for a in arrs]
dof = api.PimDofInfo(); host.DofInfo_New(C.byref(dof)); cam = scenes.pt_test_camera()
for s in range(4):
    trace.sampleWeight = 1.0 / (s + 1); host.Pt_Trace(C.byref(trace), C.byref(dof), scene, C.byref(cam))
t0 = time.perf_counter()
for s in range(4, 36):
    trace.sampleWeight = 1.0 / (s + 1); host.Pt_Trace(C.byref(trace), C.byref(dof), scene, C.byref(cam))
print("pageable" if keep else "page-locked", "Pt_Trace host layer, 1080p Cornell, images on the host: %.3f ms per call" % ((time.perf_counter() - t0) / 32 * 1e3))
