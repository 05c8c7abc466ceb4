"""Pt_Trace through the plain-C host layer (libpimhost.so, the reference's own signatures), host images included:
ms per call at 1080p.  usage: python scripts/time_host_trace.py [cornell|synth1m] [--color-only] [--4k]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pim_amd import api, scenes
from tests.test_gpu_host_api import PtTrace, Int2, ROOT
host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
host.PtSys_Ok.restype = C.c_bool; host.PtScene_NewFromSnapshot.restype = C.c_void_p
color_only = "--color-only" in sys.argv
which = "synth1m" if "synth1m" in sys.argv else "cornell"
cfg = (C.c_int32 * 5)(1, 8, 0, 0 if color_only else 1, -1)        # PtSysConfig: seed 1, 8 bounces, dt default, downloadAovs, device
host.PtSys_Configure(cfg); host.PtSys_Init()
assert host.PtSys_Ok()
if which == "cornell":
    sc, cam = scenes.cornell_box(), scenes.pt_test_camera()
else:
    sc, cam = scenes.synthetic_scene(target_tris=1_000_000, seed=1234, extent=100.0, emissive_attempts=64), scenes.synthetic_camera(100.0)
scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
trace = PtTrace(); host.PtTrace_New(C.byref(trace), Int2(W, H))
dof = api.PimDofInfo(); host.DofInfo_New(C.byref(dof))
for s in range(4):
    trace.sampleWeight = 1.0 / (s + 1); host.Pt_Trace(C.byref(trace), C.byref(dof), scene, C.byref(cam))
t0 = time.perf_counter()
for s in range(4, 36):
    trace.sampleWeight = 1.0 / (s + 1); host.Pt_Trace(C.byref(trace), C.byref(dof), scene, C.byref(cam))
print(f"Pt_Trace host layer, {W}x{H} {which}, {'color only' if color_only else 'color + albedo + normal'} on the host: "
      "%.3f ms per call" % ((time.perf_counter() - t0) / 32 * 1e3))
