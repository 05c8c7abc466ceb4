"""page-locked lightmap pages: what pimgpu_host_alloc's time is made of (62 blocks of 113 MB, as config 4's pack makes them)
usage: python scripts/micro/time_host_alloc.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pim_amd import api
lib = api.load_library()
assert lib.pimgpu_init(0) == 0
hip = C.CDLL("libamdhip64.so")
n, bytes_ = 62, 1024 * 1024 * 27 * 4
lib.pimgpu_host_alloc.restype = C.c_void_p
t0 = time.perf_counter(); ps = [lib.pimgpu_host_alloc(C.c_size_t(bytes_)) for _ in range(n)]; t1 = time.perf_counter()
for p in ps: lib.pimgpu_host_free(C.c_void_p(p))
t2 = time.perf_counter()
raw = []
for _ in range(n):
    p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), C.c_size_t(bytes_), C.c_uint(0)) == 0; raw.append(p)
t3 = time.perf_counter()
for p in raw: hip.hipHostFree(p)
print("pimgpu_host_alloc x%d: %.3f s (free %.3f s); hipHostMalloc alone: %.3f s" % (n, t1 - t0, t2 - t1, t3 - t2))
