"""profiling target: N launches of the closest-hit traversal kernel on shuffled secondary rays of the 1M-triangle scene"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pim_amd import api, scenes

ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nray = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
api.init(0)
lib = api.load_library()
sc = scenes.synthetic_scene(target_tris=ntri, seed=1234, extent=100.0, emissive_attempts=16)
g = api.Scene(sc.desc)
ro, rd = scenes.surface_rays(sc, nray, seed=1)
ro[:, 3] = 0.0; rd[:, 3] = api.RCP_EPSILON
tro = torch.from_numpy(ro).cuda(); trd = torch.from_numpy(rd).cuda()
hits = torch.zeros((nray, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    api.check(lib.pimgpu_intersect_packed_dev(g.handle, ctypes.c_void_p(tro.data_ptr()), ctypes.c_void_p(trd.data_ptr()), ctypes.c_int32(nray), ctypes.c_void_p(hits.data_ptr())))
print("done", int((hits[:, 0] >= 0).sum()))
