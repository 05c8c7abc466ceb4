"""dump rays on which the GPU traversal and the CPU brute-force checker disagree (diagnosis aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pim_amd import api, scenes
from tests.util import OracleScene
from tests.test_gpu_intersect import _rays_surface
api.init(0)
sc = scenes.cornell_box()
g = api.Scene(sc.desc); o = OracleScene(sc.desc, accel=0)
out = {}
for name, (ro, rd) in (("random", scenes.random_rays(sc, 200000, seed=11)), ("surface", _rays_surface(sc, 200000, seed=12))):
    n = ro.shape[0]
    tfar = np.full(n, api.RCP_EPSILON, np.float32)
    tfar[::7] = np.random.default_rng(5).uniform(0.01, 6.0, tfar[::7].shape).astype(np.float32)
    tnear = np.zeros(n, np.float32); tnear[::13] = 0.5
    hg = g.intersect(ro, rd, tnear, tfar); ho = o.intersect(ro, rd, tnear, tfar)
    bad = np.nonzero((hg["iVert"] != ho["iVert"]) | (hg["wuvt"].view(np.uint32) != ho["wuvt"].view(np.uint32)).any(1))[0]
    print(name, "bad", bad.size)
    out[name + "_idx"] = bad; out[name + "_ro"] = ro[bad]; out[name + "_rd"] = rd[bad]; out[name + "_tn"] = tnear[bad]; out[name + "_tf"] = tfar[bad]
    out[name + "_g_iv"] = hg["iVert"][bad]; out[name + "_o_iv"] = ho["iVert"][bad]; out[name + "_g_w"] = hg["wuvt"][bad]; out[name + "_o_w"] = ho["wuvt"][bad]
np.savez("gpurun_out/miss.npz", **out)
