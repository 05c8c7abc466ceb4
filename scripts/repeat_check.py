"""repeat the Cornell-spheres closest-hit parity check N times in one process (fresh scene and BVH each time): GPU records
against the oracle's, bit for bit.  usage: python scripts/repeat_check.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pim_amd import api, scenes
from tests.util import OracleScene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
api.init(0)
sc = scenes.cornell_box("spheres", emissive_attempts=32)
o = OracleScene(sc.desc, accel=1)
ro, rd = scenes.random_rays(sc, 20000, seed=11)
n = ro.shape[0]
tfar = np.full(n, api.RCP_EPSILON, np.float32)
tfar[::7] = np.random.default_rng(5).uniform(0.01, 6.0, tfar[::7].shape).astype(np.float32)
tnear = np.zeros(n, np.float32); tnear[::13] = 0.5
ho = o.intersect(ro, rd, tnear, tfar)
bad_runs = 0
for i in range(N):
    g = api.Scene(sc.desc)
    hg = g.intersect(ro, rd, tnear, tfar)
    bad = np.nonzero((hg.view(np.uint8).reshape(n, -1) != ho.view(np.uint8).reshape(n, -1)).any(axis=1))[0]
    if bad.size:
        bad_runs += 1
        print(f"run {i}: {bad.size} rays differ, e.g. {bad[:8]}: gpu {hg[bad[0]]} oracle {ho[bad[0]]}", flush=True)
    g.close()
print(f"{bad_runs} of {N} runs differ", flush=True)
