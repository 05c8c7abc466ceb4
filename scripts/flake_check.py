"""determinism check of the closest-hit kernel and of the BVH build: the same rays through the same scene must give the same
bits on every launch, and after every rebuild (the result does not depend on the tree, traverse.h contract)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pim_amd import api, scenes

api.init(0)
for name, sc, nray in (("cornell_spheres", scenes.cornell_box("spheres", emissive_attempts=8), 200000),
                       ("synthetic_100k", scenes.synthetic_scene(target_tris=100000, seed=5, extent=40.0, emissive_attempts=4), 400000)):
    g = api.Scene(sc.desc)
    ro, rd = scenes.surface_rays(sc, nray, seed=3, offset=1e-5)
    first = g.intersect(ro, rd)
    bad_launch = bad_build = 0
    for i in range(12):
        h = g.intersect(ro, rd)
        bad_launch += int((h.view(np.uint8).reshape(nray, -1) != first.view(np.uint8).reshape(nray, -1)).any(axis=1).sum())
    for i in range(12):
        g.rebuild_bvh()
        h = g.intersect(ro, rd)
        bad_build += int((h.view(np.uint8).reshape(nray, -1) != first.view(np.uint8).reshape(nray, -1)).any(axis=1).sum())
    print(f"{name}: rays differing over 12 relaunches {bad_launch}, over 12 rebuilds {bad_build}", flush=True)
    g.close()
