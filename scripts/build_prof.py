"""profiling target: BVH builds of the synthetic scene (rtcCommitScene's replacement): python scripts/build_prof.py <tris> [rebuilds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pim_amd import api, scenes
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
api.init(0)
sc = scenes.synthetic_scene(target_tris=ntri, seed=1234, extent=100.0, emissive_attempts=4, diffuse_only=True)
g = api.Scene(sc.desc)
ms = []
for _ in range(reps):
    g.rebuild_bvh(); ms.append(round(g.info().bvhBuildMs, 3))
print("tris", g.info().triCount, "build ms", ms)
