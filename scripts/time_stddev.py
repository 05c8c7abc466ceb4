"""time pimgpu_trace_stddev (CalcStdDev) on a rendered 1080p image"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pim_amd import api, scenes
api.init(0)
sc = scenes.cornell_box(); g = api.Scene(sc.desc); t = api.Trace(1920, 1080); dof = scenes.default_dof()
for i in range(2): t.trace(g, scenes.pt_test_camera(), dof, scenes.trace_params(i, seed=1, max_bounces=8))
t.stddev()
t0 = time.perf_counter(); r = t.stddev(); dt = time.perf_counter() - t0
print("stddev", r, "ms", round(dt * 1e3, 2))
